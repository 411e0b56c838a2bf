cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
timeout -k 10 240 python tools/exp_hamming_mfma.py > $O/exp_hamming_mfma_v2.log 2>&1; echo "mfma rc $?"; tail -4 $O/exp_hamming_mfma_v2.log
timeout -k 10 1000 bash tools/exp_lib_bench.sh nclt-slam-project_amd/csrc/libreloc_hip.so build_variants/libreloc_hip_scan104.so build_variants/libreloc_hip_scan96.so > $O/scan_vgpr_ab.log 2>&1; cat $O/scan_vgpr_ab.log

cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
timeout -k 10 900 bash tools/exp_lib_bench.sh nclt-slam-project_amd/csrc/libreloc_hip.so build_variants/libreloc_hip_fast48.so build_variants/libreloc_hip_fast40.so > $O/fast_vgpr_ab.log 2>&1; cat $O/fast_vgpr_ab.log

cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 1100 bash tools/exp_lib_bench.sh nclt-slam-project_amd/csrc/libreloc_hip.so build_variants/libreloc_hip_finish5.so build_variants/libreloc_hip_emit96.so build_variants/libreloc_hip_both96.so > $O/small_kernel_vgpr_ab2.log 2>&1; cat $O/small_kernel_vgpr_ab2.log

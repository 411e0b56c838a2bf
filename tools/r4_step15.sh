cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O
timeout -k 10 300 python tools/exp_scan_intercept.py > $O/scan_intercept.log 2>&1; cat $O/scan_intercept.log
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_tick.py -x -q -m gpu -k "scheduling or db_scan_10k or fused or global_search" > $O/scan_tests.log 2>&1; tail -2 $O/scan_tests.log

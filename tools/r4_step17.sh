cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4r; mkdir -p $O
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "suite rc $?"; tail -3 $O/gputest.log
timeout -k 10 300 python tests/dev/fuzz_gpu.py --seconds 150 --seed 404 --kinds db,db_big,match,ratio > $O/fuzz_r4d.log 2>&1; echo "fuzz rc $?"; tail -1 $O/fuzz_r4d.log
timeout -k 10 600 bash tools/exp_lib_bench.sh nclt-slam-project_amd/csrc/libreloc_hip.so build_variants/libreloc_hip_bflysel.so > $O/bfly_ab.log 2>&1; cat $O/bfly_ab.log
timeout -k 10 200 python tools/exp_scan_intercept.py > $O/scan_intercept.log 2>&1; tail -5 $O/scan_intercept.log

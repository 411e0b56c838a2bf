cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "suite rc $?"; tail -3 $O/gputest.log
timeout -k 10 300 python tests/dev/fuzz_gpu.py --seconds 150 --seed 402 --kinds orb,orb_bgr,record > $O/fuzz_r4b.log 2>&1; echo "fuzz rc $?"; tail -1 $O/fuzz_r4b.log
timeout -k 10 600 bash tools/exp_fast_grid.sh "-1 1024 512 -1 1024 512" > $O/fast_grid2.log 2>&1; cat $O/fast_grid2.log
timeout -k 10 200 python tools/exp_stage_throughput.py 4 > $O/stage_throughput.log 2>&1; tail -8 $O/stage_throughput.log

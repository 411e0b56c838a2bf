cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "suite rc $?"; tail -3 $O/gputest.log
timeout -k 10 200 python tools/exp_maxfeat_scan.py > $O/maxfeat_scan.log 2>&1; cat $O/maxfeat_scan.log
timeout -k 10 900 bash tools/exp_fast_grid.sh "-1 256 512 1024 -1 256 512 1024" > $O/fast_grid.log 2>&1; cat $O/fast_grid.log

cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4o; mkdir -p $O
timeout -k 10 420 python tests/dev/fuzz_gpu.py --seconds 330 --seed 403 > $O/fuzz_r4c.log 2>&1; echo "fuzz rc $?"; tail -2 $O/fuzz_r4c.log
timeout -k 10 400 python tools/soak_gpu.py > $O/soak_r4.log 2>&1; echo "soak rc $?"; tail -4 $O/soak_r4.log

# round 4, first GPU call: RCCL at world size 1, bench.py --gpus N self-launch, --force-dist lines
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
NOX="--no-cpu-baseline --no-ingest --no-2hz --no-extra-scans --no-matrix"
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "rccl" > $O/rccl_test.log 2>&1; echo "rccl test rc $?"; tail -3 $O/rccl_test.log
timeout -k 10 300 python bench.py --gpus 2 --rehearse --backend gloo --steps 10 $NOX > $O/bench_gpus2_selflaunch.json 2>$O/bench_gpus2_selflaunch.err; echo "selflaunch rc $?"; tail -c 600 $O/bench_gpus2_selflaunch.json
timeout -k 10 300 python bench.py --shard-db --force-dist --steps 30 > $O/bench_shard10k_rccl_world1.json 2>$O/bench_shard10k_rccl_world1.err; echo "shard force-dist rc $?"; tail -c 900 $O/bench_shard10k_rccl_world1.json
timeout -k 10 300 python bench.py --shard-db --steps 30 > $O/bench_shard10k.json 2>/dev/null; tail -c 400 $O/bench_shard10k.json
timeout -k 10 300 python bench.py --force-dist --steps 50 $NOX > $O/bench_force_dist.json 2>$O/bench_force_dist.err; echo "bench force-dist rc $?"; tail -c 700 $O/bench_force_dist.json
WORLD_SIZE=4 RANK=0 python bench.py --gpus 8; echo "mismatch rc $?"

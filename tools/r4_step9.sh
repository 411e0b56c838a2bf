cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
timeout -k 10 900 bash tools/exp_fast_grid.sh "-1 512 1024 256 -1 512 1024 256" > $O/fast_grid3.log 2>&1; cat $O/fast_grid3.log
NOX="--no-cpu-baseline --no-ingest --no-2hz --no-extra-scans --no-matrix"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 10 --warmup 2 $NOX > $O/stats.log 2>&1
head -14 $O/stats/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120

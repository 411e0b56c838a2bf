cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
F="--no-cpu-baseline --no-matrix --no-ingest --no-2hz --no-extra-scans"
for rep in 1 2; do
for cfg in "4 1 0" "4 8 1" "4 8 2" "4 8 3" "4 4 1" "3 8 1" "4 2 1"; do
  set -- $cfg
  echo -n "streams=$1 batch=$2 batch_gens=$3  "
  RELOC_DEV=1 RELOC_SCAN_BATCH_GENS=$3 timeout -k 10 120 python bench.py --streams $1 --batch $2 --steps 60 $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'frames/s  step', round(d['step_ms']['median'],3), 'p95', round(d['step_ms']['p95'],3))"
done; done 2>&1 | tee $O/bench_modes2.log
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "suite rc $?"; tail -3 $O/gputest.log

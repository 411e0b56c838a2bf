cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4l; mkdir -p $O
timeout -k 10 600 bash tools/exp_scan_gens.sh "3 -1 2 1 4 3 -1 2 1 4" > $O/scan_gens.log 2>&1; cat $O/scan_gens.log
F="--no-cpu-baseline --no-matrix --no-ingest --no-2hz --no-extra-scans"
for cfg in "4 1" "5 1" "6 1" "3 1" "4 8" "4 2" "4 1" "5 1" "6 1" "4 8"; do
  set -- $cfg
  echo -n "streams=$1 batch=$2  "
  timeout -k 10 120 python bench.py --streams $1 --batch $2 --steps 60 $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'frames/s  step', round(d['step_ms']['median'],3))"
done 2>&1 | tee $O/bench_modes.log

# Developer harness: the frame benchmark in several stream / batch shapes, same box, one after the other.
#   bash tools/exp_bench_modes.sh > gpurun_out/<round>/bench_modes.log
for cfg in "4 1" "4 1" "2 8" "3 8" "4 8" "2 4" "4 4" "3 1" "6 1"; do
  set -- $cfg
  echo -n "streams=$1 batch=$2  "
  timeout -k 10 120 python bench.py --streams $1 --batch $2 --steps 40 --no-cpu-baseline --no-matrix --no-ingest 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'frames/s  step', round(d['step_ms']['median'],3), 'ms  scan/frame', round(d['stage_us']['db_scan_per_frame'],1), 'orb', round(d['stage_us']['orb'],1), 'pnp', round(d['stage_us']['pnp'],1))"
done

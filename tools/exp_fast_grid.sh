# Developer harness (round 4): the 4-stream frame benchmark against the grid of the FAST + blur launch (RELOC_FAST_GRID: -1 = one
# workgroup per tile, n > 0 = n workgroups that walk the tiles), interleaved on one box.   bash tools/exp_fast_grid.sh "-1 512 -1 512"
for g in ${1:--1 256 512 1024 -1 256 512 1024}; do
  echo -n "fast_grid=$g  "
  RELOC_DEV=1 RELOC_FAST_GRID=$g timeout -k 10 120 python bench.py --steps 60 --no-cpu-baseline --no-matrix --no-ingest --no-2hz --no-extra-scans 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'frames/s  orb alone', round(d['stage_us']['orb'],1), 'in-config orb', round(d['roofline']['in_config']['orb_us'],1), 'in-config scan', round(d['roofline']['in_config']['scan_avg_launch_us'],1), 'step p95', round(d['step_ms']['p95'],2), 'tick global/local', round(d['latency']['tick_global_us']['median']), round(d['latency']['tick_local_us']['median']))"
done

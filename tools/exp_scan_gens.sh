# Developer harness: the 4-stream frame benchmark against the scheduling form of the whole-database scan (RELOC_SCAN_GENS:
# n > 0 = n generations of workgroups with a record quota, -1 = ONE resident generation drawing tickets until they run dry),
# interleaved.   bash tools/exp_scan_gens.sh "3 -1 3 -1 2"
for g in ${1:-3 2 3 2 4 1}; do
  echo -n "gens=$g  "
  RELOC_DEV=1 RELOC_SCAN_GENS=$g timeout -k 10 120 python bench.py --steps 60 --no-cpu-baseline --no-matrix --no-ingest --no-2hz --no-extra-scans 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'frames/s  scan alone', round(d['stage_us']['db_scan_per_frame'],1), 'in-config scan', round(d['roofline']['in_config']['scan_avg_launch_us'],1), 'step p95', round(d['step_ms']['p95'],2))"
done

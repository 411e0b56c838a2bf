# Developer harness: the frame benchmark with the scan's workgroup budgets in ROWS (+ sweepers: 1 = one per CU, 2 = one per
# XCD) against the record quota of rounds 2-3a (RELOC_SCAN_QUOTA_ROWS=0), interleaved on one box.
F="--no-cpu-baseline --no-matrix --no-ingest --no-2hz --no-extra-scans"
for a in "" "--rows ragged"; do
  for rep in 1 2; do
    for m in ${1:-0 1 2}; do
      echo -n "bench $a quota_rows=$m: "
      RELOC_DEV=1 RELOC_SCAN_QUOTA_ROWS=$m timeout -k 10 200 python bench.py --steps 60 $a $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'scan alone', round(d['stage_us']['db_scan_per_frame'],1))"
    done
  done
done

run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-matrix --no-ingest --steps 60 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['step_ms']['median'],3))"; }
for g in 2 3 4 6 8 12 -1; do echo "RELOC_SCAN_GENS=$g"; RELOC_SCAN_GENS=$g run; done
echo "default"; run
for s in 3 5 6 8; do echo "default, streams $s"; run --streams $s; done

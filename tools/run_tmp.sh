for w in 3 30 3 30; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-matrix --no-ingest --steps 100 --warmup $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('warmup', d['warmup'], round(d['value'],1), d['ms_per_step'], d['step_ms'])"
done

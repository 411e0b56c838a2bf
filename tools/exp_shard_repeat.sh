# Developer harness: run-to-run spread of the sharded / batched / default frame benchmark on one box.
F="--no-cpu-baseline --no-matrix --no-ingest --no-2hz --no-extra-scans"
for i in 1 2 3; do
  echo -n "shard10k        "; timeout -k 10 120 python bench.py --shard-db --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']))"
  echo -n "shard10k depth6 "; BENCH_SHARD_DEPTH=6 timeout -k 10 120 python bench.py --shard-db --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']))"
  echo -n "batch8 4streams "; timeout -k 10 120 python bench.py --batch 8 --steps 40 $F 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']))"
  echo -n "batch1 4streams "; timeout -k 10 120 python bench.py --steps 40 $F 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']))"
done

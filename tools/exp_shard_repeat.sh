# Developer harness: run-to-run spread of the sharded frame benchmark against the number of batches in flight, beside the
# batched / default unsharded modes, on one box.
F="--no-cpu-baseline --no-matrix --no-ingest --no-2hz --no-extra-scans"
for i in 1 2 3; do
  for d in ${1:-8 16 32}; do
    echo -n "shard10k depth $d  "; BENCH_SHARD_DEPTH=$d timeout -k 10 120 python bench.py --shard-db --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']))"
  done
  echo -n "batch8 4streams    "; timeout -k 10 120 python bench.py --batch 8 --steps 40 $F 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']))"
done

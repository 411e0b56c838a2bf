# Developer harness: the frame benchmark (4 streams) and the synchronous tick latency for several builds of the library
# (RELOC_LIB), interleaved on one box.   bash tools/exp_lib_bench.sh lib1.so lib2.so ...
F="--no-cpu-baseline --no-matrix --no-ingest --no-2hz --no-extra-scans"
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib): "
    RELOC_DEV=1 RELOC_LIB=$(realpath $lib) timeout -k 10 200 python bench.py --steps 60 $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'frames/s  tick', round(d['latency']['tick_global_us']['median'],1), round(d['latency']['tick_local_us']['median'],1))"
  done
done

# Collects the judged evidence of a round on the GPU box (run through gpurun; writes under gpurun_out/<round>/, the
# summaries are copied into profiles/ by tools/copy_round_evidence.sh afterwards).
#   usage: bash tools/collect_round_evidence.sh r4 [part]      part: all (default) | core | extra
set -e
R=${1:-r4}; PART=${2:-all}
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/$R; mkdir -p $O
NOX="--no-cpu-baseline --no-ingest --no-2hz --no-extra-scans"
if [ $PART = all ] || [ $PART = core ]; then
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1 || true
tail -2 $O/gputest.log
P=$O/pmc; mkdir -p $P
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -o fetch -- python3 tools/pmc_target.py > $P/fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -o write -- python3 tools/pmc_target.py > $P/write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/sq -o sq -- python3 tools/pmc_target.py > $P/sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $P/sq2 -o sq2 -- python3 tools/pmc_target.py > $P/sq2.log 2>&1
python tools/pmc_summarize.py $P $P/summary.json > /dev/null || true
mkdir -p profiles/pmc_$R && cp $P/summary.json profiles/pmc_$R/summary.json      # bench.py reads the round's own traffic figures
echo pmc done
timeout -k 10 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
echo bench done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 10 --warmup 2 $NOX > $O/stats.log 2>&1
# one stream: the scan in the scheduling form of the 4-stream line's `roofline` (one resident generation since round 4, in every
# context): the kernel's average here is what roofline.avg_launch_us states
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o bench1 -- python3 bench.py --streams 1 --steps 10 --warmup 2 --no-matrix $NOX > $O/stats1.log 2>&1
# the matrix kernel: trace of bench.py --matrix-only (400 pre-roll + 1 warm-up + 100 timed launches), statistics of the TIMED tail only
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/mtrace -o matrix -- python3 bench.py --matrix-only --steps 100 --warmup 1 > $O/matrix_trace_bench.json 2>$O/mtrace.err
python tools/kernel_trace_tail.py $O/mtrace k_hamming_matrix 100 > $O/matrix_timed_only.json || true
rm -rf $O/mtrace
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || true
echo core done
fi
if [ $PART = all ] || [ $PART = extra ]; then
timeout -k 10 300 python bench.py --streams 1 --steps 30 --no-matrix $NOX 2>/dev/null | tail -1 > $O/bench_1stream.json
timeout -k 10 300 python bench.py --size 720p --steps 30 --no-matrix $NOX 2>/dev/null | tail -1 > $O/bench_720p.json
timeout -k 10 300 python bench.py --batch 8 --steps 60 --no-matrix $NOX 2>/dev/null | tail -1 > $O/bench_batch8.json
timeout -k 10 300 python bench.py --rows ragged --steps 60 --no-matrix $NOX 2>/dev/null | tail -1 > $O/bench_rows_ragged.json
timeout -k 10 400 python tools/microbench.py > $O/microbench.jsonl 2>$O/microbench.err || true
timeout -k 10 300 python tools/exp_stage_throughput.py 4 > $O/stage_throughput.log 2>&1 || true
timeout -k 10 400 python bench.py --shard-db --records 100000 --steps 10 2>/dev/null | tail -1 > $O/bench_shard100k.json || true
timeout -k 10 300 python bench.py --shard-db --steps 30 2>/dev/null | tail -1 > $O/bench_shard10k.json || true
timeout -k 10 300 python bench.py --matrix-only --steps 200 2>/dev/null | tail -1 > $O/bench_matrix_only.json || true
# the multi-rank code paths with two ranks on this ONE GPU (gloo; not a scaling number)
timeout -k 10 200 python bench.py --gpus 2 --steps 10 --warmup 2 --backend gloo --rehearse --no-matrix $NOX 2>/dev/null | tail -1 > $O/bench_2ranks_one_gpu_rehearsal.json || true
# the RCCL transport at world size 1 (one rank, one GPU): both exchanges of the sharded path and the frame benchmark's barrier / all_reduce
timeout -k 10 300 python bench.py --shard-db --force-dist --steps 30 2>/dev/null | tail -1 > $O/bench_shard10k_rccl_world1.json || true
timeout -k 10 200 python tools/exp_maxfeat_scan.py > $O/scan_maxfeat.log 2>&1 || true
timeout -k 10 300 python tools/exp_small_q.py nclt-slam-project_amd/csrc/libreloc_hip.so > $O/small_q.log 2>/dev/null || true
echo extra done
fi
echo collected

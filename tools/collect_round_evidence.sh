set -e
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
O=gpurun_out/pmc_r1e; mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o fetch -- python3 tools/pmc_target.py > $O/fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o write -- python3 tools/pmc_target.py > $O/write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -o sq -- python3 tools/pmc_target.py > $O/sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq2 -o sq2 -- python3 tools/pmc_target.py > $O/sq2.log 2>&1
python tools/pmc_summarize.py $O $O/summary.json > /dev/null
mkdir -p profiles/pmc_r1e && cp $O/summary.json profiles/pmc_r1e/summary.json
timeout -k 10 300 python bench.py 2>gpurun_out/r1e_bench.err | tail -1 > gpurun_out/r1e_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1e_stats -o r1e -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r1e_stats.log 2>&1
timeout -k 10 300 python bench.py --streams 1 --no-cpu-baseline --no-matrix 2>/dev/null | tail -1 > gpurun_out/r1e_bench_1stream.json
timeout -k 10 300 python bench.py --size 720p --no-cpu-baseline --no-matrix 2>/dev/null | tail -1 > gpurun_out/r1e_bench_720p.json
echo collected

cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
# VALU instruction counts of a tick's kernels: product (FAST compaction) vs the round-3 FAST form
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/valu_compact -o v -- python3 tools/pmc_tick_target.py > $O/valu_compact.log 2>&1
python tools/pmc_per_kernel.py $O/valu_compact SQ_INSTS_VALU > $O/tick_valu_per_kernel_compact.txt; head -6 $O/tick_valu_per_kernel_compact.txt
RELOC_DEV=1 RELOC_LIB=$PWD/build_variants/libreloc_hip_nocompact.so timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/valu_nocompact -o v -- python3 tools/pmc_tick_target.py > $O/valu_nocompact.log 2>&1
python tools/pmc_per_kernel.py $O/valu_nocompact SQ_INSTS_VALU > $O/tick_valu_per_kernel_nocompact.txt; head -6 $O/tick_valu_per_kernel_nocompact.txt
rm -rf $O/valu_compact $O/valu_nocompact
timeout -k 10 900 bash tools/exp_lib_bench.sh nclt-slam-project_amd/csrc/libreloc_hip.so build_variants/libreloc_hip_nocompact.so > $O/compact_ab.log 2>&1; cat $O/compact_ab.log
for g in 512 1024 512 1024; do echo -n "compact, fast_grid=$g "; RELOC_DEV=1 RELOC_FAST_GRID=$g timeout -k 10 200 python bench.py --steps 60 --no-cpu-baseline --no-matrix --no-ingest --no-2hz --no-extra-scans 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'frames/s in-config orb', round(d['roofline']['in_config']['orb_us'],1))"; done 2>&1 | tee $O/compact_grid.log

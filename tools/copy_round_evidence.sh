# Copies what tools/collect_round_evidence.sh left under gpurun_out/<round>/ into profiles/ under the names profiles/README.md lists.
#   bash tools/copy_round_evidence.sh r2
set -e
T=${1:-r2}; R=gpurun_out/$T
cp $R/bench.json profiles/${T}_bench.json; cp $R/bench_1stream.json profiles/${T}_bench_1stream.json; cp $R/bench_720p.json profiles/${T}_bench_720p.json
cp $R/stats/bench_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv; cp $R/stats1/bench1_kernel_stats.csv profiles/${T}_bench_1stream_kernel_stats.csv
cp $R/microbench.jsonl profiles/${T}_microbench.jsonl; cp $R/gputest.log profiles/${T}_gputest.log; cp $R/smoke.log profiles/${T}_smoke.log
cp $R/bench_2ranks_one_gpu_rehearsal.json profiles/${T}_bench_2ranks_one_gpu_rehearsal.json; cp $R/bench_matrix_only.json profiles/${T}_bench_matrix_only.json
cp $R/bench_shard100k.json profiles/${T}_bench_shard100k.json; cp $R/bench_shard10k.json profiles/${T}_bench_shard10k.json
cp $R/exp_matrix2_chains.log profiles/${T}_exp_matrix2_chains.log; cp $R/exp_matrix2_spacing.log profiles/${T}_exp_matrix2_spacing.log
cp $R/matrix_prod_series.log profiles/${T}_matrix_prod_series.log; cp $R/stage_throughput.log profiles/${T}_stage_throughput.log
mkdir -p profiles/pmc_$T
for k in fetch write sq sq2; do f=$(ls $R/pmc/$k/*counter_collection.csv | head -1); cp $f profiles/pmc_$T/${k}_counter_collection.csv; done
cp $R/pmc/summary.json profiles/pmc_$T/summary.json
echo copied

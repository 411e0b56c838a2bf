# Copies what tools/collect_round_evidence.sh left under gpurun_out/<round>/ into profiles/ under the names profiles/README.md lists.
#   bash tools/copy_round_evidence.sh r4
T=${1:-r4}; R=gpurun_out/$T
cpf() { [ -s "$1" ] && cp "$1" "$2" || echo "missing: $1"; }
for n in bench bench_1stream bench_720p bench_batch8 bench_rows_ragged bench_shard100k bench_shard10k bench_matrix_only bench_2ranks_one_gpu_rehearsal bench_shard10k_rccl_world1 matrix_timed_only; do cpf $R/$n.json profiles/${T}_$n.json; done
cpf $R/stats/bench_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv; cpf $R/stats1/bench1_kernel_stats.csv profiles/${T}_bench_1stream_kernel_stats.csv
cpf $R/microbench.jsonl profiles/${T}_microbench.jsonl; cpf $R/gputest.log profiles/${T}_gputest.log; cpf $R/smoke.log profiles/${T}_smoke.log
cpf $R/stage_throughput.log profiles/${T}_stage_throughput.log
mkdir -p profiles/pmc_$T
for k in fetch write sq sq2; do f=$(ls $R/pmc/$k/*counter_collection.csv 2>/dev/null | head -1); cpf "$f" profiles/pmc_$T/${k}_counter_collection.csv; done
cpf $R/pmc/summary.json profiles/pmc_$T/summary.json
echo copied

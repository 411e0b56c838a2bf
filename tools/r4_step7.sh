cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 240 python tools/exp_hamming_mfma.py > $O/exp_hamming_mfma.log 2>&1; echo "mfma rc $?"; tail -5 $O/exp_hamming_mfma.log
bash tools/collect_round_evidence.sh r4 extra

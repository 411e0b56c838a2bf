# round 4: the rebuilt few-query kernel -- exhaustive test, fuzz campaign on it, timing, then the whole GPU suite
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_match.py tests/test_gpu_fullsize.py -x -q -m gpu -k "few_query or db_scan_10k or db_match_counts" > $O/few_query_test.log 2>&1; echo "few-query tests rc $?"; tail -3 $O/few_query_test.log
timeout -k 10 400 python tests/dev/fuzz_gpu.py --seconds 240 --seed 401 --kinds db_small > $O/fuzz_r4a.log 2>&1; echo "fuzz rc $?"; tail -2 $O/fuzz_r4a.log
EXP_FORMS=0 EXP_ROUNDS=2 timeout -k 10 300 python tools/exp_small_q.py nclt-slam-project_amd/csrc/libreloc_hip.so > $O/small_q.log 2>$O/small_q.err; cat $O/small_q.log
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "suite rc $?"; tail -3 $O/gputest.log

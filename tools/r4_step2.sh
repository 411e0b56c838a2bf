# round 4: few-query scan forms -- parity against the oracle (pytest under each form), then interleaved timing
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
O=gpurun_out/${1:-r4d}; mkdir -p $O
FORMS=${2:-1}
for f in ${FORMS//,/ }; do
  RELOC_DEV=1 RELOC_SQ_FORM=$f timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_match.py -x -q -m gpu -k "db_scan_10k or db_match_counts" > $O/parity_form$f.log 2>&1; echo "form $f parity rc $?"; tail -2 $O/parity_form$f.log
done
EXP_FORMS=${FORMS} EXP_ROUNDS=2 timeout -k 10 700 python tools/exp_small_q.py nclt-slam-project_amd/csrc/libreloc_hip.so > $O/small_q_forms.log 2>$O/small_q_forms.err; echo "exp rc $?"
cat $O/small_q_forms.log

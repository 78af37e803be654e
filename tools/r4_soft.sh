#!/bin/bash
# soft ties: exactness against the oracle, exact-block counts and timings with and without
cd $GRAFT_REPO_ROOT
for sft in 1 0; do
  echo "== TEHMM_SOFT_TIES=$sft"
  export TEHMM_SOFT_TIES=$sft
  timeout -k 10 300 python tools/spec_check.py 300000 150000 5000 700001 2>&1 | grep -E "RESULT|GPU eval|MISMATCH|DIFF" | cut -c1-250
  SINGLE=1 STAGES=viterbi,both timeout -k 10 200 python tools/stage_bench.py 10 2>/dev/null | cut -c1-300
  STAGES=viterbi,both timeout -k 10 200 python tools/stage_bench.py 100 2>/dev/null | cut -c1-300
done

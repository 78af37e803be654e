#!/bin/bash
# chain parameters (row spacing, first check): config 2 (one 10 Mb interval) and the bench geometry, per library
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  echo "== $n"
  export TEHMM_HIP_LIB=$GRAFT_REPO_ROOT/tools/exp/$n.so
  timeout -k 10 200 python tools/spec_check.py 300000 150000 2>&1 | grep -E "RESULT|GPU eval" | cut -c1-220
  SINGLE=1 STAGES=viterbi,both timeout -k 10 200 python tools/stage_bench.py 10 2>/dev/null | cut -c1-260
  STAGES=viterbi timeout -k 10 200 python tools/stage_bench.py 100 2>/dev/null | cut -c1-260
done

#!/bin/bash
# round 4: the three-wave quantised pass against the oracle and against the one-wave kernel (dev build, 36 states)
cd $GRAFT_REPO_ROOT
set -o pipefail
echo "== split=1 spec_check"; TEHMM_P2_SPLIT=1 timeout -k 10 300 python tools/spec_check.py 300000 150000 5000 700001 2>&1 | tail -n 8 || exit 1
echo "== split=0 spec_check"; TEHMM_P2_SPLIT=0 timeout -k 10 300 python tools/spec_check.py 300000 150000 2>&1 | tail -n 5 || exit 1
echo "== stage split=1"; TEHMM_P2_SPLIT=1 STAGES=viterbi,both timeout -k 10 300 python tools/stage_bench.py 100 || exit 1
echo "== stage split=0"; TEHMM_P2_SPLIT=0 STAGES=viterbi,both timeout -k 10 300 python tools/stage_bench.py 100 || exit 1

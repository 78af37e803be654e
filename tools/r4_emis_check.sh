#!/bin/bash
# three-wave emission + P0 kernel: exactness, the same exact-block counts as the one-wave kernel, timings
cd $GRAFT_REPO_ROOT
set -o pipefail
echo "== split spec_check"; timeout -k 10 300 python tools/spec_check.py 300000 150000 5000 700001 2>&1 | tail -n 7 || exit 1
echo "== one-wave spec_check"; TEHMM_EMIS_SPLIT=0 timeout -k 10 300 python tools/spec_check.py 300000 150000 5000 700001 2>&1 | tail -n 7 | head -n 1 || exit 1
for d in 1 0; do echo "== stage defer $d"; TEHMM_DEFER=$d STAGES=viterbi,both timeout -k 10 300 python tools/stage_bench.py 100 2>/dev/null | cut -c1-330 || exit 1; done
echo "== stage one-wave E"; TEHMM_EMIS_SPLIT=0 STAGES=viterbi timeout -k 10 300 python tools/stage_bench.py 100 2>/dev/null | cut -c1-330 || exit 1
echo "== bench verify"; timeout -k 10 400 python bench.py --steps 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','verified','posterior_max_rel_err')})" || exit 1

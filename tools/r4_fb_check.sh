#!/bin/bash
# round 4 fused forward / backward changes: oracle check through bench.py, E-step check, stage timings
cd $GRAFT_REPO_ROOT
set -o pipefail
echo "== bench verify (mode default)"; timeout -k 10 400 python bench.py --steps 3 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','verified','posterior_max_rel_err')})" || exit 1
echo "== estep check"; timeout -k 10 300 python tools/estep_check.py 2>&1 | tail -n 4 || exit 1
for d in 1 0; do echo "== stage defer $d"; TEHMM_DEFER=$d STAGES=posterior,both timeout -k 10 300 python tools/stage_bench.py 100 2>/dev/null | cut -c1-330 || exit 1; done

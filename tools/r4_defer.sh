#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in 1 2 0 3; do echo "== defer $d"; TEHMM_DEFER=$d STAGES=both timeout -k 10 200 python tools/stage_bench.py 100 2>/dev/null | cut -c1-330 || exit 1; done

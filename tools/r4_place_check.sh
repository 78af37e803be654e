#!/bin/bash
# device-side placement: exactness against the oracle, exact-block counts against the host placement, timings
cd $GRAFT_REPO_ROOT
set -o pipefail
echo "== device placement spec_check"; timeout -k 10 300 python tools/spec_check.py 300000 150000 5000 700001 2>&1 | tail -n 7 || exit 1
echo "== host placement spec_check"; TEHMM_DEVICE_PLACE=0 timeout -k 10 300 python tools/spec_check.py 300000 150000 5000 700001 2>&1 | tail -n 7 || exit 1
for d in 1 3 0; do
echo "== stage device place, defer $d"; TEHMM_DEFER=$d STAGES=viterbi,both timeout -k 10 300 python tools/stage_bench.py 100 2>/dev/null | cut -c1-330 || exit 1
done
echo "== stage host place"; TEHMM_DEVICE_PLACE=0 STAGES=viterbi,both timeout -k 10 300 python tools/stage_bench.py 100 2>/dev/null | cut -c1-330 || exit 1

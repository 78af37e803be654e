#!/bin/bash
# kernel timelines of the bench evaluation for the TEHMM_DEFER modes given as arguments
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for d in "$@"; do
  echo "== defer $d"
  TEHMM_DEFER=$d STAGES=both timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$d -o p -- python3 $R/tools/stage_bench.py ${MB:-100} 2>/dev/null | cut -c1-150
  python3 $R/tools/timeline.py $R/gpurun_out/tl_$d 0.3
done

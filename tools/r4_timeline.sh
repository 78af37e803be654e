#!/bin/bash
# kernel timelines of the bench evaluation: r4_timeline.sh "ENV=VAL ..." ["ENV=VAL ..." ...]   (STAGES / MB inherited)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "== $cfg"
  env $cfg timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$i -o p -- python3 $R/tools/stage_bench.py ${MB:-100} 2>/dev/null | cut -c1-150
  python3 $R/tools/timeline.py $R/gpurun_out/tl_$i ${MINMS:-0.3}
done

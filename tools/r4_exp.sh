#!/bin/bash
# times kernels (rocprofv3 --kernel-trace --stats) with each experiment library in tools/exp/ (names as arguments)
# PAT: kernel regex to print; STAGES / MB as tools/stage_bench.py
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for n in "$@"; do
  echo "== $n"
  TEHMM_HIP_LIB=$R/tools/exp/$n.so STAGES=${STAGES:-viterbi} timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/exp_$n -o p -- python3 $R/tools/stage_bench.py ${MB:-100} 2>/dev/null | cut -c1-200
  python3 - "$R/gpurun_out/exp_$n" "${PAT:-k_vit_lane3|k_emis_gain}" <<'PY'
import csv, glob, re, sys
pat = re.compile(sys.argv[2])
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat.search(r["Name"]):
            print("   %-60s calls %s avg %.3f ms" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done

#!/bin/bash
# The measurements behind profiles/rNN_*: run on the GPU box from the repository root,
#   bash tools/collect_profiles.sh r03 [part ...]        parts: stats traffic estep strong stage (default: all)
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
# rocprofv3 gets `python3 ...` directly after `--` (no shell or env hop behind the profiler), counters are collected
# in their own passes with --kernel-trace only.
set -o pipefail
tag=${1:-r03}
shift
parts=${*:-stats traffic estep esteptraffic wide strong stage pmc}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp || exit 1
has() { [[ " $parts " == *" $1 "* ]]; }
B="python3 $root/bench.py"

if has stats; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_prof" -o "$tag" -- \
    $B --steps 5 --warmup 1 --no-cpu-baseline --no-extra --no-verify > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_prof.err" || exit 2
  echo "stats done"
fi
if has traffic; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/${tag}_pmc_$c" -o "$tag" -- \
      $B --mb 20 --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-verify > "$out/${tag}_pmc_$c.json" 2> "$out/${tag}_pmc_$c.err" || exit 3
  done
  python3 "$root/tools/traffic.py" "$out/${tag}_pmc_FETCH_SIZE" "$out/${tag}_pmc_WRITE_SIZE" \
    "$(python3 -c "import json;print(json.load(open('$out/${tag}_pmc_FETCH_SIZE.json'))['config']['positions_per_gpu'])")" \
    "$out/${tag}_traffic.json" || exit 4
  echo "traffic done"
fi
if has estep; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_prof_estep" -o "${tag}_estep" -- \
    $B --mode estep --mb 50 --steps 3 --warmup 1 > "$out/${tag}_estep_under_rocprof.json" 2> "$out/${tag}_prof_estep.err" || exit 5
  timeout -k 10 400 $B --mode estep --mb 50 --steps 3 --warmup 1 > "$out/${tag}_bench_estep.json" 2> "$out/${tag}_bench_estep.err" || exit 6
  echo "estep done"
fi
if has esteptraffic; then
  # HBM bytes of one EM iteration (bench.py --mode estep, 50 Mb): FETCH_SIZE / WRITE_SIZE in separate passes
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/${tag}_epmc_$c" -o "$tag" -- \
      $B --mode estep --mb 50 --steps 1 --warmup 1 --no-cpu-baseline --no-verify > "$out/${tag}_epmc_$c.json" 2> "$out/${tag}_epmc_$c.err" || exit 9
  done
  python3 "$root/tools/traffic.py" "$out/${tag}_epmc_FETCH_SIZE" "$out/${tag}_epmc_WRITE_SIZE" \
    "$(python3 -c "import json;print(json.load(open('$out/${tag}_epmc_FETCH_SIZE.json'))['config']['positions_per_gpu'])")" \
    "$out/${tag}_estep_traffic.json" || exit 10
  echo "estep traffic done"
fi
if has pmc; then
  # SQ counters of the headline kernels (three --pmc passes each; kernels serialised by the profiler)
  STAGES=both timeout -k 10 900 bash "$root/tools/pmc.sh" "${tag}" "k_vit_lane3|k_emis_gain|k_fused_fwd|k_fused_bwd|k_vit_fix|k_tb_" tools/stage_bench.py 100 > "$out/${tag}_pmc.log" 2>&1 || exit 11
  echo "pmc done"
fi
if has wide; then
  # E-step on the item-parallel passes (100 states with ratios 2 Mb, 35 states with ratios 5 Mb) and config 5
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_prof_wide" -o "${tag}_wide" -- \
    python3 "$root/tools/wide_estep_bench.py" wide ratios --no-verify > "$out/${tag}_wide_estep_under_rocprof.txt" 2> "$out/${tag}_prof_wide.err" || exit 12
  timeout -k 10 300 python3 "$root/tools/wide_estep_bench.py" wide ratios > "$out/${tag}_wide_estep.txt" 2> "$out/${tag}_wide_estep.err" || exit 13
  timeout -k 10 120 python3 "$root/tools/config5_bench.py" > "$out/${tag}_config5.txt" 2>&1 || exit 14
  echo "wide done"
fi
if has strong; then
  timeout -k 10 400 $B --scaling strong --no-extra > "$out/${tag}_bench_strong_1gpu.json" 2> "$out/${tag}_bench_strong.err" || exit 7
  echo "strong done"
fi
if has stage; then
  timeout -k 10 300 python3 "$root/tools/stage_bench.py" 100 > "$out/${tag}_stage_bench.txt" 2> "$out/${tag}_stage_bench.err" || exit 8
  echo "stage done"
fi

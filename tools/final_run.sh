#!/bin/bash
# End-of-round validation on the GPU box: the whole GPU suite, the default bench line, the strong-scaling line on one GPU,
# the 64..128-state timings.  Results under gpurun_out/.
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r3_full8.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/r3_full8.log
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
python bench.py --scaling strong --no-extra > gpurun_out/r03_bench_strong_1gpu.json 2> gpurun_out/r03_bench_strong.err; echo "strong rc=$?"
python tools/config5_bench.py > gpurun_out/r03_config5.txt 2>&1; echo "c5 rc=$?"
python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/smoke.log
bash tools/collect_profiles.sh r03 estep; echo "estep profile rc=$?"

cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r3_full6.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/r3_full6.log
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
python tools/config5_bench.py > gpurun_out/r03_config5.txt 2>&1; echo "c5 rc=$?"
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_wide -o r03_wide -- python3 $R/tools/config5_bench.py > $R/gpurun_out/r03_config5_under_rocprof.txt 2>&1; echo "c5 prof rc=$?"
cd $R; bash tools/collect_profiles.sh r03 estep; echo "estep rc=$?"

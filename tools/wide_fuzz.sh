#!/bin/bash
# Shape fuzz of the 64..128-state exact Viterbi against the oracle (tools/wide_vit_check.py): state counts, interval counts
# on both sides of the polling limit (64), tiny intervals, sticky / sparse models, other chunk sizes.  Run on the GPU box:
#   bash tools/wide_fuzz.sh      (every line must report 1 = ALL EXACT)
set -o pipefail
cd $GRAFT_REPO_ROOT
i=0
for cfg in "64 3 20000 1" "65 70 3000 1" "80 2 50000 0" "96 90 2500 0" "101 5 33333 1" "112 1 120000 1" "120 4 25000 0" "128 66 3100 1" "77 9 9000 1" "100 130 1500 1"; do
  i=$((i+1))
  SEED=$i timeout -k 10 150 python tools/wide_vit_check.py $cfg > gpurun_out/wf_$i.log 2>&1 || { echo "cfg $cfg failed rc=$?"; tail -n 5 gpurun_out/wf_$i.log; exit 1; }
  echo "cfg [$cfg] $(grep -c 'ALL EXACT' gpurun_out/wf_$i.log) $(grep 'wide_vit=1' gpurun_out/wf_$i.log | cut -c1-160)"
done
STAY=0.999 SEED=21 timeout -k 10 150 python tools/wide_vit_check.py 100 4 40000 1 > gpurun_out/wf_s1.log 2>&1; echo "sticky .999: $(grep -c 'ALL EXACT' gpurun_out/wf_s1.log) $(grep 'wide_vit=1' gpurun_out/wf_s1.log | cut -c1-160)"
SPARSE=0.9 SEED=22 timeout -k 10 150 python tools/wide_vit_check.py 100 4 40000 1 > gpurun_out/wf_s2.log 2>&1; echo "sparse .9: $(grep -c 'ALL EXACT' gpurun_out/wf_s2.log) $(grep 'wide_vit=1' gpurun_out/wf_s2.log | cut -c1-160)"
TEHMM_SPEC_CHUNK=512 SEED=23 timeout -k 10 150 python tools/wide_vit_check.py 100 4 40000 1 > gpurun_out/wf_s3.log 2>&1; echo "chunk 512: $(grep -c 'ALL EXACT' gpurun_out/wf_s3.log) $(grep 'wide_vit=1' gpurun_out/wf_s3.log | cut -c1-160)"
TEHMM_SPEC_CHUNK=128 SEED=24 timeout -k 10 150 python tools/wide_vit_check.py 100 4 40000 1 > gpurun_out/wf_s4.log 2>&1; echo "chunk 128: $(grep -c 'ALL EXACT' gpurun_out/wf_s4.log) $(grep 'wide_vit=1' gpurun_out/wf_s4.log | cut -c1-160)"
STAGES=posterior TRACKS=2,2,3,4,5,8,12,30,20,20 timeout -k 10 120 python tools/stage_bench.py 100 > gpurun_out/sb_alllds.log 2>&1; tail -n 1 gpurun_out/sb_alllds.log | cut -c1-200
STAGES=posterior TRACKS=2,2,3,4,5,8,250g,250g timeout -k 10 120 python tools/stage_bench.py 100 > gpurun_out/sb_k8.log 2>&1; tail -n 1 gpurun_out/sb_k8.log | cut -c1-200

#!/bin/bash
# SQ counters of the round-3 kernels (E-step reductions and passes; the 64..128-state path), two --pmc passes each,
# summed per kernel by tools/pmcsum.py -> gpurun_out/r03_pmc.txt.  rocprofv3 serialises the kernels in these passes, so
# GRBM_GUI_ACTIVE / 8 XCDs is each kernel's SOLO duration in cycles.
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
A="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
B="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE"
out=$R/gpurun_out/r03_pmc.txt
echo "# tools/pmc_r03.sh: SQ_* in units of 4 cycles, percentages relative to SQ_WAVE_CYCLES; GRBM_GUI_ACTIVE summed over the 8 XCDs; kernels serialised by the profiler (solo durations)" > $out
for w in estep wide; do
  if [ $w = estep ]; then cmd="$R/tools/estep_bench.py 50"; pat="k_estep|k_fused"; else cmd="$R/tools/config5_bench.py"; pat="k_vit_wide|k_wide"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $R/gpurun_out/r03_pmc_${w}A -o p -- python3 $cmd > $R/gpurun_out/r03_pmc_${w}A.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $R/gpurun_out/r03_pmc_${w}B -o p -- python3 $cmd > $R/gpurun_out/r03_pmc_${w}B.log 2>&1 || exit 2
  echo "## $w: python3 tools/$(basename ${cmd%% *}) ${cmd#* } -- pass A" >> $out
  python3 $R/tools/pmcsum.py $R/gpurun_out/r03_pmc_${w}A "$pat" >> $out
  echo "## $w -- pass B" >> $out
  python3 $R/tools/pmcsum.py $R/gpurun_out/r03_pmc_${w}B "$pat" >> $out
done
tail -n 60 $out

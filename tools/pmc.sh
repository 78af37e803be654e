#!/bin/bash
# SQ counters per kernel, two --pmc passes (rocprofv3 serialises kernels in these passes: solo durations).
#   bash tools/pmc.sh TAG "KERNEL_REGEX" script.py [args...]     (environment is inherited by the profiled program)
# -> gpurun_out/TAG_pmc.txt.  SQ_* in units of 4 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
tag=$1; pat=$2; shift 2
A="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
B="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQC_ICACHE_BUSY_CYCLES"
out=$R/gpurun_out/${tag}_pmc.txt
echo "# tools/pmc.sh $tag '$pat' $*: SQ_* in units of 4 cycles, percentages relative to SQ_WAVE_CYCLES; GRBM_GUI_ACTIVE summed over 8 XCDs; kernels serialised" > $out
for p in A B C; do
  eval "ctr=\$$p"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/${tag}_pmc_$p -o p -- python3 $R/"$@" > $R/gpurun_out/${tag}_pmc_$p.log 2>&1 || { echo "pass $p failed"; tail -n 5 $R/gpurun_out/${tag}_pmc_$p.log; exit 1; }
  echo "## pass $p" >> $out
  python3 $R/tools/pmcsum.py $R/gpurun_out/${tag}_pmc_$p "$pat" >> $out
done
cat $out

#!/bin/bash
# SQ counters of the attention kernels at one shape (gpurun; outputs gpurun_out/r04/attn_pmc_<tag>_*.txt).  usage: run_attn_pmc.sh TAG H Nq Nk DQK DV
set -o pipefail
R=$PWD; mkdir -p gpurun_out/r04
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
P=1
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  rm -rf $R/gpurun_out/r04/attn_pmc_${TAG}_$P
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/r04/attn_pmc_${TAG}_$P -o pmc -- python $R/tools/attn_one.py "$@" > $R/gpurun_out/r04/attn_pmc_${TAG}_$P.log 2>&1 || { grep -v "^W2026\|^    @" $R/gpurun_out/r04/attn_pmc_${TAG}_$P.log | tail -5; }
  python $R/tools/pmc_table.py $R/gpurun_out/r04/attn_pmc_${TAG}_$P attn > $R/gpurun_out/r04/attn_pmc_${TAG}_$P.txt 2>&1
  rm -rf $R/gpurun_out/r04/attn_pmc_${TAG}_$P
  P=$((P+1))
done

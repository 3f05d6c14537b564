#!/bin/bash
# SQ counters of the pool-branch kernels at one block geometry (gpurun; outputs gpurun_out/pool_pmc_*.txt)
set -o pipefail
R=$PWD; mkdir -p gpurun_out
BLK=${1:-blk4-13}
cd /tmp && export TMPDIR=/tmp
P=1
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  rm -rf $R/gpurun_out/pool_pmc_$P
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pool_pmc_$P -o pmc -- python $R/tools/pool_bench.py $BLK > $R/gpurun_out/pool_pmc_$P.log 2>&1 || { grep -v "^W2026\|^    @" $R/gpurun_out/pool_pmc_$P.log | tail -5; }
  python $R/tools/pmc_table.py $R/gpurun_out/pool_pmc_$P > $R/gpurun_out/pool_pmc_$P.txt 2>&1
  rm -rf $R/gpurun_out/pool_pmc_$P
  P=$((P+1))
done

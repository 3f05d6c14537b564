#!/bin/bash
# per-kernel breakdown of single MViT blocks (gpurun): gpurun_out/blk<I>_kernels.txt
set -o pipefail
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for I in "$@"; do
  rm -rf $R/gpurun_out/blkprof_$I
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/blkprof_$I -o b -- python $R/tools/block_times.py 32 --only $I 4 > $R/gpurun_out/blkprof_$I.log 2>&1 || { tail -5 $R/gpurun_out/blkprof_$I.log; exit 1; }
  python $R/tools/prof_summary.py $R/gpurun_out/blkprof_$I 4 60 > $R/gpurun_out/blk${I}_kernels.txt 2>&1
  rm -rf $R/gpurun_out/blkprof_$I
done

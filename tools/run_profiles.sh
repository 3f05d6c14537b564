#!/bin/bash
# Round profile set on the GPU box (run from the repo root through gpurun): bench line, kernel-trace stats, PMC traffic.
# usage: tools/run_profiles.sh TAG   -> gpurun_out/TAG_*  (copy what should be judged into profiles/)
set -o pipefail
TAG=${1:-r02}
R=$PWD
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || exit 1
cat gpurun_out/${TAG}_bench_default.json
cd /tmp && export TMPDIR=/tmp
# 18 steps in that run: 5 warm-up + 10 timed + the 3 single steps bench.py launches into an empty queue for its host-time line
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_prof.log 2>&1 || exit 2
python $R/tools/prof_summary.py $R/gpurun_out/${TAG}_prof 18 80 > $R/gpurun_out/${TAG}_per_grid_summary.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/${TAG}_pmc_$C -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1 || exit 3
done
python $R/tools/pmc_traffic.py $R/gpurun_out/${TAG}_pmc_FETCH_SIZE $R/gpurun_out/${TAG}_pmc_WRITE_SIZE $R/gpurun_out/${TAG}_pmc_gemm_traffic.json bf16 8 4 16 224 attention > $R/gpurun_out/${TAG}_pmc_summary.txt 2>&1
head -3 $R/gpurun_out/${TAG}_pmc_summary.txt
rm -rf $R/gpurun_out/${TAG}_pmc_FETCH_SIZE $R/gpurun_out/${TAG}_pmc_WRITE_SIZE   # raw counter dumps are large; the summaries stay

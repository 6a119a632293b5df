#!/bin/bash
# rocprofv3 passes for the round-1 profile (run on the GPU box from the repo root)
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r${ROUND:-01}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/e2e_bench.json 2> $OUT/e2e.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kern -- python3 bench.py --kernel-only --no-cpu-baseline > $OUT/kern_bench.json 2> $OUT/kern.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --kernel-only --no-cpu-baseline --roofline-launches 24 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --kernel-only --no-cpu-baseline --roofline-launches 24 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
rm -f $OUT/e2e/*/*_kernel_trace.csv      # 35 MB of per-dispatch rows; the stats file is what is kept
find $OUT -name "*.csv" | xargs ls -la | awk '{print $5, $9}'

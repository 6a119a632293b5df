#!/bin/bash
# rocprofv3 passes for one round's profile (run on the GPU box from the repo root):  ROUND=02 bash tools/profile_round.sh
# counters in their own passes (no trace domains besides --kernel-trace), as MI355X_MICROARCH.md prescribes.
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r${ROUND:-02}
mkdir -p $OUT
NHMC_PROFILE_MARK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-by-deg --no-full-run > $OUT/e2e_bench.json 2> $OUT/e2e.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kern -- python3 bench.py --kernel-only --no-cpu-baseline > $OUT/kern_bench.json 2> $OUT/kern.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --kernel-only --no-cpu-baseline --roofline-launches 24 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --kernel-only --no-cpu-baseline --roofline-launches 24 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
# condense on the box (the e2e kernel trace is tens of MB), then drop the per-dispatch rows
mkdir -p $OUT/summary
cp -r profiles $OUT/summary/profiles_work
NHMC_PROFILE_STEPS=2 python3 profiles/summarize.py $OUT r${ROUND:-02}
cp profiles/r${ROUND:-02}_* profiles/traffic_leapfrog.json $OUT/summary/ 2>/dev/null
rm -rf $OUT/summary/profiles_work
rm -f $OUT/e2e/*/*_kernel_trace.csv $OUT/pmc_*/*/*_kernel_trace.csv
find $OUT -name "*.csv" | xargs ls -la | awk '{print $5, $9}'

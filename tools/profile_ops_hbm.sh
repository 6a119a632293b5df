#!/bin/bash
# HBM traffic of the operator data-term kernels (tools/ops_bench.py 64): FETCH_SIZE and WRITE_SIZE in their own --pmc passes,
# --kernel-trace only (MI355X_MICROARCH.md).  ROUND=03 bash tools/profile_ops_hbm.sh ; python profiles/summarize.py gpurun_out/prof_opshbm_r03 r03ops
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_opshbm_r${ROUND:-03}
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 tools/ops_bench.py 64 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 tools/ops_bench.py 64 > $OUT/pmc_write.log 2>&1
rm -f $OUT/pmc_*/*/*_kernel_trace.csv
find $OUT -name "*.csv" | xargs ls -la | awk '{print $5, $9}'

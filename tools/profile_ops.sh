#!/bin/bash
# rocprofv3 passes over the operator data terms (tools/ops_bench.py): kernel stats, LDS conflict counters.
set -x
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_ops_r${ROUND:-01}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/ops_bench.py 64 > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds -- python3 tools/ops_bench.py 16 > $OUT/pmc_lds.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 tools/ops_bench.py 16 > $OUT/pmc_mfma.log 2>&1
find $OUT -name "*.csv" | xargs ls -la | awk '{print $5, $9}'

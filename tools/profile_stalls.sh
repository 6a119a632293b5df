#!/bin/bash
# Stall attribution of the spectral pair kernel (k_pair256) and the fused SR kernel: SQ counter passes over
# tools/pair_bench.py / tools/ops_bench.py (counters in their own rocprofv3 passes, --kernel-trace only), condensed by
# tools/stall_table.py into profiles/rNN_pair256_stalls.csv.     ROUND=03 bash tools/profile_stalls.sh
set -x
export TMPDIR=/tmp
# BENCH=tools/ops_bench.py NAME=ops : the same passes over the streaming data-term kernels -> profiles/rNN_ops_stalls.csv
BENCH=${BENCH:-tools/pair_bench.py}
NAME=${NAME:-pair256}
OUT=$PWD/gpurun_out/prof_stalls_${NAME}_r${ROUND:-03}
mkdir -p $OUT
B=${B:-64}
rocprofv3 -L > $OUT/counters_available.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $OUT/pass1 -- python3 $BENCH $B > $OUT/pass1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $OUT/pass2 -- python3 $BENCH $B > $OUT/pass2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM \
  --kernel-trace --output-format csv -d $OUT/pass3 -- python3 $BENCH $B > $OUT/pass3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/pass4 -- python3 $BENCH $B > $OUT/pass4.log 2>&1
python3 tools/stall_table.py $OUT r${ROUND:-03} $NAME
rm -f $OUT/pass*/*/*_kernel_trace.csv
find $OUT -name "*.csv" | xargs ls -la | awk '{print $5, $9}'

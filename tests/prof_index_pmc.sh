#!/bin/bash
# SQ counters of the index kernels (tests/bench_index.py) for one configuration, one counter set per pass.
#   tests/prof_index_pmc.sh <outdir-under-gpurun_out> <config> [size-mib]
OUT=$PWD/gpurun_out/$1; CFG=$2; SZ=${3:-256}; R=$PWD
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
run() { timeout -s KILL 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $R/tests/bench_index.py --config $CFG --size-mib $SZ > $OUT/$1.log 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
run b "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD"
cd $R; python3 tests/pmc_summary.py "$OUT/*/runc/*counter_collection.csv" > $OUT/summary_$CFG.txt 2>&1

#!/bin/bash
# SQ counters of the region index's kernels (tests/bench_index.py under rocprofv3 --pmc, separate passes)
#   tests/prof_regions_pmc.sh <outdir-under-gpurun_out> <config> <size-mib>
OUT=$PWD/gpurun_out/$1; CFG=$2; SZ=$3; R=$PWD
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
run() { timeout -s KILL 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $R/tests/bench_index.py --config $CFG --size-mib $SZ > $OUT/$1.log 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
run b "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"
run c "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE"
cd $R; python3 tests/pmc_summary.py "$OUT/*/*/*counter_collection.csv" > $OUT/summary.txt; grep -A30 "k_rg_walk\|k_rg_guess" $OUT/summary.txt | head -80

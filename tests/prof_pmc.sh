#!/bin/bash
# usage: tests/prof_pmc.sh <outdir-under-gpurun_out> [size-mib]   (runs on the GPU box from the repo root)
OUT=$PWD/gpurun_out/$1; SZ=${2:-1024}; R=$PWD
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
run() { timeout -s KILL 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $R/bench.py --size-mib $SZ --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$1.log 2>&1; }
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
cd $R; python3 tests/pmc_summary.py "$OUT/*/runc/*counter_collection.csv" > $OUT/summary.txt

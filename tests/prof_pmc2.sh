#!/bin/bash
# second counter set: lane utilisation, memory pipeline stalls (runs on the GPU box from the repo root)
OUT=$PWD/gpurun_out/$1; SZ=${2:-1024}; R=$PWD
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
run() { timeout -s KILL 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $R/bench.py --size-mib $SZ --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/$1.log 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
run b "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT64"
run c "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_LDS_IDX_ACTIVE"
cd $R; python3 tests/pmc_summary.py "$OUT/*/runc/*counter_collection.csv" > $OUT/summary.txt

#!/bin/bash
# memory-pipeline counters of the aec kernels (runs on the GPU box from the repo root); few counters
# per pass, every pass bounded by `timeout` (a rejected counter set makes rocprofv3 hang in its exit path)
OUT=$PWD/gpurun_out/$1; SZ=${2:-1024}; R=$PWD
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
run() { timeout -s KILL 150 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 $R/bench.py --size-mib $SZ --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$1.log 2>&1 || echo "pass $1 failed"; }
run a "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"
run b "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
run c "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
run d "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
run e "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum"
run f "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
cd $R; python3 tests/pmc_summary.py "$OUT/*/runc/*counter_collection.csv" > $OUT/summary.txt; cat $OUT/summary.txt | grep -v "^k_scan\|^k_dec_result\|^k_seg" | head -90

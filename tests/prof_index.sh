#!/bin/bash
# rocprofv3 kernel statistics of the bare-stream index pass (tests/bench_index.py) per configuration.
#   tests/prof_index.sh <outdir-under-gpurun_out> [size-mib]
OUT=$PWD/gpurun_out/$1; SZ=${2:-1024}; R=$PWD
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  for c in c2 c5 c3 typical; do
    s=$SZ; [ $c = typical ] && s=$((SZ / 4))
    timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/idx_$c -- python3 $R/tests/bench_index.py --config $c --size-mib $s > $OUT/bench_index_$c.txt 2>&1
    f=$(find $OUT/idx_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_index_$c.csv
    rm -rf $OUT/idx_$c
  done )

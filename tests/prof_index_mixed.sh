#!/bin/bash
# rocprofv3 kernel statistics of the index pass on mixed content (tests/bench_index_mixed.py), one noise level per run.
#   tests/prof_index_mixed.sh <outdir-under-gpurun_out> [size-mib] [permille ...]
OUT=$PWD/gpurun_out/$1; SZ=${2:-1024}; R=$PWD; shift; shift
PMS=${@:-"10 50"}
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  for pm in $PMS; do
    timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mix_$pm -- python3 $R/tests/bench_index_mixed.py --size-mib $SZ --noise-permille $pm > $OUT/bench_index_mixed_$pm.txt 2>&1
    f=$(find $OUT/mix_$pm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_index_mixed_$pm.csv
    rm -rf $OUT/mix_$pm
  done )

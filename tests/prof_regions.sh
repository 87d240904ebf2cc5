#!/bin/bash
# The region index on the GPU box: index-pass times per configuration (product library), the scheme's own statistics
# (tuning library, AEC_IDX_STATS), and the kernel statistics of one configuration under rocprofv3.
#   tests/prof_regions.sh <outdir-under-gpurun_out> [size-mib]
OUT=$PWD/gpurun_out/$1; SZ=${2:-1024}; R=$PWD
mkdir -p $OUT
for c in c2 c5 c3 typical; do
  s="64 $SZ"; [ $c = typical ] && s="64 $((SZ / 4))"
  timeout -s KILL 600 python3 tests/bench_index.py --config $c --size-mib $s > $OUT/bench_index_$c.txt 2>&1
  AEC_AMD_LIB=$R/libaec_amd/lib/tuning/libaec.so.0 AEC_IDX_STATS=1 timeout -s KILL 600 python3 tests/bench_index.py --config $c --size-mib $SZ > $OUT/stats_index_$c.txt 2>&1
done
( cd /tmp && export TMPDIR=/tmp
  for c in c2 c3; do
    timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/idx_$c -- python3 $R/tests/bench_index.py --config $c --size-mib $SZ > $OUT/prof_index_$c.txt 2>&1
    f=$(find $OUT/idx_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_index_$c.csv
    rm -rf $OUT/idx_$c
  done )
tail -n 3 $OUT/bench_index_*.txt $OUT/stats_index_*.txt

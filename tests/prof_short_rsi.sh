# rocprofv3 kernel statistics of aec_buffer_decode on streams with short RSIs (tests/bench_short_rsi.py)
#   tests/prof_short_rsi.sh <outdir-under-gpurun_out> [size-mib]
O=$PWD/gpurun_out/$1; SZ=${2:-16}; R=$PWD
mkdir -p $O
( cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/tests/bench_short_rsi.py --size-mib $SZ > $O/bench_short_rsi_under_rocprof.txt 2>&1
  f=$(find $O/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_short_rsi.csv
  rm -rf $O/tr )
head -14 $O/kernel_stats_short_rsi.csv | cut -c1-160

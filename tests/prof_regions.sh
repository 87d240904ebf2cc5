#!/bin/bash
# The region index on the GPU box: index-pass times per configuration (product library), the scheme's own statistics
# (tuning library, AEC_IDX_STATS), and the kernel statistics of two configurations under rocprofv3.
#   tests/prof_regions.sh <outdir-under-gpurun_out> [size-mib]
OUT=$PWD/gpurun_out/$1; SZ=${2:-1024}; R=$PWD
mkdir -p $OUT
for c in c2 c5 c3 typical; do
  s="$SZ"; [ $c = c3 ] && s="$SZ $((SZ * 4))"
  timeout -s KILL 900 python3 tests/bench_index.py --config $c --size-mib $s > $OUT/bench_index_$c.txt 2>&1
  s2=$SZ; [ $c = c3 ] && s2=$((SZ * 4))
  AEC_AMD_LIB=$R/libaec_amd/lib/tuning/libaec.so.0 AEC_IDX_STATS=1 timeout -s KILL 900 python3 tests/bench_index.py --config $c --size-mib $s2 > $OUT/stats_index_$c.txt 2>&1
done
( cd /tmp && export TMPDIR=/tmp
  for c in c2 c3; do
    s2=$SZ; [ $c = c3 ] && s2=$((SZ * 4))
    timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/idx_$c -- python3 $R/tests/bench_index.py --config $c --size-mib $s2 > $OUT/prof_index_$c.txt 2>&1
    f=$(find $OUT/idx_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_index_$c.csv
    rm -rf $OUT/idx_$c
  done )
grep -h "index \|regions:" $OUT/bench_index_*.txt $OUT/stats_index_*.txt | grep -v "^==" | cut -c1-420
python3 - $OUT <<'PY'
import csv,re,sys
for c in ("c2","c3"):
    try: rows=list(csv.DictReader(open(f"{sys.argv[1]}/kernel_stats_index_{c}.csv")))
    except Exception as e: print(c, e); continue
    print(c)
    for r in rows[:8]:
        m=re.search(r'(k_\w+)', r["Name"]); n=m.group(1) if m else r["Name"][:30]
        print("  %-16s calls %5s total/3 %9.3f ms avg %9.3f us" % (n, r["Calls"], float(r["TotalDurationNs"])/3e6, float(r["AverageNs"])/1e3))
PY

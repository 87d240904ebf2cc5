#!/bin/bash
# kernel times of the region index under rocprofv3 for a few settings of the tuning library's knobs
#   tests/prof_regions_sweep.sh <outdir-under-gpurun_out> <config> <size-mib> "VAR=val VAR=val" ...
OUT=$PWD/gpurun_out/$1; CFG=$2; SZ=$3; R=$PWD; shift 3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for setting in "$@"; do
  i=$((i+1))
  ( export $setting AEC_AMD_LIB=$R/libaec_amd/lib/tuning/libaec.so.0
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run_$i -- python3 $R/tests/bench_index.py --config $CFG --size-mib $SZ > $OUT/run_$i.txt 2>&1 )
  f=$(find $OUT/run_$i -name "*kernel_stats.csv" | head -1)
  echo "== $setting"; grep "index " $OUT/run_$i.txt | tail -1
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    if 'k_rg' in r["Name"]:
        print("   %-12s calls %4s avg %9.1f us" % (r["Name"].split("::")[-1].split("(")[0], r["Calls"], float(r["AverageNs"])/1e3))
PY
  rm -rf $OUT/run_$i
done

#!/bin/bash
# per-dispatch kernel durations of one index pass (tests/bench_index.py), in launch order.
#   tests/prof_index_trace.sh <outdir-under-gpurun_out> <config> [size-mib] [--decode]
OUT=$PWD/gpurun_out/$1; CFG=$2; SZ=${3:-1024}; EXTRA=$4; R=$PWD
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$CFG -- python3 $R/tests/bench_index.py --config $CFG --size-mib $SZ $EXTRA > $OUT/bench_index_$CFG.txt 2>&1 )
f=$(find $OUT/tr_$CFG -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $OUT/trace_$CFG.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows:
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    if not m: continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None or s - t0 > 50_000_000: print("----"); t0 = s
    print("%-22s start %9.1f us  dur %9.1f us  grid %s" % (m.group(1), (s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size", "")))
    
PY
rm -rf $OUT/tr_$CFG

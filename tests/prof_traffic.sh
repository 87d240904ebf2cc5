#!/bin/bash
# HBM traffic of every aec kernel on the bench workload (default 4096 MiB), one counter per pass as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Run on the GPU box from the repo root:
#   tests/prof_traffic.sh <outdir-under-gpurun_out> [size-mib] [config]
# writes <outdir>/traffic.json, to be committed as profiles/<round>/traffic_<config>_<size>.json
OUT=$PWD/gpurun_out/$1; SZ=${2:-4096}; CFG=${3:-c2}; R=$PWD
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/bench.py --config $CFG --size-mib $SZ --steps 2 --warmup 0 --no-cpu-baseline --no-extras > $OUT/$c.log 2>&1
done
cd $R; python3 - "$OUT" "$SZ" "$CFG" <<'PY'
import sys, json, glob
sys.path.insert(0, 'tests')
from pmc_summary import summarise
out, sz, cfg = sys.argv[1], int(sys.argv[2]), sys.argv[3]
res = summarise(glob.glob(out + '/*/runc/*counter_collection.csv'))
kernels = sorted({k for k, _ in res})
doc = {"config": cfg, "size_mib": sz, "unit": "bytes per launch", "note": "FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE of the wide coalesced input "
       "streams (k_analyze, k_pack) is doubled per the gfx950 correction; k_decode's narrow loads are uncorrected (lower bound)", "kernels": {}}
for k in kernels:
    name = k.split('<')[0]
    f = res.get((k, 'FETCH_SIZE'), 0.0) * 1024
    w = res.get((k, 'WRITE_SIZE'), 0.0) * 1024
    corr = 2.0 if name in ('k_analyze', 'k_pack') else 1.0
    doc["kernels"][name] = {"fetch_raw": int(f), "fetch": int(f * corr), "write": int(w), "traffic": int(f * corr + w)}
json.dump(doc, open(out + '/traffic.json', 'w'), indent=1)
print(json.dumps(doc["kernels"], indent=1))
PY

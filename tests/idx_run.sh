timeout 300 python tests/bench_index.py --config c2 --size-mib 1 64 1024 2>&1 | tail -3
timeout 300 python tests/bench_index.py --config c5 --size-mib 64 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/idxprof3 -- python3 $GRAFT_REPO_ROOT/tests/bench_index.py --config c2 --size-mib 1024 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv,glob,re
f=sorted(glob.glob("gpurun_out/idxprof3/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    m=re.search(r"(k_\w+)", r["Name"])
    if m and m.group(1) in ("k_spec","k_index","k_expand"): print(m.group(1), r["Calls"], round(float(r["AverageNs"])/1e3,1),"us")
PY

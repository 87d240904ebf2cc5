# HIP API trace of small aec_buffer_decode calls (run on the GPU box): which host-side calls a 64 KiB decode is made of
cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out/$1; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/trace -- python3 $R/tests/bench_abi_small.py --chunk-kib 64 --reps 30 > $O/out.txt 2>&1
f=$(find $O -name "*hip_api_trace.csv" | head -1)
python3 - "$f" <<'PY' > $O/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last decode calls: find hipStreamSynchronize events; print the API calls between the last few of them
syncs = [i for i, r in enumerate(rows) if r["Function"] == "hipStreamSynchronize"]
for k in (-3, -2):
    a, b = syncs[k - 1], syncs[k]
    base = int(rows[a]["End_Timestamp"])
    print("---- between two synchronisations")
    for r in rows[a:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - base) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {r['Function']}")
PY
head -120 $O/summary.txt

# kernel trace of small aec_buffer_decode calls with short RSIs (run on the GPU box): which kernels a 64 KiB decode is made of
#   prof_small_rsi.sh <outdir> <rsi>
cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out/$1; mkdir -p $O; R=$PWD
( cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/tests/bench_short_rsi.py --size-kib 64 --rsi $2 > $O/out.txt 2>&1 )
f=$(find $O -name "*kernel_trace.csv" | head -1); m=$(find $O -name "*memory_copy_trace.csv" | head -1)
python3 - "$f" "$m" <<'PY' > $O/summary.txt
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("aec::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
try:
    for r in csv.DictReader(open(sys.argv[2])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", "") ))
except Exception as e:
    print("no copy trace", e)
rows.sort()
fin = [i for i, r in enumerate(rows) if r[2].startswith("k_small_finish")]
for which in (len(fin) // 3, 2 * len(fin) // 3):
    i = fin[which]
    j0 = i
    while j0 > 0 and not rows[j0][2].startswith("k_small_parse"): j0 -= 1
    j0 = max(0, j0 - 3)
    base = rows[j0][0]
    print("---- call", which)
    for (s, e, n) in rows[j0:i + 9]:
        print(f"{(s - base) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n[:60]}")
PY
cat $O/summary.txt | head -70

# kernel trace of small aec_buffer_decode calls (run on the GPU box): which kernels a 64 KiB decode is made of
cd $GRAFT_REPO_ROOT; O=$PWD/gpurun_out/$1; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tests/bench_abi_small.py > $O/out.txt 2>&1
f=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' > $O/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40 kernels of the 64 KiB decode phase: find decode kernels with tiny grids
out = []
for i, r in enumerate(rows):
    name = r["Kernel_Name"].split("(")[0].replace("aec::(anonymous namespace)::", "").replace("void ", "")
    out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
# print a window of kernels around each of the first few k_decode launches following a k_index
cnt = 0
for i, (s, e, n, g, w) in enumerate(out):
    if (n.startswith("k_decode<") or n.startswith("k_decode_wave<")) and cnt < 40:
        cnt += 1
        if cnt in (5, 25, 38):
            j0 = max(0, i - 9)
            base = out[j0][0]
            print("---- window", cnt)
            for (s2, e2, n2, g2, w2) in out[j0:i + 3]:
                print(f"{(s2 - base) / 1e3:9.1f} us  +{(e2 - s2) / 1e3:7.1f} us  {n2[:60]}  grid {g2} wg {w2}")
PY
cat $O/summary.txt | head -80

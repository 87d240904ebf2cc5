# rocprofv3 kernel statistics of ONE shape of tests/bench_sz_chunks.py: prof_sz_one.sh <outdir> <chunks> <chunk KiB> <scan line>
O=$PWD/gpurun_out/$1; R=$PWD; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/tests/bench_sz_chunks.py --only $2 $3 $4 > $O/out.txt 2>&1
  f=$(find $O/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv; rm -rf $O/tr )
grep -v amdgpu $O/out.txt | grep "compress" | head -6
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    n = r["Name"].replace("aec::(anonymous namespace)::", "").split("(")[0]
    print(f'{n[:38]:38s} calls {r["Calls"]:>6s} total {int(r["TotalDurationNs"]) / 1e6:9.3f} ms avg {float(r["AverageNs"]) / 1e3:9.1f} us min {int(r["MinNs"]) / 1e3:9.1f} us max {int(r["MaxNs"]) / 1e3:9.1f} us')
PY

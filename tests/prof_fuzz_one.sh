# rocprofv3 kernel statistics of ONE case of tests/fuzz_index_gpu.py: prof_fuzz_one.sh <outdir> <seed> <case>
O=$PWD/gpurun_out/$1; R=$PWD; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp
  timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/tests/fuzz_index_gpu.py --cases 300 --seed $2 --only $3 --time > $O/out.txt 2>&1
  f=$(find $O/tr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv; rm -rf $O/tr )
grep "case $3" $O/out.txt | tail -1
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    n = r["Name"].replace("aec::(anonymous namespace)::", "").split("(")[0]
    print(f'{n[:38]:38s} calls {r["Calls"]:>5s} total {int(r["TotalDurationNs"]) / 1e6:9.3f} ms avg {float(r["AverageNs"]) / 1e3:9.1f} us min {int(r["MinNs"]) / 1e3:9.1f} us max {int(r["MaxNs"]) / 1e3:9.1f} us')
PY

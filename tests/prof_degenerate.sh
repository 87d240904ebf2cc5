#!/bin/bash
# GPU box: the degenerate inputs' table, and a kernel timeline of one decode per slow shape (tests/bench_degenerate.py)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/deg
python tests/bench_degenerate.py > gpurun_out/deg/bench_degenerate.txt 2>&1
cat gpurun_out/deg/bench_degenerate.txt
for shape in ${SHAPES:-noise ramp}; do
rocprofv3 --kernel-trace -d gpurun_out/deg/prof -o $shape -- python3 tests/bench_degenerate.py --only $shape > gpurun_out/deg/$shape.log 2>&1
python - $shape <<'PY'
import sqlite3, re, sys
db = sqlite3.connect(f'gpurun_out/deg/prof/{sys.argv[1]}_results.db')
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
t0 = rows[0][1]
prev = None
out = open(f'gpurun_out/deg/{sys.argv[1]}_timeline.txt', 'w')
for n, s, e in rows:
    m = re.search(r'(k_\w+?)(I|E)', n)
    gap = (s - prev) / 1e6 if prev else 0
    if (e - s) > 50e3 or gap > 1:
        print(f"{(s - t0) / 1e6:10.3f} ms  gap {gap:8.3f}  {(e - s) / 1e3:10.1f} us  {m.group(1) if m else n[:30]}", file=out)
    prev = e
PY
done

#!/usr/bin/env python3
"""Prints the kernels around a few decode launches of a rocprofv3 --kernel-trace csv (tests/prof_small.sh)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def nm(r):
    n = r["Kernel_Name"].replace("aec::(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)


out = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r), r.get("Grid_Size", ""), r.get("Workgroup_Size", "")) for r in rows]
idxs = [i for i, o in enumerate(out) if o[2].startswith("k_decode<") or o[2].startswith("k_decode_wave<")]
picks = [int(x) for x in sys.argv[2:]] or [5, 25, 45]
for k in picks:
    if k < len(idxs):
        i = idxs[k]
        j0 = max(0, i - 10)
        base = out[j0][0]
        print("---- window", k)
        for (s2, e2, n2, g2, w2) in out[j0:i + 4]:
            print(f"{(s2 - base) / 1e3:9.1f} us  +{(e2 - s2) / 1e3:7.1f} us  {n2[:70]}  grid {g2} wg {w2}")

"""Summarise rocprofv3 counter_collection.csv files: mean counter value per aec kernel."""
import collections
import csv
import glob
import sys


def summarise(paths):
    agg = collections.defaultdict(list)
    for f in paths:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "aec::" not in k:
                continue
            short = k.split("aec::(anonymous namespace)::")[1].split("(")[0]
            agg[(short, row["Counter_Name"])].append(float(row["Counter_Value"]))
    return {kc: sum(v) / len(v) for kc, v in agg.items()}


if __name__ == "__main__":
    paths = []
    for a in sys.argv[1:]:
        paths += glob.glob(a, recursive=True)
    res = summarise(paths)
    kernels = sorted({k for k, _ in res})
    counters = sorted({c for _, c in res})
    for k in kernels:
        print(k)
        for c in counters:
            if (k, c) in res:
                print(f"   {c:28s} {res[(k, c)]:18.1f}")

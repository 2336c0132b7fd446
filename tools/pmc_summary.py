"""Aggregates a rocprofv3 --pmc counter_collection.csv by kernel: mean of every counter per dispatch.
usage: python tools/pmc_summary.py <dir or csv> [name-substring ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    pats = sys.argv[2:]
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if pats and not any(p in k for p in pats):
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
    for k in sorted(acc):
        print(k[:110])
        for c in sorted(acc[k]):
            print(f"    {c:32s} {acc[k][c] / cnt[k][c]:16.1f}   (n={cnt[k][c]})")


if __name__ == "__main__":
    main()

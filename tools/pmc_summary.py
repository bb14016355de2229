#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per-kernel mean counter values,
and per-environment figures for the wildfire step kernel."""
import collections
import csv
import glob
import sys

path = sys.argv[1]
envs = float(sys.argv[2]) if len(sys.argv) > 2 else 65536.0
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "wf_" not in k:
        continue
    print(k[:90])
    for c, vals in sorted(v.items()):
        m = sum(vals) / len(vals)
        print(f"    {c:24s} n={len(vals):3d} mean={m:14.1f}  per_env={m / envs:10.2f}")

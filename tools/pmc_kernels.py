#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv by kernel family: mean counter value per dispatch.
    python tools/pmc_kernels.py <counter_collection.csv> [substring ...]"""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].split("(")[0]
        rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
subs = sys.argv[2:]
for k, cs in sorted(rows.items(), key=lambda kv: -sum(len(v) for v in kv[1].values())):
    if subs and not any(s in k for s in subs):
        continue
    n = max(len(v) for v in cs.values())
    print(f"{k[:90]}  dispatches {n}")
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} mean {sum(v) / len(v):14.1f}")

#!/usr/bin/env python3
"""Timeline of ONE steady step from a rocprofv3 --kernel-trace CSV: start / end of every kernel relative to the step's first launch,
with the stream it ran on (which lane of the plan).  python tools/step_timeline.py <kernel_trace.csv> [step index from the end = 2]"""
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rtd::", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
# a step starts at the uint8 stem kernel
starts = [i for i, r in enumerate(rows) if "stem0_u8" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
i0, i1 = starts[-k - 1], starts[-k]
t0 = rows[i0][0]
prev_end = {}
for s, e, n, q in rows[i0:i1]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:7.1f} us  q{q}  {n[:70]}")
print("step span us", (rows[i1 - 1][1] - t0) / 1e3, " next step starts at", (rows[i1][0] - t0) / 1e3)

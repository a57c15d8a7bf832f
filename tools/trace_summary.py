"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) median/min/mean duration in us.
Usage: python tools/trace_summary.py <kernel_trace.csv> [name-substring ...]"""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
pats = sys.argv[2:]
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if pats and not any(p in n for p in pats):
        continue
    agg[(n[:60], r["Grid_Size_X"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(f"{k[0]:60s} grid={k[1]:>9s} wg={k[2]:>4s} n={len(v):6d} median={v[len(v)//2]/1e3:9.2f}us min={v[0]/1e3:9.2f} mean={sum(v)/len(v)/1e3:9.2f} total={sum(v)/1e6:9.2f}ms")

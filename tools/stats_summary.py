"""Summarise a rocprofv3 *_kernel_stats.csv. Usage: python tools/stats_summary.py <csv> [n_diffusion_steps] [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms ; per diffusion step {tot/1e6/nsteps:.3f} ms")
for r in rows[:top]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>7s} avg_us={float(r['AverageNs'])/1e3:9.1f} "
          f"per_step_ms={int(r['TotalDurationNs'])/1e6/nsteps:7.3f} pct={r['Percentage']}")

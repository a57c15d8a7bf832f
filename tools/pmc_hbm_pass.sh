#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the kernels whose name contains one of <substrings> (comma separated), separate --pmc passes with
# --kernel-trace only (MI355X_MICROARCH.md, HBM section: FETCH_SIZE x 2 on gfx950): tools/pmc_hbm_pass.sh <substrings> <python script> [args...]
cd /tmp && export TMPDIR=/tmp
KN=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmch_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmch_$c -- python3 "$@" > /tmp/pmch_$c.log 2>&1
  python3 - "$(find /tmp/pmch_$c -name '*counter_collection.csv' | head -1)" "$KN" $c <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
pats = sys.argv[2].split(",")
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if any(p in n for p in pats):
        short = next(p for p in pats if p in n) + ("<" + n.split("<", 1)[1].split(">")[0] + ">" if "<" in n.split("(anonymous namespace)::")[-1].split("(")[0] else "")
        agg[(short[:60], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%s per dispatch (KB) %-62s grid=%-9s n=%-4d mean=%.1f" % (sys.argv[3], k[0], k[1], len(v), sum(v) / len(v)))
PY
done

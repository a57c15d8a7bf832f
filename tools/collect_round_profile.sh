#!/bin/bash
# Collects the judged evidence of a round on the GPU box into gpurun_out/<tag>_*:
#   bench JSON line, rocprofv3 --kernel-trace --stats summary of the same command, per-kernel trace summary of our
#   kernels, and FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, --kernel-trace only) for the roofline kernels.
TAG=${1:-r02}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
OURS="backbone_kernel backbone_lp_kernel conv_tower tower_lp gru_bidir gru_lp value_tail tail_lp candidate_windows compact_flags propose_kernel select_kernel transform advance_rows gather_rows x0hat epilogue_ln conv1d_cl"
python3 /root/repo/bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python3 /root/repo/bench.py --steps 2 --warmup 1 --cpu-steps 0 > /tmp/prof_$TAG.log 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
python3 /root/repo/tools/trace_summary.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) $OURS > $OUT/${TAG}_own_kernels_trace_summary.txt
: > $OUT/${TAG}_pmc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 /root/repo/bench.py --steps 1 --warmup 0 --cpu-steps 0 --alt-precision f16x3 --alt-steps 1 > /tmp/pmc_$c.log 2>&1
  python3 - $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $c >> $OUT/${TAG}_pmc.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if any(p in n for p in ("backbone_kernel", "backbone_lp_kernel", "propose_kernel", "conv_tower", "tower_lp", "gru_bidir", "gru_lp", "value_tail", "tail_lp")):
        agg[(n[:90], r["Grid_Size_X"] if "Grid_Size_X" in r else "")].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%s per dispatch (KB) %-92s grid=%-8s n=%d mean=%.3f min=%.3f max=%.3f" % (sys.argv[2], k[0], k[1], len(v), sum(v) / len(v), min(v), max(v)))
PY
done
cat $OUT/${TAG}_bench.json | cut -c1-400; cat $OUT/${TAG}_pmc.txt; cat $OUT/${TAG}_own_kernels_trace_summary.txt

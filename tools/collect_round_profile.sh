#!/bin/bash
# Collects the judged evidence of a round on the GPU box into gpurun_out/<tag>_*:
#   bench JSON line, rocprofv3 --kernel-trace --stats summary of the same command, per-kernel trace summary of our
#   kernels, and FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, --kernel-trace only) for the roofline kernels.
TAG=${1:-r06}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
OURS="backbone_kernel backbone_lp_kernel backbone_lp_t_kernel conv_tower tower_lp gru_bidir gru_pc gru_lp value_tail tail_lp candidate_windows compact_flags propose_kernel select_kernel select_rows_kernel tds_cdf tds_gather transform advance_rows gather_rows x0hat epilogue_ln conv1d_cl"
python3 /root/repo/bench.py --steps 3 --warmup 1 --full-json $OUT/${TAG}_bench_full.json 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json   # the compact line (what the driver parses) + the long record
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python3 /root/repo/bench.py --steps 2 --warmup 1 --cpu-steps 0 --c4-steps 0 --c3-steps 0 --c5-steps 0 --extra-legs 0 > /tmp/prof_$TAG.log 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
python3 /root/repo/tools/trace_summary.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1) $OURS > $OUT/${TAG}_own_kernels_trace_summary.txt
: > $OUT/${TAG}_pmc.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 /root/repo/bench.py --steps 1 --warmup 0 --cpu-steps 0 --c4-steps 0 --c3-steps 0 --c5-steps 0 --extra-legs 0 --alt-precision f16x3,bf16x3,bf16 --alt-steps 1 > /tmp/pmc_$c.log 2>&1
  python3 - $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $c >> $OUT/${TAG}_pmc.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if any(p in n for p in ("backbone_kernel", "backbone_lp_kernel", "backbone_lp_t_kernel", "propose_kernel", "select_rows_kernel", "conv_tower", "tower_lp", "gru_bidir", "gru_lp", "value_tail", "tail_lp")):
        agg[(n[:90], r["Grid_Size_X"] if "Grid_Size_X" in r else "")].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%s per dispatch (KB) %-92s grid=%-8s n=%d mean=%.3f min=%.3f max=%.3f" % (sys.argv[2], k[0], k[1], len(v), sum(v) / len(v), min(v), max(v)))
PY
done
# HBM bytes per launch of the roofline kernels -> ${TAG}_pmc.json (read by bench.py; FETCH_SIZE x 2: MI355X_MICROARCH.md, HBM section)
python3 - $OUT/${TAG}_pmc.txt $OUT/${TAG}_pmc.json $TAG <<'PY'
import json, re, sys
rows = {}
for ln in open(sys.argv[1]):
    m = re.match(r"(FETCH_SIZE|WRITE_SIZE) per dispatch \(KB\) (.*?)\s+grid=\s*\S*\s+n=(\d+) mean=([\d.]+) min=([\d.]+) max=([\d.]+)", ln)
    if m:
        rows.setdefault(m.group(2).strip(), {})[m.group(1)] = (float(m.group(4)), float(m.group(5)), int(m.group(3)), float(m.group(6)))
def traffic(pat, use_min=False, use_max=False):
    for name, d in rows.items():
        if re.search(pat, name) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            k = 3 if use_max else 1 if use_min else 0
            return int(2 * d["FETCH_SIZE"][k] * 1024 + d["WRITE_SIZE"][k] * 1024), name, d
    return None, None, None
out = {"how": "two separate passes per counter (tools/collect_round_profile.sh %s): rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace "
              "--output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-steps 0 --c4-steps 0 --c3-steps 0 --c5-steps 0 --extra-legs 0 --alt-precision f16x3,bf16x3,bf16 --alt-steps 1 ; "
              "per-dispatch means in profiles/%s_pmc.txt (rocprofv3 reports KB)" % (sys.argv[3], sys.argv[3]),
       "fetch_correction": "x2: on gfx950 FETCH_SIZE tallies 128-B requests as 64 B for wide coalesced streams (MI355X_MICROARCH.md, HBM section)"}
t, name, d = traffic(r"backbone_kernel<true(, false)?>")
if t: out["backbone_traffic_bytes_per_launch"] = t; out["backbone"] = {"kernel": name, "FETCH_SIZE_KB": d["FETCH_SIZE"][0], "WRITE_SIZE_KB": d["WRITE_SIZE"][0], "algorithmic_bytes_per_launch": 14405104}
t, name, d = traffic(r"propose_kernel<false, false>", use_min=True)
if t: out["k1_traffic_bytes_per_launch"] = t; out["k1"] = {"kernel": name + " (min over the dispatches: the list also holds bench.py's saturated launches)", "FETCH_SIZE_KB": d["FETCH_SIZE"][1], "WRITE_SIZE_KB": d["WRITE_SIZE"][1], "algorithmic_bytes_per_launch": 9779200}
t, name, d = traffic(r"propose_kernel<false, false>", use_max=True)
if t: out["k1_saturated_traffic_bytes_per_launch"] = t; out["k1_saturated"] = {"kernel": name + " (max over the dispatches: bench.py's saturated launches, B = 16384)", "FETCH_SIZE_KB": d["FETCH_SIZE"][3], "WRITE_SIZE_KB": d["WRITE_SIZE"][3], "algorithmic_bytes_per_launch": 16384 * 200 * (21 + 17 * 10)}
lp = {}
for mode, pat in (("f16x3", r"backbone_lp_t_kernelIDF16_Li3E"), ("bf16x3", r"backbone_lp_t_kernelIDF16bLi3E"), ("bf16", r"backbone_lp_t_kernelIDF16bLi1E|backbone_lp_t_kernel<bool _Accum")):
    t, name, d = traffic(pat)
    if t: lp[mode] = t; out["backbone_lp_" + mode] = {"kernel": name, "FETCH_SIZE_KB": d["FETCH_SIZE"][0], "WRITE_SIZE_KB": d["WRITE_SIZE"][0]}
out["backbone_lp_traffic_bytes_per_launch"] = lp
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
# K2 at saturation: HBM bytes actually moved, one pass per counter and candidate-row stride (200 = dense: the gather of 200-byte rows
# over-fetches; 256 = whole lines) -> ${TAG}_pmc_k2_raw.txt
: > $OUT/${TAG}_pmc_k2_raw.txt
for ld in 0 256; do
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_k2
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_k2 -- python3 /root/repo/tools/resample_microbench.py one 10 $ld > /dev/null 2>&1
  python3 - $(find /tmp/pmc_k2 -name "*counter_collection.csv" | head -1) $c $ld >> $OUT/${TAG}_pmc_k2_raw.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "select_rows" in n:
        agg[(n[:70], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("row stride %s: %s per dispatch (KB) %-72s grid=%-8s n=%d mean=%.1f" % (sys.argv[3] if sys.argv[3] != "0" else "200", sys.argv[2], k[0], k[1], len(v), sum(v) / len(v)))
PY
done
done
cat $OUT/${TAG}_bench.json | cut -c1-400; cat $OUT/${TAG}_pmc.txt; cat $OUT/${TAG}_own_kernels_trace_summary.txt

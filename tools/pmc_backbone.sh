#!/bin/bash
# PMC passes over the one-launch backbone kernel (tools/backbone_microbench.py). Output: gpurun_out/pmc_bb_*.csv
cd /tmp && export TMPDIR=/tmp
B=${1:-16}
i=0
for set in "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcbb$i -- python3 /root/repo/tools/backbone_microbench.py $B 200 > /tmp/pmcbb$i.log 2>&1
  f=$(find /tmp/pmcbb$i -name "*counter_collection.csv" | head -1); [ -z "$f" ] && tail -5 /tmp/pmcbb$i.log
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "backbone_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%-28s n=%d mean=%.1f" % (k, len(v), sum(v) / len(v)))
PY
done

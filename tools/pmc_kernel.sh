#!/bin/bash
# SQ PMC passes for one kernel: tools/pmc_kernel.sh <kernel-name-substring> <python script> [args...]
# (separate rocprofv3 --pmc runs with --kernel-trace only; prints the per-dispatch mean of every counter)
cd /tmp && export TMPDIR=/tmp
KN=$1; shift
i=0
for set in "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmck$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmck$i -- python3 "$@" > /tmp/pmck$i.log 2>&1
  f=$(find /tmp/pmck$i -name "*counter_collection.csv" | head -1); [ -z "$f" ] && tail -5 /tmp/pmck$i.log
  python3 - "$f" "$KN" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%-28s n=%d mean=%.1f" % (k, len(v), sum(v) / len(v)))
PY
done

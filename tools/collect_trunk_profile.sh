#!/bin/bash
# Evidence for the config-4 value trunk (csrc/svdd_trunk.hip) on the GPU box -> gpurun_out/<tag>_trunk_*:
#   per-GEMM timing of one forward at 3840 live candidates for the GEMM kernel versions (1 = 128 x 128 tiles, 2 = default,
#   13 / 14 = 256 x 256 kernel without epilogue / with one K block: K loop and epilogue timed apart), rocprofv3 kernel
#   stats of the default, and an SQ counter pass of trunk_gemm256_kernel.
TAG=${1:-r03}
OUT=/root/repo/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_trunk_gemm_versions.txt
for v in 1 2; do
  echo "## SVDD_OPT_TRUNK_GEMM_VERSION=$v" >> $OUT/${TAG}_trunk_gemm_versions.txt
  SVDD_TRUNK_GEMM=$v python3 /root/repo/tools/trunk_microbench.py 3840 bf16x3 --gemms 2>/dev/null >> $OUT/${TAG}_trunk_gemm_versions.txt
  SVDD_TRUNK_GEMM=$v python3 /root/repo/tools/trunk_microbench.py 3840 bf16 2>/dev/null | tail -1 >> $OUT/${TAG}_trunk_gemm_versions.txt
done
: > $OUT/${TAG}_trunk_gemm_epilogue.txt
for v in 13 14 15 16; do
  echo "## debug version $v (13: no epilogue, 14: one K block + epilogue, 15: no epilogue + no LDS-DMA in the K loop, 16: no epilogue + no fragment reads; results are garbage, timing only)" >> $OUT/${TAG}_trunk_gemm_epilogue.txt
  SVDD_TRUNK_GEMM=$v python3 /root/repo/tools/trunk_microbench.py 3840 bf16x3 --gemms 2>/dev/null >> $OUT/${TAG}_trunk_gemm_epilogue.txt
done
rm -rf /tmp/prof_trunk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trunk -o t -- python3 /root/repo/tools/trunk_microbench.py 3840 bf16x3 > /dev/null 2>&1
python3 /root/repo/tools/kernel_stats_top.py /tmp/prof_trunk 14 > $OUT/${TAG}_trunk_kernel_split.txt
: > $OUT/${TAG}_pmc_trunk_gemm256.txt
for set in "SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_t
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_t -- python3 /root/repo/tools/trunk_microbench.py 3840 bf16x3 > /dev/null 2>&1
  python3 - $(find /tmp/pmc_t -name "*counter_collection.csv" | head -1) >> $OUT/${TAG}_pmc_trunk_gemm256.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "trunk_gemm256" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%-28s n=%d mean=%.1f sum=%.1f" % (k, len(v), sum(v) / len(v), sum(v)))
PY
done

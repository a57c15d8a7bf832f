#!/bin/bash
# Which measurement scripts under tools/ still run against today's library (ABI in svdd_amd/_lib.py)? Each Python tool is started on the
# GPU box with a 40 s budget: "ok" = finished with exit code 0, "runs" = still running without an error when the budget ended (the long
# soaks / whole-decode sweeps), "FAILS" = non-zero exit (last line of its output kept). Tools that need arguments, an instrumented
# library (tools/exp_variants.py build ...) or input files are listed as such, not run. -> gpurun_out/<tag>_tools_check.txt
# Usage (GPU box): bash tools/check_tools.sh r06
cd "$(dirname "$0")/.."
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_tools_check.txt
SKIP="exp_variants.py lpt_phase_timing.py lpt_step_stamps.py tower_lp_timing.py gen_lpt_taps.py stats_summary.py trace_summary.py kernel_stats_top.py perturbation_f16x3_droplo.py"
: > $OUT
for f in tools/*.py; do
  b=$(basename $f)
  if echo " $SKIP " | grep -q " $b "; then echo "$b  not-run (needs arguments / an instrumented build / input files)" >> $OUT; continue; fi
  timeout 40 python $f > /tmp/tool_out.txt 2>&1
  rc=$?
  last=$(grep -v "amdgpu.ids" /tmp/tool_out.txt | tail -1 | cut -c1-160)
  if [ $rc -eq 0 ]; then echo "$b  ok" >> $OUT
  elif [ $rc -eq 124 ]; then
    if grep -q "Traceback\|Error" /tmp/tool_out.txt; then echo "$b  FAILS (before the budget ended): $last" >> $OUT; else echo "$b  runs (40 s budget ended without an error)" >> $OUT; fi
  else echo "$b  FAILS rc=$rc: $last" >> $OUT; fi
done
cat $OUT

#!/bin/bash
# which reference-run tests turn red under a deliberately perturbed kernel -> profiles/r06_perturbation_check.txt
# (builds: python tools/exp_variants.py build perturb). Usage (GPU box): bash tools/perturbation_check.sh
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_perturbation_check.txt
K='(unguided or m20 or tds_baseline or c2_against or c3_against) and f32'
{
echo "# fp32 backbone, LayerNorm epsilon 1e-5 -> 1.2e-5 / 2e-5 / 1e-4 (patched copies under build/exp/perturb/): pytest tests/test_e2e_gpu.py -k \"$K\""
for v in eps1.2 eps2 eps10; do
  echo "## $v"
  SVDD_HIP_LIB=build/exp/perturb/$v/libsvdd_hip.so python -m pytest tests/test_e2e_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "$K" 2>&1 | grep -E "^FAILED|passed|failed" 
done
echo "# f16x3: the lo plane of backbone layer 7's weights zeroed (tools/perturbation_f16x3_droplo.py): C2 / C3 reference-run tests"
python tools/perturbation_f16x3_droplo.py 2>&1 | grep -E "^FAILED|passed|failed|assert .*<=|Error" | head -8
echo "# control: the tracked kernels"
python -m pytest tests/test_e2e_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "$K or ((c2_against or c3_against) and f16x3)" 2>&1 | grep -E "^FAILED|passed|failed"
} > $OUT 2>&1
cat $OUT

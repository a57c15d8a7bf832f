"""a perturbation the headline-config tests DO catch, in the split-precision path. The lo plane of ONE backbone layer's
weights (layer 7) is zeroed in the f16x3 operand image — i.e. that layer's hi*lo pass multiplies by zero: its products keep 11 bits of
the weights instead of 22 — by patching svdd_amd.fused.pack_backbone_lp in this process only (no product code changes), then the C2 /
C3 reference-run tests run in-process. Usage (GPU box): python tools/perturbation_f16x3_droplo.py [layer]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from svdd_amd import fused

LAYER = int(sys.argv[1]) if len(sys.argv) > 1 else 7
_orig = fused.pack_backbone_lp


def perturbed(cnn, precision):
    pk = _orig(cnn, precision)
    if precision == "f16x3":
        nl = len(cnn.convs)
        per_layer = 4 * 9 * 4 * 4 * 16 * 2 * 2 * 8                       # [c][t][cg][g][j][ct][parts][e]
        t = pk["tiles"]
        t[: nl * per_layer].view(nl, 4, 9, 4, 4, 16, 2, 2, 8)[LAYER, :, :, :, :, :, :, 1, :] = 0
    return pk


fused.pack_backbone_lp = perturbed
sys.exit(pytest.main(["tests/test_e2e_gpu.py", "-m", "gpu", "-q", "-x", "--no-header", "-p", "no:cacheprovider",
                      "-k", "(c2_against or c3_against) and f16x3 and not bf16x3"]))

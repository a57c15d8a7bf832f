"""Race soak of the small-batch backbone (backbone_split_kernel: 2 / 4 workgroups per sequence exchanging the LayerNorm'd image
through system-scope accesses + an arrival counter): many launches at many batch sizes, under a concurrent HBM-heavy side stream,
every output compared bit for bit with the one-workgroup kernel. Usage: python tools/split_soak.py [--iters 300]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, backbone, config, fused

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
args = ap.parse_args()
DEV = "cuda:0"
torch.manual_seed(1)
cnn = backbone.CNNModel(config.dna_config().model, alphabet_size=5).to(DEV).eval()
pk = fused.pack_backbone(cnn)
lib = _lib.lib()
side = torch.cuda.Stream()
noise = torch.empty(64 << 20, device=DEV)
bad = 0
launches = 0
for n, L in ((4, 200), (17, 200), (32, 200), (64, 200), (64, 187), (100, 200), (128, 200), (300, 200)):
    x = torch.randint(0, 5, (n, L), device=DEV, dtype=torch.uint8)
    _lib.set_option(7, 1)
    ref = fused.backbone_cnn(x, pk).clone()
    _lib.set_option(7, 0)
    for it in range(args.iters):
        if it % 3 == 0:
            with torch.cuda.stream(side):                      # something else hammering the memory system meanwhile
                noise.add_(1.0)
        if it % 50 == 25:
            x = torch.randint(0, 5, (n, L), device=DEV, dtype=torch.uint8)
            _lib.set_option(7, 1)
            ref = fused.backbone_cnn(x, pk).clone()
            _lib.set_option(7, 0)
        out = fused.backbone_cnn(x, pk)
        launches += 1
        if not torch.equal(out, ref):
            bad += 1
            print(f"MISMATCH n={n} L={L} it={it} max|d|={float((out - ref).abs().max()):.3e}")
torch.cuda.synchronize()
err = ctypes.c_int(-1)
lib.svdd_backbone_split_status(ctypes.byref(err))
print(f"{launches} split launches, {bad} mismatches, barrier time-outs: {err.value}")
sys.exit(1 if bad or err.value else 0)

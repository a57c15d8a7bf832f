"""Where tower_lp_kernel's cycles go on candidate windows (instrumented build: tools/exp_variants.py build tower_lp_timing; run with
SVDD_HIP_LIB=build/exp/tower_lp_timing/timing/libsvdd_hip.so): cycles summed over the waves of one launch of the config-2 step's
window batch (2560 candidates). Usage: python tools/tower_lp_timing.py [mode] [changes_per_candidate]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from svdd_amd import _lib, fused, synthetic

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
chg = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
dev = "cuda:0"
B, M, L = 256, 10, 200
model, emb, head, _ = synthetic.build("dna", dev)
fv = fused.FusedValueNet(emb, head).to(dev).eval()
fv.precision = mode
pk = fv._lp_pack()
g = torch.Generator(device=dev).manual_seed(1)
x = torch.where(torch.rand(B, L, device=dev, generator=g) < 0.7, 4, torch.randint(0, 4, (B, L), device=dev, generator=g)).to(torch.uint8)
cand = x[:, None, :].repeat(1, M, 1)
flip = (torch.rand(B, M, L, device=dev, generator=g) < chg / (0.7 * L)) & (cand == 4)
cand = torch.where(flip, torch.randint(0, 4, (B, M, L), device=dev, generator=g).to(torch.uint8), cand).contiguous()
win = fused.candidate_windows(cand, x)
parent = fused.conv_tower_lp(x, pk["tiles"], fv.tw_bias, pk["tinv"], fv.tw_resmask, pk["prec"])
out = torch.empty((B * M, L, parent.shape[2], 64), dtype=parent.dtype, device=dev)
tiles = float(((win[:, 1] - win[:, 0]) // 16).float().mean())
run = lambda: fused.conv_tower_windows_lp(cand, win, parent, pk["tiles"], fv.tw_bias, pk["tinv"], fv.tw_resmask, pk["prec"], out=out)
for _ in range(20):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record(); torch.cuda.synchronize()
lib = _lib.lib()
buf = np.zeros(4096 * 8 * 8, dtype=np.uint64)
assert lib.svdd_internal_tw_dbg(ctypes.c_void_p(buf.ctypes.data), 1) == 0          # zero the device table
run(); torch.cuda.synchronize()
assert lib.svdd_internal_tw_dbg(ctypes.c_void_p(buf.ctypes.data), 0) == 0
d = buf.reshape(4096, 8, 8).astype(np.float64)
ran = d[:, :, 6] > 0
t = np.array([d[:, :, k][ran].sum() for k in range(8)])
wgs = ran.sum() / 8.0
print(f"{mode}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch, {tiles:.2f} live row tiles per candidate window; {wgs:.0f} workgroups ran the layers")
names = ("prologue (parent rows, one-hot, zeroing) [per wave]", "MFMA loops", "barrier after a layer's loop", "epilogues + image barriers", "output copy", "whole kernel")
for k, nm in enumerate(names):
    print(f"  {nm:52s} {t[k] / t[6]:10.0f} cycles per wave   ({t[k] / t[5] * 100:5.1f} %)")

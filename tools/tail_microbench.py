"""value_tail_kernel alone (direction sum + LayerNorm + 64 -> 128 + ReLU + collapsed head + mean over length): per-dispatch time at the
row counts of the C2 decode, with the valid-row count on the device as the work-skipping decode passes it.
Usage: python tools/tail_microbench.py [n ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib
from svdd_amd.fused import pack_tail, value_tail
dev = "cuda:0"
torch.manual_seed(0)
L = 200
w1p, b1f = pack_tail(torch.randn(128, 64, device=dev) * 0.1, torch.randn(128, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1)
weff, beff = torch.randn(128, 1, device=dev) * 0.1, torch.randn(1, device=dev)
h = torch.randn(2, 2560, L, 64, device=dev)
ref = None
for live in [int(a) for a in sys.argv[1:]] or (2560, 2048, 1980, 1536, 512):
    cnt = torch.tensor([live], dtype=torch.int32, device=dev)
    for _ in range(3): out = value_tail(h, w1p, b1f, weff, beff, count=cnt)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(20): value_tail(h, w1p, b1f, weff, beff, count=cnt)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    tot, k = _lib.profile_collect(7)
    us = tot / k * 1e3
    print(f"count={live} of 2560 rows, L={L}: {us:.1f} us  ({2*live*L*64*4/(us*1e-6)/1e12:.2f} TB/s read, {live*L*2*64*128/(us*1e-6)/1e12:.1f} TFLOP/s)  sum={float(out[:live].double().sum()):.9f}")

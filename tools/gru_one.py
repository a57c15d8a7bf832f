"""fp32 GRU alone (svdd_gru_bidir_f32, default kernel): per-launch time at n sequences. Usage: python tools/gru_one.py [n ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib
from svdd_amd.fused import gru_bidir, pack_gru

gru = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).to("cuda").eval()
wp, bp = pack_gru(gru)
res = []
for n in [int(a) for a in sys.argv[1:]] or [2048, 2560]:
    x = torch.randn(n, 200, 64, device="cuda")
    for _ in range(4):
        gru_bidir(x, wp, bp)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(10):
        gru_bidir(x, wp, bp)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    tot, k = _lib.profile_collect(3)
    res.append(f"n={n}: {tot / k * 1e3:.1f} us")
print(" | ".join(res))

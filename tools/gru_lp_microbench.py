"""Time of the split-precision GRU kernel alone (svdd_gru_bidir_lp). Usage: python tools/gru_lp_microbench.py [mode] [n ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, fused

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
ns = [int(a) for a in sys.argv[2:]] or [2560]
dev = "cuda:0"
torch.manual_seed(0)
gru = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).to(dev).eval()
wp, bp, inv = fused.pack_gru_lp(gru, mode)
for n in ns:
    x = torch.randn(n, 200, 64, device=dev).relu()
    dt, parts = fused.LP_DTYPES[mode]
    if os.environ.get("GRU_LP_FP32_INPUT") != "1":          # the tower's output format: [n, L, P, 64] 16-bit planes
        x = fused._split16(x, dt, parts).permute(0, 1, 3, 2).contiguous()
    out = torch.empty(2, n, 200, 64, device=dev)
    for _ in range(3):
        fused.gru_bidir_lp(x, wp, bp, inv, _lib.PRECISIONS[mode], out=out)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(10):
        fused.gru_bidir_lp(x, wp, bp, inv, _lib.PRECISIONS[mode], out=out)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    tot, k = _lib.profile_collect(3)
    print(f"mode={mode} n={n} L=200: {tot / k * 1e3:.1f} us")

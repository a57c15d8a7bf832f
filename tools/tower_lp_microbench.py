"""Time of the split-precision conv tower on candidate windows (svdd_conv_tower_windows_lp) and on whole sequences.
Usage: python tools/tower_lp_microbench.py [mode] [changes_per_candidate]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    tot, k = _lib.profile_collect(5)
    return tot / k * 1e3


t_win = timed(lambda: fused.conv_tower_windows_lp(cand, win, parent, pk["tiles"], fv.tw_bias, pk["tinv"], fv.tw_resmask, pk["prec"], out=out))
t_full = timed(lambda: fused.conv_tower_lp(cand.view(B * M, L), pk["tiles"], fv.tw_bias, pk["tinv"], fv.tw_resmask, pk["prec"]))
t_par = timed(lambda: fused.conv_tower_lp(x, pk["tiles"], fv.tw_bias, pk["tinv"], fv.tw_resmask, pk["prec"]))
print(f"mode={mode} changes/cand {float((cand != x[:, None, :]).float().sum(2).mean()):.2f} live tiles/cand {tiles:.2f}: "
      f"windows {t_win:.1f} us  full(2560) {t_full:.1f} us  parents(256) {t_par:.1f} us")

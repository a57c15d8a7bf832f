"""A/B of the generations of the fp32 conv tower (svdd_set_tower_version): whole sequences and candidate windows.
Usage: python tools/tower_ab.py [changes_per_candidate] [only this generation]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, fused, ops, synthetic

chg = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
dev = "cuda:0"
B, M, L = 256, 10, 200
model, emb, head, _ = synthetic.build("dna", dev)
fv = fused.FusedValueNet(emb, head).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
x = torch.where(torch.rand(B, L, device=dev, generator=g) < 0.7, 4, torch.randint(0, 4, (B, L), device=dev, generator=g)).to(torch.uint8)
cand = x[:, None, :].repeat(1, M, 1)
flip = (torch.rand(B, M, L, device=dev, generator=g) < chg / (0.7 * L)) & (cand == 4)
cand = torch.where(flip, torch.randint(0, 4, (B, M, L), device=dev, generator=g).to(torch.uint8), cand).contiguous()
win = fused.candidate_windows(cand, x)
onehot = ops.transform_samples(cand.view(B * M, L))
xo = ops.transform_samples(x)
parent = fused.conv_tower(xo, fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
tiles = float(((win[:, 1] - win[:, 0]) // 16).float().mean())


def timed(fn, slot, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    tot, k = _lib.profile_collect(slot)
    return tot / k * 1e3


for v in ((int(sys.argv[2]),) if len(sys.argv) > 2 else (1, 2, 3)):
    _lib.lib().svdd_set_tower_version(v)
    t_win = timed(lambda: fused.conv_tower_windows(onehot, win, parent, M, fv.tw_tiles, fv.tw_bias, fv.tw_resmask), 5)
    t_full = timed(lambda: fused.conv_tower(onehot, fv.tw_tiles, fv.tw_bias, fv.tw_resmask), 5)
    t_par = timed(lambda: fused.conv_tower(xo, fv.tw_tiles, fv.tw_bias, fv.tw_resmask), 5)
    print(f"generation {v}: live tiles/cand {tiles:.2f}: windows {t_win:.1f} us  full(2560) {t_full:.1f} us  parents(256) {t_par:.1f} us")

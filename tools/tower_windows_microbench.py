"""Windowed (parent-sharing) conv tower vs the full tower on SVDD-like candidates (each masked position unmasks with
probability 1/remaining-steps; here: a fixed expected number of changes per candidate)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic, ops, fused
dev = "cuda:0"
B, M, L = 256, 10, 200
model, emb, head, _ = synthetic.build("dna", dev)
fv = fused.FusedValueNet(emb, head).to(dev).eval()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for lam in (0.0, 1.56, 200.0):
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randint(0, 4, (B, L), generator=g).to(torch.uint8)
    x[torch.rand(B, L, generator=g) < 0.5] = 4
    chg = (torch.rand(B, M, L, generator=g) < lam / max(1.0, float((x == 4).sum(1).float().mean()))) & (x == 4)[:, None, :]
    cand = torch.where(chg, torch.randint(0, 4, (B, M, L), generator=g).to(torch.uint8), x[:, None, :].expand(B, M, L)).contiguous()
    x, cand = x.to(dev), cand.to(dev)
    onehot = ops.transform_samples(cand.view(B * M, L))
    win = fused.candidate_windows(cand, x)
    parent = fused.conv_tower(ops.transform_samples(x), fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
    frac = float(((win[:, 1] - win[:, 0]) // 16).float().mean()) / 13
    t_full = timeit(lambda: fused.conv_tower(onehot, fv.tw_tiles, fv.tw_bias, fv.tw_resmask))
    t_par = timeit(lambda: fused.conv_tower(ops.transform_samples(x), fv.tw_tiles, fv.tw_bias, fv.tw_resmask))
    t_win = timeit(lambda: fused.candidate_windows(cand, x))
    t_tw = timeit(lambda: fused.conv_tower_windows(onehot, win, parent, M, fv.tw_tiles, fv.tw_bias, fv.tw_resmask))
    print(f"changes/candidate {lam:6.2f}: row tiles computed {frac:5.1%} ; full tower {t_full:7.1f} us ; parent pass {t_par:6.1f} + windows {t_win:5.1f} + windowed tower {t_tw:7.1f} us")

"""Where a DPS gradient (diffusion_gosai.py:1321-1330) spends its time at B rows: the backbone half (forward2 + backward to the
one-hot input) and the reward-net half (ConvGRU forward + backward to its input), each alone. Usage: python tools/dps_split.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dna, emb, head, rew = synthetic.build("dna", "cuda:0")
x = torch.randint(0, 5, (B, 200), device="cuda:0")
x[:, ::3] = 4
oh = torch.nn.functional.one_hot(x, 5).float()
sig = torch.zeros(B, device="cuda:0")
cf = (x != 4).long()


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def backbone_half():
    with torch.enable_grad():
        xo = oh.clone().requires_grad_(True)
        dna.forward2(xo, x, sig).sum().backward()


def reward_half():
    grus = [m for m in rew.modules() if isinstance(m, torch.nn.GRU)]
    for m in grus:
        m.train()
    try:
        with torch.enable_grad():
            p = torch.softmax(oh.clone(), dim=2).requires_grad_(True)
            rew(p.transpose(1, 2)[:, 0:4, :])[:, 0].mean().backward()
    finally:
        for m in grus:
            m.eval()


def whole():
    with torch.enable_grad():
        dna.compute_gradient_DPS(oh.clone(), x, rew, sig, cf)


print(f"B={B}: backbone forward2 + backward {timed(backbone_half):.1f} ms ; reward net forward + backward {timed(reward_half):.1f} ms ; "
      f"compute_gradient_DPS {timed(whole):.1f} ms")


# ---- inside the reward-net half: the bidirectional GRU (MIOpen's fused RNN, fwd + bwd) against the rest
gru = [m for m in rew.modules() if isinstance(m, torch.nn.GRU)][0]
xin = torch.randn(B, 200, gru.input_size, device="cuda:0")


def gru_only():
    gru.train()
    try:
        with torch.enable_grad():
            xi = xin.clone().requires_grad_(True)
            gru(xi)[0].sum().backward()
    finally:
        gru.eval()


def gru_fwd_only():
    with torch.no_grad():
        gru(xin)


print(f"      nn.GRU({gru.input_size}, {gru.hidden_size}, bidirectional) alone: forward + backward {timed(gru_only):.1f} ms, forward {timed(gru_fwd_only):.1f} ms")

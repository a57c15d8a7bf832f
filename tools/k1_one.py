"""K1 alone at a saturating size (B = 16384, L = 200, M = 10): per-launch time. Usage: python tools/k1_one.py [masked_frac]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, ops

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
dev = "cuda:0"
torch.manual_seed(0)
B, L, M = 16384, 200, 10
logits = torch.randn(B, 5, L, device=dev).transpose(1, 2)
x = torch.where(torch.rand(B, L, device=dev) < frac, torch.full((B, L), 4, device=dev), torch.randint(0, 4, (B, L), device=dev)).to(torch.uint8)
cand = torch.empty(B, M, L, dtype=torch.uint8, device=dev)
onehot = torch.empty(B * M, L, 4, device=dev)
rng = ops.Rng(seed=1, step=3)
for _ in range(5):
    ops.propose(logits, x, 0.0078, 0.5, M, rng, cand=cand, onehot=onehot)
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(50):
    ops.propose(logits, x, 0.0078, 0.5, M, rng, cand=cand, onehot=onehot)
torch.cuda.synchronize()
_lib.profile_enable(False)
tot, n = _lib.profile_collect(0)
us = tot * 1e3 / n
print(f"masked {frac:.0%}: K1 {us:7.2f} us  {B * L * (21 + 17 * M) / us / 1e3:7.1f} GB/s")

"""Micro-benchmark of K1 (propose) and K2 (select) alone, at the bench workload's size and at a size
that saturates the chip. Prints per-launch time (HIP events over many back-to-back launches) and the
achieved algorithmic bandwidth. Usage: python tools/k1_microbench.py [--masked-frac 0.5]"""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--masked-frac", type=float, default=0.5)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--exact", action="store_true")
ap.add_argument("--msplit", type=int, default=0)
args = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
_lib.set_force_exact(args.exact)
_lib.set_option(1, args.msplit)
for (B, L, M) in [(256, 200, 10), (2048, 200, 10), (16384, 200, 10), (2048, 200, 20)]:
    logits = torch.randn(B, 5, L, device=dev).transpose(1, 2)
    x = torch.where(torch.rand(B, L, device=dev) < args.masked_frac, torch.full((B, L), 4, device=dev), torch.randint(0, 4, (B, L), device=dev)).to(torch.uint8)
    cand = torch.empty(B, M, L, dtype=torch.uint8, device=dev)
    onehot = torch.empty(B * M, L, 4, device=dev)
    scores = torch.randn(B, M, device=dev)
    rng = ops.Rng(seed=1, step=3)
    for _ in range(5):
        ops.propose(logits, x, 0.0078, 0.5, M, rng, cand=cand, onehot=onehot)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _lib.profile_enable(True)
    e0.record()
    for _ in range(args.iters):
        ops.propose(logits, x, 0.0078, 0.5, M, rng, cand=cand, onehot=onehot)
    e1.record(); torch.cuda.synchronize()
    _lib.profile_enable(False)
    tot, n = _lib.profile_collect(0)
    us_wall = e0.elapsed_time(e1) * 1e3 / args.iters
    us = tot * 1e3 / n
    nbytes = B * L * (21 + 17 * M)
    _lib.profile_enable(True)
    for _ in range(args.iters):
        ops.select(scores, cand, want_soft=False)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    tot, n = _lib.profile_collect(1)
    us2 = tot * 1e3 / n
    nb2 = B * (4 * M + 2 * L + 4)
    # reference point: a pure fill of the one-hot buffer (the write-bandwidth ceiling for this many bytes)
    for _ in range(3):
        onehot.fill_(1.0)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        onehot.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    us_fill = e0.elapsed_time(e1) * 1e3 / 20
    fill_gbs = onehot.numel() * 4 / us_fill / 1e3
    print(f"   fill of onehot ({onehot.numel()*4/1e6:.1f} MB): {us_fill:.2f} us/launch back-to-back = {fill_gbs:.0f} GB/s")
    print(f"B={B:6d} L={L} M={M:3d}  K1 {us:8.2f} us (wall/launch {us_wall:6.2f})  {nbytes/us/1e3:8.1f} GB/s ({nbytes/1e6:.1f} MB)   K2 {us2:7.2f} us {nb2/us2/1e3:7.1f} GB/s")

"""Race soak of the config-4 trunk's shared levels + two streams: every iteration evolves the parents by a few
positions (so the parents' own step-to-step update runs), draws candidates that differ from them at 1-4 positions and a random
live count, and compares forward_tokens(shared=..., two streams) with whole sequences on one chain of kernels, bit for bit.
Usage: python tools/trunk_shared_soak.py [iterations] [precision]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic
from svdd_amd.fused_trunk import FusedEnformerValueNet

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
dev = "cuda:0"
_, emb, head, _ = synthetic.build("dna", dev, value="enformer")
fn = FusedEnformerValueNet(emb, head, prec)
ref = FusedEnformerValueNet(emb, head, prec)
ref.tower_streams = 1
if "--no-share" in sys.argv:
    fn.share_level0 = False
if "--one-stream" in sys.argv:
    fn.tower_streams = 1
if "--same-instance" in sys.argv:
    ref = fn
B, M, L = 256, 15, 200
n = B * M
g = torch.Generator(device=dev).manual_seed(7)
par = torch.randint(0, 5, (B, L), device=dev, generator=g, dtype=torch.uint8)
bad = 0
with torch.no_grad():
    for it in range(iters):
        # parents: 0-3 positions per row move on
        k = torch.randint(0, 4, (B,), device=dev, generator=g)
        pos = torch.randint(0, L, (B, 3), device=dev, generator=g)
        sel = torch.arange(3, device=dev)[None, :] < k[:, None]
        newv = ((par.gather(1, pos).long() + 1 + torch.randint(0, 4, (B, 3), device=dev, generator=g)) % 5).to(torch.uint8)
        par = par.scatter(1, pos, torch.where(sel, newv, par.gather(1, pos)))
        # candidates: 1-4 changed positions
        tok = par[:, None, :].repeat(1, M, 1).view(n, L)
        kc = torch.randint(1, 5, (n,), device=dev, generator=g)
        posc = torch.randint(0, L, (n, 4), device=dev, generator=g)
        selc = torch.arange(4, device=dev)[None, :] < kc[:, None]
        newc = ((tok.gather(1, posc).long() + 1 + torch.randint(0, 4, (n, 4), device=dev, generator=g)) % 5).to(torch.uint8)
        tok = tok.scatter(1, posc, torch.where(selc, newc, tok.gather(1, posc))).contiguous()
        live = int(torch.randint(n // 3, n + 1, (1,), device=dev, generator=g))
        cnt = torch.tensor([live], dtype=torch.int32, device=dev)
        idx = torch.arange(n, dtype=torch.int32, device=dev)
        a = fn.forward_tokens(tok, count=cnt, shared=(par, idx, M)).reshape(n)[:live]
        if ref is fn:
            a = a.clone()
            s_, fn.tower_streams = fn.tower_streams, 1
            b = fn.forward_tokens(tok, count=cnt).reshape(n)[:live]
            fn.tower_streams = s_
        else:
            b = ref.forward_tokens(tok, count=cnt).reshape(n)[:live]
        if not torch.equal(a, b):
            bad += 1
            print(f"iteration {it}: {int((a != b).sum())} of {live} scores differ, max {float((a - b).abs().max()):.3e}")
print(f"{prec}: {iters} iterations (n = {n}, random live counts, evolving parents, streams {fn.last_streams}): {bad} mismatches")

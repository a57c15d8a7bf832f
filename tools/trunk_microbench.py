"""The fused Enformer-shaped value trunk (svdd_amd/fused_trunk.py) at the BASELINE configs[3] shard size: time per forward
against the PyTorch module.  Usage: python tools/trunk_microbench.py [n] [precision] [--module] [--gemms] [--shared]
--shared: the rows are candidates of n / 15 parents that differ from them at 1-4 positions (42 / 33 / 17 / 8 %: a Poisson-like
mix that reproduces the compact-row fractions of a real C4 decode, tools/trunk_window_stats.py), scored with the first levels shared with the parent (forward_tokens(shared=...))
(run under `rocprofv3 --kernel-trace --stats` for the per-kernel split)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic
from svdd_amd.fused_trunk import FusedEnformerValueNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
dev = "cuda:0"
_, emb, head, _ = synthetic.build("dna", dev, value="enformer")
tok = torch.randint(0, 5, (n, 200), device=dev, dtype=torch.uint8)
fn = FusedEnformerValueNet(emb, head, prec)
if os.environ.get("SVDD_TRUNK_GEMM"):                  # A/B: 1 = 128 x 128 tiles everywhere, 3 = 256 x 256 everywhere
    from svdd_amd import _lib
    _lib.set_option(4, int(os.environ["SVDD_TRUNK_GEMM"]))


def bench(f, it=3):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / it * 1e3


shared = None
if "--shared" in sys.argv:
    M = 15
    B = n // M
    g = torch.Generator(device="cpu").manual_seed(1)
    par = torch.randint(0, 5, (B, 200), generator=g, dtype=torch.uint8)
    tok = par[:, None, :].repeat(1, M, 1).view(n, 200)
    k = torch.multinomial(torch.tensor([0.42, 0.33, 0.17, 0.08]), n, replacement=True, generator=g) + 1
    for c in range(n):
        pos = torch.randint(0, 200, (int(k[c]),), generator=g)
        tok[c, pos] = (tok[c, pos] + 1 + torch.randint(0, 4, (int(k[c]),), generator=g, dtype=torch.uint8)) % 5
    tok = tok.to(dev).contiguous()
    shared = (par.to(dev), torch.arange(n, dtype=torch.int32, device=dev), M)
    if os.environ.get("SVDD_SHARE_LEVELS"):
        fn.share_levels = int(os.environ["SVDD_SHARE_LEVELS"])
    if os.environ.get("SVDD_SHARE_PARENT_STEPS"):
        fn.share_parent_steps = bool(int(os.environ["SVDD_SHARE_PARENT_STEPS"]))
    if os.environ.get("SVDD_SHARE_SLOTS"):
        fn.share_slots = int(os.environ["SVDD_SHARE_SLOTS"])
count = None
if os.environ.get("SVDD_LIVE"):                           # a compacted batch: only the first SVDD_LIVE rows are live (device scalar)
    count = torch.tensor([int(os.environ["SVDD_LIVE"])], dtype=torch.int32, device=dev)
fl = emb.flops_per_sequence() * (n if count is None else int(count))
ms = bench(lambda: fn.forward_tokens(tok, count=count, shared=shared))
if shared:
    print("compact rows per shared level:", fn.last_window_rows.tolist(), "of", [n * (200 >> d) for d in range(len(fn.last_window_rows))], "windows per candidate <=", fn.share_slots)
print(f"fused trunk {prec} n={n}: {ms:.1f} ms  = {fl / ms / 1e9:.1f} TFLOP/s fp32-equivalent ({fl / 1e12:.2f} TFLOP per forward)")
if "--module" in sys.argv:
    oh = (torch.nn.functional.one_hot(tok.long().clamp(max=3), 4) * (tok != 4)[..., None]).float()
    with torch.no_grad():
        ms = bench(lambda: head(emb(oh)), it=2)
    print(f"PyTorch module fp32 n={n}: {ms:.1f} ms  = {fl / ms / 1e9:.1f} TFLOP/s")

if "--gemms" in sys.argv:
    fn.timing = []
    fn.forward_tokens(tok, count=count, shared=shared)
    torch.cuda.synchronize()
    agg = {}
    for M, N, C, T, e0, e1 in fn.timing:
        k = (M, N, C, T)
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    tot = 0.0
    passes = 3 if prec == "bf16x3" else 1
    for (M, N, C, T), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fl = 2.0 * M * N * C * T * cnt
        tot += ms
        print(f"  M={M:8d} N={N:5d} Cin={C:5d} T={T}  x{cnt:3d}  {ms:8.2f} ms  {fl / ms / 1e9:8.1f} TFLOP/s fp32-equiv  ({fl * passes / ms / 1e9:7.1f} on the MFMA)")
    print(f"  all GEMMs: {tot:.1f} ms")

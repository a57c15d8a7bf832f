"""Split-precision value-net kernels (conv tower, GRU, tail) against the exact-fp32 kernels on SVDD-MC-like candidates:
score error per mode, per-kernel time.  Usage: python tools/value_lp_check.py [B] [M] [L]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, fused, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = "cuda:0"
model, emb, head, _ = synthetic.build("dna" if L == 200 else "rna", dev)
torch.manual_seed(1)
x = torch.where(torch.rand(B, L, device=dev) < 0.7, 4, torch.randint(0, 4, (B, L), device=dev)).to(torch.uint8)
cand = x[:, None, :].repeat(1, M, 1)
flip = (torch.rand(B, M, L, device=dev) < 0.015) & (cand == 4)
cand = torch.where(flip, torch.randint(0, 4, (B, M, L), device=dev).to(torch.uint8), cand).contiguous()
from svdd_amd import ops
onehot = ops.transform_samples(cand.view(B * M, L))
fv = fused.FusedValueNet(emb, head).to(dev).eval()
with torch.no_grad():
    ref64 = head.double()(emb.double()(onehot[:256].double())).reshape(-1).float()
    emb.float(); head.float()
    s_full = fv(onehot).reshape(-1)
    s_win = fv.forward_candidates(onehot, cand, x).reshape(-1)
print(f"B={B} M={M} L={L}: scores rms {s_full.pow(2).mean().sqrt().item():.4f}  f32 fused vs fp64 module {(s_full[:256] - ref64).abs().max().item():.3e}  "
      f"windows == full (f32): {torch.equal(s_full, s_win)}")


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


def kernel_times(fn):
    fn(); torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    names = {3: "gru", 5: "tower", 7: "tail"}
    out = []
    for k, nm in names.items():
        tot, n = _lib.profile_collect(k)
        out.append(f"{nm} {tot / 5 * 1e3:.0f} us ({n // 5} launches)")
    return "  ".join(out)


with torch.no_grad():
    print(f"  f32     {bench(lambda: fv.forward_candidates(onehot, cand, x)):7.3f} ms  [{kernel_times(lambda: fv.forward_candidates(onehot, cand, x))}]")
    for mode in ("f16x3", "bf16x3", "f16", "bf16"):
        fv.precision = mode
        s_lp = fv.forward_candidates(onehot, cand, x).reshape(-1)
        s_lp_full = fv.forward_tokens(cand.view(B * M, L)).reshape(-1)
        again = fv.forward_candidates(onehot, cand, x).reshape(-1)
        t = bench(lambda: fv.forward_candidates(onehot, cand, x))
        print(f"  {mode:7s} {t:7.3f} ms  max|lp - f32| {(s_lp - s_full).abs().max().item():.3e}  max|lp - fp64| {(s_lp[:256] - ref64).abs().max().item():.3e}  "
              f"windows == full: {torch.equal(s_lp, s_lp_full)}  deterministic {torch.equal(s_lp, again)}  "
              f"argmax over M agrees with f32: {(s_lp.view(B, M).argmax(1) == s_full.view(B, M).argmax(1)).float().mean().item():.4f}  [{kernel_times(lambda: fv.forward_candidates(onehot, cand, x))}]")

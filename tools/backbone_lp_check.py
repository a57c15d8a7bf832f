"""Split-precision backbone kernel (svdd_backbone_cnn_lp) against the exact-fp32 one-launch kernel and an fp64 evaluation
of the PyTorch module: max / rms logit error per mode, and time per forward.  Usage: python tools/backbone_lp_check.py [B] [L]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch
from svdd_amd import backbone, config, fused

torch.manual_seed(0)
dev = "cuda"
if "--rg" in sys.argv:                            # A/B: row groups (waves per SIMD) of the transposed kernel: 2, 3 or 4
    from svdd_amd import _lib
    k = sys.argv.index("--rg")
    _lib.set_option(3, 20 + int(sys.argv[k + 1]))
    del sys.argv[k:k + 2]
if "--v1" in sys.argv:                            # A/B: the round-2 kernel (svdd_set_option SVDD_OPT_BACKBONE_LP_VERSION = 1)
    from svdd_amd import _lib
    _lib.set_option(3, 1)
    sys.argv.remove("--v1")
if "--time-only" in sys.argv:                    # ablation experiments: time of one mode, nothing else
    import time as _t
    mode = sys.argv[sys.argv.index("--time-only") + 1]
    B, L = int(sys.argv[1]), int(sys.argv[2])
    cnn = backbone.CNNModel((config.dna_config() if L > 104 else config.rna_config()).model, alphabet_size=5).to(dev).eval()
    x = torch.randint(0, 5, (B, L), device=dev, dtype=torch.uint8)
    pk = fused.pack_backbone(cnn) if mode == "f32" else fused.pack_backbone_lp(cnn, mode)
    fn = (lambda: fused.backbone_cnn(x, pk)) if mode == "f32" else (lambda: fused.backbone_cnn_lp(x, pk))
    for _ in range(5):
        fn()
    if "--events" in sys.argv:                   # kernel time: events around 50 back-to-back launches
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{mode} B={B} L={L}: {e0.elapsed_time(e1) / 50:.3f} ms (events)")
        sys.exit(0)
    torch.cuda.synchronize(); t0 = _t.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print(f"{mode} B={B} L={L}: {(_t.perf_counter() - t0) / 20 * 1e3:.3f} ms")
    sys.exit(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = config.dna_config() if L > 104 else config.rna_config()
cnn = backbone.CNNModel(cfg.model, alphabet_size=5).to(dev).eval()
with torch.no_grad():
    for nm in cnn.norms:
        nm.weight.uniform_(0.5, 1.5); nm.bias.uniform_(-0.3, 0.3)
x = torch.randint(0, 5, (B, L), device=dev, dtype=torch.uint8)
x[:, : L // 2] = 4
with torch.no_grad():
    ref64 = copy.deepcopy(cnn).double()(x[:32], torch.zeros(32, device=dev, dtype=torch.float64), zero_sigma=False).float()
    f32 = fused.backbone_cnn(x, fused.pack_backbone(cnn))
print(f"B={B} L={L}  logits rms {ref64.pow(2).mean().sqrt().item():.3f}  max|f32 kernel - fp64| {(f32[:32] - ref64).abs().max().item():.3e}")


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


pk32 = fused.pack_backbone(cnn)
t32 = bench(lambda: fused.backbone_cnn(x, pk32))
print(f"  f32     {t32:7.3f} ms")
for mode in ("f16x3", "bf16x3", "f16", "bf16"):
    pk = fused.pack_backbone_lp(cnn, mode)
    out = fused.backbone_cnn_lp(x, pk)
    torch.cuda.synchronize()
    again = fused.backbone_cnn_lp(x, pk)
    e64 = (out[:32] - ref64).abs()
    e32 = (out - f32).abs()
    print(f"  {mode:7s} {bench(lambda: fused.backbone_cnn_lp(x, pk)):7.3f} ms   max|lp - fp64| {e64.max().item():.3e}  "
          f"max|lp - f32 kernel| {e32.max().item():.3e}  rms {e32.pow(2).mean().sqrt().item():.3e}  deterministic {torch.equal(out, again)}  "
          f"argmax agreement with f32 {(out[..., :4].argmax(-1) == f32[..., :4].argmax(-1)).float().mean().item():.6f}")

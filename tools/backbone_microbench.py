"""Fused one-launch backbone (svdd_backbone_cnn_f32) vs the layer-wise FusedBackbone and the plain CNNModel."""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from svdd_amd import _lib, fused, config, backbone

torch.manual_seed(0)
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cfg = config.dna_config()
cnn = backbone.CNNModel(cfg.model, alphabet_size=5).to(dev).eval()
with torch.no_grad():                       # non-trivial LayerNorm affine so the test means something
    for nm in cnn.norms:
        nm.weight.uniform_(0.5, 1.5); nm.bias.uniform_(-0.3, 0.3)
x = torch.randint(0, 5, (B, L), device=dev, dtype=torch.uint8)
pk = fused.pack_backbone(cnn)
fb = fused.FusedBackbone(cnn).to(dev)
with torch.no_grad():
    ref = cnn(x, torch.zeros(B, device=dev), zero_sigma=True).contiguous()
    lay = fb(x)
    out = fused.backbone_cnn(x, pk)
torch.cuda.synchronize()
print("max|fused1 - cnn|", (out - ref).abs().max().item(), " max|layerwise - cnn|", (lay - ref).abs().max().item(),
      " scale", ref.abs().max().item())

def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    print("one-launch  %.3f ms" % bench(lambda: fused.backbone_cnn(x, pk)))
    print("layer-wise  %.3f ms" % bench(lambda: fb(x)))
    print("CNNModel    %.3f ms" % bench(lambda: cnn(x, None, zero_sigma=True)))
fl = backbone.CNNModel.flops_per_position() * B * L
print("useful TF at one-launch: %.1f" % (fl / (bench(lambda: fused.backbone_cnn(x, pk)) * 1e-3) / 1e12))

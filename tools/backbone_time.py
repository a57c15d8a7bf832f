"""Per-launch time of svdd_backbone_cnn_f32 / _lp (HIP events over back-to-back launches) and a checksum of its logits (so that
two builds can be compared bit for bit: SVDD_HIP_LIB=<other libsvdd_hip.so>).
Usage: python tools/backbone_time.py [B] [L] [f32|f16x3|bf16x3|f16|bf16]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, backbone, config, fused

if os.environ.get("SVDD_BB_LP_VERSION"):               # A/B of the split-precision backbone kernels (SVDD_OPT_BACKBONE_LP_VERSION)
    _lib.set_option(3, int(os.environ["SVDD_BB_LP_VERSION"]))

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = sys.argv[3] if len(sys.argv) > 3 else "f32"
dev = "cuda:0"
torch.manual_seed(0)
cnn = backbone.CNNModel((config.dna_config() if L > 104 else config.rna_config()).model, alphabet_size=5).to(dev).eval()
x = torch.randint(0, 5, (B, L), device=dev, dtype=torch.uint8)
pk = fused.pack_backbone(cnn) if mode == "f32" else fused.pack_backbone_lp(cnn, mode)
fwd = fused.backbone_cnn if mode == "f32" else fused.backbone_cnn_lp
out = torch.empty(B, L, 5, device=dev)
for rnd in range(2):
    for _ in range(3):
        fwd(x, pk, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fwd(x, pk, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"svdd_backbone_cnn {mode} B={B} L={L}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per forward   "
          f"sha1(logits) {hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]}")

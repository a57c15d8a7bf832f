"""Race soak of svdd_backbone_cnn_lp: many launches on fresh inputs, each repeated, one digest over everything. Run it under two builds
(SVDD_HIP_LIB=<other libsvdd_hip.so>) or two kernel variants (SVDD_BB_LP_VERSION=21 / 22 / 23) and compare the digests; within a
run every repeat of a launch must give the same bits.  Usage: python tools/lp_backbone_soak.py [mode] [launches] [B] [L]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, backbone, config, fused

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
L = int(sys.argv[4]) if len(sys.argv) > 4 else 200
if os.environ.get("SVDD_BB_LP_VERSION"):
    _lib.set_option(3, int(os.environ["SVDD_BB_LP_VERSION"]))
dev = "cuda:0"
torch.manual_seed(11)
cnn = backbone.CNNModel((config.dna_config() if L > 104 else config.rna_config()).model, alphabet_size=5).to(dev).eval()
pk = fused.pack_backbone_lp(cnn, mode)
h = hashlib.sha1()
unstable = 0
g = torch.Generator(device=dev).manual_seed(3)
for i in range(n):
    x = torch.randint(0, 5, (B, L), device=dev, dtype=torch.uint8, generator=g)
    a = fused.backbone_cnn_lp(x, pk).clone()
    b = fused.backbone_cnn_lp(x, pk).clone()
    c = fused.backbone_cnn_lp(x, pk)
    unstable += int(not (torch.equal(a, b) and torch.equal(a, c)))
    h.update(a.cpu().numpy().tobytes())
print(f"{mode} B={B} L={L}: {n} inputs x 3 launches, {unstable} unstable, sha1 {h.hexdigest()[:20]}")

"""Time stamps of every step of one tap of backbone_lp_t_kernel (instrumented build bb_lpt_timing/stamps[_solo0]): ticks between
consecutive stamps, per wave of workgroup 0. Stamp order in a tap: tap head, weight prefetch, then [step x 7 (6), prefetch] x 4."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from svdd_amd import _lib, backbone, config, fused

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
dev = "cuda:0"
torch.manual_seed(0)
cnn = backbone.CNNModel(config.dna_config().model, alphabet_size=5).to(dev).eval()
x = torch.randint(0, 5, (256, 200), device=dev, dtype=torch.uint8)
pk = fused.pack_backbone(cnn) if mode == "f32" else fused.pack_backbone_lp(cnn, mode)
for _ in range(3):
    (fused.backbone_cnn if mode == "f32" else fused.backbone_cnn_lp)(x, pk)
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 32, dtype=np.uint64)
assert _lib.lib().svdd_internal_lpt_dbg(ctypes.c_void_p(buf.ctypes.data)) == 0
st = buf[8192:8192 + 2 * 8 * 64].reshape(2, 8, 64).astype(np.int64)
for blk in range(1):
    for w in (0, 4, 1, 5):
        s = st[blk, w]
        n = int((s > 0).sum())
        d = np.diff(s[:n])
        print(f"wg {blk} wave {w} (rg {w >> 2}): {n} stamps, tap total {s[n - 1] - s[0]} ticks, offset of its first stamp from wave 0's {s[0] - st[blk, 0, 0]}")
        print("   " + " ".join(f"{v:4d}" for v in d))

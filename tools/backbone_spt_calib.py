"""Calibration of csrc/svdd_spt.h: time of one backbone launch with exactly s sequences per tile and 256 x s sequences (one
full round of workgroups), L = 50 and L = 33, fp32 and f16x3.  Usage: python tools/backbone_spt_calib.py [mode ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, backbone, config, fused

dev = "cuda:0"
cnn = backbone.CNNModel(config.rna_config().model, alphabet_size=5).to(dev).eval()
for mode in (sys.argv[1:] or ["f32", "f16x3"]):
    pk = fused.pack_backbone(cnn) if mode == "f32" else fused.pack_backbone_lp(cnn, mode)
    fwd = fused.backbone_cnn if mode == "f32" else fused.backbone_cnn_lp
    for L in (50, 33):
        res = []
        for s in range(1, 208 // L + 1):
            x = torch.randint(0, 5, (256 * s, L), device=dev, dtype=torch.uint8)
            _lib.lib().svdd_set_backbone_packing(-s)
            for _ in range(3):
                fwd(x, pk)
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for _ in range(10):
                fwd(x, pk)
            torch.cuda.synchronize()
            _lib.profile_enable(False)
            tot, k = _lib.profile_collect(6)
            res.append(f"s={s} ({(s * L + 15) // 16} row tiles): {tot / k * 1e3:.0f} us")
        _lib.lib().svdd_set_backbone_packing(0)
        print(f"{mode} L={L}: " + " | ".join(res))

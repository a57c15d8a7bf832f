"""Where a workgroup of backbone_lp_t_kernel spends its cycles (instrumented build: tools/exp_variants.py build bb_lpt_timing,
run with SVDD_HIP_LIB=build/exp/bb_lpt_timing/timing/libsvdd_hip.so): per wave the LayerNorm phases, the (tap, chunk) loop,
the barrier wait after it and the epilogues, and the loop of every layer against its MFMA floor.
Usage: python tools/lpt_phase_timing.py [mode] [B] [--L 50 --spt s]   (--L <= 104: `spt` sequences per tile, B = 256 x spt: one full round)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from svdd_amd import _lib, backbone, config, fused

args = [a for a in sys.argv[1:] if not a.startswith("--")]
mode = args[0] if args else "f16x3"
def _opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


L, spt = _opt("--L", 200), _opt("--spt", 1)
args = [a for a in args if a not in (str(L), str(spt))] if ("--L" in sys.argv or "--spt" in sys.argv) else args
B = int(args[1]) if len(args) > 1 else 256 * spt
dev = "cuda:0"
if os.environ.get("SVDD_BB_LP_VERSION"):               # 21 / 22 / 23: waves per SIMD of the transposed kernel
    _lib.set_option(3, int(os.environ["SVDD_BB_LP_VERSION"]))
torch.manual_seed(0)
cnn = backbone.CNNModel((config.dna_config() if L > 104 else config.rna_config()).model, alphabet_size=5).to(dev).eval()
if L <= 104:
    _lib.lib().svdd_set_backbone_packing(-spt)
x = torch.randint(0, 5, (B, L), device=dev, dtype=torch.uint8)
pk = fused.pack_backbone(cnn) if mode == "f32" else fused.pack_backbone_lp(cnn, mode)
fwd = fused.backbone_cnn if mode == "f32" else fused.backbone_cnn_lp
for _ in range(30):                                    # clocks ramp for tens of ms after idle
    fwd(x, pk)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fwd(x, pk)
e1.record(); torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 32, dtype=np.uint64)
rc = _lib.lib().svdd_internal_lpt_dbg(ctypes.c_void_p(buf.ctypes.data))
assert rc == 0, rc
d = buf.reshape(256, 8, 32).astype(np.float64)[:min(B // spt, 256)]
tot = d[:, :, :4].sum(-1)
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"{mode} B={B} L={L} spt={spt}: launch {us:.1f} us (20 back to back); cycles per wave (mean over workgroups): total {tot.mean():.0f} = {tot.mean() / us / 1e3:.2f} GHz if the launch were all of it")
for k, name in enumerate(("LayerNorm finalize + image write", "(tap, chunk) loop", "wait at the first LayerNorm barrier", "epilogue, statistics, first layer")):
    print(f"  {name:34s} rg0 {d[:, :4, k].mean():9.0f}  rg1 {d[:, 4:, k].mean():9.0f}   ({d[:, :, k].mean() / tot.mean() * 100:4.1f} %)")
if d[:, :, 4:7].sum() > 0:
    for k, name in ((4, "steps between weight prefetches"), (5, "weight prefetch + counted wait"), (6, "tap head (schedule, addresses)")):
        print(f"  {name:34s} rg0 {d[:, :4, k].mean():9.0f}  rg1 {d[:, 4:, k].mean():9.0f}")
if "--brief" in sys.argv:
    print("  loop cycles rg0 / rg1 at layers 0, 8, 12, 16: " + "   ".join(f"{d[:, :4, 8 + ly].mean():7.0f} / {d[:, 4:, 8 + ly].mean():7.0f}" for ly in (0, 8, 12, 16)))
    sys.exit(0)
dil = [1] * 4 + [1] * 4 + [4] * 4 + [16] * 4 + [64] * 4
nl = len(cnn.convs)
print("  layer  dilation  loop cycles rg0 / rg1   MFMA floor of the SIMD (16 cycles each)   occupancy while rg0 is in the loop")
for ly in range(nl + 1):
    dl = cnn.convs[ly].dilation[0] if ly < nl else 0
    # live (tap, row tile) pairs of both row groups
    live = [0, 0]
    for t in range(9 if ly < nl else 1):
        dd = (t - 4) * dl * spt if ly < nl else 0
        lo, hi = max(0, -dd), min(L * spt, L * spt - dd)
        for r in range(13):
            if lo < hi and lo < 16 * r + 16 and hi > 16 * r:
                live[r & 1] += 1
    if ly == nl:
        nt = (L * spt + 15) // 16
        live = [(nt + 1) // 2, nt // 2]
    npass = 3 if mode.endswith("x3") else 1
    floor = (live[0] + live[1]) * 4 * 2 * (8 * 32 if mode == "f32" else npass * 16)       # f32: 8 v_mfma_f32_16x16x4_f32 of 32 cycles per (tile, column tile, chunk, tap)
    print(f"  {ly:5d}  {dl:8d}  {d[:, :4, 8 + ly].mean():9.0f} / {d[:, 4:, 8 + ly].mean():9.0f}   {floor:9d}   {floor / d[:, :4, 8 + ly].mean():.2f}")

"""Fuzz of the interleaved short-sequence backbone kernels (round 5): random L <= 104, random batch, device-side count and row
index lists, fp32 against the PyTorch module (2e-5) and every packing against the one-sequence-per-tile packing (same bits);
f16x3 / bf16 against fp32 within their tolerances and packing-invariant. Usage: python tools/interleave_fuzz.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, backbone, config, fused

dev = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(0)
torch.manual_seed(0)
cnn = backbone.CNNModel(config.rna_config().model, alphabet_size=5).to(dev).eval()
pk = fused.pack_backbone(cnn)
lp = {m: fused.pack_backbone_lp(cnn, m) for m in ("f16x3", "bf16")}
worst = {"f32_vs_torch": 0.0, "f16x3_vs_f32": 0.0, "bf16_vs_f32": 0.0}
Ls = [104, 69, 52, 51, 50, 34, 26, 13, 7, 1]
for c in range(cases):
    L = Ls[c] if c < len(Ls) else int(torch.randint(1, 105, (1,), generator=g))
    n = int(torch.randint(1, 1400, (1,), generator=g))
    x = torch.randint(0, 5, (n, L), generator=g).to(torch.uint8).to(dev)
    live = int(torch.randint(1, n + 1, (1,), generator=g))
    cnt = torch.tensor([live], dtype=torch.int32, device=dev)
    idx = torch.randperm(n, generator=g)[:live].sort().values.to(torch.int32).to(dev)
    with torch.no_grad():
        ref = cnn(x.long(), torch.zeros(n, device=dev))
    outs = {}
    for full in (-1, 0):                              # one sequence per tile ; the planner's choice
        _lib.lib().svdd_set_backbone_packing(full)
        a = fused.backbone_cnn(x, pk)
        b = fused.backbone_cnn(x, pk, count=cnt, row_idx=idx, scatter=True, out=torch.zeros(n, L, 5, device=dev))
        d = fused.backbone_cnn(x, pk, count=cnt, out=torch.zeros(n, L, 5, device=dev))
        al = {m: fused.backbone_cnn_lp(x, lp[m]) for m in lp}
        bl = {m: fused.backbone_cnn_lp(x, lp[m], count=cnt, row_idx=idx, scatter=False, out=torch.zeros(n, L, 5, device=dev)) for m in lp}
        outs[full] = (a, b, d, al, bl)
    _lib.lib().svdd_set_backbone_packing(0)
    a, b, d, al, bl = outs[0]
    a1, b1, d1, al1, bl1 = outs[-1]
    assert torch.equal(a, a1) and torch.equal(b, b1) and torch.equal(d, d1), (L, n, "fp32 packing")
    assert torch.equal(b[idx.long()], a[idx.long()]) and torch.equal(d[:live], a[:live]), (L, n, "fp32 compaction")
    for m in lp:
        assert torch.equal(al[m], al1[m]) and torch.equal(bl[m], bl1[m]), (L, n, m, "lp packing")
        assert torch.equal(bl[m][:live], al[m][idx.long()]), (L, n, m, "lp compaction")
    worst["f32_vs_torch"] = max(worst["f32_vs_torch"], float((a - ref).abs().max()))
    worst["f16x3_vs_f32"] = max(worst["f16x3_vs_f32"], float((al["f16x3"] - a).abs().max()))
    worst["bf16_vs_f32"] = max(worst["bf16_vs_f32"], float((al["bf16"] - a).abs().max()))
    assert worst["f32_vs_torch"] <= 2e-5 and worst["f16x3_vs_f32"] <= 2e-5 and worst["bf16_vs_f32"] <= 0.1, (L, n, worst)
print(f"{cases} cases ok (L in 1..104, n in 1..1399, device-side counts, index lists, three precisions): max errors", worst)

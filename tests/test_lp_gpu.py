"""-m gpu: the split-precision net kernels (svdd_amd/csrc/svdd_lp_*.hip, Diffusion.precision) against the exact-fp32
kernels and fp64 evaluations of the PyTorch modules.

Tolerances: the north star allows 1e-4 on reward / soft-value tensors. The x3 modes (operands split hi + lo, three
MFMAs per product) must meet it on logits AND scores at the full BASELINE configs[1] size; the one-pass modes are
plain 16-bit arithmetic and only have to stay within the error such arithmetic implies (bounds below; the measured
values are recorded in profiles/r02_lp_check.txt). The sampler kernels are untouched by the precision knob: whatever
logits / scores the nets produce, the decoded tokens must equal the oracle's on the recorded logits / scores."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
X3 = ("f16x3", "bf16x3")
TOL_LOGITS = {"f16x3": 1e-4, "bf16x3": 1e-4, "f16": 2e-2, "bf16": 1e-1}
TOL_SCORES = {"f16x3": 1e-4, "bf16x3": 1e-4, "f16": 2e-3, "bf16": 1e-2}


@pytest.fixture(scope="module")
def nets():
    from svdd_amd import synthetic
    return synthetic.build("dna", DEV)


def _perturbed_cnn(L):
    from svdd_amd import backbone, config
    torch.manual_seed(5)
    cfg = config.dna_config() if L > 104 else config.rna_config()
    cnn = backbone.CNNModel(cfg.model, alphabet_size=5).to(DEV).eval()
    with torch.no_grad():
        for nm in cnn.norms:                         # non-trivial LayerNorm affine
            nm.weight.uniform_(0.5, 1.5)
            nm.bias.uniform_(-0.3, 0.3)
    return cnn


@pytest.mark.parametrize("mode", ["f16x3", "bf16x3", "f16", "bf16"])
@pytest.mark.parametrize("B,L", [(256, 200), (37, 50), (3, 200), (9, 128)])
def test_backbone_lp_vs_f32_and_fp64(mode, B, L):
    import copy
    from svdd_amd import fused
    cnn = _perturbed_cnn(L)
    x = torch.randint(0, 5, (B, L), device=DEV, dtype=torch.uint8)
    x[:, : L // 3] = 4
    f32 = fused.backbone_cnn(x, fused.pack_backbone(cnn))
    pk = fused.pack_backbone_lp(cnn, mode)
    out = fused.backbone_cnn_lp(x, pk)
    again = fused.backbone_cnn_lp(x, pk)
    torch.cuda.synchronize()
    assert torch.equal(out, again)
    assert torch.isfinite(out).all()
    err = (out - f32).abs().max().item()
    assert err <= TOL_LOGITS[mode], (mode, err)
    nb = min(B, 16)
    with torch.no_grad():
        ref64 = copy.deepcopy(cnn).double()(x[:nb], torch.zeros(nb, device=DEV, dtype=torch.float64)).float()
    e64 = (out[:nb] - ref64).abs().max().item()
    assert e64 <= TOL_LOGITS[mode], (mode, e64)
    if mode == "f16x3":                              # at least as good as the fp32 kernel itself (wide MFMA accumulate)
        assert e64 <= 2.0 * (f32[:nb] - ref64).abs().max().item() + 1e-6


@pytest.mark.parametrize("mode", ["f16x3", "bf16"])
@pytest.mark.parametrize("L", [200, 128, 105])
def test_backbone_lp_transposed_variants_same_bits(mode, L):
    """backbone_lp_t_kernel with one, two (default) and three waves per SIMD — 13 / 7 + 6 / 5 + 4 + 4 row tiles per wave, each a
    generated step listing (tools/gen_lpt_taps.py) in an all-live and a partly-live form — accumulates every output in the same
    order: the three must agree bit for bit, at lengths that leave 0, 5 and 7 whole row tiles in the zero padding."""
    from svdd_amd import _lib, fused
    cnn = _perturbed_cnn(L)
    x = torch.randint(0, 5, (19, L), device=DEV, dtype=torch.uint8)
    pk = fused.pack_backbone_lp(cnn, mode)
    outs = {}
    try:
        for v in (22, 21, 23):
            _lib.check(_lib.lib().svdd_set_option(3, v), "svdd_set_option")
            outs[v] = fused.backbone_cnn_lp(x, pk).clone()
    finally:
        _lib.check(_lib.lib().svdd_set_option(3, 22), "svdd_set_option")
    torch.cuda.synchronize()
    assert torch.isfinite(outs[22]).all()
    assert torch.equal(outs[21], outs[22]) and torch.equal(outs[23], outs[22])


@pytest.mark.parametrize("dils", [(4, 1, 64, 16, 1), (1, 64, 4, 1, 16), (64, 16, 4, 1, 1)])
def test_backbone_lp_other_dilation_orders(dils, monkeypatch):
    """The kernels take the dilation of every layer as data. With the reference's order the last conv layer has dilation 64, a
    shift by four row tiles — which hides any read / write overlap between the two row groups of a SIMD at the one layer
    boundary without a LayerNorm (the 1x1 stage) behind the tile parity. Other orders do not: every variant must still match the
    fp32 kernel, and each other bit for bit."""
    from svdd_amd import _lib, backbone, fused
    monkeypatch.setattr(backbone, "DILATIONS", dils)
    cnn = _perturbed_cnn(200)
    assert [c.dilation[0] for c in cnn.convs][-1] == dils[-1]
    x = torch.randint(0, 5, (40, 200), device=DEV, dtype=torch.uint8)
    f32 = fused.backbone_cnn(x, fused.pack_backbone(cnn))
    for mode in ("f16x3", "bf16"):
        pk = fused.pack_backbone_lp(cnn, mode)
        outs = {}
        try:
            for v in (22, 21, 23):
                _lib.check(_lib.lib().svdd_set_option(3, v), "svdd_set_option")
                outs[v] = torch.stack([fused.backbone_cnn_lp(x, pk) for _ in range(3)])     # three launches each: a race is timing
        finally:
            _lib.check(_lib.lib().svdd_set_option(3, 22), "svdd_set_option")
        torch.cuda.synchronize()
        assert (outs[22][0] - f32).abs().max().item() <= TOL_LOGITS[mode], (mode, dils)
        for v in (22, 21, 23):
            assert torch.equal(outs[v], outs[22][:1].expand_as(outs[v])), (mode, dils, v)


def test_backbone_lp_row_placement_independent():
    """A sequence's logits do not depend on which tile / batch position it lands in (needed by the exact work-skipping
    paths, which re-pack rows)."""
    from svdd_amd import fused
    cnn = _perturbed_cnn(50)
    x = torch.randint(0, 5, (64, 50), device=DEV, dtype=torch.uint8)
    pk = fused.pack_backbone_lp(cnn, "f16x3")
    full = fused.backbone_cnn_lp(x, pk)
    perm = torch.randperm(64, device=DEV)
    part = fused.backbone_cnn_lp(x[perm][:23].contiguous(), pk)
    assert torch.equal(part, full[perm][:23])


def _candidates(B, M, L, seed=1, p_flip=0.015):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.where(torch.rand(B, L, device=DEV, generator=g) < 0.7, 4,
                    torch.randint(0, 4, (B, L), device=DEV, generator=g)).to(torch.uint8)
    cand = x[:, None, :].repeat(1, M, 1)
    flip = (torch.rand(B, M, L, device=DEV, generator=g) < p_flip) & (cand == 4)
    cand = torch.where(flip, torch.randint(0, 4, (B, M, L), device=DEV, generator=g).to(torch.uint8), cand).contiguous()
    return x, cand


@pytest.mark.parametrize("mode", ["f16x3", "bf16x3", "f16", "bf16"])
@pytest.mark.parametrize("B,M,L", [(256, 10, 200), (5, 3, 200), (16, 4, 50)])
def test_value_net_lp_vs_f32(nets, mode, B, M, L):
    from svdd_amd import fused, ops
    _, emb, head, _ = nets
    g = torch.Generator().manual_seed(3)
    for m in emb.modules():                          # non-trivial BatchNorm statistics (folded into the tower weights)
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g).to(DEV) * 0.1)
            m.running_var.copy_((torch.rand(m.num_features, generator=g) + 0.5).to(DEV))
    fv = fused.FusedValueNet(emb, head).to(DEV).eval()
    x, cand = _candidates(B, M, L)
    onehot = ops.transform_samples(cand.view(B * M, L))
    with torch.no_grad():
        s32 = fv(onehot).reshape(-1)
        fv.precision = mode
        s_tok = fv.forward_tokens(cand.view(B * M, L)).reshape(-1)
        s_oh = fv(onehot).reshape(-1)                # the generic entry converts the one-hot back to tokens
        assert torch.equal(s_tok, s_oh)
        err = (s_tok - s32).abs().max().item()
        assert err <= TOL_SCORES[mode], (mode, err)
        if fv.candidates_ok(L, M):
            s_win = fv.forward_candidates(onehot, cand, x).reshape(-1)
            assert torch.equal(s_win, s_tok)         # parent-sharing windows are bit-identical in every mode


@pytest.mark.parametrize("mode", ["f16x3", "bf16x3"])
@pytest.mark.parametrize("n,L", [(16, 200), (37, 50), (1, 7), (300, 50)])
def test_gru_lp_vs_fp64(mode, n, L):
    from svdd_amd import _lib, fused
    torch.manual_seed(n)
    gru = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).to(DEV).eval()
    with torch.no_grad():
        for p in gru.parameters():
            p.mul_(2.0)                              # larger gates: exercise saturation
    x = torch.randn(n, L, 64, device=DEV).relu()
    wp, bp, inv = fused.pack_gru_lp(gru, mode)
    out = fused.gru_bidir_lp(x, wp.to(DEV), bp.to(DEV), inv.to(DEV), _lib.PRECISIONS[mode])
    g64 = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).double()
    g64.load_state_dict({k: v.double().cpu() for k, v in gru.state_dict().items()})
    with torch.no_grad():
        ref64 = g64(x.double().cpu())[0]
    assert (out[0].double().cpu() - ref64[:, :, :64]).abs().max().item() <= 5e-5
    assert (out[1].double().cpu() - ref64[:, :, 64:]).abs().max().item() <= 5e-5


@pytest.mark.parametrize("mode", ["f16x3", "bf16"])
@pytest.mark.parametrize("rng_mode", ["replay", "philox"])
def test_lp_decode_sampler_still_exact(mode, rng_mode):
    """The precision knob changes the nets only: on the logits / scores the 16-bit nets produced, the oracle recomputes
    every propose / select / finalize step and must arrive at the same tokens."""
    from oracle import svdd_oracle as orc
    from svdd_amd import synthetic
    model, emb, head, _ = synthetic.build("dna", DEV)
    B, L, M, S = 6, 200, 3, 12
    sched = model._schedule(S, 1e-5)[0]
    model.precision, model.rng_mode, model.philox_seed, model.trace = mode, rng_mode, 77, []
    torch.manual_seed(0)
    x_gpu = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
    torch.cuda.synchronize()
    trace = [(lg.cpu().numpy(), None if sc is None else sc.cpu().numpy()) for lg, sc in model.trace]
    uf = None
    if rng_mode == "replay":
        torch.manual_seed(0)
        uf = lambda i, M_, B_, L_: torch.rand(M_, B_, 5, L_).numpy().transpose(0, 1, 3, 2).copy()   # noqa: E731
    x_cpu = orc.replay_controlled_sample(trace, sched, B, L, M, uniform_fn=uf, seed=77)
    assert np.array_equal(x_gpu.cpu().numpy(), x_cpu)
    assert int(x_gpu.max()) <= 3


def test_precision_knob_switches_and_restores(nets):
    """f32 -> f16x3 -> f32 on one model object: the fp32 decode is bit-identical before and after, the f16x3 one differs
    from it in bits of the logits but stays within 1e-4."""
    model, emb, head, _ = nets
    model.rng_mode, model.philox_seed = "philox", 3
    x = torch.randint(0, 5, (8, 200), device=DEV, dtype=torch.uint8)
    model.precision = "f32"
    a = model._backbone_logits(x).clone()
    model.precision = "f16x3"
    b = model._backbone_logits(x).clone()
    model.precision = "f32"
    c = model._backbone_logits(x).clone()
    assert torch.equal(a, c)
    assert not torch.equal(a, b) and (a - b).abs().max().item() <= 1e-4

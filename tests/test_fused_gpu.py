"""-m gpu: the MI355X formulations of the nets (svdd_amd/fused.py: channels-last convs, folded BN,
hand-written MFMA GRU kernel) compute the same functions as the plain PyTorch modules.
fp32 vs fp32: tolerance 2e-5 absolute on O(1) activations (re-association + hardware exp/rcp in the
GRU gates); the north-star tolerance for soft values is 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def nets():
    from svdd_amd import synthetic
    return synthetic.build("dna", DEV)


@pytest.mark.parametrize("mode", [0, 1, 2])         # 0: producer / consumer waves (default), 1: one wave does both halves,
                                                    # 2: both directions per workgroup
@pytest.mark.parametrize("n,L", [(16, 200), (37, 50), (2560, 200), (1, 7), (300, 50), (4000, 20)])
def test_gru_kernel_vs_torch(n, L, mode):
    from svdd_amd import _lib
    from svdd_amd.fused import gru_bidir, pack_gru
    _lib.lib().svdd_gru_set_mode(mode)
    torch.manual_seed(n)
    gru = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).to(DEV).eval()
    with torch.no_grad():
        for p in gru.parameters():
            p.mul_(2.0)                                  # larger gates: exercise saturation
    x = torch.randn(n, L, 64, device=DEV)
    wpack, bpack = pack_gru(gru)
    with torch.no_grad():
        ref = gru(x)[0]
    out = gru_bidir(x, wpack.to(DEV), bpack.to(DEV))
    torch.cuda.synchronize()
    _lib.lib().svdd_gru_set_mode(0)
    assert (out[0] - ref[:, :, :64]).abs().max().item() <= 2e-5
    assert (out[1] - ref[:, :, 64:]).abs().max().item() <= 2e-5
    # the recurrence itself, independently of MIOpen: fp64 on the CPU
    g64 = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).double()
    g64.load_state_dict({k: v.double().cpu() for k, v in gru.state_dict().items()})
    nn_ = min(n, 8)
    with torch.no_grad():
        ref64 = g64(x[:nn_].double().cpu())[0]
    assert (out[0, :nn_].double().cpu() - ref64[:, :, :64]).abs().max().item() <= 2e-5
    assert (out[1, :nn_].double().cpu() - ref64[:, :, 64:]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("n,L,live", [(2560, 200, None), (37, 50, None), (1, 7, None), (3, 1, None), (5, 2, None), (17, 3, None),
                                      (320, 200, 200), (64, 33, 17)])
def test_gru_producer_consumer_same_bits(n, L, live):
    """gru_pc_kernel (producer waves compute the input projections one step ahead, consumer waves run the recurrence) keeps
    every accumulator's order of products: same bits as the single-role kernel, also on a compacted batch."""
    from svdd_amd import _lib
    from svdd_amd.fused import gru_bidir, pack_gru
    torch.manual_seed(n + L)
    gru = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).to(DEV).eval()
    x = torch.randn(n, L, 64, device=DEV)
    wpack, bpack = pack_gru(gru)
    cnt = None if live is None else torch.tensor([live], dtype=torch.int32, device=DEV)
    outs = []
    try:
        for mode in (1, 4):
            _lib.lib().svdd_gru_set_mode(mode)
            outs.append(gru_bidir(x, wpack.to(DEV), bpack.to(DEV), count=cnt, out=torch.zeros(2, n, L, 64, device=DEV)))
    finally:
        _lib.lib().svdd_gru_set_mode(0)
    assert torch.equal(outs[0], outs[1])
    if live is not None:
        assert float(outs[1][:, live:].abs().max()) == 0.0              # rows beyond the count are not touched


def test_fused_value_net_vs_plain(nets):
    from svdd_amd.fused import FusedValueNet
    model, emb, head, reward = nets
    # non-trivial BatchNorm statistics
    g = torch.Generator().manual_seed(3)
    for m in emb.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g).to(DEV) * 0.1)
            m.running_var.copy_((torch.rand(m.num_features, generator=g) + 0.5).to(DEV))
    fv = FusedValueNet(emb, head).to(DEV).eval()
    tok = torch.randint(0, 5, (96, 200), device=DEV)
    oh = (torch.nn.functional.one_hot(tok.clamp(max=3), 4) * (tok != 4)[..., None]).float()
    with torch.no_grad():
        ref = head(emb(oh))
        fv(oh)                                          # warm-up (MIOpen solver search on first use)
        out = fv(oh)
        out2 = fv(oh)
        out_t = fv(oh.transpose(1, 2).contiguous())     # reward-model layout
    assert out.shape == ref.shape == (96, 1, 1)
    assert (out - ref).abs().max().item() <= 2e-5
    print("repeat-call max diff", (out - out2).abs().max().item(), "layout max diff", (out - out_t).abs().max().item())
    # MIOpen's split-K convolutions use atomics at small batch: call-to-call noise at the 1e-8 level
    assert (out - out2).abs().max().item() <= 1e-6
    assert (out - out_t).abs().max().item() <= 2e-6
    model.clear_fused()


@pytest.mark.parametrize("B", [32, 256])          # 256: the hand-written conv path; 32: MIOpen convs
def test_fused_backbone_vs_plain(nets, B):
    from svdd_amd.fused import FusedBackbone
    model = nets[0]
    fb = FusedBackbone(model.backbone).to(DEV).eval()
    x = torch.randint(0, 5, (B, 200), device=DEV).to(torch.uint8)
    with torch.no_grad():
        ref = model.backbone(x, None, zero_sigma=True)
        out = fb(x)
    assert out.shape == (B, 200, 5) and out.is_contiguous()
    assert (out - ref).abs().max().item() <= 2e-5
    if B >= 192:
        assert torch.equal(out, fb(x))                  # our kernels are run-to-run deterministic (MIOpen's split-K is not)


def test_engine_uses_fused_nets_and_rows_are_batch_invariant(nets):
    """Row independence of the fused value net: scoring rows in one [B*M] call or in chunks gives
    the same scores bit-for-bit (needed for the batched-vs-per-candidate deviation to be exact)."""
    model, emb, head, _ = nets
    model.fuse_nets = True
    fn = model.value_callable(emb, head)
    assert isinstance(fn, torch.nn.Module)
    oh = torch.zeros(64, 200, 4, device=DEV)
    oh.scatter_(2, torch.randint(0, 4, (64, 200, 1), device=DEV), 1.0)
    with torch.no_grad():
        full = fn(oh)
        parts = torch.cat([fn(oh[:16]), fn(oh[16:48]), fn(oh[48:])])
    assert (full - parts).abs().max().item() <= 1e-6


@pytest.mark.parametrize("n,L,cin,cout,T,dil", [
    (8, 200, 128, 128, 9, 1), (8, 200, 128, 128, 9, 4), (5, 200, 128, 128, 9, 16), (5, 200, 128, 128, 9, 64),
    (9, 50, 128, 128, 9, 64), (7, 50, 128, 128, 9, 4), (33, 200, 64, 64, 5, 1), (6, 50, 64, 64, 5, 1),
    (3, 37, 64, 128, 3, 2), (3, 224, 128, 64, 1, 1)])
def test_conv1d_cl_kernel_vs_torch(n, L, cin, cout, T, dil):
    from svdd_amd import _lib
    from svdd_amd.fused import conv1d_cl, pack_conv
    torch.manual_seed(n * L + T)
    x = torch.randn(n, L, cin, device=DEV)
    w = torch.randn(cout, cin, T, device=DEV) / (cin * T) ** 0.5
    y = conv1d_cl(x, pack_conv(w), cout, T, dil)
    ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), padding=(T // 2) * dil, dilation=dil).transpose(1, 2)
    err = (y.double() - ref).abs().max().item()
    assert y.shape == (n, L, cout) and err <= 2e-5, err
    _lib.lib().svdd_conv1d_set_dynamic(1)              # the generic (runtime-scheduled) kernel on the same problem
    try:
        y2 = conv1d_cl(x, pack_conv(w), cout, T, dil)
    finally:
        _lib.lib().svdd_conv1d_set_dynamic(0)
    assert (y2.double() - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("n,L,cin,cout,T,dil,act", [(6, 200, 128, 128, 9, 4, 0), (6, 200, 128, 128, 9, 64, 0),
                                                    (9, 200, 64, 64, 5, 1, 1), (9, 50, 64, 64, 5, 1, 1), (5, 200, 64, 64, 5, 1, 2)])
def test_conv1d_cl_fused_epilogue(n, L, cin, cout, T, dil, act):
    from svdd_amd.fused import conv1d_cl, pack_conv
    torch.manual_seed(act + n)
    x = torch.randn(n, L, cin, device=DEV)
    w = torch.randn(cout, cin, T, device=DEV) / (cin * T) ** 0.5
    b = torch.randn(cout, device=DEV)
    fp = torch.randn(n, L, cout, device=DEV)
    y = conv1d_cl(x, pack_conv(w), cout, T, dil, bias=b, f_prev=fp, act=act)
    c = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), b.double(), padding=(T // 2) * dil, dilation=dil).transpose(1, 2)
    ref = {0: torch.relu(c) + fp.double(), 1: torch.relu(c + fp.double()), 2: c + fp.double()}[act]
    assert (y.double() - ref).abs().max().item() <= 2e-5
    # + the next layer's LayerNorm in the same kernel
    tb, gm, bt = torch.randn(cout, device=DEV), torch.randn(cout, device=DEV), torch.randn(cout, device=DEV)
    y2, hn = conv1d_cl(x, pack_conv(w), cout, T, dil, bias=b, f_prev=fp, act=act, ln=(tb, gm, bt))
    ref_hn = torch.nn.functional.layer_norm(ref + tb.double(), (cout,), gm.double(), bt.double(), eps=1e-5)
    assert torch.equal(y2, y)
    assert (hn.double() - ref_hn).abs().max().item() <= 5e-5


@pytest.mark.parametrize("n,L", [(7, 200), (2560, 200), (9, 50), (3, 37), (5, 208)])
def test_fused_conv_tower_vs_layerwise(nets, n, L):
    """One-launch LDS-resident conv tower vs the layer-by-layer path (MIOpen convs + epilogue kernels) and vs the
    plain PyTorch modules."""
    from svdd_amd.fused import FusedValueNet
    model, emb, head, _ = nets
    g = torch.Generator().manual_seed(5)
    for m in emb.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g).to(DEV) * 0.1)
            m.running_var.copy_((torch.rand(m.num_features, generator=g) + 0.5).to(DEV))
    fv = FusedValueNet(emb, head).to(DEV).eval()
    assert fv.tower_ok
    tok = torch.randint(0, 5, (n, L), device=DEV)
    oh = (torch.nn.functional.one_hot(tok.clamp(max=3), 4) * (tok != 4)[..., None]).float()
    with torch.no_grad():
        fv.use_fused_tower = True
        a = fv(oh)
        a2 = fv(oh)
        fv.use_fused_tower = False
        b = fv(oh)
        ref = head(emb(oh[: min(n, 64)]))
    assert torch.equal(a, a2)                                        # deterministic
    assert (a - b).abs().max().item() <= 2e-5
    assert (a[: min(n, 64)] - ref).abs().max().item() <= 2e-5
    model.clear_fused()


@pytest.mark.parametrize("n,L", [(7, 200), (3, 187), (1, 208), (9, 50), (5, 33), (2, 9), (300, 200)])
def test_one_launch_backbone_vs_plain(n, L):
    """svdd_backbone_cnn_f32 (whole backbone in one launch) against the plain CNNModel on ragged shapes: L = 200
    (one sequence per workgroup, clamped tap addressing, dead-tile skipping), several short sequences per tile,
    L shorter than the widest dilation (dead taps), and more sequences than fit the tile grid evenly."""
    from svdd_amd import backbone, config, fused
    torch.manual_seed(5)
    cnn = backbone.CNNModel(config.dna_config().model, alphabet_size=5).to(DEV).eval()
    with torch.no_grad():
        for nm in cnn.norms:
            nm.weight.uniform_(0.5, 1.5)
            nm.bias.uniform_(-0.3, 0.3)
    x = torch.randint(0, 5, (n, L), device=DEV, dtype=torch.uint8)
    pk = fused.pack_backbone(cnn)
    with torch.no_grad():
        ref = cnn(x, torch.zeros(n, device=DEV), zero_sigma=True).contiguous()
        out = fused.backbone_cnn(x, pk)
    assert out.shape == (n, L, 5)
    assert (out - ref).abs().max().item() <= 2e-5
    assert torch.equal(out, fused.backbone_cnn(x, pk))                # deterministic
    # rows are independent: a sub-batch gives the same bits whatever tile it lands in
    if n >= 5:
        assert torch.equal(out[2:5], fused.backbone_cnn(x[2:5].contiguous(), pk))


@pytest.mark.parametrize("n,L", [(4, 200), (1, 208), (33, 200), (64, 200), (65, 187), (128, 200), (7, 105), (300, 200), (257, 200), (384, 200)])
def test_backbone_on_several_workgroups_per_sequence_same_bits(n, L):
    """Round 4: small batches of one-sequence tiles run svdd_backbone_cnn_f32 on 2 / 4 workgroups per sequence (row-split,
    the LayerNorm'd image of every layer exchanged through a caller-owned scratch, agent-scope group barriers). Same bits as
    the one-workgroup kernel (A/B through SVDD_OPT_BACKBONE_SPLIT), for every split that fits, repeated launches (the
    arrival counters are reset per launch), and no group barrier ever timed out."""
    import ctypes
    from svdd_amd import _lib, backbone, config, fused
    torch.manual_seed(6)
    cnn = backbone.CNNModel(config.dna_config().model, alphabet_size=5).to(DEV).eval()
    with torch.no_grad():
        for nm in cnn.norms:
            nm.weight.uniform_(0.5, 1.5)
            nm.bias.uniform_(-0.3, 0.3)
    x = torch.randint(0, 5, (n, L), device=DEV, dtype=torch.uint8)
    x[0, L // 2:] = 4
    pk = fused.pack_backbone(cnn)
    lib = _lib.lib()
    try:
        _lib.check(lib.svdd_set_option(7, 1), "split off")
        one = fused.backbone_cnn(x, pk).clone()
        outs = {}
        for R in (2, 4):
            if R * (n % 256 if n > 256 else n) > 256:      # (n > 256: the tail round after the whole rounds of one-workgroup tiles)
                continue
            _lib.check(lib.svdd_set_option(7, R), "split forced")
            outs[R] = [fused.backbone_cnn(x, pk).clone() for _ in range(3)]
        _lib.check(lib.svdd_set_option(7, 0), "split auto")
        auto = fused.backbone_cnn(x, pk).clone()
        torch.cuda.synchronize()
    finally:
        lib.svdd_set_option(7, 0)
    err = ctypes.c_int(-1)
    _lib.check(lib.svdd_backbone_split_status(ctypes.byref(err)), "status")
    assert err.value == 0
    assert torch.isfinite(one).all()
    for R, got in outs.items():
        for o in got:
            assert torch.equal(o, one), (R, float((o - one).abs().max()))
    assert torch.equal(auto, one)
    with torch.no_grad():
        ref = cnn(x, torch.zeros(n, device=DEV), zero_sigma=True).contiguous()
    assert (one - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("n,L,T", [(5, 200, 1), (64, 200, 1), (7, 50, 1), (3, 33, 2), (9, 17, 4), (2, 1, 3)])
def test_value_tail_vs_torch(n, L, T):
    """svdd_value_tail_f32 (direction sum + LayerNorm + dense1 + ReLU + collapsed dense2/head + mean over length)
    against the same maps as plain torch ops, incl. ragged last row tiles and several tasks."""
    from svdd_amd import fused
    g = torch.Generator(device="cpu").manual_seed(11)
    h = torch.randn(2, n, L, 64, generator=g).to(DEV)
    w1 = (torch.randn(128, 64, generator=g) * 0.2).to(DEV)
    b1 = (torch.randn(128, generator=g) * 0.1).to(DEV)
    gamma = (torch.rand(64, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(64, generator=g) * 0.2).to(DEV)
    w_eff = (torch.randn(128, T, generator=g) * 0.2).to(DEV)
    b_eff = torch.randn(T, generator=g).to(DEV)
    wp, bf = fused.pack_tail(w1, b1, gamma, beta)
    out = fused.value_tail(h, wp, bf, w_eff, b_eff)
    hn = torch.nn.functional.layer_norm((h[0] + h[1]).double(), (64,), gamma.double(), beta.double(), 1e-5)
    ref = (torch.relu(hn @ w1.double().t() + b1.double()) @ w_eff.double()).mean(dim=1) + b_eff.double()
    assert out.shape == (n, T)
    assert (out.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(out, fused.value_tail(h, wp, bf, w_eff, b_eff))


@pytest.mark.parametrize("B,M,L,nchg", [(6, 5, 200, 2), (3, 10, 200, 6), (4, 3, 187, 1), (2, 4, 120, 3), (5, 2, 208, 0)])
def test_tower_windows_equal_full_tower(nets, B, M, L, nchg):
    """svdd_conv_tower_windows_f32 (parent tower + per-candidate row windows) is bit-identical to the full tower on
    every candidate: candidates that equal the parent, changes at the sequence ends, many scattered changes."""
    from svdd_amd import fused, ops
    model, emb, head, _ = nets
    fv = fused.FusedValueNet(emb, head).to(DEV).eval()
    g = torch.Generator(device="cpu").manual_seed(3 + nchg)
    x = torch.randint(0, 5, (B, L), generator=g).to(torch.uint8)
    x[:, ::3] = 4                                                   # plenty of MASKs to replace
    cand = x[:, None, :].repeat(1, M, 1).clone()
    for b in range(B):
        for m in range(M):
            if m == 0 and nchg:                                     # one candidate changes both sequence ends
                pos = torch.tensor([0, L - 1])
            else:
                masked = (x[b] == 4).nonzero().flatten()
                k = int(torch.randint(0, nchg + 1, (1,), generator=g))
                pos = masked[torch.randperm(len(masked), generator=g)[:k]]
            cand[b, m, pos] = torch.randint(0, 4, (len(pos),), generator=g).to(torch.uint8)
    x, cand = x.to(DEV), cand.to(DEV).contiguous()
    onehot = ops.transform_samples(cand.view(B * M, L))
    full = fused.conv_tower(onehot, fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
    parent = fused.conv_tower(ops.transform_samples(x), fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
    win = fused.candidate_windows(cand, x)
    out = fused.conv_tower_windows(onehot, win, parent, M, fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
    assert torch.equal(out, full)
    w = win.cpu()
    diff = (cand != x[:, None, :]).view(B * M, L).cpu()
    for c in range(B * M):                                          # window = changed positions +- 27, 16-aligned
        p = diff[c].nonzero().flatten()
        if len(p) == 0:
            assert w[c].tolist() == [0, 0]
        else:
            assert w[c, 0] == max(0, int(p.min()) - 27) // 16 * 16 and w[c, 1] >= min(L, int(p.max()) + 28)
    with torch.no_grad():                                           # and the scores through the whole value net
        assert torch.equal(fv.forward_candidates(onehot, cand, x), fv(onehot))


def test_full_size_c2_net_paths(nets):
    """Config-2 sizes (B=256, M=10, L=200): the kernels the benchmark actually runs — one-launch backbone, parent-sharing
    tower, GRU, value tail — against the plain PyTorch modules on one real propose step."""
    from svdd_amd import fused, ops
    model, emb, head, _ = nets
    B, M, L = 256, 10, 200
    g = torch.Generator(device="cpu").manual_seed(17)
    x = torch.randint(0, 4, (B, L), generator=g).to(torch.uint8)
    x[torch.rand(B, L, generator=g) < 0.6] = 4
    x = x.to(DEV)
    fb = fused.FusedBackbone(model.backbone).to(DEV).eval()
    fv = fused.FusedValueNet(emb, head).to(DEV).eval()
    with torch.no_grad():
        logits = fb(x)                                              # one-launch kernel (B >= 192)
        ref_logits = model.backbone(x, None, zero_sigma=True)
        assert (logits - ref_logits).abs().max().item() <= 2e-5
        cand, onehot, _ = ops.propose(logits, x, 0.999 / 128, 0.5, M, ops.Rng(seed=5, step=3))
        scores = fv.forward_candidates(onehot, cand, x)             # parent tower + windows + GRU + tail
        assert torch.equal(scores, fv(onehot))                      # sharing the parent's tower changes no bit
        ref = head(emb(onehot))
        assert scores.shape == ref.shape == (B * M, 1, 1)
        assert (scores - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("n,L", [(2560, 200), (9, 50), (3, 37), (5, 208)])
def test_tower_generations_bit_identical(nets, n, L):
    """The second-generation tower kernel (two column tiles per wave, live-tile count as a template parameter) keeps
    every output element's accumulation order: same bits as the first generation, whole sequences and windows."""
    from svdd_amd import _lib, fused, ops
    model, emb, head, _ = nets
    fv = fused.FusedValueNet(emb, head).to(DEV).eval()
    g = torch.Generator(device="cpu").manual_seed(n + L)
    M = 3
    B = max(1, n // M)
    x = torch.randint(0, 5, (B, L), generator=g).to(torch.uint8)
    x[:, ::2] = 4
    cand = x[:, None, :].repeat(1, M, 1).clone()
    flip = (torch.rand(B, M, L, generator=g) < 0.03) & (cand == 4)
    cand[flip] = torch.randint(0, 4, (int(flip.sum()),), generator=g).to(torch.uint8)
    x, cand = x.to(DEV), cand.to(DEV).contiguous()
    onehot = ops.transform_samples(cand.view(B * M, L))
    win = fused.candidate_windows(cand, x)
    outs = []
    try:
        for v in (1, 2, 3):
            _lib.lib().svdd_set_tower_version(v)
            full = fused.conv_tower(onehot, fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
            parent = fused.conv_tower(ops.transform_samples(x), fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
            outs.append((full, fused.conv_tower_windows(onehot, win, parent, M, fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
                         if L > 104 else full))                     # windows: one sequence per 208-row tile only
    finally:
        _lib.lib().svdd_set_tower_version(0)
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])
    assert torch.equal(outs[1][0], outs[1][1])


@pytest.mark.parametrize("n,L", [(256, 50), (1100, 50), (1408, 50), (1650, 50), (2560, 50), (37, 33), (700, 9), (3, 104)])
@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_backbone_sequences_per_tile_choice_same_bits(n, L, mode):
    """Short sequences (several fit a 208-row tile): the number a workgroup takes is chosen to minimise rounds x tile cost
    (csrc/svdd_spt.h) — on the host for a known row count, by the workgroups for a device-side count, with or without a
    row index list; since round 5 the fp32 kernel may mix TWO tile sizes in one launch (full rounds of one size, the remainder at
    another: 1100 sequences of L = 50 = 256 tiles of four + 76 of one). Every variant gives the bits of the always-full tiles of round 1."""
    from svdd_amd import _lib, backbone, config, fused
    torch.manual_seed(L)
    cnn = backbone.CNNModel(config.rna_config().model, alphabet_size=5).to(DEV).eval()
    x = torch.randint(0, 5, (n, L), device=DEV, dtype=torch.uint8)
    pk = fused.pack_backbone(cnn) if mode == "f32" else fused.pack_backbone_lp(cnn, mode)
    fwd = fused.backbone_cnn if mode == "f32" else fused.backbone_cnn_lp
    live = max(1, (2 * n) // 3)
    cnt = torch.tensor([live], dtype=torch.int32, device=DEV)
    idx = torch.randperm(n, device=DEV)[:live].sort().values.to(torch.int32).contiguous()
    outs = []
    try:
        for full in (1, 0):
            _lib.lib().svdd_set_backbone_packing(full)
            a = fwd(x, pk)
            b = fwd(x, pk, count=cnt, out=torch.zeros(n, L, 5, device=DEV))
            c = fwd(x, pk, count=cnt, row_idx=idx, scatter=True, out=torch.zeros(n, L, 5, device=DEV))
            outs.append((a, b, c))
    finally:
        _lib.lib().svdd_set_backbone_packing(0)
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)
    a, b, c = outs[1]
    assert torch.equal(b[:live], a[:live]) and float(b[live:].abs().max() if live < n else 0.0) == 0.0
    assert torch.equal(c[idx.long()], a[idx.long()])


@pytest.mark.parametrize("n,L", [(16, 200), (37, 50), (5, 7), (300, 33), (1, 1)])
def test_gru_train_forward_and_backward_kernels(n, L):
    """csrc/svdd_gru_train.hip: the GRU with a gradient to its input (reward-net half of the DPS baseline, reference
    diffusion_gosai.py:1321-1330 through Enformer.py:1595-1602). Forward = the inference kernel's bits; backward (BPTT, d/dx)
    against torch autograd through nn.GRU on the same weights, for ragged tiles and degenerate lengths."""
    from svdd_amd import fused
    torch.manual_seed(n * 1000 + L)
    gru = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).to(DEV)
    x = torch.randn(n, L, 64, device=DEV)
    gout = torch.randn(2, n, L, 64, device=DEV)
    wpack, bpack = fused.pack_gru(gru)
    wpack, bpack, wb = wpack.to(DEV), bpack.to(DEV), fused.pack_gru_bwd(gru).to(DEV)
    xr = x.clone().requires_grad_(True)
    out = fused.GruBidirFunction.apply(xr, wpack, bpack, wb)
    assert torch.equal(out.detach(), fused.gru_bidir(x, wpack, bpack))           # the inference kernel's bits
    (out * gout).sum().backward()
    xt = x.clone().requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):                              # native cells: plain autograd
        y = gru(xt)[0]                                                           # [n, L, 128] = fwd | bwd halves
    (y[:, :, :64] * gout[0] + y[:, :, 64:] * gout[1]).sum().backward()
    assert float((out.detach() - torch.stack([y[:, :, :64], y[:, :, 64:]]).detach()).abs().max()) <= 2e-5
    scale = float(xt.grad.abs().max())
    assert float((xr.grad - xt.grad).abs().max()) <= 2e-4 * max(scale, 1.0), (float((xr.grad - xt.grad).abs().max()), scale)


@pytest.mark.parametrize("B,L", [(5, 200), (3, 50)])
def test_backbone_forward2_on_the_hip_conv_kernel_both_directions(B, L):
    """CNNModel.forward2 (the differentiable backbone entry of the DPS baseline, reference dnaconv.py:212-247) in channels-last
    rows with every dilated convolution on svdd_conv1d_cl_f32 forward AND backward (fused.DilatedConvFunction: backward-data =
    the same convolution with flipped taps and swapped channel axes), and with the element-wise ops between two convolutions
    as one hand-written pass per direction (fused.BackboneLayersFunction: svdd_bb_layer_fwd_f32 / svdd_bb_layer_bwd_f32),
    against the plain PyTorch / MIOpen trunk: logits and the gradient with respect to the relaxed one-hot input.
    A gradient through 20 ReLU layers is discontinuous in the pre-activations: two fp32 evaluations that differ at round-off
    decide a handful of ReLUs differently and then differ by 1e-3 .. 1e-2 of the gradient scale (all of them do against fp64,
    tools/diag_bb_grad.py) — so the gradients are compared with the ReLU decisions pinned: the torch reference multiplies by the
    masks the hand-written forward pass took instead of calling relu."""
    from svdd_amd import backbone, config, fused
    import torch.nn.functional as F
    torch.manual_seed(L)
    cnn = backbone.CNNModel((config.dna_config() if L == 200 else config.rna_config()).model, alphabet_size=5).to(DEV).eval()
    with torch.no_grad():
        for nm in cnn.norms:
            nm.weight.uniform_(0.5, 1.5)
            nm.bias.uniform_(-0.3, 0.3)
    for p in cnn.parameters():
        p.requires_grad_(False)
    x = torch.softmax(torch.randn(B, L, 5, device=DEV), dim=-1)
    g = torch.randn(B, L, 5, device=DEV)
    t = torch.linspace(0.0, 1.0, B, device=DEV)                                  # a time bias of its own per sequence
    res = {}
    fused.BackboneLayersFunction.keep_masks = True
    for hip, fused_layers in ((False, False), (True, False), (True, True)):
        cnn.hip_convs, cnn.fused_layers = hip, fused_layers
        xi = x.clone().requires_grad_(True)
        y = cnn.forward2(xi, t)
        (y * g).sum().backward()
        res[hip, fused_layers] = (y.detach(), xi.grad.clone())
    masks = fused.BackboneLayersFunction.last_masks
    fused.BackboneLayersFunction.keep_masks, fused.BackboneLayersFunction.last_masks = False, None
    cnn.hip_convs, cnn.fused_layers = False, True
    for k in ((True, False), (True, True)):
        assert float((res[k][0] - res[False, False][0]).abs().max()) <= 2e-5, k

    def pinned(conv_fn):
        """the trunk in torch ops with the ReLU decisions of the hand-written pass (first conv and tail as in _trunk_cl)"""
        xi = x.clone().requires_grad_(True)
        tbs = cnn._time_biases(t)
        feat = F.relu(cnn.linear(xi.permute(0, 2, 1))).permute(0, 2, 1)
        for i in range(cnn.num_layers):
            h = F.layer_norm(feat + tbs[i].transpose(1, 2), (feat.shape[2],), cnn.norms[i].weight, cnn.norms[i].bias, cnn.norms[i].eps)
            feat = conv_fn(i, h) * masks[i] + feat
        out = cnn.final_conv(feat.permute(0, 2, 1)).permute(0, 2, 1)
        (out * g).sum().backward()
        return xi.grad

    packs = cnn._conv_packs()
    H = cnn.args.hidden_dim
    g_hip = pinned(lambda i, h: fused.DilatedConvFunction.apply(h, packs[i][0], packs[i][1], H, H, 9, packs[i][2]) + cnn.convs[i].bias)
    g_mio = pinned(lambda i, h: cnn.convs[i](h.permute(0, 2, 1)).permute(0, 2, 1))
    got = res[True, True][1]
    scale = max(1.0, float(g_mio.abs().max()))
    assert float((got - g_hip).abs().max()) <= 2e-5 * scale, (float((got - g_hip).abs().max()), scale)
    assert float((got - g_mio).abs().max()) <= 1e-4 * scale, (float((got - g_mio).abs().max()), scale)
    # ... and with free ReLU decisions the three gradients agree wherever no decision differs: the typical element is at round-off
    for k in ((True, False), (True, True)):
        assert float((res[k][1] - res[False, False][1]).abs().median()) <= 2e-6 * scale, k


@pytest.mark.parametrize("B", [3, 37])
def test_backbone_one_launch_forward_with_saved_statistics_and_its_gradient_kernel(B):
    """The DPS pair of round 5 (reference diffusion_gosai.py:1321-1330 through models/dnaconv.py:212-247 on one_hot(x_t)):
    svdd_backbone_cnn_save_f32 must give the inference kernel's logits BIT FOR BIT (it is the same kernel plus stores), and
    svdd_backbone_cnn_grad_f32 the gradient with respect to the one-hot input — against torch autograd in fp64 over the plain
    modules with the ReLU decisions pinned to the ones the forward kernel took (decoded from its saved masks; a gradient through
    20 ReLU layers is discontinuous in the pre-activations, see the test above)."""
    from svdd_amd import backbone, config, fused
    import torch.nn.functional as F
    L = 200
    torch.manual_seed(B)
    cnn = backbone.CNNModel(config.dna_config().model, alphabet_size=5).to(DEV).eval()
    with torch.no_grad():
        for nm in cnn.norms:
            nm.weight.uniform_(0.5, 1.5)
            nm.bias.uniform_(-0.3, 0.3)
    for p in cnn.parameters():
        p.requires_grad_(False)
    fb = fused.FusedBackbone(cnn).to(DEV).eval()
    assert fb.grad_ok(L)
    tok = torch.randint(0, 5, (B, L), device=DEV, dtype=torch.uint8)
    tok[0] = 4                                                                    # an all-MASK row
    pk = fb.ol_pack()
    ref = fused.backbone_cnn(tok, pk)
    logits, saved = fused.backbone_cnn_save(tok, pk)
    assert torch.equal(logits, ref)
    g = torch.randn(B, L, 5, device=DEV)
    onehot = F.one_hot(tok.long(), 5).float().requires_grad_(True)
    out = fb.forward_with_grad(onehot, tok)
    assert torch.equal(out.detach(), ref)
    (out * g).sum().backward()
    got = onehot.grad.clone()
    # fp64 reference with pinned ReLU decisions
    masks = fused.decode_backbone_masks(saved[2], L).double()                     # [B, nl + 2, L, 128]
    c64 = __import__("copy").deepcopy(cnn).double()
    c64.clear_time_bias_cache()
    xi = F.one_hot(tok.long(), 5).double().requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        tbs = c64._time_biases(torch.zeros(B, device=DEV, dtype=torch.float64))
        feat = c64.linear(xi.permute(0, 2, 1)).permute(0, 2, 1) * masks[:, 0]
        nl = c64.num_layers
        for i in range(nl):
            h = F.layer_norm(feat + tbs[i].transpose(1, 2), (128,), c64.norms[i].weight, c64.norms[i].bias, c64.norms[i].eps)
            feat = c64.convs[i](h.permute(0, 2, 1)).permute(0, 2, 1) * masks[:, 1 + i] + feat
        h1 = c64.final_conv[0](feat.permute(0, 2, 1)).permute(0, 2, 1) * masks[:, nl + 1]
        out64 = c64.final_conv[2](h1.permute(0, 2, 1)).permute(0, 2, 1)
    assert float((out64.detach() - ref.double()).abs().max()) <= 2e-5             # the pinned fp64 forward is the kernel's function
    (out64 * g.double()).sum().backward()
    want = xi.grad
    scale = max(1.0, float(want.abs().max()))
    err = float((got.double() - want).abs().max())
    assert err <= 5e-5 * scale, (err, scale)


@pytest.mark.parametrize("B,task", [(6, "dna"), (5, "rna")])
def test_value_net_forward_grad_vs_torch_autograd(B, task):
    """FusedValueNet.forward_grad — the DPS reward call on a RELAXED input (reference diffusion_gosai.py:1326-1329) with the convolutions
    on svdd_conv1d_cl_f32 in both directions and the GRU on the BPTT kernels — against torch autograd through the plain modules
    (native GRU cells): scores and the gradient with respect to the input."""
    from svdd_amd import synthetic
    model, _, _, reward = synthetic.build(task, DEV)
    L = model.config.model.length
    fn = model.reward_callable(reward)
    assert fn.grad_ok(L)
    torch.manual_seed(B)
    x = torch.softmax(2.0 * torch.randn(B, L, 5, device=DEV), dim=-1)[:, :, :4].contiguous()
    xa = x.clone().requires_grad_(True)
    sa = fn.forward_grad(xa)
    sa[:, 0].mean().backward()
    xb = x.clone().requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        sb = reward(xb.transpose(1, 2))
    sb[:, 0].mean().backward()
    assert sa.shape == sb.shape
    assert float((sa - sb).abs().max()) <= 2e-5, float((sa - sb).abs().max())
    scale = float(xb.grad.abs().max())
    err = float((xa.grad - xb.grad).abs().max())
    assert err <= 5e-4 * scale, (err, scale)
    assert float((xa.grad - xb.grad).abs().median()) <= 2e-5 * scale


@pytest.mark.parametrize("B,task", [(6, "dna"), (37, "dna"), (5, "rna")])
def test_value_net_mean_score_input_grad_without_autograd(B, task):
    """FusedValueNet.mean_score_input_grad (round 6: the DPS reward call's gradient on 16 hand-written launches, no autograd — stem,
    conv + fused epilogue, GRU with saved gates, tail forward + backward in one pass, BPTT, gated transposed convs, stem transpose)
    against (a) the autograd form of the same kernels (forward_grad) and (b) torch autograd through the plain modules in fp64."""
    import copy
    from svdd_amd import synthetic
    model, _, _, reward = synthetic.build(task, DEV)
    L = model.config.model.length
    fn = model.reward_callable(reward)
    assert fn.grad_ok(L)
    torch.manual_seed(B)
    x = torch.softmax(2.0 * torch.randn(B, L, 5, device=DEV), dim=-1)[:, :, :4].contiguous()
    x[0, : L // 3] = 0.0                                                          # rows of zeros (what a MASK position's one-hot is)
    got = fn.mean_score_input_grad(x)
    assert got.shape == (B, L, 4) and torch.isfinite(got).all()
    fn.gru_off_chain = False                                                      # A/B: the GRU's non-recurrent halves inside the chains (round 5's kernels)
    try:
        got_in = fn.mean_score_input_grad(x)
    finally:
        fn.gru_off_chain = True
    assert float((got - got_in).abs().max()) <= 2e-6 * float(got_in.abs().max())
    xa = x.clone().requires_grad_(True)
    fn.forward_grad(xa)[:, 0].mean().backward()
    r64 = copy.deepcopy(reward).double()
    xb = x.double().clone().requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        r64(xb.transpose(1, 2))[:, 0].mean().backward()
    scale = float(xb.grad.abs().max())
    err_auto = float((got - xa.grad).abs().max())
    err64 = float((got.double() - xb.grad).abs().max())
    err64_auto = float((xa.grad.double() - xb.grad).abs().max())
    print(f"mean_score_input_grad {task} B={B}: vs forward_grad {err_auto / scale:.2e}, vs fp64 {err64 / scale:.2e} (forward_grad vs fp64 {err64_auto / scale:.2e})")
    bad = (got - xa.grad).abs() > 5e-5 * scale
    print("  elements off by > 5e-5 of the scale:", int(bad.sum()), "of", bad.numel(), "; rows", sorted(set(bad.nonzero()[:, 0].tolist()))[:8],
          "positions", sorted(set(bad.nonzero()[:, 1].tolist()))[:12])
    # Equal to a few 1e-7 of the scale — except where ONE ReLU decides differently: the stem is a GEMM in forward_grad and a direct sum
    # here, so a pre-activation within an ulp of zero can land on either side, and the gradient then differs inside that unit's
    # receptive field only (measured: B = 6 has one such unit, row 4 around position 189, 1.5e-2 of the scale over 18 positions; both
    # are gradients of the same function at a kink). Accepted: at most one such window per case.
    if bad.any():
        rows, pos = bad.nonzero()[:, 0], bad.nonzero()[:, 1]
        assert len(set(rows.tolist())) == 1 and int(pos.max() - pos.min()) <= 48 and err_auto <= 5e-2 * scale, (err_auto, scale)
    assert float((got - xa.grad).abs().median()) <= 2e-6 * scale
    assert float((got.double() - xb.grad).abs().median()) <= 2e-5 * scale
    if not bad.any():
        assert err64 <= max(5e-4 * scale, 2.0 * err64_auto), (err64, err64_auto, scale)


@pytest.mark.parametrize("scale", [0.0, 300.0, 25600.0])
def test_dps_step_without_autograd_equals_the_autograd_path(scale):
    """Diffusion._dps_guided_q with dps_fused (round 6: one-launch backbone pair + svdd_dps_probs / _probs_bwd / _guided_q + the reward
    net's gradient pass, no autograd) against round 5's path (torch autograd between the same big kernels) on a half-unmasked state:
    the guided q_xs, and the un-guided base (scale 0) bit for bit."""
    from svdd_amd import synthetic
    model, _, _, reward = synthetic.build("dna", DEV)
    B, L = 24, 200
    torch.manual_seed(5)
    x = torch.where(torch.rand(B, L, device=DEV) < 0.5, 4, torch.randint(0, 4, (B, L), device=DEV)).to(torch.uint8)
    x[0] = 4
    x[1] = torch.randint(0, 4, (L,), device=DEV).to(torch.uint8)                 # a fully unmasked row
    sched = model._schedule(128, 1e-5)[0]
    assert model._dps_fused_nets(x, reward) is not None
    model.dps_fused = True
    q_new = model._dps_guided_q(x, sched[40, 1], sched[40, 2], reward, scale)
    model.dps_fused = False
    try:
        q_old = model._dps_guided_q(x, sched[40, 1], sched[40, 2], reward, scale)
    finally:
        model.dps_fused = True
    assert q_new.shape == q_old.shape == (B, L, 5) and torch.isfinite(q_new).all()
    if scale == 0.0:
        assert torch.equal(q_new, q_old)
    rel = float(((q_new - q_old).abs() / q_old.abs().clamp(min=1e-12)).max())
    print(f"dps fused vs autograd path, scale {scale}: max rel diff of q {rel:.2e}")
    assert rel <= 2e-4, rel


@pytest.mark.parametrize("n,L", [(5, 50), (37, 200), (256, 200)])
def test_gru_pair_off_the_chain_equals_the_in_chain_kernels(n, L):
    """svdd_gru_bidir_train2_f32 / _bwd2_f32 (round 6: W_i x of every step and W_i^T da of every step as whole-chip launches, the serial
    chains keep only the recurrent 48 MFMAs per step) against the in-chain kernels of round 5: hidden states and saved gates bit for bit
    (the accumulators see the same operations in the same order), the gated input gradient to fp32 rounding."""
    import ctypes
    from svdd_amd import _lib, synthetic
    from svdd_amd.fused import pack_gru, pack_gru_bwd
    _, _, _, reward = synthetic.build("dna" if L == 200 else "rna", DEV)
    gru = reward.embedding.gru_tower.gru
    wpack, bpack = (t.to(DEV) for t in pack_gru(gru))
    wbwd = pack_gru_bwd(gru).to(DEV)
    lib, st = _lib.lib(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(n)
    x = torch.relu(torch.randn(n, L, 64, device=DEV))
    gout = torch.randn(2, n, L, 64, device=DEV) * 1e-3
    out1, out2 = torch.empty(2, n, L, 64, device=DEV), torch.empty(2, n, L, 64, device=DEV)
    save1, save2 = torch.empty(2, n, L, 4, 64, device=DEV), torch.empty(2, n, L, 4, 64, device=DEV)
    scratch = torch.empty(2, n * L, 192, device=DEV)
    _lib.check(lib.svdd_gru_bidir_train_f32(x.data_ptr(), wpack.data_ptr(), bpack.data_ptr(), out1.data_ptr(), save1.data_ptr(), n, L, st), "train")
    _lib.check(lib.svdd_gru_bidir_train2_f32(x.data_ptr(), wpack.data_ptr(), bpack.data_ptr(), scratch.data_ptr(), out2.data_ptr(),
                                             save2.data_ptr(), n, L, st), "train2")
    assert torch.equal(out1, out2) and torch.equal(save1, save2)
    dx = torch.empty(2, n, L, 64, device=DEV)
    _lib.check(lib.svdd_gru_bidir_bwd_f32(gout.data_ptr(), out1.data_ptr(), save1.data_ptr(), wbwd.data_ptr(), dx.data_ptr(), n, L, st), "bwd")
    want = torch.where(x > 0, dx[0] + dx[1], torch.zeros_like(x))
    got = torch.empty(n, L, 64, device=DEV)
    _lib.check(lib.svdd_gru_bidir_bwd2_f32(gout.data_ptr(), out1.data_ptr(), save1.data_ptr(), wbwd.data_ptr(), scratch.data_ptr(), x.data_ptr(),
                                           got.data_ptr(), n, L, st), "bwd2")
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-6 * scale, (float((got - want).abs().max()), scale)
    assert torch.equal(got == 0, want == 0) or float(((got == 0) != (want == 0)).float().mean()) < 1e-6
    nogate = torch.empty(n, L, 64, device=DEV)
    _lib.check(lib.svdd_gru_bidir_bwd2_f32(gout.data_ptr(), out1.data_ptr(), save1.data_ptr(), wbwd.data_ptr(), scratch.data_ptr(), None,
                                           nogate.data_ptr(), n, L, st), "bwd2 no gate")
    assert float((nogate - (dx[0] + dx[1])).abs().max()) <= 2e-6 * float((dx[0] + dx[1]).abs().max())


def test_reward_gradient_kernels_one_by_one_vs_torch():
    """The ABI-11 entry points of the reward net's gradient pass, each alone against torch (autograd where it is a backward):
    svdd_reward_stem_f32 / _bwd_f32 (4 -> 64 x 15 taps on real-valued rows), svdd_conv1d_cl_gated_f32 (transposed 64 -> 64 x 5
    convolution + residual + ReLU gate), svdd_reward_tail_grad_f32 (direction sum + LayerNorm + FFN + head + means: forward and
    backward in one pass), svdd_sum_gate_f32."""
    import ctypes
    import torch.nn.functional as F
    from svdd_amd import _lib
    from svdd_amd.fused import pack_conv
    lib, st = _lib.lib(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    n, L = 7, 200
    # stem
    x = torch.rand(n, L, 4, device=DEV)
    w = torch.randn(64, 4, 15, device=DEV) * 0.2
    b = torch.randn(64, device=DEV) * 0.1
    wk = w.permute(2, 1, 0).reshape(-1, 64).contiguous()                        # [(t, c)][co]
    f = torch.empty(n, L, 64, device=DEV)
    _lib.check(lib.svdd_reward_stem_f32(x.data_ptr(), wk.data_ptr(), b.data_ptr(), f.data_ptr(), n, L, 15, st), "stem")
    xr = x.clone().requires_grad_(True)
    want = F.relu(F.conv1d(xr.transpose(1, 2), w, b, padding=7)).transpose(1, 2)
    assert float((f - want.detach()).abs().max()) <= 1e-5
    g = torch.randn(n, L, 64, device=DEV) * 1e-3
    gm = torch.where(want > 0, g, torch.zeros_like(g)).contiguous()             # the gradient at the pre-activation
    want.backward(g)
    dx = torch.empty(n, L, 4, device=DEV)
    _lib.check(lib.svdd_reward_stem_bwd_f32(gm.data_ptr(), wk.data_ptr(), dx.data_ptr(), n, L, 15, st), "stem bwd")
    assert float((dx - xr.grad).abs().max()) <= 1e-6 * max(1.0, float(xr.grad.abs().max()) * 1e3)
    # gated transposed convolution: y = gate > 0 ? conv^T(g) + g : 0
    w5 = torch.randn(64, 64, 5, device=DEV) * 0.1
    wt = pack_conv(w5.flip(2).transpose(0, 1).contiguous())
    fin = torch.randn(n, L, 64, device=DEV)
    fr = fin.clone().requires_grad_(True)
    out = F.conv1d(F.relu(fr).transpose(1, 2), w5, None, padding=2).transpose(1, 2) + F.relu(fr)   # conv(relu(f)) + relu(f): residual block on a gated input
    out.backward(g)
    y = torch.empty(n, L, 64, device=DEV)
    gc = g.contiguous()
    _lib.check(lib.svdd_conv1d_cl_gated_f32(gc.data_ptr(), wt.data_ptr(), y.data_ptr(), n, L, 64, 64, 5, 1, gc.data_ptr(), fin.data_ptr(), st), "gated conv")
    assert float((y - fr.grad).abs().max()) <= 2e-6
    # tail: d mean_n(mean_l(w_eff . relu(W1 LN(h0 + h1) + b1))) / d (h0 + h1)
    h = torch.randn(2, n, L, 64, device=DEV)
    w1, b1 = torch.randn(128, 64, device=DEV) * 0.2, torch.randn(128, device=DEV) * 0.1
    gam, bet, weff = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.1, torch.randn(128, device=DEV)
    hr = h.clone().requires_grad_(True)
    z = F.relu(F.linear(F.layer_norm(hr[0] + hr[1], (64,), gam, bet, 1e-5), w1, b1))
    ((z @ weff).mean(dim=1)).mean().backward()
    gout = torch.empty_like(h)
    _lib.check(lib.svdd_reward_tail_grad_f32(h[0].data_ptr(), h[1].data_ptr(), w1.data_ptr(), b1.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                                             weff.data_ptr(), 1e-5, n, L, gout[0].data_ptr(), gout[1].data_ptr(), st), "tail grad")
    sc = float(hr.grad.abs().max())
    assert float((gout[0] - hr.grad[0]).abs().max()) <= 2e-5 * sc and torch.equal(gout[0], gout[1])
    # sum + gate
    a_, b_ = torch.randn(n, L, 64, device=DEV), torch.randn(n, L, 64, device=DEV)
    sg = torch.empty_like(a_)
    _lib.check(lib.svdd_sum_gate_f32(a_.data_ptr(), b_.data_ptr(), fin.data_ptr(), sg.data_ptr(), a_.numel(), st), "sum gate")
    assert torch.equal(sg, torch.where(fin > 0, a_ + b_, torch.zeros_like(a_)))

"""-m gpu: the hand-written kernels of the Enformer-shaped value trunk (csrc/svdd_trunk.hip, svdd_amd/fused_trunk.py;
BASELINE.json configs[3], reference decode.py:78-80 / Enformer.py:1271-1334, 1807-2007) against the PyTorch module they
replace, in fp32 on the same weights. (The trunk's attention / pooling blocks come from an un-vendored dependency of the
reference — SURVEY section 8c: parity unpinned by construction — so the PyTorch module IS the definition here.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _randomise(emb, head, seed):
    """Non-trivial BatchNorm statistics, attention output projections (zero at init) and pooling logits."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for m in emb.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight.copy_(1.0 + 0.2 * torch.randn(m.num_features, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.num_features, generator=g))
        for blk in emb.transformer_tower:
            w = blk.mha.to_out.weight
            w.copy_(torch.randn(w.shape, generator=g) * (w.shape[1] ** -0.5))
            blk.mha.to_out.bias.copy_(torch.randn(w.shape[0], generator=g) * 0.05)
        for blk in emb.conv_tower.blocks:
            pw = blk[1].pool.to_attn_logits.weight
            pw.add_((torch.randn(pw.shape, generator=g) * 0.05).to(pw.device))


@pytest.mark.parametrize("precision,tol", [("f32", 2e-5), ("bf16x3", 1e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("L,kw", [(200, dict(n_conv=4, channels=768, n_transformers=2, n_heads=4, key_len=16)),
                                  (50, dict(n_conv=4, channels=768, n_transformers=2, n_heads=4, key_len=16)),
                                  (37, dict(n_conv=3, channels=768, n_transformers=1, n_heads=2, key_len=32))])
def test_fused_trunk_equals_the_module_small(precision, tol, L, kw):
    from svdd_amd import synthetic
    from svdd_amd.fused_trunk import FusedEnformerValueNet
    _, emb, head, _ = synthetic.build("dna", DEV, hidden_dim=32, num_cnn_stacks=1, value="enformer", enformer_kwargs=kw)
    _randomise(emb, head, L)
    g = torch.Generator(device=DEV).manual_seed(L)
    n = 37
    tok = torch.randint(0, 5, (n, L), device=DEV, generator=g, dtype=torch.uint8)
    tok[0] = 4                                                       # the all-MASK prior
    onehot = (torch.nn.functional.one_hot(tok.long().clamp(max=3), 4) * (tok != 4)[..., None]).float()
    with torch.no_grad():
        ref = head(emb(onehot)).reshape(n)
        fn = FusedEnformerValueNet(emb, head, precision)
        out = fn.forward_tokens(tok).reshape(n)
        out2 = fn(onehot).reshape(n)
        # a compacted batch: only the first `live` rows are computed, and they are the same bits
        live = 11
        cnt = torch.tensor([live], dtype=torch.int32, device=DEV)
        part = fn.forward_tokens(tok, count=cnt).reshape(n)[:live]
    scale = float(ref.abs().max())
    assert torch.isfinite(out).all()
    assert float((out - ref).abs().max()) <= tol * max(1.0, scale), (float((out - ref).abs().max()), scale)
    assert torch.equal(out, out2)
    assert torch.equal(part, out[:live])


def test_fused_trunk_full_size_c4():
    """BASELINE configs[3]: the 230 M-parameter trunk (7 conv levels to 1536 channels, 11 transformer blocks on 2 tokens),
    160 candidates of length 200."""
    from svdd_amd import synthetic
    from svdd_amd.fused_trunk import FusedEnformerValueNet
    _, emb, head, _ = synthetic.build("dna", DEV, value="enformer")
    _randomise(emb, head, 3)
    g = torch.Generator(device=DEV).manual_seed(1)
    n, L = 160, 200
    tok = torch.randint(0, 5, (n, L), device=DEV, generator=g, dtype=torch.uint8)
    tok[:8, 40:] = 4
    onehot = (torch.nn.functional.one_hot(tok.long().clamp(max=3), 4) * (tok != 4)[..., None]).float()
    with torch.no_grad():
        ref = head(emb(onehot)).reshape(n)
        out = FusedEnformerValueNet(emb, head, "bf16x3").forward_tokens(tok).reshape(n)
        one = FusedEnformerValueNet(emb, head, "bf16").forward_tokens(tok).reshape(n)
        f32 = FusedEnformerValueNet(emb, head, "f32").forward_tokens(tok).reshape(n)
    scale = max(1.0, float(ref.abs().max()))
    err3, err1, err32 = float((out - ref).abs().max()), float((one - ref).abs().max()), float((f32 - ref).abs().max())
    print(f"full-size trunk: |score| max {float(ref.abs().max()):.3f}  f32 err {err32:.2e}  bf16x3 err {err3:.2e}  bf16 err {err1:.2e}")
    assert err32 <= 2e-5 * scale          # fp32 MFMAs against the fp32 modules: summation order only
    assert err3 <= 1e-4 * scale
    assert err1 <= 5e-2 * scale


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "f32"])
def test_tower_on_two_streams_same_bits(precision):
    """The candidates as 2 / 3 / 4 parts on as many streams (fused_trunk.tower_streams, from 2048 token rows on) against
    one chain of kernels: the same bits, with and without a device-side live count (small ones included), and twice in a row (the streams join before the next forward touches the buffers)."""
    from svdd_amd import synthetic
    from svdd_amd.fused_trunk import FusedEnformerValueNet
    kw = dict(n_conv=7, channels=768, n_transformers=3, n_heads=4, key_len=16)
    _, emb, head, _ = synthetic.build("dna", DEV, hidden_dim=32, num_cnn_stacks=1, value="enformer", enformer_kwargs=kw)
    _randomise(emb, head, 9)
    g = torch.Generator(device=DEV).manual_seed(2)
    n, L = 1100, 200                                                       # 2200 token rows
    tok = torch.randint(0, 5, (n, L), device=DEV, generator=g, dtype=torch.uint8)
    with torch.no_grad():
        fn = FusedEnformerValueNet(emb, head, precision)
        assert fn.tower_streams == 2
        outs = {}
        for live in (None, 1100, 901, 550, 300):
            cnt = None if live is None else torch.tensor([live], dtype=torch.int32, device=DEV)
            k = n if live is None else live
            got = []
            for streams in (2, 2, 3, 4):
                fn.tower_streams = streams
                got.append(fn.forward_tokens(tok, count=cnt).reshape(n)[:k].clone())
                assert fn.last_streams == streams
            fn.tower_streams = 1
            c = fn.forward_tokens(tok, count=cnt).reshape(n)[:k].clone()
            assert fn.last_streams == 1
            assert torch.isfinite(c).all() and all(torch.equal(a, c) for a in got), live
            outs[live] = c
    assert torch.equal(outs[None][:300], outs[300]) and outs[None].unique().numel() > n // 2


def test_scores_do_not_depend_on_the_row_position_full_size():
    """A candidate's score must be the same bits wherever the compaction / the two-stream interleave puts its row: full-size
    trunk, 2048 candidates scored in order, in reversed order, and as two half batches on two streams. (Caught by
    tools/trunk_shared_soak.py: the head's 1 x 1 convolution as a hipBLASLt GEMM changed the last bit of a row's result with the
    row's position; it is a row-wise reduction now.)"""
    from svdd_amd import synthetic
    from svdd_amd.fused_trunk import FusedEnformerValueNet
    _, emb, head, _ = synthetic.build("dna", DEV, value="enformer")
    _randomise(emb, head, 4)
    g = torch.Generator(device=DEV).manual_seed(3)
    n, L = 2048, 200
    tok = torch.randint(0, 5, (n, L), device=DEV, generator=g, dtype=torch.uint8)
    with torch.no_grad():
        fn = FusedEnformerValueNet(emb, head, "bf16x3")
        fn.tower_streams = 1
        a = fn.forward_tokens(tok).reshape(n).clone()
        b = fn.forward_tokens(tok.flip(0).contiguous()).reshape(n).flip(0).clone()
        fn.tower_streams = 2
        c = fn.forward_tokens(tok).reshape(n).clone()
        assert fn.last_streams == 2
    assert torch.isfinite(a).all() and a.unique().numel() > n // 2
    assert torch.equal(a, b)
    assert torch.equal(a, c)


def test_mc_decode_with_the_fused_trunk_vs_oracle():
    """SVDD-MC (M = 20, L = 200) with the fused trunk as value function, precision bf16x3: the work-skipping loop (live
    candidates gathered on the device, `count` fed to every trunk kernel) against the plain loop, and the oracle's replay of
    the recorded logits / scores."""
    from oracle import svdd_oracle as orc
    from svdd_amd import synthetic
    from svdd_amd.fused_trunk import FusedEnformerValueNet
    model, emb, head, _ = synthetic.build("dna", DEV, value="enformer",
                                          enformer_kwargs=dict(n_conv=4, channels=768, n_transformers=2, n_heads=4, key_len=16))
    _randomise(emb, head, 5)
    B, L, M, S = 6, 200, 20, 24
    sched = model._schedule(S, 1e-5)[0]
    model.precision, model.rng_mode, model.philox_seed = "bf16x3", "philox", 8
    assert isinstance(model.value_callable(emb, head), FusedEnformerValueNet)
    outs = {}
    for skip in (False, True):
        model.skip_unchanged, model.trace, model.skip_stats = skip, [], ({} if skip else None)
        x = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M).cpu().numpy()
        tr = [(lg.cpu().numpy(), None if sc is None else sc.cpu().numpy()) for lg, sc in model.trace]
        outs[skip] = (x, tr, model.skip_stats)
    model.precision, model.skip_unchanged, model.trace, model.skip_stats, model.rng_mode = "f32", True, None, None, "replay"
    x_on, tr_on, st = outs[True]
    assert st["kind"] == "mc" and st["live_candidates"] < st["candidates"]
    assert np.array_equal(x_on, orc.replay_controlled_sample(tr_on, sched, B, L, M, seed=8))
    assert np.array_equal(x_on, outs[False][0])
    for (la, sa), (lb, sb) in zip(tr_on, outs[False][1]):
        assert np.array_equal(la, lb) and (sa is None or np.array_equal(sa, sb))


@pytest.mark.parametrize("M,N,Cin,T,rps,live", [(1000, 768, 96, 1, 8, None), (2100, 896, 64, 5, 14, None),
                                                (777, 256, 160, 5, 7, 40), (5000, 1152, 64, 1, 10, 333),
                                                (100, 128, 32, 1, 5, None), (513, 384, 32, 3, 9, 0)])
@pytest.mark.parametrize("parts", [2, 1])
def test_gemm_256_tiles_equal_128_tiles(M, N, Cin, T, rps, live, parts):
    """svdd_trunk_gemm's two kernels (128 x 128 register-staged tiles, 256 x 256 LDS-DMA tiles with staggered wave groups)
    accumulate every output element in the same order: bit-identical results, for ragged M, a half-wide last column block
    (N = 896, 1152), 5 row-shifted taps, bias / residual / GELU epilogue and a device-side live-row count; and both agree
    with an fp64 product of the hi + lo operands."""
    from svdd_amd import _lib
    from svdd_amd.fused_trunk import pack_gemm_weight, GUARD, TAIL
    lib = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(M + N)
    a = torch.randn(M, Cin, generator=g)
    w = torch.randn(N, Cin, T, generator=g) * (Cin * T) ** -0.5
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g).to(DEV)
    a_hi = a.to(torch.bfloat16)
    a_lo = (a - a_hi.float()).to(torch.bfloat16)
    planes = []
    for pl in ((a_hi, a_lo) if parts == 2 else (a_hi,)):
        buf = torch.zeros((GUARD + M + TAIL) * Cin, dtype=torch.bfloat16, device=DEV)
        buf[GUARD * Cin:(GUARD + M) * Cin] = pl.reshape(-1).to(DEV)
        planes.append(buf[GUARD * Cin:])
    wp = pack_gemm_weight(w, parts).to(DEV)
    cnt = None if live is None else torch.tensor([live], dtype=torch.int32, device=DEV)
    m_live = M if live is None else min(M, live * rps)
    outs, pls = {}, {}
    ps, pb = (1.0 + 0.1 * torch.randn(N, generator=g)).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
    for ver in (1, 3, 42):                                 # 42: the LDS-DMA kernel with 192-row tiles everywhere (round 4)
        _lib.check(lib.svdd_set_option(4, 3 if ver == 42 else ver), "svdd_set_option")
        _lib.check(lib.svdd_set_option(4, 42 if ver == 42 else 41), "svdd_set_option")
        out = torch.full((M, N), 7.0, device=DEV)
        o_hi = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
        o_lo = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
        rc = lib.svdd_trunk_gemm(planes[0].data_ptr(), planes[1].data_ptr() if parts == 2 else None, wp.data_ptr(),
                                 bias.to(DEV).data_ptr(), resid.data_ptr(), out.data_ptr(), M, N, Cin, T, Cin, N, 2,
                                 None if cnt is None else cnt.data_ptr(), rps,
                                 o_hi.data_ptr(), o_lo.data_ptr() if parts == 2 else None, ps.data_ptr(), pb.data_ptr(), 2, 2, None)
        _lib.check(rc, "svdd_trunk_gemm")
        torch.cuda.synchronize()
        outs[ver] = out.cpu()
        pls[ver] = (o_hi.float().cpu(), o_lo.float().cpu())
    # the fused second output = what the separate element-wise pass writes from the fp32 output
    r_hi = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
    r_lo = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
    rc = lib.svdd_trunk_act_split(out.data_ptr(), ps.data_ptr(), pb.data_ptr(), 2, m_live if live is None else M, N, rps, 2,
                                  r_hi.data_ptr(), r_lo.data_ptr() if parts == 2 else None, None if cnt is None else cnt.data_ptr(), None)
    _lib.check(rc, "svdd_trunk_act_split")
    torch.cuda.synchronize()
    for ver in (1, 3, 42):
        assert torch.equal(pls[ver][0], r_hi.float().cpu())
        if parts == 2:
            assert torch.equal(pls[ver][1], r_lo.float().cpu())
    _lib.check(lib.svdd_set_option(4, 2), "svdd_set_option")
    _lib.check(lib.svdd_set_option(4, 40), "svdd_set_option")           # tile height by cost again
    assert torch.equal(outs[1][:m_live], outs[3][:m_live])
    assert torch.equal(outs[42][:m_live], outs[3][:m_live]) and bool((outs[42][m_live:] == 7.0).all())
    assert bool((outs[3][m_live:] == 7.0).all())                       # rows beyond the live count are not written
    # fp64 reference on the operands the kernels see (x3: hi + lo of both; dropped lo * lo term ~ 2^-16 relative)
    a_eff = (a_hi.double() + (a_lo.double() if parts == 2 else 0.0))
    w_hi = w.to(torch.bfloat16)
    w_eff = w_hi.double() + ((w - w_hi.float()).to(torch.bfloat16).double() if parts == 2 else 0.0)
    ap = torch.zeros(M + 4, Cin, dtype=torch.float64)
    ap[2:M + 2] = a_eff
    acc = sum(ap[t - T // 2 + 2: t - T // 2 + 2 + M] @ w_eff[:, :, t].t() for t in range(T)) + bias.double()
    ref = acc / (1.0 + torch.exp(-1.702 * acc)) + resid.cpu().double()
    if m_live > 0:
        err = float((outs[3][:m_live].double() - ref[:m_live]).abs().max())
        assert err <= (2e-5 if parts == 2 else 1e-4), err


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "f32"])
@pytest.mark.parametrize("kw", [dict(n_conv=4, channels=768, n_transformers=2, n_heads=4, key_len=16), None])
def test_first_levels_shared_with_the_parent_same_bits(precision, kw):
    """forward_tokens(shared=...): the first levels of the conv tower only on the window of rows around the
    positions where a candidate differs from its parent, the parent's planes elsewhere — the same bits as the whole-sequence
    path, for windows at both ends, single positions, differences spread over the whole sequence (more of them than window
    slots), copies of the parent (no window) and a device-side live count. kw None: the full-size trunk of BASELINE.json configs[3]."""
    from svdd_amd import synthetic
    from svdd_amd.fused_trunk import FusedEnformerValueNet
    extra = dict(enformer_kwargs=kw) if kw else {}
    small = dict(hidden_dim=32, num_cnn_stacks=1) if kw else {}
    _, emb, head, _ = synthetic.build("dna", DEV, value="enformer", **small, **extra)
    _randomise(emb, head, 3)
    g = torch.Generator(device="cpu").manual_seed(11)
    B, M, L = 5, 6, 200
    x = torch.randint(0, 5, (B, L), generator=g, dtype=torch.uint8)
    x[0] = 4
    cand = x[:, None, :].repeat(1, M, 1)
    edits = {(0, 0): [0], (0, 1): [199], (0, 2): [0, 199], (0, 3): [7, 8], (0, 4): [100], (1, 0): [6], (1, 1): [193], (1, 2): [14, 15, 16],
             (1, 3): [1, 60, 120, 180], (2, 0): [99, 101], (2, 5): [198], (3, 1): list(range(0, 200, 9)), (3, 2): [50], (4, 0): [150, 151],
             (4, 3): [8], (4, 4): [191], (4, 5): [33, 77]}
    for (b, m), ps in edits.items():
        for p in ps:
            cand[b, m, p] = (int(cand[b, m, p]) + 1 + (p % 3)) % 5
    ids = sorted(b * M + m for (b, m) in edits) + [2 * M + 1]              # one copy of its parent: an empty window
    x, cand = x.to(DEV), cand.to(DEV)
    n = B * M
    toks = torch.zeros((n, L), dtype=torch.uint8, device=DEV)
    live = len(ids)
    idx = torch.zeros(n, dtype=torch.int32, device=DEV)
    idx[:live] = torch.tensor(ids, dtype=torch.int32, device=DEV)
    toks[:live] = cand.view(n, L)[idx[:live].long()]
    cnt = torch.tensor([live], dtype=torch.int32, device=DEV)
    with torch.no_grad():
        fn = FusedEnformerValueNet(emb, head, precision)
        whole = fn.forward_tokens(toks, count=cnt).reshape(n)[:live].clone()
        rows = {}
        for depth, slots in ((1, 1), (2, 1), (3, 1), (1, 4), (3, 2), (3, 4), (4, 1), (4, 4)):  # shared levels (lengths 200, 100, 50, 25) x window slots
            fn.share_levels, fn.share_slots = depth, slots
            shared = fn.forward_tokens(toks, count=cnt, shared=(x, idx, M)).reshape(n)[:live].clone()
            assert torch.equal(shared, whole), (depth, slots)
            rows[depth, slots] = fn.last_window_rows.tolist()
        fn.share_level0 = False
        off = fn.forward_tokens(toks, count=cnt, shared=(x, idx, M)).reshape(n)[:live].clone()
        assert torch.equal(off, whole)
        fn.share_level0 = True
        # the parents' own levels step by step: x_t from x_(t-1) on the windows that changed (a few positions per row, none in one
        # row, then parents that have nothing to do with the previous ones), against parents computed whole
        fn.share_levels, fn.share_slots = 4, 4
        prows = []
        for step in range(3):
            x2 = x.clone()
            if step < 2:
                for b in range(B - 1):
                    for p in (3 + 17 * b + 40 * step, 199 - 11 * b - step):
                        x2[b, p] = (x2[b, p] + 1 + step) % 5
            else:
                x2 = torch.randint(0, 5, (B, L), generator=g, dtype=torch.uint8).to(DEV)
            toks2 = toks.clone()
            par = (idx[:live] // M).long()
            toks2[:live] = torch.where(toks[:live] != x[par], toks[:live], x2[par])      # the same edits on the new parents
            x = x2
            want = fn.forward_tokens(toks2, count=cnt).reshape(n)[:live].clone()
            fn.share_parent_steps = True
            got = fn.forward_tokens(toks2, count=cnt, shared=(x, idx, M)).reshape(n)[:live].clone()
            prows.append(None if fn.last_parent_rows is None else fn.last_parent_rows.tolist())
            assert torch.equal(got, want), step
            fn.share_parent_steps = False
            got = fn.forward_tokens(toks2, count=cnt, shared=(x, idx, M)).reshape(n)[:live].clone()
            assert torch.equal(got, want) and fn.last_parent_rows is None, step
    assert prows[0] is not None and 0 < prows[0][0] < B * L // 2 and prows[2][0] == B * L, prows
    assert torch.isfinite(whole).all() and whole.unique().numel() > live // 2
    assert len(rows[3, 1]) == 3 and rows[3, 1][0] == rows[1, 1][0]
    assert 0 < rows[3, 1][0] < live * L // 2 and 0 < rows[3, 1][2] < live * 54, rows   # the windows are a fraction of the rows
    assert rows[3, 4][0] < rows[3, 2][0] < rows[3, 1][0], rows                   # ... a smaller one with a window per changed position


def test_cli_mc_with_the_enformer_value_trunk(tmp_path):
    """decode.py --model enformer (reference decode.py:72-80,149: the Enformer-shaped trunk as value function) through the CLI
    mirror with --precision bf16x3: the hand-written trunk kernels score the candidates; npz contract of decode.py:117."""
    from svdd_amd import cli
    from svdd_amd.config import SamplingConfig
    from svdd_amd.fused_trunk import FusedEnformerValueNet
    import svdd_amd.synthetic as syn
    orig = syn.build
    seen = {}

    def small_build(task, device, seed=44, value="convgru", **kw):
        m = orig(task, device, seed=seed, value=value,
                 enformer_kwargs=dict(n_conv=4, channels=768, n_transformers=2, n_heads=4, key_len=16))
        m[0].config.sampling = SamplingConfig(steps=4)
        seen["model"], seen["emb"], seen["head"] = m[0], m[1], m[2]
        return m

    syn.build = small_build
    try:
        path, out = cli.main("mc", ["--task", "dna", "--model", "enformer", "--precision", "bf16x3", "--batch_size", "3",
                                    "--sample_M", "4", "--val_batch_num", "1", "--out_dir", str(tmp_path), "--rng", "philox"])
    finally:
        syn.build = orig
    assert isinstance(seen["model"].value_callable(seen["emb"], seen["head"]), FusedEnformerValueNet)
    z = np.load(path)
    assert path.endswith("dna-HepG2.npz") and set(z.files) == {"decoding", "baseline"}
    assert z["decoding"].shape == (3,) and np.isfinite(z["decoding"]).all() and np.isfinite(z["baseline"]).all()

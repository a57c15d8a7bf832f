"""-m gpu: exact work-skipping (Diffusion.skip_unchanged, logits_cache).

A candidate that unmasked nothing is its parent (reference diffusion_gosai.py:1203) and the nets do not see the time step
(:334-335), so its value score (:1207-1209), its Tweedie forward, x0-hat and reward (:1413-1436) are the parent's; a row
whose selected candidate is such a copy keeps its logits. The engine skips exactly that work. The bar: decodes are
bit-identical with skipping on and off — tokens, and (traced) the logits and scores of every step — at the BASELINE
configs[1] (SVDD-MC, B=256, L=200, M=10) and configs[2] (SVDD-PM, B=256, L=50, M=10) sizes, 128 steps, in replay and
Philox modes, in fp32 and in a split-precision mode. Plus the device-side compaction primitives against torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ------------------------------------------------------------------------------------------- primitives ----
@pytest.mark.parametrize("n", [1, 63, 64, 1023, 1024, 1025, 2560, 40960])
@pytest.mark.parametrize("p", [0.0, 0.3, 1.0])
def test_compact_flags_vs_torch(n, p):
    from svdd_amd import ops
    g = torch.Generator(device=DEV).manual_seed(n)
    flags = (torch.rand(n, device=DEV, generator=g) < p).to(torch.int32) * 7      # any non-zero value counts
    live_idx = torch.full((n,), -5, dtype=torch.int32, device=DEV)
    slot = torch.full((n,), -5, dtype=torch.int32, device=DEV)
    count = torch.full((1,), -5, dtype=torch.int32, device=DEV)
    ops.compact_flags(flags, live_idx, slot, count)
    ref = torch.nonzero(flags).flatten().to(torch.int32)
    k = int(count)
    assert k == ref.numel()
    assert torch.equal(live_idx[:k], ref)                                         # stable: ascending order
    exp_slot = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    exp_slot[ref.long()] = torch.arange(k, dtype=torch.int32, device=DEV)
    assert torch.equal(slot, exp_slot)


@pytest.mark.parametrize("n", [1, 63, 64, 1023, 1024, 1025, 2560, 40960])
@pytest.mark.parametrize("p", [0.0, 0.3, 1.0])
def test_compact_by_key_vs_torch(n, p):
    """The compaction ordered by key (window size in row tiles), largest first, stable inside a key: against torch's stable sort."""
    from svdd_amd import ops
    g = torch.Generator(device=DEV).manual_seed(n + 1)
    key = torch.randint(1, 21, (n,), device=DEV, generator=g, dtype=torch.int32)          # keys above 15 clamp to 15
    key = torch.where(torch.rand(n, device=DEV, generator=g) < p, key, torch.zeros_like(key))
    if n > 3:
        key[1] = -2                                                                        # negative = not live
    live_idx = torch.full((n,), -5, dtype=torch.int32, device=DEV)
    slot = torch.full((n,), -5, dtype=torch.int32, device=DEV)
    count = torch.full((1,), -5, dtype=torch.int32, device=DEV)
    ops.compact_by_key(key, live_idx, slot, count)
    kc = key.clamp(0, 15)
    live = torch.nonzero(kc).flatten()
    order = torch.sort(kc[live], descending=True, stable=True).indices
    ref = live[order].to(torch.int32)
    k = int(count)
    assert k == ref.numel()
    assert torch.equal(live_idx[:k], ref)
    exp_slot = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    exp_slot[ref.long()] = torch.arange(k, dtype=torch.int32, device=DEV)
    assert torch.equal(slot, exp_slot)


def test_mc_decode_same_tokens_whatever_the_order_of_the_live_candidates():
    """FusedValueNet.sort_live_by_window (the live candidates handed to the windowed tower by descending window size) must not
    change a token or a traced score: a row's result does not depend on its place in the compacted batch."""
    from svdd_amd import synthetic
    model, emb, head, _ = synthetic.build("dna", DEV)
    model.rng_mode, model.philox_seed = "philox", 11
    fn = model.value_callable(emb, head)
    outs = []
    for srt in (True, False):
        fn.sort_live_by_window = srt
        model.trace = []
        x0 = model.controlled_sample(emb, head, num_steps=24, eval_sp_size=64, sample_M=10)
        outs.append((x0, [sc for _, sc in model.trace if sc is not None]))
        model.trace = None
    fn.sort_live_by_window = True
    assert torch.equal(outs[0][0], outs[1][0])
    assert len(outs[0][1]) == len(outs[1][1]) == 24
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


def test_gather_and_advance_rows():
    from svdd_amd import ops
    g = torch.Generator(device=DEV).manual_seed(0)
    B, M, R = 37, 5, 250                                                          # 1000-byte rows: not a multiple of 16
    src = torch.randn(B * M, R, device=DEV, generator=g)
    flags = (torch.rand(B * M, device=DEV, generator=g) < 0.6).to(torch.int32)
    live_idx = torch.empty(B * M, dtype=torch.int32, device=DEV)
    slot = torch.empty(B * M, dtype=torch.int32, device=DEV)
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.compact_flags(flags, live_idx, slot, count)
    k = int(count)
    comp = torch.full_like(src, 9.0)
    ops.gather_rows(src, live_idx, count, comp)
    assert torch.equal(comp[:k], src[live_idx[:k].long()]) and bool((comp[k:] == 9.0).all())
    sel = torch.randint(0, M, (B,), device=DEV, generator=g).to(torch.int32)
    dst = torch.zeros(B, R, device=DEV)
    ops.advance_rows(comp, slot, sel, dst, M)
    for b in range(B):
        c = b * M + int(sel[b])
        exp = src[c] if int(flags[c]) else torch.zeros(R, device=DEV)
        assert torch.equal(dst[b], exp), b
    tok = torch.randint(0, 5, (B * M, 200), device=DEV, dtype=torch.uint8, generator=g)   # u8 rows, 200 bytes
    tc = torch.zeros_like(tok)
    ops.gather_rows(tok, live_idx, count, tc)
    assert torch.equal(tc[:k], tok[live_idx[:k].long()])


@pytest.mark.parametrize("mode", ["argmax", "multinomial"])
def test_select_compact_equals_select_on_dense_scores(mode):
    from svdd_amd import ops
    g = torch.Generator(device=DEV).manual_seed(4)
    B, M, L = 41, 10, 200
    cand = torch.randint(0, 5, (B, M, L), device=DEV, dtype=torch.uint8, generator=g)
    flags = (torch.rand(B * M, device=DEV, generator=g) < 0.7).to(torch.int32)
    flags[:M] = 0                                                                 # a row whose candidates are all copies
    live_idx, slot = torch.empty(B * M, dtype=torch.int32, device=DEV), torch.empty(B * M, dtype=torch.int32, device=DEV)
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.compact_flags(flags, live_idx, slot, count)
    sc = torch.randn(B * M, device=DEV, generator=g) * 0.01
    sc[int(count):] = float("nan")                                                # beyond count: must never be read
    parent = torch.randn(B, device=DEV, generator=g) * 0.01
    dense = torch.where(slot >= 0, sc[slot.clamp(min=0).long()], parent.repeat_interleave(M)).view(B, M)
    m = ops.SELECT_ARGMAX if mode == "argmax" else ops.SELECT_MULTINOMIAL
    rng = ops.Rng(seed=3, step=9, row_offset=100) if mode == "multinomial" else None
    x_ref, _, idx_ref = ops.select(dense, cand, mode=m, rng=rng, want_soft=False)
    x_c, idx_c, sel_score, changed = ops.select_compact(sc, slot, parent, cand, mode=m, rng=rng)
    assert torch.equal(x_c, x_ref) and torch.equal(idx_c, idx_ref)
    assert torch.equal(sel_score, dense.gather(1, idx_ref.long()[:, None])[:, 0])
    assert torch.equal(changed.bool(), flags.view(B, M).gather(1, idx_ref.long()[:, None])[:, 0].bool())


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("L", [200, 50])
def test_backbone_on_compacted_rows(precision, L):
    """The one-launch backbone on a device-side row list: compacted output (candidate compaction) and in-place
    scatter (per-row logits cache) give the same bits as the full forward for the listed rows and touch nothing else."""
    from svdd_amd import ops, synthetic
    model, _, _, _ = synthetic.build("dna" if L == 200 else "rna", DEV)
    model.precision = precision
    fb = model._fused_backbone()
    n = 77
    x = torch.randint(0, 5, (n, L), device=DEV, dtype=torch.uint8)
    full = fb.forward_rows(x).clone()
    flags = (torch.rand(n, device=DEV) < 0.4).to(torch.int32)
    idx, slot, count = torch.empty(n, dtype=torch.int32, device=DEV), torch.empty(n, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.compact_flags(flags, idx, slot, count)
    k = int(count)
    comp = torch.full((n, L, 5), 7.0, device=DEV)
    fb.forward_rows(x, count=count, out=comp, row_idx=idx, scatter=False)
    assert torch.equal(comp[:k], full[idx[:k].long()])
    tiles = (k + (208 // L) - 1) // (208 // L)
    assert bool((comp[tiles * (208 // L):] == 7.0).all())                         # rows of untouched tiles are untouched
    inplace = torch.full((n, L, 5), 7.0, device=DEV)
    fb.forward_rows(x, count=count, out=inplace, row_idx=idx, scatter=True)
    live = flags.bool()
    assert torch.equal(inplace[live], full[live]) and bool((inplace[~live] == 7.0).all())
    model.precision = "f32"


# ------------------------------------------------------------------------------------------- whole decodes ----
def _decode(model, kind, emb, head, reward, B, M, S, skip, cache="auto", trace=False):
    model.skip_unchanged, model.logits_cache = skip, cache
    model.trace = [] if trace else None
    model.skip_stats = {} if skip else None
    torch.manual_seed(0)
    if kind == "mc":
        x0 = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
    else:
        x0 = model.controlled_sample_tweedie(reward, num_steps=S, eval_sp_size=B, sample_M=M, options="True", task="rna")
    torch.cuda.synchronize()
    tr, st = model.trace, model.skip_stats
    model.trace, model.skip_stats, model.skip_unchanged, model.logits_cache = None, None, True, "auto"
    return x0, tr, st


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_mc_decode_config2_skipping_is_bit_identical(precision):
    from svdd_amd import synthetic
    model, emb, head, reward = synthetic.build("dna", DEV)
    model.rng_mode, model.philox_seed, model.precision = "philox", 7, precision
    B, M, S = 256, 10, 128
    off, _, _ = _decode(model, "mc", emb, head, reward, B, M, S, skip=False)
    on, _, st = _decode(model, "mc", emb, head, reward, B, M, S, skip=True)
    cached, _, _ = _decode(model, "mc", emb, head, reward, B, M, S, skip=True, cache="on")
    assert torch.equal(on, off) and torch.equal(cached, off)
    assert st["kind"] == "mc" and 0 < st["live_candidates"] < st["candidates"] and 0 < st["changed_row_steps"] < st["row_steps"]
    model.precision = "f32"


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_mc_decode_short_sequences_skipping_is_bit_identical(precision):
    """SVDD-MC at L = 50 (several sequences per tile): the live candidates' token rows are gathered and scored as a compact
    batch, and the per-row logits cache (logits_cache = "auto" applies here) recomputes only the rows the last select
    changed. Tokens and every step's logits / scores equal the plain loop's, bit for bit."""
    from svdd_amd import synthetic
    model, emb, head, reward = synthetic.build("rna", DEV)
    model.rng_mode, model.philox_seed, model.precision = "philox", 21, precision
    B, M, S = 256, 10, 128
    assert model._use_logits_cache(50) and not model._use_logits_cache(200)
    off, tr_off, _ = _decode(model, "mc", emb, head, reward, B, M, S, skip=False, trace=True)
    on, tr_on, st = _decode(model, "mc", emb, head, reward, B, M, S, skip=True, trace=True)
    nocache, _, _ = _decode(model, "mc", emb, head, reward, B, M, S, skip=True, cache="off")
    assert torch.equal(on, off) and torch.equal(nocache, off)
    assert st["kind"] == "mc" and st["live_candidates"] < 0.8 * st["candidates"] and st["changed_row_steps"] < st["row_steps"]
    for (la, sa), (lb, sb) in zip(tr_on, tr_off):
        assert torch.equal(la, lb)
        assert (sa is None and sb is None) or torch.equal(sa, sb)
    model.precision = "f32"


def test_mc_decode_config4_shard_skipping_is_bit_identical():
    """BASELINE.json configs[3] per-GPU shard (SVDD-MC, B = 2048 / 8 = 256, L = 200, M = 20), ConvGRU value net."""
    from svdd_amd import synthetic
    model, emb, head, reward = synthetic.build("dna", DEV)
    model.rng_mode, model.philox_seed, model.row_offset = "philox", 3, 5 * 256          # the 6th rank's rows
    off, _, _ = _decode(model, "mc", emb, head, reward, 256, 20, 128, skip=False)
    on, _, st = _decode(model, "mc", emb, head, reward, 256, 20, 128, skip=True)
    model.row_offset = 0
    assert torch.equal(on, off) and int(on.max()) <= 3
    assert st["live_candidates"] < st["candidates"]


def test_mc_decode_skipping_trace_is_bit_identical():
    """Every step's logits and [B, M] scores, not just the final tokens (replay mode, so the uniforms are the reference's)."""
    from svdd_amd import synthetic
    model, emb, head, reward = synthetic.build("dna", DEV)
    model.rng_mode = "replay"
    B, M, S = 24, 6, 40
    off, tr_off, _ = _decode(model, "mc", emb, head, reward, B, M, S, skip=False, trace=True)
    on, tr_on, _ = _decode(model, "mc", emb, head, reward, B, M, S, skip=True, cache="on", trace=True)
    assert torch.equal(on, off) and len(tr_on) == len(tr_off) == S + 1
    for (la, sa), (lb, sb) in zip(tr_on, tr_off):
        assert torch.equal(la, lb)
        assert (sa is None and sb is None) or torch.equal(sa, sb)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_pm_decode_config3_skipping_is_bit_identical(precision):
    from svdd_amd import synthetic
    model, emb, head, reward = synthetic.build("rna", DEV)
    model.rng_mode, model.philox_seed, model.precision = "philox", 11, precision
    B, M, S = 256, 10, 128
    off, _, _ = _decode(model, "pm", emb, head, reward, B, M, S, skip=False)
    on, _, st = _decode(model, "pm", emb, head, reward, B, M, S, skip=True)
    assert torch.equal(on, off)
    assert st["kind"] == "pm" and st["live_candidates"] < 0.8 * st["candidates"]     # L = 50: most candidates are copies
    model.precision = "f32"


def test_pm_decode_skipping_trace_is_bit_identical():
    from svdd_amd import synthetic
    model, emb, head, reward = synthetic.build("rna", DEV)
    model.rng_mode = "replay"
    B, M, S = 16, 5, 32
    off, tr_off, _ = _decode(model, "pm", emb, head, reward, B, M, S, skip=False, trace=True)
    on, tr_on, _ = _decode(model, "pm", emb, head, reward, B, M, S, skip=True, trace=True)
    assert torch.equal(on, off) and len(tr_on) == len(tr_off)
    for (la, sa), (lb, sb) in zip(tr_on, tr_off):
        assert torch.equal(la, lb)
        assert (sa is None and sb is None) or torch.equal(sa, sb)


def test_generic_skipping_with_an_opaque_value_net():
    """skip_generic: an arbitrary nn.Module as value function (here the Enformer-shaped trunk in miniature). The live
    candidates are scored in a smaller batch; on the recorded logits / scores the oracle reproduces the tokens, the scores
    of copies equal their parent's, and the scores agree with the unskipped evaluation to fp32 round-off."""
    from oracle import svdd_oracle as orc
    from svdd_amd import synthetic
    model, emb, head, _ = synthetic.build("dna", DEV, hidden_dim=32, num_cnn_stacks=1, value="enformer",
                                          enformer_kwargs=dict(n_conv=4, channels=384, n_transformers=2, n_heads=2, key_len=16))
    B, L, M, S = 6, 200, 5, 96                     # ~2 unmaskings per candidate and step: ~10 % of the candidates are copies
    sched = model._schedule(S, 1e-5)[0]
    model.rng_mode, model.philox_seed = "philox", 13
    runs = {}
    for generic in (False, True):
        model.skip_generic, model.trace, model.state_trace = generic, [], []
        model.skip_stats = {} if generic else None
        x0 = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
        torch.cuda.synchronize()
        runs[generic] = (x0, model.trace, model.state_trace, model.skip_stats)
    model.skip_generic, model.trace, model.state_trace, model.skip_stats = False, None, None, None
    x_on, tr_on, st_on, stats = runs[True]
    assert stats["kind"] == "mc-generic" and stats["live_candidates"] < stats["candidates"]
    trace = [(lg.cpu().numpy(), None if sc is None else sc.cpu().numpy()) for lg, sc in tr_on]
    assert np.array_equal(x_on.cpu().numpy(), orc.replay_controlled_sample(trace, sched, B, L, M, seed=13))
    # step 0 has identical inputs in both runs: same logits; scores equal to round-off although the batch differs
    assert torch.equal(tr_on[0][0], runs[False][1][0][0])
    assert (tr_on[0][1] - runs[False][1][0][1]).abs().max().item() <= 1e-5


@pytest.mark.parametrize("B,L,task", [(256, 200, "dna"), (96, 50, "rna")])
def test_tds_carry_is_bit_identical(B, L, task):
    """SMC/TDS: the resampled particles are copies of proposals, so the next step's forward(x) and denominator reward are
    row gathers of this step's forward(sample) and numerator reward (2 of the 3 net evaluations of a step). Same tokens
    as evaluating everything, bit for bit."""
    import numpy as np
    from svdd_amd import synthetic
    model, _, _, reward = synthetic.build(task, DEV)
    model.rng_mode, model.philox_seed = "philox", 5
    outs = []
    for skip in (False, True):
        model.skip_unchanged = skip
        np.random.seed(11)
        outs.append(model.controlled_sample_TDS(reward, 0.5, num_steps=32, eval_sp_size=B))
    model.skip_unchanged = True
    assert model._tds_carry(reward, L) == {"keep_logits": True, "keep_den": True}
    assert torch.equal(outs[0], outs[1])
    assert len(torch.unique(outs[0], dim=0)) < B                    # the resample did duplicate particles


@pytest.mark.parametrize("kind,precision", [("mc", "f32"), ("mc", "f16x3"), ("pm", "f32"), ("tds", "f32"), ("plain", "f32"), ("mc_rna", "f32")])
def test_prior_rows_run_once_bit_identical(kind, precision):
    """Diffusion.dedup_prior: the rows of the prior x_T are B copies of the all-MASK row, so the first backbone forward of a decode
    (and the parents' first value score / tower pass) runs on ONE row and is broadcast. Tokens and every traced logit / score
    must equal the decode that forwards all B rows."""
    from svdd_amd import synthetic
    task = "rna" if kind in ("pm", "mc_rna") else "dna"
    model, emb, head, reward = synthetic.build(task, DEV)
    model.rng_mode, model.philox_seed, model.precision = "philox", 5, precision
    B, M, S = 48, 6, 12
    outs = []
    for on in (True, False):
        model.dedup_prior = on
        model.trace = []
        np.random.seed(0)
        if kind in ("mc", "mc_rna"):
            x0 = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
        elif kind == "pm":
            x0 = model.controlled_sample_tweedie(reward, num_steps=S, eval_sp_size=B, sample_M=M, options="True", task="rna")
        elif kind == "tds":
            x0 = model.controlled_sample_TDS(reward, 0.5, num_steps=S, eval_sp_size=B)
        else:
            x0 = model.decode_sample(num_steps=S, eval_sp_size=B)
        torch.cuda.synchronize()
        outs.append((x0, model.trace))
        model.trace = None
    model.dedup_prior, model.precision = True, "f32"
    assert torch.equal(outs[0][0], outs[1][0])
    assert len(outs[0][1]) == len(outs[1][1])
    for (la, sa), (lb, sb) in zip(outs[0][1], outs[1][1]):
        assert torch.equal(la, lb)
        assert (sa is None) == (sb is None) and (sa is None or torch.equal(sa, sb))


def test_mc_decode_two_part_value_net_is_bit_identical():
    """FusedValueNet.split_gru_rounds: in the late steps of a C2 decode (more live candidates than one round of GRU units) the
    compacted list runs as two parts on two streams (tower(B) -> [GRU(B), tail(B) || tower(A)] -> GRU(A) -> tail(A)). Forced on
    for EVERY step here (late_steps_from = 0: the second part is then often empty), against the one-part decode: tokens and every
    traced score identical, and the skip statistics too."""
    from svdd_amd import synthetic
    model, emb, head, _ = synthetic.build("dna", DEV)
    model.rng_mode, model.philox_seed = "philox", 3
    fn = model.value_callable(emb, head)
    outs = []
    # ("auto", round 6: two parts whenever the last step's live count — read back asynchronously — came within 3 % of one GRU round)
    for on, frm in ((True, 0.0), (True, "auto"), (False, 0.8)):
        fn.split_gru_rounds, model.late_steps_from = on, frm
        model.trace, model.skip_stats = [], {}
        x0 = model.controlled_sample(emb, head, num_steps=128, eval_sp_size=256, sample_M=10)
        torch.cuda.synchronize()
        outs.append((x0, [sc for _, sc in model.trace if sc is not None], dict(model.skip_stats)))
        model.trace, model.skip_stats = None, None
    fn.split_gru_rounds, model.late_steps_from = True, "auto"
    for o in outs[:2]:
        assert torch.equal(o[0], outs[2][0])
        assert len(o[1]) == len(outs[2][1]) == 128
        for a, b in zip(o[1], outs[2][1]):
            assert torch.equal(a, b)
        assert o[2]["live_candidates"] == outs[2][2]["live_candidates"]

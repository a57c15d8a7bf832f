"""-m gpu parity tests: the HIP kernels, called through the C ABI, against the CPU oracle on the
same seeded inputs, against the golden fixtures captured from the reference, and — at
BASELINE.json's full sizes — through size-independent properties.

Bar: bit-exact for tokens / indices / one-hots; floats (q_xs, log-probs, soft values) equal to the
oracle up to 1 ulp (both sides evaluate exp/log correctly rounded; spec tolerance is 1e-4)."""
import numpy as np
import pytest
import torch

from oracle import svdd_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from svdd_amd import ops as _ops
    from svdd_amd import _lib
    arch, ncu = _lib.device_info()
    assert arch.startswith("gfx950"), arch
    return _ops


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def bvl_view(logits_np):
    """numpy logical [R,L,5] -> GPU tensor with logical shape [R,L,5] and [R,5,L] memory."""
    return dev(np.ascontiguousarray(np.swapaxes(logits_np, 1, 2))).transpose(1, 2)


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def rand_case(rng, B, L, frac_unmasked=0.5, scale=2.0):
    logits = (rng.standard_normal((B, L, 5)) * scale).astype(np.float32)
    x = np.where(rng.random((B, L)) < frac_unmasked, rng.integers(0, 4, (B, L)), 4).astype(np.uint8)
    return logits, x


@pytest.mark.parametrize("B,L,M", [(8, 200, 6), (3, 50, 5), (1, 7, 1), (5, 64, 4), (2, 129, 13)])
@pytest.mark.parametrize("layout", [orc.BLV, orc.BVL])
def test_propose_replay_vs_oracle(ops, B, L, M, layout):
    rng = np.random.default_rng(100 + B * L + M)
    logits, x = rand_case(rng, B, L)
    logits[0, : min(L, 4)] = 0.0                       # exact ties between categories
    dm, mcs = np.float32(0.0078), np.float32(0.61)
    shape = (M, B, L, 5) if layout == orc.BLV else (M, B, 5, L)
    uni = rng.random(shape, dtype=np.float32)
    uni.flat[:3] = [0.0, (2 ** 24 - 1) / 2 ** 24, 2.0 ** -24]    # extremes of the 24-bit grid
    lg_np = logits if layout == orc.BLV else np.ascontiguousarray(np.swapaxes(logits, 1, 2))
    c_ref, oh_ref, q_ref = orc.propose(lg_np, x, dm, mcs, M, uniforms=uni, layout=layout)
    lg = dev(logits) if layout == orc.BLV else bvl_view(logits)
    cand, onehot, q = ops.propose(lg, dev(x), dm, mcs, M, ops.Rng(uniforms=dev(uni)), want_q=True)
    torch.cuda.synchronize()
    assert np.array_equal(cand.cpu().numpy(), c_ref)
    assert np.array_equal(onehot.cpu().numpy(), oh_ref)
    q_log = q.cpu().numpy()                            # logical [B,L,5]
    q_ref_log = q_ref if layout == orc.BLV else np.swapaxes(q_ref, 1, 2)
    assert ulp_diff(q_log, q_ref_log).max() <= 1
    # and without q_xs (the fast path that skips unmasked positions)
    cand2, onehot2, _ = ops.propose(lg, dev(x), dm, mcs, M, ops.Rng(uniforms=dev(uni)))
    assert torch.equal(cand2, cand) and torch.equal(onehot2, onehot)


@pytest.mark.parametrize("B,L,M", [(8, 200, 10), (3, 50, 3)])
def test_propose_philox_vs_oracle(ops, B, L, M):
    rng = np.random.default_rng(7)
    logits, x = rand_case(rng, B, L, frac_unmasked=0.3)
    dm, mcs = np.float32(0.0078), np.float32(0.2)
    for step, seed, off in [(0, 1234, 0), (77, 2 ** 63 + 5, 1000)]:
        c_ref, oh_ref, _ = orc.propose(logits, x, dm, mcs, M, seed=seed, row_offset=off, step=step, want_q=False)
        cand, onehot, _ = ops.propose(dev(logits), dev(x), dm, mcs, M, ops.Rng(seed=seed, row_offset=off, step=step))
        assert np.array_equal(cand.cpu().numpy(), c_ref)
        assert np.array_equal(onehot.cpu().numpy(), oh_ref)


@pytest.mark.parametrize("M,msplit", [(70, 1), (129, 1), (64, 1), (65, 2), (300, 0), (1000, 4)])
def test_propose_many_candidates_vs_oracle(ops, M, msplit):
    """More candidates per position than one pass of K1's LDS token table holds (64 per unit): the chunked path, for every
    way of splitting the candidates over waves (msplit forced through svdd_set_option; 0 = the host's choice), in both RNG
    modes, against the oracle."""
    from svdd_amd import _lib
    rng = np.random.default_rng(M)
    B, L = 3, 77
    logits, x = rand_case(rng, B, L, frac_unmasked=0.4)
    dm, mcs = np.float32(0.0078), np.float32(0.35)
    uni = rng.random((M, B, L, 5), dtype=np.float32)
    c_ref, oh_ref, _ = orc.propose(logits, x, dm, mcs, M, uniforms=uni, layout=orc.BLV)
    p_ref, _, _ = orc.propose(logits, x, dm, mcs, M, seed=11, row_offset=5, step=9, want_q=False)
    _lib.check(_lib.lib().svdd_set_option(1, msplit), "msplit")
    try:
        cand, onehot, _ = ops.propose(dev(logits), dev(x), dm, mcs, M, ops.Rng(uniforms=dev(uni)))
        pc, _, _ = ops.propose(dev(logits), dev(x), dm, mcs, M, ops.Rng(seed=11, row_offset=5, step=9))
        torch.cuda.synchronize()
    finally:
        _lib.check(_lib.lib().svdd_set_option(1, 0), "msplit")
    assert np.array_equal(cand.cpu().numpy(), c_ref)
    assert np.array_equal(onehot.cpu().numpy(), oh_ref)
    assert np.array_equal(pc.cpu().numpy(), p_ref)


def test_propose_philox_shard_invariance(ops):
    """Rows keyed by global index: decoding a shard with row_offset equals slicing the full batch."""
    rng = np.random.default_rng(8)
    B, L, M = 16, 200, 10
    logits, x = rand_case(rng, B, L, frac_unmasked=0.2)
    full, _, _ = ops.propose(dev(logits), dev(x), 0.0078, 0.5, M, ops.Rng(seed=9, step=3))
    for lo, hi in [(0, 8), (8, 16), (4, 5)]:
        part, _, _ = ops.propose(dev(logits[lo:hi]), dev(x[lo:hi]), 0.0078, 0.5, M, ops.Rng(seed=9, step=3, row_offset=lo))
        assert torch.equal(part, full[lo:hi])


@pytest.mark.parametrize("name,layout", [("g5_step_mc.npz", orc.BLV), ("g5_step_mc_bvl.npz", orc.BVL)])
def test_golden_g5_step(ops, golden, name, layout):
    g = golden(name)
    x = g["x"].astype(np.uint8)
    M = g["scores"].shape[1]
    lg = dev(g["logits"]) if layout == orc.BLV else bvl_view(g["logits"])
    cand, onehot, q = ops.propose(lg, dev(x), float(g["dm"]), float(g["mcs"]), M, ops.Rng(uniforms=dev(g["uniforms"])), want_q=True)
    assert np.array_equal(cand.cpu().numpy(), g["cand"])
    B, L = x.shape
    assert np.array_equal(onehot.cpu().numpy().reshape(B, M, L, 4), g["onehot"])
    assert np.allclose(q.cpu().numpy(), g["q_xs"], rtol=2e-6, atol=0)
    x_next, soft, idx = ops.select(dev(g["scores"]), cand)
    assert np.array_equal(idx.cpu().numpy(), g["idx"])
    assert np.array_equal(x_next.cpu().numpy(), g["x_next"])
    assert np.abs(soft.cpu().numpy() - g["soft"]).max() <= 1e-6        # spec: 1e-4


@pytest.mark.parametrize("M", [1, 2, 10, 20, 63, 64, 65, 200, 1024])
@pytest.mark.parametrize("mode", [0, 1])
def test_select_vs_oracle(ops, M, mode):
    rng = np.random.default_rng(M)
    B, L = 37, 50 if M > 100 else 200
    scores = (rng.standard_normal((B, M)) * 0.3).astype(np.float32)
    scores[1] = 0.25                                   # all tied -> first index
    if M > 2:
        scores[2, M - 1] = scores[2, 1] = scores[2].max() + 1.0   # exact tie -> lower index
        scores[3] = np.float32(0.1) + (rng.integers(0, 3, M) * 2.0 ** -27).astype(np.float32)  # ulp near-ties
    scores[4] *= 100.0
    cand = rng.integers(0, 5, (B, M, L)).astype(np.uint8)
    x_ref, soft_ref, idx_ref = orc.select(scores, cand, mode=mode, seed=11, row_offset=5, step=9)
    x_next, soft, idx = ops.select(dev(scores), dev(cand), mode=mode, rng=ops.Rng(seed=11, row_offset=5, step=9))
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.array_equal(x_next.cpu().numpy(), x_ref)
    assert ulp_diff(soft.cpu().numpy(), soft_ref).max() <= 1


@pytest.mark.parametrize("M,L", [(10, 200), (20, 200), (10, 50), (3, 7), (2, 200), (64, 36), (1, 200)])
@pytest.mark.parametrize("mode", [0, 1])
def test_select_rows_per_wave_equals_one_row_per_wave_and_oracle(ops, M, L, mode):
    """K2 for M <= 64 packs 64 / pow2(M) rows into a wave and skips the softmax where the best score leads by >= 2^-18
    (exact shortcut). Against the one-wave-per-row kernel (svdd_set_option A/B) on 50 k rows, and against the oracle on a
    slice: near-uniform scores (what random-init value nets give), gaps straddling the shortcut threshold, exact ties,
    clear winners; ragged last wave; row copies in 8-, 4-, 2- and 1-byte units."""
    from svdd_amd import _lib
    rng = np.random.default_rng(1000 * M + L + mode)
    B = 50001
    scores = (rng.standard_normal((B, M)) * 0.3).astype(np.float32)
    scores[:10000] = (rng.standard_normal((10000, M)) * 1e-7).astype(np.float32) + np.float32(0.01)      # near-uniform
    thr = np.float32(2.0 ** -18)
    for k, d in enumerate([thr, np.nextafter(thr, np.float32(0)), np.nextafter(thr, np.float32(1)), thr / 2, thr * 2]):
        rows = slice(10000 + 1000 * k, 11000 + 1000 * k)
        base = (rng.standard_normal((1000, M)) * 0.2).astype(np.float32)
        if M > 1:
            top = base.max(axis=1)
            j = rng.integers(0, M, 1000)
            base[np.arange(1000), j] = top + d * rng.choice([1.0, -1.0], 1000).astype(np.float32)
        scores[rows] = base
    scores[20000:21000] = np.float32(0.25)                                                          # all tied
    scores[21000:22000] = (rng.integers(-2, 3, (1000, M)) * 0.125).astype(np.float32)             # many exact ties
    cand = rng.integers(0, 5, (B, M, L)).astype(np.uint8)
    r = ops.Rng(seed=11, row_offset=5, step=9)
    sc_d, cand_d = dev(scores), dev(cand)
    fast = ops.select(sc_d, cand_d, mode=mode, rng=r, want_soft=False)
    fast_soft = ops.select(sc_d, cand_d, mode=mode, rng=r, want_soft=True)
    _lib.check(_lib.lib().svdd_set_option(2, 1), "one row per wave")
    try:
        slow = ops.select(sc_d, cand_d, mode=mode, rng=r, want_soft=True)
        torch.cuda.synchronize()
    finally:
        _lib.check(_lib.lib().svdd_set_option(2, 0), "rows per wave")
    assert torch.equal(fast[2], slow[2]) and torch.equal(fast_soft[2], slow[2])        # idx
    assert torch.equal(fast[0], slow[0]) and torch.equal(fast_soft[0], slow[0])        # gathered rows
    assert torch.equal(fast_soft[1], slow[1])                                          # soft values, bit for bit
    # round 4: four row groups per wave with the gathers batched (what batches of >= 2^21 (row, candidate) slots take)
    _lib.check(_lib.lib().svdd_set_option(2, 2), "four row groups per wave")
    try:
        wide = ops.select(sc_d, cand_d, mode=mode, rng=r, want_soft=False)
        wide_soft = ops.select(sc_d, cand_d, mode=mode, rng=r, want_soft=True)
        torch.cuda.synchronize()
    finally:
        _lib.check(_lib.lib().svdd_set_option(2, 0), "rows per wave")
    assert torch.equal(wide[2], slow[2]) and torch.equal(wide[0], slow[0])
    assert torch.equal(wide_soft[2], slow[2]) and torch.equal(wide_soft[0], slow[0]) and torch.equal(wide_soft[1], slow[1])
    sl = np.r_[0:300, 9990:15100:7, 20000:20050, 21000:21300, B - 70:B]
    x_ref, soft_ref, idx_ref = orc.select(scores[sl], cand[sl], mode=mode, seed=11, row_offset=0, step=9)
    if mode == 0:                                                                      # (Philox is keyed by the row: argmax only)
        assert np.array_equal(fast[2].cpu().numpy()[sl], idx_ref)
        assert np.array_equal(fast[0].cpu().numpy()[sl], x_ref)
    assert ulp_diff(fast_soft[1].cpu().numpy()[sl], soft_ref).max() <= 1


@pytest.mark.parametrize("layout", [orc.BLV, orc.BVL])
def test_pointwise_kernels_vs_oracle(ops, layout):
    rng = np.random.default_rng(21)
    R, L = 9, 200
    logits, x = rand_case(rng, R, L)
    logits[0, :8, :4] = 1.5                            # ties among the 4 real tokens
    lg_np = logits if layout == orc.BLV else np.ascontiguousarray(np.swapaxes(logits, 1, 2))
    lg = dev(logits) if layout == orc.BLV else bvl_view(logits)
    oh_ref, xh_ref = orc.x0hat(lg_np, x, layout=layout)
    oh, xh = ops.x0hat(lg, dev(x), want_tokens=True)
    assert np.array_equal(oh.cpu().numpy(), oh_ref) and np.array_equal(xh.cpu().numpy(), xh_ref)
    assert np.array_equal(ops.finalize(lg, dev(x)).cpu().numpy(), orc.finalize(lg_np, x, layout=layout))
    lp = ops.subs_logp(lg, dev(x)).cpu().numpy()
    lp_ref = orc.subs_logp(lg_np, x, layout=layout)
    lp_ref = lp_ref if layout == orc.BLV else np.swapaxes(lp_ref, 1, 2)
    assert ulp_diff(lp, lp_ref).max() <= 1
    for tr in (False, True):
        assert np.array_equal(ops.transform_samples(dev(x), transposed=tr).cpu().numpy(), orc.transform_samples(x, transposed=tr))


def _sched(golden, S):
    return golden("g3_schedule.npz")[f"S{S}"]


@pytest.mark.parametrize("name", ["g6_traj_mc_c1.npz", "g6_traj_mc_s16.npz"])
def test_golden_g6_trajectory(ops, golden, name):
    """The reference's own controlled_sample run (recorded logits + scores per step) replayed
    through the HIP kernels: every x_t and the final x_0 bit-exact."""
    g = golden(name)
    S, B, L, M = int(g["S"]), int(g["B"]), int(g["L"]), int(g["M"])
    sched = _sched(golden, S)
    torch.manual_seed(int(g["seed"]))
    x = torch.full((B, L), 4, dtype=torch.uint8, device=DEV)
    for i in range(S):
        assert np.array_equal(x.cpu().numpy(), g["xs"][i]), f"x_t step {i}"
        uni = torch.rand(M, B, 5, L)                   # torch CPU generator == the reference's stream
        cand, _, _ = ops.propose(bvl_view(g["logits"][i]), x, sched[i, 5], sched[i, 4], M, ops.Rng(uniforms=uni.to(DEV)))
        assert np.array_equal(cand.cpu().numpy(), g["cand"][i]), f"cand step {i}"
        x, _, _ = ops.select(dev(g["scores"][i]), cand)
    assert np.array_equal(ops.finalize(bvl_view(g["logits"][S]), x).cpu().numpy(), g["x0"])


def test_golden_g7_tweedie_trajectory(ops, golden):
    g = golden("g7_traj_pm.npz")
    S, B, L, M = int(g["S"]), int(g["B"]), int(g["L"]), int(g["M"])
    sched = _sched(golden, S)
    torch.manual_seed(int(g["seed"]))
    x = torch.full((B, L), 4, dtype=torch.uint8, device=DEV)
    for i in range(S):
        uni = torch.rand(M, B, 5, L)
        cand, _, _ = ops.propose(bvl_view(g["logits"][i]), x, sched[i, 5], sched[i, 4], M, ops.Rng(uniforms=uni.to(DEV)))
        assert np.array_equal(cand.cpu().numpy(), g["cand"][i])
        cl = g["cand_logits"][i].reshape(B * M, L, 5)
        oh, _ = ops.x0hat(bvl_view(cl), cand.reshape(B * M, L))
        assert np.array_equal(oh.cpu().numpy().reshape(B, M, 4, L), g["x0hat_onehot_t"][i].astype(np.float32))
        x, _, _ = ops.select(dev(g["scores"][i]), cand)
        assert np.array_equal(x.cpu().numpy(), g["xs"][i + 1])
    assert np.array_equal(ops.finalize(bvl_view(g["logits"][S]), x).cpu().numpy(), g["x0"])


def test_golden_g8_tds_trajectory(ops, golden):
    g = golden("g8_traj_tds.npz")
    S, B, L = int(g["S"]), int(g["B"]), int(g["L"])
    sched = _sched(golden, S)
    torch.manual_seed(int(g["seed"]))
    np.random.seed(int(g["np_seed"]))
    x = torch.full((B, L), 4, dtype=torch.uint8, device=DEV)
    for i in range(S):
        uni = torch.rand(1, B, 5, L)
        cand, _, _ = ops.propose(bvl_view(g["logits"][i]), x, sched[i, 5], sched[i, 4], 1, ops.Rng(uniforms=uni.to(DEV)))
        sample = cand[:, 0].contiguous()
        assert np.array_equal(sample.cpu().numpy(), g["samples"][i])
        u = np.random.random_sample(B)
        x, idx = ops.tds_resample(dev(g["num"][i]), dev(g["den"][i]), float(g["alpha"]), sample, dev(u))
        assert np.array_equal(x.cpu().numpy(), g["xs"][i + 1])
    assert np.array_equal(ops.finalize(bvl_view(g["logits"][S]), x).cpu().numpy(), g["x0"])


def test_golden_g10_decode_sample(ops, golden):
    g = golden("g10_decode_sample.npz")
    S, B, L = int(g["S"]), int(g["B"]), int(g["L"])
    sched = _sched(golden, S)
    torch.manual_seed(int(g["seed"]))
    x = torch.full((B, L), 4, dtype=torch.uint8, device=DEV)
    for i in range(S):
        assert np.array_equal(x.cpu().numpy(), g["xs"][i])
        cand, _, _ = ops.propose(bvl_view(g["logits"][i]), x, sched[i, 5], sched[i, 4], 1, ops.Rng(uniforms=torch.rand(1, B, 5, L).to(DEV)))
        x = cand[:, 0].contiguous()
    assert np.array_equal(ops.finalize(bvl_view(g["logits"][S]), x).cpu().numpy(), g["x0"])


@pytest.mark.parametrize("B,L", [(1, 200), (6, 200), (63, 50), (64, 200), (65, 7), (129, 200), (130, 200), (257, 50), (2048, 200),
                                 (5000, 33), (65536, 200), (140001, 8)])
def test_tds_resample_vs_oracle(ops, B, L):
    """K4 against the oracle's numpy-order restatement: pairwise float32 sum (one block, several blocks, beyond the
    parallel block capacity of 131072), serial float64 cumsum in 64-element chunks (ragged tails), searchsorted, row
    gather in 8 / 4 / 2 / 1-byte units. BASELINE configs[4] is B = 2048."""
    rng = np.random.default_rng(B)
    num = rng.standard_normal(B).astype(np.float32)
    den = rng.standard_normal(B).astype(np.float32)
    sample = rng.integers(0, 5, (B, L)).astype(np.uint8)
    u = rng.random(B)
    u[0] = 0.0
    x_ref, idx_ref, _, _ = orc.tds_resample(num, den, 0.5, sample, u)
    x, idx = ops.tds_resample(dev(num), dev(den), 0.5, dev(sample), dev(u))
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.array_equal(x.cpu().numpy(), x_ref)


def test_full_size_properties_c2(ops):
    """BASELINE config 2 sizes (B=256, L=200, M=10): size-independent invariants + a sampled
    oracle comparison on a row subset (Philox makes rows independent of the batch they sit in)."""
    rng = np.random.default_rng(2)
    B, L, M = 256, 200, 10
    logits, x = rand_case(rng, B, L, frac_unmasked=0.5)
    lg, xd = bvl_view(logits), dev(x)
    cand, onehot, _ = ops.propose(lg, xd, 0.0078, 0.5, M, ops.Rng(seed=3, step=64))
    c = cand.cpu().numpy()
    assert c.max() <= 4
    un = x != 4
    assert np.array_equal(c[un[:, None, :].repeat(M, 1)], x[:, None, :].repeat(M, 1)[un[:, None, :].repeat(M, 1)])  # copy_flag
    oh = onehot.cpu().numpy().reshape(B, M, L, 4)
    assert np.array_equal(oh.sum(-1), (c != 4).astype(np.float32))           # MASK rows are zero rows
    assert np.array_equal(oh.argmax(-1)[c != 4], c[c != 4])
    rows = [0, 17, 255]
    for r in rows:
        lg_r = np.ascontiguousarray(np.swapaxes(logits[r:r + 1], 1, 2))
        c_ref, _, _ = orc.propose(lg_r, x[r:r + 1], np.float32(0.0078), np.float32(0.5), M, seed=3, row_offset=r, step=64, want_q=False, layout=orc.BVL)
        assert np.array_equal(c[r:r + 1], c_ref)
    scores = rng.standard_normal((B, M)).astype(np.float32)
    x_next, soft, idx = ops.select(dev(scores), cand)
    i = idx.cpu().numpy()
    assert np.array_equal(i, scores.argmax(1))                               # no ties in this draw
    assert np.array_equal(x_next.cpu().numpy(), c[np.arange(B), i])
    assert np.allclose(soft.cpu().numpy().sum(1), 1.0, atol=1e-6)
    # idempotence: proposing from a fully unmasked state changes nothing
    xf = dev(rng.integers(0, 4, (B, L)).astype(np.uint8))
    cand2, _, _ = ops.propose(lg, xf, 0.0078, 0.5, M, ops.Rng(seed=3, step=65))
    assert torch.equal(cand2, xf[:, None, :].expand(B, M, L))


# ------------------------------------------------------------------ K1 filter (fast exact path)
def test_fastmath_bounds_exhaustive(ops):
    """The error bounds K1's filter assumes, measured on the device: the fast Gumbel-norm over ALL 2^24
    possible uniforms, fast exp on [-80,0], fast log on (1,4]. Margins in the kernel: 2^-21, 2^-21, 2^-22."""
    from svdd_amd import _lib
    eg, ee, el = _lib.selftest_fastmath()
    print(f"max rel err: g~ {eg:.3e} (2^{np.log2(eg):.2f}), exp {ee:.3e} (2^{np.log2(ee):.2f}), log {el:.3e} (2^{np.log2(el):.2f})")
    assert eg <= 2.0 ** -21
    assert ee <= 2.0 ** -21
    assert el <= 2.0 ** -22


@pytest.mark.parametrize("layout", [orc.BLV, orc.BVL])
def test_filter_equals_forced_exact_large(ops, layout):
    """A/B: the filtered fast path and the forced-exact path produce identical candidates on 2.5M draws
    (Philox) incl. large-magnitude logits that must take the guard."""
    from svdd_amd import _lib
    rng = np.random.default_rng(31)
    B, L, M = 128, 200, 20
    logits, x = rand_case(rng, B, L, frac_unmasked=0.1, scale=3.0)
    logits[0] *= 40.0                                   # |z| > 60: guard -> exact path
    logits[1, :, :4] = 0.5                              # uniform q: many near-ties between categories
    lg = dev(logits) if layout == orc.BLV else bvl_view(logits)
    outs = []
    for force in (False, True):
        _lib.set_force_exact(force)
        try:
            cand, onehot, _ = ops.propose(lg, dev(x), 0.0078, 0.0021, M, ops.Rng(seed=5, step=100))
            torch.cuda.synchronize()
        finally:
            _lib.set_force_exact(False)
        outs.append((cand.clone(), onehot.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_filter_adversarial_ties_vs_oracle(ops):
    """Replay uniforms crafted so that categories tie exactly or within an ulp (equal q, equal or
    adjacent u): the filter must hand these lanes to the exact path and reproduce the oracle's
    first-index rule."""
    rng = np.random.default_rng(32)
    B, L, M = 4, 192, 8
    logits = np.zeros((B, L, 5), np.float32)
    logits[..., :4] = rng.integers(-2, 3, (B, L, 1)).astype(np.float32)      # z0=z1=z2=z3 -> equal q
    x = np.full((B, L), 4, np.uint8)
    k = rng.integers(1, 2 ** 24 - 2, (M, B, L, 1))
    jitter = rng.integers(-1, 2, (M, B, L, 5))
    jitter[..., 4] = 0
    u = ((k + jitter) / 2.0 ** 24).astype(np.float32)                         # equal / adjacent uniforms
    dm = np.float32(0.0078)
    mcs = np.float32(dm * 0.25)                                               # MASK ties with the tokens too
    u[:, 1] = u[:, 1, :, :1]                                                  # row 1: all five uniforms equal
    c_ref, oh_ref, _ = orc.propose(logits, x, dm, mcs, M, uniforms=u, want_q=False)
    cand, onehot, _ = ops.propose(dev(logits), dev(x), dm, mcs, M, ops.Rng(uniforms=dev(u)))
    assert np.array_equal(cand.cpu().numpy(), c_ref)
    assert np.array_equal(onehot.cpu().numpy(), oh_ref)
    assert len(np.unique(c_ref)) >= 3                                         # the ties do exercise several outcomes


def test_golden_g11_dps_sampling(ops, golden):
    """DPS baseline: the reference's guided q_xs and the uniforms it drew (fixture g11, q laid out [B,5,L]
    like the reference's) through svdd_sample_categorical: next states bit-exact."""
    g = golden("g11_traj_dps.npz")
    S, B, L = int(g["S"]), int(g["B"]), int(g["L"])
    assert int(g["q_is_bvl"]) == 1
    for i in range(S):
        x = g["xs"][i]
        u_stream = np.ascontiguousarray(np.swapaxes(g["u"][i], 1, 2))[None]          # [1,B,5,L] = the stream order
        cand, onehot = ops.sample_categorical(bvl_view(g["q"][i]), dev(x), 1, ops.Rng(uniforms=dev(u_stream)))
        ref = orc.sample_categorical_merged(np.ascontiguousarray(np.swapaxes(g["q"][i], 1, 2)), x, u_stream, layout=orc.BVL)
        assert np.array_equal(cand.cpu().numpy(), ref)
        if i + 1 < S:
            assert np.array_equal(cand.cpu().numpy()[:, 0], g["xs"][i + 1])
        assert np.array_equal(onehot.cpu().numpy(), orc.transform_samples(ref[:, 0]))


def test_sample_categorical_philox_vs_forced_exact(ops):
    from svdd_amd import _lib
    rng = np.random.default_rng(41)
    B, L, M = 32, 200, 8
    q = rng.random((B, L, 5)).astype(np.float32) * np.float32(0.01)
    q[..., 4] = 0.5
    q[0, :, :4] = 0.0                                   # zeros: ties at 0 among losers
    x = np.where(rng.random((B, L)) < 0.3, rng.integers(0, 4, (B, L)), 4).astype(np.uint8)
    outs = []
    for force in (False, True):
        _lib.set_force_exact(force)
        try:
            cand, _ = ops.sample_categorical(dev(q), dev(x), M, ops.Rng(seed=3, step=1))
            torch.cuda.synchronize()
        finally:
            _lib.set_force_exact(False)
        outs.append(cand.clone())
    assert torch.equal(outs[0], outs[1])
    un = x != 4
    assert np.array_equal(outs[0].cpu().numpy()[un[:, None, :].repeat(M, 1)], x[:, None, :].repeat(M, 1)[un[:, None, :].repeat(M, 1)])


def test_propose_select_fuzz_vs_oracle(ops):
    """Random ragged shapes (L not a multiple of the 64-position tile, M = 1 .. 37, tiny and odd B, anything from
    all-MASK to fully decoded rows, both logits layouts, extreme move chances): propose + select through the C ABI
    equal the oracle token for token."""
    rng = np.random.default_rng(2024)
    for case in range(24):
        B = int(rng.integers(1, 20)); L = int(rng.integers(1, 260)); M = int(rng.integers(1, 38))
        frac = float(rng.choice([0.0, 0.05, 0.5, 0.95, 1.0]))
        logits, x = rand_case(rng, B, L, frac_unmasked=frac)
        logits *= np.float32(rng.choice([0.1, 1.0, 8.0]))            # flat to very peaked proposals
        dm = np.float32(rng.choice([1e-6, 0.0078, 0.3])); mcs = np.float32(rng.choice([1e-5, 0.2, 0.69]))
        seed, step, off = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 128)), int(rng.integers(0, 5000))
        layout = orc.BLV if case % 2 == 0 else orc.BVL
        lg_ref = logits if layout == orc.BLV else np.ascontiguousarray(np.swapaxes(logits, 1, 2))
        c_ref, oh_ref, _ = orc.propose(lg_ref, x, dm, mcs, M, seed=seed, row_offset=off, step=step, want_q=False,
                                       layout=layout)
        lg = dev(logits) if layout == orc.BLV else bvl_view(logits)
        cand, onehot, _ = ops.propose(lg, dev(x), float(dm), float(mcs), M, ops.Rng(seed=seed, row_offset=off, step=step))
        assert np.array_equal(cand.cpu().numpy(), c_ref), (case, B, L, M)
        assert np.array_equal(onehot.cpu().numpy(), oh_ref), (case, B, L, M)
        scores = rng.standard_normal((B, M)).astype(np.float32)
        if case % 3 == 0:
            scores[:, rng.integers(0, M)] = scores.max()             # ties: the first maximum wins
        x_ref, soft_ref, idx_ref = orc.select(scores, c_ref)
        x_next, soft, idx = ops.select(dev(scores), cand)
        assert np.array_equal(idx.cpu().numpy(), idx_ref) and np.array_equal(x_next.cpu().numpy(), x_ref), (case, B, L, M)
        assert np.abs(soft.cpu().numpy() - soft_ref).max() <= 1e-6


# ------------------------------------------------------------------ the reference's primitive fixtures, straight into the kernels
def test_g1_reference_sample_categorical_on_the_kernel(ops, golden):
    """g1: `_sample_categorical(q)` as the reference computed it (diffusion_gosai.py:30-34: q incl. rows in the unmasked
    pattern, its own `rand_like` uniforms) -> `svdd_sample_categorical` with the recorded uniforms. x is all-MASK so that no
    position takes the copy branch (:1203 is the caller's, not this primitive's); both memory layouts. Tokens bit-exact."""
    g = golden("g1_sample_categorical.npz")
    q, u, tok = g["q"], g["u"], g["tokens"]
    B, L = tok.shape
    x = dev(np.full((B, L), 4, dtype=np.uint8))
    cand, onehot = ops.sample_categorical(dev(q), x, 1, ops.Rng(uniforms=dev(u[None])))
    assert np.array_equal(cand.cpu().numpy()[:, 0], tok)
    assert np.array_equal(onehot.cpu().numpy().reshape(B, L, 4), orc.transform_samples(tok))
    cand2, _ = ops.sample_categorical(bvl_view(q), x, 1, ops.Rng(uniforms=dev(np.ascontiguousarray(np.swapaxes(u, 1, 2))[None])))
    assert np.array_equal(cand2.cpu().numpy()[:, 0], tok)


def test_g2_reference_subs_parameterization_on_the_kernel(ops, golden):
    """g2: `_subs_parameterization` outputs of the reference (:286-304, incl. exact ties) -> `svdd_subs_logp`: unmasked rows
    are exact constants, masked rows within 1 ulp (correctly-rounded exp / log here, SLEEF u10 there — DESIGN section 2)."""
    g = golden("g2_subs.npz")
    x = g["xt"].astype(np.uint8)
    for lg in (dev(g["logits"]), bvl_view(g["logits"])):
        lp = ops.subs_logp(lg, dev(x)).cpu().numpy()
        un = x != 4
        assert np.array_equal(lp[un], g["logp"][un])
        d = ulp_diff(lp, g["logp"])
        assert d.max() <= 1 and (d == 0).mean() > 0.99


def test_g4_reference_transform_samples_on_the_kernel(ops, golden):
    """g4: `transform_samples` of the reference (:1462-1470; MASK rows all-zero) -> `svdd_transform_samples`, both outputs."""
    g = golden("g4_transform.npz")
    tok = dev(g["tokens"].astype(np.uint8))
    assert np.array_equal(ops.transform_samples(tok).cpu().numpy(), g["onehot"].astype(np.float32))
    assert np.array_equal(ops.transform_samples(tok, transposed=True).cpu().numpy(),
                          g["onehot"].astype(np.float32).transpose(0, 2, 1))


# ------------------------------------------------------------------ K8: torch's CPU mt19937 stream on the device
def _mt_device_state(seed, drawn):
    """(device state [625] of torch.manual_seed(seed) after `drawn` floats, oracle generator in the same state)."""
    from svdd_amd import ops as _ops
    torch.manual_seed(seed)
    if drawn:
        torch.rand(drawn)
    wp = _ops.mt_state_from_torch(torch.get_rng_state())
    m = orc.MT19937(0)
    for k in range(624):
        m.mt[k] = int(wp[k])
    m.pos = int(wp[624])
    return torch.from_numpy(wp.astype(np.uint32).view(np.int32).copy()).to(DEV), m


@pytest.mark.parametrize("seed,drawn,ns", [(0, 0, (10, 1, 700, 5)), (44, 1010, (624,)), (7, 623, (1, 1, 2)), (3, 0, (227, 454, 681, 9999)),
                                          (11, 17, (2_560_000, 3, 256_000)), (1, 624, (1247,))])
def test_mt19937_device_stream_equals_the_host_generator(ops, seed, drawn, ns):
    """svdd_mt19937_uniform_f32 against the oracle's restatement of at::mt19937 + uniform_real_distribution<float> (pinned to
    torch by g9): consecutive calls of every length class — inside a block, across one, across many generations, the config-2
    step size (M*B*5*L = 2.56 M) — from every kind of start (fresh seed: twist first; mid-block; last word of a block)."""
    import ctypes
    from svdd_amd import _lib
    state, m = _mt_device_state(seed, drawn)
    for n in ns:
        out = torch.empty(n, dtype=torch.float32, device=DEV)
        rc = _lib.lib().svdd_mt19937_uniform_f32(state.data_ptr(), out.data_ptr(), n,
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        assert np.array_equal(out.cpu().numpy(), m.torch_rand(n)), (seed, drawn, n)
        pos = int(state[624].item())
        assert 1 <= pos <= 624


def test_mt19937_device_stream_g9_and_state_write_back(ops, golden):
    """g9 (torch.manual_seed(s); torch.rand(n) recorded from torch itself) through DeviceReplayStream, and the write-back: after
    close() torch's own generator continues the stream exactly where a host-only run would be."""
    g = golden("g9_rng.npz")
    for seed, n in [(0, 140), (44, 5000), (123456789, 256000)]:
        torch.manual_seed(seed)
        st = ops.DeviceReplayStream(DEV)
        a = st.uniforms(n // 2, prefetch=True).clone()              # a prefetch of the wrong size is rolled back
        b = st.uniforms(n - n // 2, prefetch=False).clone()
        st.close()
        r = torch.cat([a, b]).cpu().numpy()
        assert np.array_equal(r[:256], g[f"torch_s{seed}_n{n}_head"][: min(256, n)])
        assert np.array_equal(r[-256:], g[f"torch_s{seed}_n{n}_tail"])
        assert r.astype(np.float64).sum() == g[f"torch_s{seed}_n{n}_sum"]
        after = torch.rand(1000)
        torch.manual_seed(seed)
        torch.rand(n)
        assert torch.equal(after, torch.rand(1000))
    torch.manual_seed(7)
    st = ops.DeviceReplayStream(DEV)
    r = st.uniforms(3 * 50 * 5, prefetch=False).clone()
    st.close()
    assert np.array_equal(r.cpu().numpy().reshape(3, 50, 5), g["torch_s7_randlike_3_50_5"])
    # an unused prefetch of the RIGHT size is rolled back too
    torch.manual_seed(123)
    st = ops.DeviceReplayStream(DEV)
    a = st.uniforms(5000).clone()
    st.close()
    after = torch.rand(10)
    torch.manual_seed(123)
    ref = torch.rand(5010)
    assert torch.equal(a.cpu(), ref[:5000]) and torch.equal(after, ref[5000:])


@pytest.mark.parametrize("B,L", [(3, 50), (16, 200), (257, 200)])
def test_dps_kernels_vs_oracle(ops, B, L):
    """K9 (svdd_dps_probs / _probs_bwd / _guided_q: the per-position arithmetic of a DPS step, reference diffusion_gosai.py:1306-1314,
    1321-1330) against the numpy oracle, which the CPU suite pins to the reference's g11 run and to autograd of the reference's
    expressions. fp32 tolerance: exp / log differ in the last ulps (the kernels: ocml, the oracle: numpy)."""
    from oracle import svdd_oracle as orc
    rng = np.random.default_rng(B + L)
    logits = (rng.standard_normal((B, L, 5)) * 2).astype(np.float32)
    x = rng.integers(0, 5, (B, L)).astype(np.uint8)
    x[0] = 4
    if B > 1:
        x[1] = rng.integers(0, 4, L)
    tl, tx = torch.from_numpy(logits).cuda(), torch.from_numpy(x).cuda()
    p = ops.dps_probs(tl, tx)
    assert np.allclose(p.cpu().numpy(), orc.dps_probs(logits, x), rtol=2e-6, atol=1e-7)
    dp = (rng.standard_normal((B, L, 4)) * 1e-3).astype(np.float32)
    dl, dr = ops.dps_probs_bwd(tl, tx, torch.from_numpy(dp).cuda())
    wl, wr = orc.dps_probs_bwd(logits, x, dp)
    assert np.allclose(dl.cpu().numpy(), wl, rtol=1e-5, atol=1e-9) and np.allclose(dr.cpu().numpy(), wr, rtol=1e-5, atol=1e-9)
    assert float(dl[tx != 4].abs().max()) == 0.0 and float(dr[tx == 4].abs().max()) == 0.0      # each position feeds exactly one of the two paths
    ga, gb = (rng.standard_normal((B, L, 5)) * 1e-4).astype(np.float32), (rng.standard_normal((B, L, 5)) * 1e-4).astype(np.float32)
    for scale in (0.0, 300.0):
        q = ops.dps_guided_q(tl, tx, torch.from_numpy(ga).cuda(), torch.from_numpy(gb).cuda(), 0.0078, 0.31, scale)
        assert np.allclose(q.cpu().numpy(), orc.dps_guided_q(logits, x, ga + gb, 0.0078, 0.31, scale), rtol=2e-6, atol=1e-12)
    assert float(q[..., 4].min()) > 0 and torch.isfinite(q).all()


def test_side_streams_sit_on_hardware_queues_of_their_own():
    """ops.side_stream (round 6): the engine's three side streams per device are chosen so that a marker on any of them is not held up
    by a sleeping kernel on the current stream nor on another side stream — HIP has four hardware queues, and streams that share one
    run in order (the C4 trunk's two chains lost 20 % on such a pair, profiles/r06_c4_stream_probe.txt). Also: the list is process-wide
    (the same objects on every call) and a stream that IS on the current stream's queue is recognised as such."""
    from svdd_amd import ops as o
    dev = torch.device("cuda", 0)
    s = [o.side_stream(dev, k) for k in range(o.SIDE_SLOTS)]
    assert [o.side_stream(dev, k) for k in range(o.SIDE_SLOTS)] == s and o.side_stream(dev, o.SIDE_SLOTS) == s[0]
    cur = torch.cuda.current_stream(dev)
    assert len({x.cuda_stream for x in s}) == o.SIDE_SLOTS and cur.cuda_stream not in {x.cuda_stream for x in s}
    free = lambda busy, cand: sum(o._held_up_by(busy, cand) for _ in range(3)) <= 1      # noqa: E731  (majority of three timings)
    for a in s[:2]:                                     # the two slots the C4 trunk's chains use
        assert free(cur, a)
    assert free(s[0], s[1]) and free(s[1], s[0])
    assert not free(cur, cur)                           # the test itself sees a shared queue when there is one


@pytest.mark.parametrize("B,M", [(256, 10), (1 << 18, 10), (70000, 20)])
def test_select_decision_only_form_equals_the_full_kernel(ops, B, M):
    """svdd_select with x_next = NULL (round 6's measurement of what the index gather costs): the decision alone — idx — must be the
    full kernel's, at the decode's size and at the saturated sizes (the 4-row-group launch), near-tied scores included."""
    import ctypes
    from svdd_amd import _lib
    L = 200
    g = torch.Generator(device="cuda").manual_seed(B + M)
    scores = torch.randn(B, M, device="cuda", generator=g) * 1e-7 + 0.01
    scores[::7] = torch.randn((B + 6) // 7, M, device="cuda", generator=g)
    cand = torch.randint(0, 5, (B, M, L), device="cuda", generator=g, dtype=torch.uint8)
    _, _, idx = ops.select(scores, cand, want_soft=False)
    idx2 = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    rc = _lib.lib().svdd_select(scores.data_ptr(), cand.data_ptr(), B, L, M, ops.SELECT_ARGMAX, None, None, None, idx2.data_ptr(),
                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0 and torch.equal(idx, idx2)
    assert _lib.lib().svdd_select(scores.data_ptr(), cand.data_ptr(), B, L, M, ops.SELECT_ARGMAX, None, None, None, None,
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == _lib.E_ARG      # neither output: refused

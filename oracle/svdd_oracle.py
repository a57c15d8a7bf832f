"""numpy/ctypes front end of the CPU ORACLE (oracle/svdd_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (svdd_amd/) never does; it fails loudly without its HIP library.

Besides thin wrappers over the C functions, this file restates the reference's *outer loops*
(`Diffusion.controlled_sample*`, diffusion_gosai.py:1021-1145, 938-978, 888-936) on CPU so
that whole trajectories can be compared; the nets are opaque torch callables, as in the reference.

Parity status: PINNED against golden vectors generated from the reference itself
(tests/golden/make_golden.py, tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsvdd_oracle.so")

MASK = 4
VOCAB = 5
BLV = 0  # [B,L,5] contiguous: (b,l,v) at (b*L+l)*5+v
BVL = 1  # [B,5,L] contiguous: what the reference's CNN backbone output (a permuted view) is in memory


def _layout_of(logits, x_shape, layout):
    """logits is a [B,L,5] array for BLV or a [B,5,L] array for BVL."""
    B, L = x_shape
    want = (B, L, 5) if layout == BLV else (B, 5, L)
    assert tuple(logits.shape) == want, (logits.shape, want, layout)


def build(force=False):
    src = os.path.join(_HERE, "svdd_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get("SVDD_ORACLE_LIB")     # e.g. the ASan/UBSan build (oracle/Makefile: asan)
        if override:
            _lib = ctypes.CDLL(os.path.abspath(override))
        else:
            build()
            _lib = ctypes.CDLL(_SO)
        _lib.orc_philox_select_uniform.restype = ctypes.c_float
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


class MT19937(ctypes.Structure):
    _fields_ = [("mt", ctypes.c_uint32 * 624), ("pos", ctypes.c_int)]

    def __init__(self, seed):
        super().__init__()
        lib().orc_mt_seed(ctypes.byref(self), ctypes.c_uint32(seed & 0xFFFFFFFF))

    def torch_rand(self, *shape):
        """torch.manual_seed(seed); torch.rand(*shape) on CPU (float32)."""
        out = np.empty(shape, dtype=np.float32)
        lib().orc_torch_rand_f32(ctypes.byref(self), _p(out, ctypes.c_float), ctypes.c_int64(out.size))
        return out

    def numpy_random_sample(self, n):
        """np.random.seed(seed); np.random.random_sample(n)."""
        out = np.empty(n, dtype=np.float64)
        lib().orc_numpy_random_sample(ctypes.byref(self), _p(out, ctypes.c_double), ctypes.c_int64(n))
        return out


def philox_uniform5(seed, pos, step, m):
    u = np.empty(5, dtype=np.float32)
    lib().orc_philox_uniform5(ctypes.c_uint64(seed), ctypes.c_uint64(pos), ctypes.c_uint32(step),
                              ctypes.c_uint32(m), _p(u, ctypes.c_float))
    return u


def philox_select_uniform(seed, row, step):
    return float(lib().orc_philox_select_uniform(ctypes.c_uint64(seed), ctypes.c_uint64(row), ctypes.c_uint32(step)))


def subs_logp(logits, x, layout=BLV):
    logits = _f32(logits); x = _u8(x)
    _layout_of(logits, x.shape, layout)
    out = np.empty_like(logits)
    lib().orc_subs_logp(_p(logits, ctypes.c_float), _p(x, ctypes.c_uint8), x.shape[0], x.shape[1], layout, _p(out, ctypes.c_float))
    return out


# ---- DPS (gradient guidance), the per-position arithmetic around the two net passes (reference diffusion_gosai.py:1306-1314, 1321-1330):
# numpy restatement of what svdd_dps_probs / svdd_dps_probs_bwd / svdd_dps_guided_q compute (K9). Pinned on the CPU against the
# reference's recorded guided q_xs (g11: its own controlled_sample_DPS run) and against torch autograd of the reference's expressions
# (tests/test_oracle_golden.py); the kernels are checked against THIS on the GPU (tests/test_kernels_gpu.py).
def _dps_expected_probs(logits, x):
    """log p(x0 | x_t) (SUBS, :286-304), E = copy * onehot(x) + (1 - copy) * log p (:1325), softmax(E, dim=2) (:1326) — all [B, L, 5]."""
    lp = subs_logp(logits, x).astype(np.float32)
    keep = (np.asarray(x) != MASK)[..., None]
    onehot = np.eye(5, dtype=np.float32)[np.asarray(x, dtype=np.int64)]
    E = np.where(keep, onehot, lp).astype(np.float32)
    e = np.exp(E - E.max(axis=2, keepdims=True), dtype=np.float32)
    return lp, keep, (e / e.sum(axis=2, keepdims=True, dtype=np.float32)).astype(np.float32)


def dps_probs(logits, x):
    """-> softmax(E)[..., 0:4] fp32 [B, L, 4]: the reward model's input (:1327, before its transpose)."""
    return np.ascontiguousarray(_dps_expected_probs(logits, x)[2][..., :4])


def dps_probs_bwd(logits, x, dprobs4):
    """d loss / d probs4 -> (d loss / d logits through forward2's SUBS — zero at unmasked positions —, the direct term copy * dE)."""
    lp, keep, pr = _dps_expected_probs(logits, x)
    dp = np.concatenate([np.asarray(dprobs4, np.float32), np.zeros(pr.shape[:2] + (1,), np.float32)], axis=2)
    dE = pr * (dp - (pr * dp).sum(axis=2, keepdims=True, dtype=np.float32))              # softmax backward
    dlogits = np.where(keep, 0.0, dE - np.exp(lp, dtype=np.float32) * dE.sum(axis=2, keepdims=True, dtype=np.float32))
    return dlogits.astype(np.float32), np.where(keep, dE, 0.0).astype(np.float32)


def dps_guided_q(logits, x, x_grad, dm, mcs, scale):
    """q_xs = exp(log p) * (mct - mcs) ; q_xs[MASK] = mcs ; q_xs * exp(scale * (x_grad - x_grad[MASK]))   (:1306-1314) -> fp32 [B, L, 5]."""
    lp = subs_logp(logits, x).astype(np.float32)
    q = (np.exp(lp, dtype=np.float32) * np.float32(dm)).astype(np.float32)
    q[..., MASK] = np.float32(mcs)
    g = np.asarray(x_grad, np.float32)
    return (q * np.exp(np.float32(scale) * (g - g[..., MASK:MASK + 1]), dtype=np.float32)).astype(np.float32)


def sample_categorical(q, u):
    q = _f32(q); u = _f32(u)
    tok = np.empty(q.shape[:-1], dtype=np.uint8)
    lib().orc_sample_categorical(_p(q, ctypes.c_float), _p(u, ctypes.c_float), ctypes.c_int64(tok.size), _p(tok, ctypes.c_uint8))
    return tok


def sample_categorical_merged(q, x, uniforms, layout=BLV):
    """M x (copy_flag * x + (1 - copy_flag) * _sample_categorical(q)) for a caller-built q
    (DPS, diffusion_gosai.py:1316-1319). q: [B,L,5] (BLV) or [B,5,L] (BVL); uniforms [M, *q.shape]. -> cand u8 [B,M,L]."""
    q = _f32(q); x = _u8(x); uniforms = _f32(uniforms)
    B, L = x.shape
    _layout_of(q, x.shape, layout)
    M = uniforms.shape[0]
    ql = q if layout == BLV else np.ascontiguousarray(np.swapaxes(q, 1, 2))
    out = np.empty((B, M, L), dtype=np.uint8)
    for m in range(M):
        ul = uniforms[m] if layout == BLV else np.ascontiguousarray(np.swapaxes(uniforms[m], 1, 2))
        tok = sample_categorical(ql, ul)
        out[:, m] = np.where(x != MASK, x, tok)
    return out


def propose(logits, x, dm, mcs, M, uniforms=None, seed=0, row_offset=0, step=0, want_q=True, layout=BLV):
    """-> (cand u8 [B,M,L], onehot f32 [B*M,L,4], q_xs f32 (same layout as logits) | None).

    layout BLV: logits [B,L,5], uniforms [M,B,L,5]; layout BVL: logits [B,5,L], uniforms [M,B,5,L]."""
    logits = _f32(logits); x = _u8(x)
    B, L = x.shape
    _layout_of(logits, x.shape, layout)
    cand = np.empty((B, M, L), dtype=np.uint8)
    onehot = np.empty((B * M, L, 4), dtype=np.float32)
    q = np.empty(logits.shape, dtype=np.float32) if want_q else None
    kind = 0 if uniforms is not None else 1
    if uniforms is not None:
        uniforms = _f32(uniforms)
        assert uniforms.shape == (M,) + tuple(logits.shape), uniforms.shape
    lib().orc_propose(_p(logits, ctypes.c_float), _p(x, ctypes.c_uint8), ctypes.c_float(dm), ctypes.c_float(mcs),
                      B, L, M, layout, kind, _p(uniforms, ctypes.c_float), ctypes.c_uint64(seed), ctypes.c_uint64(row_offset),
                      ctypes.c_uint32(step), _p(cand, ctypes.c_uint8), _p(onehot, ctypes.c_float), _p(q, ctypes.c_float))
    return cand, onehot, q


def select(scores, cand, mode=0, seed=0, row_offset=0, step=0):
    """-> (x_next u8 [B,L], soft f32 [B,M], idx i32 [B])."""
    scores = _f32(scores); cand = _u8(cand)
    B, M, L = cand.shape
    assert scores.shape == (B, M)
    x_next = np.empty((B, L), dtype=np.uint8)
    soft = np.empty((B, M), dtype=np.float32)
    idx = np.empty(B, dtype=np.int32)
    lib().orc_select(_p(scores, ctypes.c_float), _p(cand, ctypes.c_uint8), B, L, M, mode, ctypes.c_uint64(seed),
                     ctypes.c_uint64(row_offset), ctypes.c_uint32(step), _p(x_next, ctypes.c_uint8),
                     _p(soft, ctypes.c_float), _p(idx, ctypes.c_int32))
    return x_next, soft, idx


def x0hat(logits, xt, layout=BLV):
    """-> (onehot_t f32 [R,4,L], x0hat u8 [R,L])."""
    logits = _f32(logits); xt = _u8(xt)
    R, L = xt.shape
    _layout_of(logits, xt.shape, layout)
    oh = np.empty((R, 4, L), dtype=np.float32)
    xh = np.empty((R, L), dtype=np.uint8)
    lib().orc_x0hat(_p(logits, ctypes.c_float), _p(xt, ctypes.c_uint8), R, L, layout, _p(oh, ctypes.c_float), _p(xh, ctypes.c_uint8))
    return oh, xh


def finalize(logits, x, layout=BLV):
    logits = _f32(logits); x = _u8(x)
    _layout_of(logits, x.shape, layout)
    out = np.empty(x.shape, dtype=np.int64)
    lib().orc_finalize(_p(logits, ctypes.c_float), _p(x, ctypes.c_uint8), x.shape[0], x.shape[1], layout, _p(out, ctypes.c_int64), None)
    return out


def transform_samples(tok, transposed=False):
    tok = _u8(tok)
    R, L = tok.shape
    out = np.empty((R, 4, L) if transposed else (R, L, 4), dtype=np.float32)
    lib().orc_transform_samples(_p(tok, ctypes.c_uint8), R, L, int(bool(transposed)), _p(out, ctypes.c_float))
    return out


def tds_resample(reward_num, reward_den, alpha, sample, u):
    """-> (x_next u8 [B,L], idx i32 [B], ratio f32 [B], cdf f64 [B])."""
    num = _f32(reward_num); den = _f32(reward_den); sample = _u8(sample)
    u = np.ascontiguousarray(u, dtype=np.float64)
    B, L = sample.shape
    x_next = np.empty((B, L), dtype=np.uint8)
    idx = np.empty(B, dtype=np.int32)
    ratio = np.empty(B, dtype=np.float32)
    cdf = np.empty(B, dtype=np.float64)
    lib().orc_tds_resample(_p(num, ctypes.c_float), _p(den, ctypes.c_float), ctypes.c_double(alpha),
                           _p(sample, ctypes.c_uint8), _p(u, ctypes.c_double), B, L, _p(x_next, ctypes.c_uint8),
                           _p(idx, ctypes.c_int32), _p(ratio, ctypes.c_float), _p(cdf, ctypes.c_double))
    return x_next, idx, ratio, cdf


def move_chances(t, dt):
    """Correctly-rounded fp32 (mct, mcs, mct-mcs) for one step (cross-check of the host table)."""
    out = np.empty(3, dtype=np.float32)
    lib().orc_move_chances(ctypes.c_float(t), ctypes.c_float(dt), _p(out, ctypes.c_float))
    return out


# ----------------------------------------------------------------------------------------
# Outer loops (CPU).  Nets are torch modules/callables on CPU; everything between them is
# the C oracle.  `schedule` is the (S, 3) fp32 table of (mct, mcs, mct - mcs) and
# `uniform_fn(step, M, B, L)` returns the [M,B,L,5] uniforms of that step (replay) or None
# for Philox.
# ----------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def controlled_sample(backbone, value_fn, schedule, B, L, M, uniform_fn=None, seed=0,
                      row_offset=0, mode=0, batched_value=False, record=None):
    """SVDD-MC: Diffusion.controlled_sample, diffusion_gosai.py:1021-1061 + :1174-1228.

    backbone(x_int64[B,L]) -> raw logits f32 [B,L,5]; value_fn(onehot f32 [n,L,4]) -> [n] scores.
    The reference calls the value net M times with batch B (:1207-1209); `batched_value`
    calls it once with B*M rows instead (the engine's documented deviation)."""
    torch = _torch()
    S = schedule.shape[0]
    x = np.full((B, L), MASK, dtype=np.uint8)                                   # :1033, :751-753
    for i in range(S):
        with torch.no_grad():
            logits = backbone(torch.from_numpy(x.astype(np.int64))).float().numpy()  # :1189
        mct, mcs, dm = (float(v) for v in schedule[i])
        uni = uniform_fn(i, M, B, L) if uniform_fn is not None else None
        cand, onehot, _ = propose(logits, x, dm, mcs, M, uniforms=uni, seed=seed, row_offset=row_offset,
                                  step=i, want_q=False)
        with torch.no_grad():
            if batched_value:
                sc = value_fn(torch.from_numpy(onehot)).reshape(B, M).float().numpy()
            else:
                oh = onehot.reshape(B, M, L, 4)
                sc = np.stack([value_fn(torch.from_numpy(np.ascontiguousarray(oh[:, m]))).reshape(B).float().numpy()
                               for m in range(M)], axis=1)                      # :1207-1209,1219
        x_next, soft, idx = select(sc, cand, mode=mode, seed=seed, row_offset=row_offset, step=i)
        if record is not None:
            record.append(dict(x=x.copy(), logits=logits, scores=sc, cand=cand, soft=soft, idx=idx, x_next=x_next))
        x = x_next
    with torch.no_grad():
        logits = backbone(torch.from_numpy(x.astype(np.int64))).float().numpy()  # :1049-1060
    return finalize(logits, x)


def decode_sample(backbone, schedule, B, L, uniform_fn=None, seed=0, row_offset=0):
    """Un-guided ancestral decode: Diffusion.decode_sample, diffusion_gosai.py:888-936 + :1147-1172."""
    torch = _torch()
    S = schedule.shape[0]
    x = np.full((B, L), MASK, dtype=np.uint8)
    for i in range(S):
        with torch.no_grad():
            logits = backbone(torch.from_numpy(x.astype(np.int64))).float().numpy()
        mct, mcs, dm = (float(v) for v in schedule[i])
        uni = uniform_fn(i, 1, B, L) if uniform_fn is not None else None
        cand, _, _ = propose(logits, x, dm, mcs, 1, uniforms=uni, seed=seed, row_offset=row_offset, step=i, want_q=False)
        x = cand[:, 0]
    with torch.no_grad():
        logits = backbone(torch.from_numpy(x.astype(np.int64))).float().numpy()
    return finalize(logits, x)


def controlled_sample_tweedie(backbone, reward_fn, schedule, B, L, M, uniform_fn=None, seed=0,
                              row_offset=0, mode=0, record=None):
    """SVDD-PM: controlled_sample_tweedie, diffusion_gosai.py:1105-1145 + :1373-1460 (options == "True").

    reward_fn(onehot f32 [n,4,L]) -> [n] (task 0)."""
    torch = _torch()
    S = schedule.shape[0]
    x = np.full((B, L), MASK, dtype=np.uint8)
    for i in range(S):
        with torch.no_grad():
            logits = backbone(torch.from_numpy(x.astype(np.int64))).float().numpy()
        mct, mcs, dm = (float(v) for v in schedule[i])
        uni = uniform_fn(i, M, B, L) if uniform_fn is not None else None
        cand, _, _ = propose(logits, x, dm, mcs, M, uniforms=uni, seed=seed, row_offset=row_offset, step=i, want_q=False)
        sc = np.empty((B, M), dtype=np.float32)
        for m in range(M):                                                         # :1413-1436
            xm = np.ascontiguousarray(cand[:, m])
            with torch.no_grad():
                lg = backbone(torch.from_numpy(xm.astype(np.int64))).float().numpy()  # :1415
                oh, _ = x0hat(lg, xm)                                                  # :1416-1419
                sc[:, m] = reward_fn(torch.from_numpy(oh)).reshape(B).float().numpy()  # :1430
        x_next, soft, idx = select(sc, cand, mode=mode, seed=seed, row_offset=row_offset, step=i)
        if record is not None:
            record.append(dict(x=x.copy(), logits=logits, scores=sc, cand=cand, idx=idx, x_next=x_next))
        x = x_next
    with torch.no_grad():
        logits = backbone(torch.from_numpy(x.astype(np.int64))).float().numpy()
    return finalize(logits, x)


def controlled_sample_tds(backbone, reward_fn, schedule, alpha, B, L, uniform_fn, choice_u_fn, record=None):
    """SMC/TDS baseline: controlled_sample_TDS, diffusion_gosai.py:938-978 + :1230-1284.

    choice_u_fn(step, B) -> the B float64 uniforms np.random.choice consumes in that step."""
    torch = _torch()
    S = schedule.shape[0]
    x = np.full((B, L), MASK, dtype=np.uint8)
    for i in range(S):
        with torch.no_grad():
            logits = backbone(torch.from_numpy(x.astype(np.int64))).float().numpy()
        mct, mcs, dm = (float(v) for v in schedule[i])
        cand, _, _ = propose(logits, x, dm, mcs, 1, uniforms=uniform_fn(i, 1, B, L), want_q=False)
        sample = np.ascontiguousarray(cand[:, 0])
        with torch.no_grad():
            lg = backbone(torch.from_numpy(sample.astype(np.int64))).float().numpy()   # :1263
            oh, _ = x0hat(lg, sample)
            num = reward_fn(torch.from_numpy(oh)).reshape(B).float().numpy()           # :1269
            lg = backbone(torch.from_numpy(x.astype(np.int64))).float().numpy()        # :1273
            oh, _ = x0hat(lg, x)
            den = reward_fn(torch.from_numpy(oh)).reshape(B).float().numpy()           # :1277
        x_next, idx, ratio, _ = tds_resample(num, den, alpha, sample, choice_u_fn(i, B))
        if record is not None:
            record.append(dict(x=x.copy(), sample=sample, num=num, den=den, idx=idx, x_next=x_next))
        x = x_next
    with torch.no_grad():
        logits = backbone(torch.from_numpy(x.astype(np.int64))).float().numpy()
    return finalize(logits, x)


def replay_controlled_sample(trace, schedule, B, L, M, uniform_fn=None, seed=0, row_offset=0, mode=0, layout=BLV):
    """SVDD-MC / SVDD-PM outer loop on RECORDED per-step (logits, scores): `trace` is a list of S
    (logits, scores[B,M]) pairs followed by the (logits, None) of the noise-removal forward. Everything
    but the nets is recomputed here, so the result is independent of whether the nets are run-to-run
    deterministic."""
    S = schedule.shape[0]
    assert len(trace) == S + 1, (len(trace), S)
    x = np.full((B, L), MASK, dtype=np.uint8)
    for i in range(S):
        logits, scores = trace[i]
        mct, mcs, dm = (float(v) for v in schedule[i])
        uni = uniform_fn(i, M, B, L) if uniform_fn is not None else None
        cand, _, _ = propose(logits, x, dm, mcs, M, uniforms=uni, seed=seed, row_offset=row_offset, step=i,
                             want_q=False, layout=layout)
        x, _, _ = select(scores, cand, mode=mode, seed=seed, row_offset=row_offset, step=i)
    return finalize(trace[S][0], x, layout=layout)

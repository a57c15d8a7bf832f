"""Torch-tensor front end of the C ABI (include/svdd_hip.h): raw pointers of CUDA(HIP) tensors
are handed to libsvdd_hip.so on torch's current stream. No fallbacks: tensors must live on the
GPU, and the library must be present.

Token tensors are uint8 (0..3 = A,C,G,T; 4 = MASK). `[B,L,5]` float tensors may be contiguous
(layout BLV) or a permuted view of a `[B,5,L]` buffer (layout BVL, what a Conv1d backbone
returns after `.permute(0, 2, 1)`, reference models/dnaconv.py:201); both are consumed in place.
"""
import ctypes
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import (LAYOUT_BLV, LAYOUT_BVL, RNG_PHILOX, RNG_REPLAY, SELECT_ARGMAX,  # noqa: F401
                   SELECT_MULTINOMIAL, SvddError, SvddRng)

MASK = 4


@dataclass
class Rng:
    """Uniform source for the categorical draws.

    replay : uniforms tensor = M consecutive blocks in the same memory layout as the logits
             (the bytes `M x rand_like(q_xs)` produce from torch's CPU mt19937 stream).
    philox : counter-based, keyed by (seed, step, row_offset + b, m, l)."""
    uniforms: Optional[torch.Tensor] = None
    seed: int = 0
    row_offset: int = 0
    step: int = 0
    uniforms_layout: Optional[int] = None      # None: same layout as the logits passed to propose()
    uniforms_rows: int = 0                     # replay: > 0 = the blocks hold a WHOLE batch of that many rows; ours start at row_offset

    def c_struct(self, logits_layout=LAYOUT_BLV):
        if self.uniforms is not None:
            ul = logits_layout if self.uniforms_layout is None else self.uniforms_layout
            return SvddRng(RNG_REPLAY, 0, self.uniforms.data_ptr(), 0, self.row_offset if self.uniforms_rows else 0, ul,
                           self.uniforms_rows)
        return SvddRng(RNG_PHILOX, self.step, None, self.seed & 0xFFFFFFFFFFFFFFFF, self.row_offset, 0, 0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise SvddError(f"{name} must be a GPU tensor (the SVDD hot path has no CPU fallback)")
    if t.dtype != dtype:
        raise SvddError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def layout_of(t):
    """(tensor-to-pass, layout) for a logical [R,L,5] fp32 tensor."""
    assert t.dim() == 3 and t.shape[2] == 5, t.shape
    if t.is_contiguous():
        return t, LAYOUT_BLV
    if t.transpose(1, 2).is_contiguous():
        return t, LAYOUT_BVL
    return t.contiguous(), LAYOUT_BLV


def _empty_like_layout(t, layout):
    if layout == LAYOUT_BLV:
        return torch.empty(t.shape, dtype=torch.float32, device=t.device)
    R, L, V = t.shape
    return torch.empty((R, V, L), dtype=torch.float32, device=t.device).transpose(1, 2)


def propose(logits, x, dm, mcs, M, rng, want_q=False, cand=None, onehot=None):
    """-> (cand u8 [B,M,L], onehot f32 [B*M,L,4], q_xs f32 [B,L,5] (same strides as logits) | None)."""
    logits = _need(logits, torch.float32, "logits")
    x = _need(x, torch.uint8, "x").contiguous()
    B, L = x.shape
    logits, layout = layout_of(logits)
    assert logits.shape == (B, L, 5), (logits.shape, x.shape)
    if cand is None:
        cand = torch.empty((B, M, L), dtype=torch.uint8, device=x.device)
    if onehot is None:
        onehot = torch.empty((B * M, L, 4), dtype=torch.float32, device=x.device)
    q = _empty_like_layout(logits, layout) if want_q else None
    if rng.uniforms is not None:
        u = _need(rng.uniforms, torch.float32, "uniforms")
        rows = rng.uniforms_rows if rng.uniforms_rows else B       # (a shard of a batch of uniforms_rows rows: the whole batch's blocks)
        assert u.is_contiguous() and u.numel() == M * rows * L * 5 and rng.row_offset * bool(rng.uniforms_rows) + B <= rows, (u.shape, M, B, L)
    rs = rng.c_struct(layout)
    rc = _lib.lib().svdd_propose(logits.data_ptr(), x.data_ptr(), float(dm), float(mcs), B, L, M, layout,
                                 ctypes.byref(rs), cand.data_ptr(), onehot.data_ptr(),
                                 q.data_ptr() if q is not None else None, _stream())
    _lib.check(rc, "svdd_propose")
    return cand, onehot, q


def sample_categorical(q, x, M, rng):
    """M categorical draws from a caller-built q [B,L,5] (>= 0), merged with copy_flag -> (cand, onehot)."""
    q = _need(q, torch.float32, "q")
    x = _need(x, torch.uint8, "x").contiguous()
    B, L = x.shape
    q, layout = layout_of(q)
    cand = torch.empty((B, M, L), dtype=torch.uint8, device=x.device)
    onehot = torch.empty((B * M, L, 4), dtype=torch.float32, device=x.device)
    rs = rng.c_struct(layout)
    rc = _lib.lib().svdd_sample_categorical(q.data_ptr(), x.data_ptr(), B, L, M, layout, ctypes.byref(rs),
                                            cand.data_ptr(), onehot.data_ptr(), _stream())
    _lib.check(rc, "svdd_sample_categorical")
    return cand, onehot


def select(scores, cand, mode=SELECT_ARGMAX, rng=None, want_soft=True, x_next=None):
    """-> (x_next u8 [B,L], soft f32 [B,M] | None, idx i32 [B])."""
    cand = _need(cand, torch.uint8, "cand").contiguous()
    B, M, L = cand.shape
    scores = _need(scores, torch.float32, "scores").contiguous()
    assert scores.numel() == B * M, (scores.shape, cand.shape)
    if x_next is None:
        x_next = torch.empty((B, L), dtype=torch.uint8, device=cand.device)
    soft = torch.empty((B, M), dtype=torch.float32, device=cand.device) if want_soft else None
    idx = torch.empty((B,), dtype=torch.int32, device=cand.device)
    rs = rng.c_struct() if rng is not None else None
    rc = _lib.lib().svdd_select(scores.data_ptr(), cand.data_ptr(), B, L, M, mode,
                                ctypes.byref(rs) if rs is not None else None, x_next.data_ptr(),
                                soft.data_ptr() if soft is not None else None, idx.data_ptr(), _stream())
    _lib.check(rc, "svdd_select")
    return x_next, soft, idx


def select_compact(scores_c, slot, parent_score, cand, mode=SELECT_ARGMAX, rng=None, x_next=None, sel_score=None,
                   changed=None, idx=None):
    """svdd_select_compact: select on the scores of the LIVE candidates only. scores_c f32 [>= count] (compacted),
    slot i32 [B*M] (position in scores_c, or -1 for a copy of the parent), parent_score f32 [B].
    -> (x_next u8 [B,L], idx i32 [B], sel_score f32 [B], changed i32 [B])."""
    cand = _need(cand, torch.uint8, "cand").contiguous()
    B, M, L = cand.shape
    dev = cand.device
    scores_c = _need(scores_c, torch.float32, "scores").contiguous()
    x_next = torch.empty((B, L), dtype=torch.uint8, device=dev) if x_next is None else x_next
    idx = torch.empty((B,), dtype=torch.int32, device=dev) if idx is None else idx
    sel_score = torch.empty((B,), dtype=torch.float32, device=dev) if sel_score is None else sel_score
    changed = torch.empty((B,), dtype=torch.int32, device=dev) if changed is None else changed
    rs = rng.c_struct() if rng is not None else None
    rc = _lib.lib().svdd_select_compact(scores_c.data_ptr(), slot.data_ptr(), parent_score.data_ptr(), cand.data_ptr(),
                                        B, L, M, mode, ctypes.byref(rs) if rs is not None else None, x_next.data_ptr(),
                                        None, idx.data_ptr(), sel_score.data_ptr(), changed.data_ptr(), _stream())
    _lib.check(rc, "svdd_select_compact")
    return x_next, idx, sel_score, changed


def compact_flags(flags, live_idx, slot, count):
    """Stable device-side compaction (svdd_compact_flags): live_idx[k] = i, slot[i] = k for the k-th non-zero flag,
    slot[i] = -1 otherwise, count[0] = number of non-zero flags. All int32 device tensors; nothing returns to the host."""
    rc = _lib.lib().svdd_compact_flags(flags.data_ptr(), flags.numel(), live_idx.data_ptr(), slot.data_ptr(), count.data_ptr(),
                                       _stream())
    _lib.check(rc, "svdd_compact_flags")


def compact_by_key(key, live_idx, slot, count, split=0):
    """The same compaction ordered by key, largest first, stable inside a key (svdd_compact_by_key): key[i] > 0 = live.
    split > 0: count has 3 entries and also receives the lengths of the list's parts [0, split) and [split, ...)."""
    assert split == 0 or count.numel() >= 3
    rc = _lib.lib().svdd_compact_by_key(key.data_ptr(), key.numel(), live_idx.data_ptr(), slot.data_ptr(), count.data_ptr(), int(split),
                                        _stream())
    _lib.check(rc, "svdd_compact_by_key")


def gather_rows(src, idx, count, dst):
    """dst[i] = src[idx[i]] for i < count[0] (row tensors of equal row size, contiguous)."""
    n = src.shape[0]
    row_bytes = src[0].numel() * src.element_size()
    rc = _lib.lib().svdd_gather_rows(src.data_ptr(), idx.data_ptr(), count.data_ptr() if count is not None else None, n,
                                     row_bytes, dst.data_ptr(), _stream())
    _lib.check(rc, "svdd_gather_rows")
    return dst


def advance_rows(src, slot, sel, dst, M):
    """The selected candidate becomes the next parent: dst[b] = src[slot[b*M + sel[b]]] where that slot is >= 0."""
    B = dst.shape[0]
    row_bytes = dst[0].numel() * dst.element_size()
    rc = _lib.lib().svdd_advance_rows(src.data_ptr(), slot.data_ptr(), sel.data_ptr(), B, M, row_bytes, dst.data_ptr(), _stream())
    _lib.check(rc, "svdd_advance_rows")
    return dst


def x0hat(logits, xt, want_tokens=False, want_onehot=True):
    """-> (onehot_t f32 [R,4,L] | None, x0hat u8 [R,L] | None)."""
    logits = _need(logits, torch.float32, "logits")
    xt = _need(xt, torch.uint8, "xt").contiguous()
    R, L = xt.shape
    logits, layout = layout_of(logits)
    oh = torch.empty((R, 4, L), dtype=torch.float32, device=xt.device) if want_onehot else None
    xh = torch.empty((R, L), dtype=torch.uint8, device=xt.device) if want_tokens else None
    rc = _lib.lib().svdd_x0hat(logits.data_ptr(), xt.data_ptr(), R, L, layout, oh.data_ptr() if oh is not None else None,
                               xh.data_ptr() if xh is not None else None, _stream())
    _lib.check(rc, "svdd_x0hat")
    return oh, xh


def finalize(logits, x):
    """Noise-removal argmax -> int64 [B,L] (the API's LongTensor)."""
    logits = _need(logits, torch.float32, "logits")
    x = _need(x, torch.uint8, "x").contiguous()
    B, L = x.shape
    logits, layout = layout_of(logits)
    out = torch.empty((B, L), dtype=torch.int64, device=x.device)
    rc = _lib.lib().svdd_finalize(logits.data_ptr(), x.data_ptr(), B, L, layout, out.data_ptr(), None, _stream())
    _lib.check(rc, "svdd_finalize")
    return out


def transform_samples(tok, transposed=False):
    """tokens u8 [R,L] -> one-hot f32 [R,L,4] (or [R,4,L]); MASK rows are zero."""
    tok = _need(tok, torch.uint8, "tok").contiguous()
    R, L = tok.shape
    out = torch.empty((R, 4, L) if transposed else (R, L, 4), dtype=torch.float32, device=tok.device)
    rc = _lib.lib().svdd_transform_samples(tok.data_ptr(), R, L, int(bool(transposed)), out.data_ptr(), _stream())
    _lib.check(rc, "svdd_transform_samples")
    return out


def subs_logp(logits, x):
    """Diffusion.forward()'s SUBS re-parameterisation -> log p(x0|xt), same strides as logits."""
    logits = _need(logits, torch.float32, "logits")
    x = _need(x, torch.uint8, "x").contiguous()
    B, L = x.shape
    logits, layout = layout_of(logits)
    out = _empty_like_layout(logits, layout)
    rc = _lib.lib().svdd_subs_logp(logits.data_ptr(), x.data_ptr(), B, L, layout, out.data_ptr(), _stream())
    _lib.check(rc, "svdd_subs_logp")
    return out


def dps_probs(logits, x):
    """softmax(E)[..., 0:4] of a DPS step, E = keep * onehot(x) + (1 - keep) * log p(x0 | x) (reference diffusion_gosai.py:1325-1328):
    logits fp32 [B, L, 5] contiguous (the backbone's raw output), x u8 [B, L] -> fp32 [B, L, 4], the reward net's input."""
    logits = _need(logits, torch.float32, "logits")
    x = _need(x, torch.uint8, "x").contiguous()
    assert logits.is_contiguous() and logits.shape == (*x.shape, 5)
    B, L = x.shape
    out = torch.empty((B, L, 4), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().svdd_dps_probs(logits.data_ptr(), x.data_ptr(), B, L, out.data_ptr(), _stream()), "svdd_dps_probs")
    return out


def dps_probs_bwd(logits, x, dprobs4):
    """d loss / d probs4 [B, L, 4] -> (d loss / d logits [B, L, 5] — what the backbone's gradient kernel takes; zero at unmasked
    positions — , direct [B, L, 5] = the gradient through `keep * x_onehot`, zero at masked positions)."""
    x = _need(x, torch.uint8, "x").contiguous()
    B, L = x.shape
    g = dprobs4.contiguous().float()
    assert logits.is_contiguous() and logits.shape == (B, L, 5) and g.shape == (B, L, 4)
    dlogits = torch.empty((B, L, 5), dtype=torch.float32, device=x.device)
    direct = torch.empty_like(dlogits)
    _lib.check(_lib.lib().svdd_dps_probs_bwd(logits.data_ptr(), x.data_ptr(), g.data_ptr(), B, L, dlogits.data_ptr(), direct.data_ptr(),
                                             _stream()), "svdd_dps_probs_bwd")
    return dlogits, direct


def dps_guided_q(logits, x, grad_backbone, grad_direct, dm, mcs, scale):
    """The guided transition weights of a DPS step (reference :1306-1314): exp(log p) * dm with q[MASK] = mcs, times
    exp(scale * (x_grad - x_grad[MASK])), x_grad = grad_backbone + grad_direct -> fp32 [B, L, 5]."""
    x = _need(x, torch.uint8, "x").contiguous()
    B, L = x.shape
    gb, gd = grad_backbone.contiguous().float(), grad_direct.contiguous().float()
    assert logits.is_contiguous() and logits.shape == (B, L, 5) and gb.shape == (B, L, 5) and gd.shape == (B, L, 5)
    q = torch.empty((B, L, 5), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().svdd_dps_guided_q(logits.data_ptr(), x.data_ptr(), gb.data_ptr(), gd.data_ptr(), float(dm), float(mcs),
                                            float(scale), B, L, q.data_ptr(), _stream()), "svdd_dps_guided_q")
    return q


def tds_resample(reward_num, reward_den, alpha, sample, u):
    """-> (x_next u8 [B,L], idx i32 [B]); u = the B float64 uniforms np.random.choice consumes."""
    num = _need(reward_num, torch.float32, "reward_num").contiguous()
    den = _need(reward_den, torch.float32, "reward_den").contiguous()
    sample = _need(sample, torch.uint8, "sample").contiguous()
    u = _need(u, torch.float64, "u").contiguous()
    B, L = sample.shape
    assert num.numel() == B and den.numel() == B and u.numel() == B
    x_next = torch.empty_like(sample)
    idx = torch.empty((B,), dtype=torch.int32, device=sample.device)
    work = torch.empty((2 * B,), dtype=torch.float64, device=sample.device)
    rc = _lib.lib().svdd_tds_resample(num.data_ptr(), den.data_ptr(), float(alpha), sample.data_ptr(), u.data_ptr(),
                                      B, L, x_next.data_ptr(), idx.data_ptr(), work.data_ptr(), _stream())
    _lib.check(rc, "svdd_tds_resample")
    return x_next, idx


# ------------------------------------------------------------------------------------------------ replay RNG on the device
_TORCH_STATE_BYTES = 5056       # THGeneratorState: u64 seed | i32 left | i32 seeded | u64 next | u64 state[624] | 3 doubles + i32 (Box-Muller cache)


def mt_state_from_torch(state_u8):
    """torch.get_rng_state() (CPU generator, the legacy 5056-byte layout) -> int64 numpy [625]: 624 words + pos.
    at::mt19937 draws with `if (--left == 0) next_state(); y = state[next++]`: pos = 625 - left (manual_seed leaves left = 1:
    pos = 624, twist first)."""
    import numpy as np
    a = state_u8.numpy() if isinstance(state_u8, torch.Tensor) else np.asarray(state_u8)
    if a.size != _TORCH_STATE_BYTES or not int(a[12:16].view(np.int32)[0]):
        raise SvddError("unexpected torch CPU generator state layout (want the 5056-byte mt19937 state, seeded)")
    left = int(a[8:12].view(np.int32)[0])
    words = a[24:24 + 624 * 8].view(np.uint64)
    if not (1 <= left <= 624) or int(words.max()) >> 32:
        raise SvddError("unexpected torch CPU generator state contents")
    out = np.empty(625, dtype=np.int64)
    out[:624] = words
    out[624] = 625 - left
    return out


def mt_state_to_torch(words_pos, template_u8):
    """Inverse: [625] (624 words + pos) -> a 5056-byte torch CPU generator state (seed and the Box-Muller cache of
    `template_u8` kept)."""
    import numpy as np
    a = np.array(template_u8.numpy() if isinstance(template_u8, torch.Tensor) else template_u8, dtype=np.uint8, copy=True)
    pos = int(words_pos[624])
    if not 1 <= pos <= 624:      # (the kernel returns pos in [1, 624] whenever it drew anything: left = 625 - pos in [1, 624])
        raise SvddError(f"mt19937 state with pos = {pos} has no torch representation")
    a[8:12].view(np.int32)[0] = 625 - pos
    a[16:24].view(np.uint64)[0] = pos
    a[24:24 + 624 * 8].view(np.uint64)[:] = np.asarray(words_pos[:624], dtype=np.uint64)
    return torch.from_numpy(a)





# ---- side streams, process-wide -----------------------------------------------------------------------------------------------
# Every component that runs parts of a step concurrently (the C4 trunk's two chains, the late steps' second GRU part, the SVDD-PM
# two-part step, the device replay generator) takes its side streams from ONE list per device, for the life of the process.
# Why a list and a test (round 6, tools/c4_stream_probe.py, profiles/r06_c4_stream_probe.txt): HIP maps streams onto FOUR hardware
# queues, in the order in which they are first used, and two streams on one hardware queue do not run concurrently — nor does a side
# stream that shares the queue of the stream the decode itself runs on. Which queue a stream got used to depend on how many streams
# anything in the process had touched before: the C4 trunk's two chains measured 65 instead of 80 seq/s whenever 4 k other streams
# had been used first, and bench.py's C4 leg lost 20 % in round 6 when an earlier leg stopped creating a stream per decode.
# side_stream() therefore PROBES its candidates once: a candidate is kept only if a marker on it is not held up by a sleeping kernel
# on the current stream, nor by one on a candidate already kept (same hardware queue = in-order = the marker waits). ~10 ms, once.
_SIDE_STREAMS = {}
SIDE_SLOTS = 3                                        # three other hardware queues exist beside the main stream's
SIDE_TRUNK_A, SIDE_TRUNK_B, SIDE_REPLAY = 0, 1, 2     # who uses which slot (the split GRU / PM paths share slot 0)


def _held_up_by(busy, cand, ms=0.4):
    """True when a marker on stream `cand` has to wait for a sleeping kernel on stream `busy` (they share a hardware queue)."""
    e0, e1, ec = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    e0.record(busy)
    with torch.cuda.stream(busy):
        torch.cuda._sleep(int(ms * 2.0e6))            # ~ms at 2 GHz
    e1.record(busy)
    ec.record(cand)
    torch.cuda.synchronize()
    return e0.elapsed_time(ec) > 0.5 * e0.elapsed_time(e1)


def _probe_side_streams(dev, want):
    import os
    cur = torch.cuda.current_stream(dev)
    kept, tried = [], 0
    probe = os.environ.get("SVDD_SIDE_STREAM_PROBE", "1") != "0" and not torch.cuda.is_current_stream_capturing()
    while len(kept) < want and tried < 12:
        st = torch.cuda.Stream(device=dev)
        tried += 1
        with torch.cuda.stream(st):
            torch.zeros(1, device=dev)                # first use: this is when HIP binds the stream to a hardware queue
        try:
            if probe and (_held_up_by(cur, st) or any(_held_up_by(k, st) for k in kept)):
                continue
        except Exception:                             # noqa: BLE001  (no _sleep in this build, ...): take the streams as they come
            probe = False
        kept.append(st)
    while len(kept) < want:                           # fewer free hardware queues than slots: the remaining slots share
        kept.append(kept[len(kept) % max(len(kept), 1)] if kept else torch.cuda.Stream(device=dev))
    return kept


def side_stream(device, slot=0):
    """The `slot`-th side stream of `device` (slots 0 .. SIDE_SLOTS - 1; made and probed at the first call, never destroyed)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = _probe_side_streams(torch.device("cuda", key), SIDE_SLOTS)
    return _SIDE_STREAMS[key][slot % SIDE_SLOTS]


class DeviceReplayStream:
    """torch's global CPU generator continued ON THE DEVICE for the span of one decode (rng_mode = "replay"): the state
    (2.5 KB) is uploaded at open(), `uniforms(n)` returns the next n floats of the stream as a device tensor (K8
    `svdd_mt19937_uniform_f32`: one workgroup, launched on a side stream ONE CALL AHEAD so that it runs under the nets of the
    current diffusion step), and close() writes the advanced state back into torch's generator — after it, torch.rand() on
    the host continues exactly where the reference's run would. Token-exact with the host replay (tests/test_kernels_gpu.py).
    Contract: between open and close() nothing else may draw from torch's CPU generator (the reference's eval-mode hot path does
    not either: rand_like in _sample_categorical is its only consumer); a value function that did (dropout in
    train mode) would see the pre-decode state — use Diffusion.replay_rng = "host" for such a net."""

    def __init__(self, device):
        self.dev = torch.device(device)
        self.template = torch.get_rng_state()
        import numpy as np
        st = mt_state_from_torch(self.template)
        self.state = torch.from_numpy(st.astype(np.uint32).view(np.int32).copy()).to(self.dev)          # [625] u32 bits
        key = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        self.side = side_stream(self.dev, SIDE_REPLAY)      # one side stream per device for the life of the process
        self.side.wait_stream(torch.cuda.current_stream(self.dev))
        self.bufs = {}            # n -> [two device buffers]
        self.flip = 0
        self.ahead = None         # (n, tensor, state snapshot before it was drawn, event) of the block generated ahead of need
        self.drawn = 0

    def _launch(self, n, out):
        rc = _lib.lib().svdd_mt19937_uniform_f32(self.state.data_ptr(), out.data_ptr(), n,
                                                 ctypes.c_void_p(self.side.cuda_stream))
        _lib.check(rc, "svdd_mt19937_uniform_f32")

    def _generate(self, n, snapshot):
        """Enqueue the next n floats on the side stream -> (tensor, snapshot | None, event)."""
        pair = self.bufs.setdefault(n, [torch.empty(n, dtype=torch.float32, device=self.dev) for _ in range(2)])
        out = pair[self.flip]
        self.flip ^= 1
        with torch.cuda.stream(self.side):
            snap = self.state.clone() if snapshot else None
            self._launch(n, out)
            ev = torch.cuda.Event()
            ev.record(self.side)
        return out, snap, ev

    def uniforms(self, n, prefetch=True):
        """The next n floats of the stream, usable on the CURRENT stream. prefetch: also start the n after them (a decode asks
        for the same n at every step); an unused prefetch is rolled back at close()."""
        cur = torch.cuda.current_stream(self.dev)
        if self.ahead is not None and self.ahead[0] == n:
            _, out, _, ev = self.ahead
            self.ahead = None
        else:
            self._rollback()
            self.side.wait_stream(cur)          # the buffer about to be overwritten may still be read by an earlier K1
            out, _, ev = self._generate(n, False)
        cur.wait_event(ev)
        self.drawn += n
        if prefetch:
            # the prefetch overwrites the OTHER buffer of the pair, last read by the K1 of the previous call: order after it
            self.side.wait_stream(cur)
            o2, snap, e2 = self._generate(n, True)
            self.ahead = (n, o2, snap, e2)
        return out

    def _rollback(self):
        if self.ahead is not None:
            _, _, snap, _ = self.ahead
            with torch.cuda.stream(self.side):
                self.state.copy_(snap)
            self.ahead = None

    def close(self):
        """Write the advanced state back into torch's global CPU generator (one small D2H copy + sync)."""
        self._rollback()
        self.side.synchronize()
        if self.drawn:
            import numpy as np
            st = self.state.cpu().numpy().view(np.uint32).astype(np.int64)
            torch.set_rng_state(mt_state_to_torch(st, self.template))
        torch.cuda.current_stream(self.dev).wait_stream(self.side)

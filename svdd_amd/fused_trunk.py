"""The Enformer-shaped value trunk of BASELINE.json configs[3] on hand-written kernels (csrc/svdd_trunk.hip).

`FusedEnformerValueNet(trunk, head)` computes the same function as `head(trunk(onehot))` for an
`enformer_value.EnformerTrunk` + `value_nets.ConvHead` pair (reference decode.py:78-80; layer structure
Enformer.py:1271-1334, :1807-1884 conv tower, :1887-2007 transformer tower, :2176-2292 ConvBlock "NACDR"), with every
matrix product on the 16-bit matrix cores in split precision:

    precision "bf16x3"  operands split hi + lo in bf16 (a 16-bit operand: 1e-5-class error, NOT fp32-class), 3 MFMAs per product, fp32 accumulate;
              "bf16"    one pass on hi.

Layout and kernels (see the header of svdd_trunk.hip): channels-last rows with two zero rows behind every sequence (and two
guard rows in front of the first), so a k = 5 convolution is five row-shifted GEMMs accumulated in one launch; BatchNorm (eval) + GELU + the hi / lo
split are one element-wise pass that writes the next GEMM's operand planes; the attention pooling is a 1x1 GEMM for
the logits plus one pair-softmax pass. The transformer tower works on the 2 tokens a 200-long sequence is pooled down to:
its projections and FFNs are the same GEMM kernel (q, k, v fused into one launch), its 2 x 2 attention is a handful of
tiny tensor ops. Every kernel takes the number of live sequences as a device scalar (`count`), so the exact work-skipping
of the SVDD-MC loop needs no host round trip.

The module takes TOKENS ([n, L] uint8, 4 = MASK): the engine's one-hot rows are exact, and the stem's k = 15 convolution
over a one-hot input is a K = 60 GEMM whose A operand is exactly representable in bf16."""
import ctypes
import threading
import os

import torch
from torch import nn

from . import _lib
from .enformer_value import EnformerTrunk, _positional_features, _relative_shift
from .value_nets import ConvHead

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
GUARD = 4            # operand-plane rows in front of row 0 (a k = 5 tap reads rows -2 .. of the first tile)
TAIL = 136           # ... and behind the last row (the last 128-row tile + 2 tap rows)
# (side streams: ops.side_stream — process-wide and probed; a second instance with streams of its own once ran 20 % slower than one
#  chain, and round 6 found why: HIP has four hardware queues and streams that share one do not overlap)
WIN_K = 4            # window slots per candidate and level of the shared levels (svdd_trunk.hip)


def _level_len(L, d):
    """Sequence length at conv-tower level d (every level halves it, rounding up)."""
    for _ in range(d):
        L = (L + 1) // 2
    return L


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack_gemm_weight(w, parts):
    """Conv1d / Linear weight [N, Cin, T] (or [N, Cin]) fp32 -> bf16 fragments for svdd_trunk_gemm:
    [KB = T * Cin/32][N/128][8 n-tiles][parts][64 lanes = 16 g + j][8 e] = part(W[128 nb + 16 nt + j][32 c + 8 g + e][t]),
    kb = c * T + t (chunk major, tap minor) ; part 0 = bf16(W), part 1 = bf16(W - part 0)."""
    if w.dim() == 2:
        w = w[:, :, None]
    N, Cin, T = w.shape
    assert N % 128 == 0 and Cin % 32 == 0, (N, Cin)
    w = w.detach().float()
    hi = w.to(torch.bfloat16)
    ps = [hi] if parts == 1 else [hi, (w - hi.float()).to(torch.bfloat16)]
    v = torch.stack(ps, dim=0).reshape(parts, N // 128, 8, 16, Cin // 32, 4, 8, T)     # [p][nb][nt][j][c][g][e][t]
    return v.permute(4, 7, 1, 2, 0, 5, 3, 6).contiguous().reshape(-1)                  # [c][t][nb][nt][p][g][j][e]


def pack_gemm_weight_f32(w, parts=1):
    """The fp32 twin (precision "f32": one fp32 operand plane, v_mfma_f32_16x16x4_f32): Conv1d / Linear weight [N, Cin, T] ->
    [KB = T * Cin/32][N/128][8 n-tiles][2 pieces][64 lanes = 16 g + j][4 e] = W[128 nb + 16 nt + j][32 c + 8 g + 4 p + e][t] —
    the 16-byte pieces svdd_trunk_gemm's LDS-DMA copies straight into fragment order (a K block's 32 channels are dealt
    8 to a lane group: MFMA step (p, e) multiplies channel 8 g + 4 p + e)."""
    if w.dim() == 2:
        w = w[:, :, None]
    N, Cin, T = w.shape
    assert N % 128 == 0 and Cin % 32 == 0, (N, Cin)
    v = w.detach().float().reshape(N // 128, 8, 16, Cin // 32, 4, 2, 4, T)             # [nb][nt][j][c][g][p][e][t]
    return v.permute(3, 7, 0, 1, 5, 4, 2, 6).contiguous().reshape(-1)                  # [c][t][nb][nt][p][g][j][e]


class _Planes:
    """(hi, lo) bf16 operand planes (or the one fp32 plane of precision "f32") inside one allocation, with guard rows so that
    shifted / overrunning tile reads stay inside it."""

    def __init__(self, max_elems, parts, dev, dtype=torch.bfloat16):
        self.buf = [torch.zeros(max_elems, dtype=dtype, device=dev) for _ in range(parts)]

    FRONT = GUARD * 4096      # elements in front of row 0, the same for every channel count: the two rows a k = 5 tap reads in
                              # front of the first sequence must be zero, and with a C-dependent offset they would alias the
                              # data rows of a narrower view of the same buffer

    def view(self, rows, C):
        need = self.FRONT + (rows + TAIL) * C
        assert C <= 4096 and need <= self.buf[0].numel(), (rows, C, self.buf[0].numel())
        return [b[self.FRONT:] for b in self.buf]


class _PlanesAt:
    """A _Planes seen from `base` elements further on (the second half batch's region: the rows in front of its row 0 are
    zeroed by the caller, the first half never writes them)."""

    def __init__(self, planes, base):
        self.buf = [b[base:] for b in planes.buf]

    def view(self, rows, C):
        assert _Planes.FRONT + (rows + TAIL) * C <= self.buf[0].numel(), (rows, C, self.buf[0].numel())
        return [b[_Planes.FRONT:] for b in self.buf]


_OPTION_LOCK = threading.RLock()     # the plane format / concurrency hint are process-wide switches of the library: one forward at a time sets them


class FusedEnformerValueNet(nn.Module):
    @staticmethod
    def supports(trunk, head):
        """(ok, why): whether svdd_trunk_gemm takes every GEMM of this trunk — output channels in 128s, input channels in 32s
        (a 384-channel toy trunk has a 192-channel stem) — and the block structure is the one the kernels were written for.
        Diffusion.value_callable falls back to the PyTorch modules ONLY on a False here; any assertion raised while packing a
        supported trunk is a bug and propagates."""
        try:
            blocks = trunk.conv_tower.blocks
            stem = blocks[0][0]
            if tuple(stem.weight.shape[1:]) != (4, 15):
                return False, f"stem weight {tuple(stem.weight.shape)} is not [*, 4, 15]"
            shapes = [("stem", stem.weight.shape[0], 64)]
            for i, blk in enumerate(blocks):
                if i > 0:
                    a = blk[0]
                    if a.conv.kernel_size[0] != 5 or a.residual or not isinstance(a.pool, nn.Identity):
                        return False, f"conv block {i}a is not a plain 5-tap block"
                    shapes.append((f"conv{i}a", a.conv.out_channels, a.conv.in_channels))
                b = blk[1]
                if b.conv.kernel_size[0] != 1 or not b.residual:
                    return False, f"conv block {i}b is not a residual 1x1 block"
                shapes.append((f"conv{i}b", b.conv.out_channels, b.conv.in_channels))
                shapes.append((f"pool{i}", *b.pool.to_attn_logits.weight.shape[:2]))
            for k, tb in enumerate(trunk.transformer_tower):
                m = tb.mha
                shapes += [(f"qkv{k}", 2 * m.to_q.weight.shape[0] + m.to_v.weight.shape[0], m.to_q.weight.shape[1]),
                           (f"out{k}", *m.to_out.weight.shape), (f"ffn1_{k}", *tb.ffn1.weight.shape), (f"ffn2_{k}", *tb.ffn2.weight.shape)]
            pw = trunk.pointwise_conv
            if pw.conv.kernel_size[0] != 1 or pw.residual:
                return False, "pointwise block is not a plain 1x1 block"
            shapes.append(("pointwise", pw.conv.out_channels, pw.conv.in_channels))
            head.channel_transform.conv.layer.weight                # noqa: B018  (the ConvHead layout the tail reads)
        except AttributeError as e:
            return False, f"not the EnformerTrunk / ConvHead layout ({e})"
        for name, N, Cin in shapes:
            if N % 128 or Cin % 32:
                return False, f"GEMM {name}: {N} output / {Cin} input channels (need multiples of 128 / 32)"
        return True, ""

    def __init__(self, trunk: EnformerTrunk, head: ConvHead, precision="bf16x3"):
        super().__init__()
        assert precision in ("bf16x3", "bf16", "f32")
        self.precision = precision
        self.parts = 2 if precision == "bf16x3" else 1
        # "f32" (round 4): ONE fp32 operand plane and fp32 MFMAs — the trunk at the reference's precision (decode.py:78-80 runs it
        # in fp32). Same kernels (svdd_set_option(SVDD_OPT_TRUNK_PLANES_F32) selects their fp32-plane instantiations), same
        # layout, same exact work-skipping; element counts are the bf16 modes', bytes those of bf16x3.
        self.f32 = precision == "f32"
        self.plane_dtype = torch.float32 if self.f32 else torch.bfloat16
        self._ws = {}
        dev = next(trunk.parameters()).device
        P = self.parts
        pk = (lambda w: pack_gemm_weight_f32(w).to(dev)) if self.f32 else (lambda w: pack_gemm_weight(w, P).to(dev))   # noqa: E731

        def bn_affine(bn):
            s = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).float()
            return s.contiguous(), (bn.bias.detach() - bn.running_mean * s).float().contiguous()

        with torch.no_grad():
            blocks = trunk.conv_tower.blocks
            stem = blocks[0][0]
            assert tuple(stem.weight.shape[1:]) == (4, 15)
            half = stem.weight.shape[0]
            wk = torch.zeros(half, 64, device=dev)
            wk[:, :60] = stem.weight.detach().float().permute(0, 2, 1).reshape(half, 60)      # K index 4 t + ci
            self.stem_w, self.stem_b = pk(wk), stem.bias.detach().float().contiguous()
            self.levels = []
            for i, blk in enumerate(blocks):
                lv = {}
                if i > 0:
                    a = blk[0]
                    assert a.conv.kernel_size[0] == 5 and not a.residual and isinstance(a.pool, nn.Identity)
                    lv["a_bn"], lv["a_w"], lv["a_b"] = bn_affine(a.norm), pk(a.conv.weight), a.conv.bias.detach().float().contiguous()
                    lv["a_cin"], lv["a_cout"] = a.conv.in_channels, a.conv.out_channels
                b = blk[1]
                assert b.conv.kernel_size[0] == 1 and b.residual
                lv["b_bn"], lv["b_w"], lv["b_b"] = bn_affine(b.norm), pk(b.conv.weight), b.conv.bias.detach().float().contiguous()
                lv["C"] = b.conv.out_channels
                lv["pool_w"] = pk(b.pool.to_attn_logits.weight.detach()[:, :, 0, 0])
                self.levels.append(lv)
            self.C = trunk.pointwise_conv.conv.in_channels
            self.tf = []
            for tb in trunk.transformer_tower:
                m = tb.mha
                d = {"ln1": (tb.norm.weight.detach().float().contiguous(), tb.norm.bias.detach().float().contiguous(), tb.norm.eps),
                     "qkv_w": pk(torch.cat([m.to_q.weight, m.to_k.weight, m.to_v.weight], dim=0)),
                     "nq": m.to_q.weight.shape[0], "nv": m.to_v.weight.shape[0],
                     "out_w": pk(m.to_out.weight), "out_b": m.to_out.bias.detach().float().contiguous(),
                     "heads": m.heads, "dk": m.dim_key, "dv": m.dim_value,
                     "content_bias": m.rel_content_bias.detach().float().contiguous(), "pos_bias": m.rel_pos_bias.detach().float().contiguous(),
                     "rel_w": m.to_rel_k.weight.detach().float(), "nfeat": m.num_rel_pos_features,
                     "ln2": (tb.ffn_norm.weight.detach().float().contiguous(), tb.ffn_norm.bias.detach().float().contiguous(), tb.ffn_norm.eps),
                     "f1_w": pk(tb.ffn1.weight), "f1_b": tb.ffn1.bias.detach().float().contiguous(),
                     "f2_w": pk(tb.ffn2.weight), "f2_b": tb.ffn2.bias.detach().float().contiguous()}
                assert (2 * d["nq"] + d["nv"]) % 128 == 0
                self.tf.append(d)
            pw = trunk.pointwise_conv
            assert pw.conv.kernel_size[0] == 1 and not pw.residual
            self.pw_bn, self.pw_w, self.pw_b = bn_affine(pw.norm), pk(pw.conv.weight), pw.conv.bias.detach().float().contiguous()
            self.pw_out = pw.conv.out_channels
            hw = head.channel_transform.conv.layer
            self.head_w = hw.weight.detach().float()[:, :, 0].t().contiguous()        # [3072, n_tasks]
            self.head_b = hw.bias.detach().float()
        self._relk = {}
        self.timing = None
        self.share_level0 = True        # forward_tokens(shared=...): the first levels on the changed windows only (exact)
        self.share_levels = 4           # ... how many of them (all but the last must have an even length: 200, 100, 50, 25 at L = 200)
        self.share_slots = WIN_K        # ... windows per candidate and level (1: one window around every changed position)
        self.share_parent_steps = os.environ.get("SVDD_TRUNK_PARENT_STEPS", "1") != "0"   # ... and the parents' own levels from the previous call's, the same way (env: A/B runs)
        self.last_parent_rows = None
        self.gemm_conc_hint = False     # tell svdd_trunk_gemm how many chains share the chip (A/B knob; measured: pricing a launch
                                        # against the WHOLE chip picks better tile heights even with two chains — tools/trunk_tile_ab.py)
        self.tower_streams = int(os.environ.get("SVDD_TRUNK_TOWER_STREAMS", "2"))   # the candidates as this many parts on as many streams (1: one chain of kernels; at most 4)
        self._side = None
        self.last_window_rows = None

    # ------------------------------------------------------------------ thin kernel wrappers
    def _gemm(self, planes, w, bias, resid, out, M, N, Cin, T, act, count, rps, nxt=None, post=None, post_act=ACT_NONE, pad=0):
        """out (fp32, may be None) = act(A W + bias) (+ resid); nxt: the operand planes the NEXT GEMM reads, written by this
        GEMM's epilogue as post_act(post[0] y + post[1]) with the `pad` rows of every sequence zeroed."""
        if self.timing is not None:                                # tools/trunk_microbench.py --gemms: per-launch events
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._gemm_launch(planes, w, bias, resid, out, M, N, Cin, T, act, count, rps, nxt, post, post_act, pad)
            e1.record()
            self.timing.append((M, N, Cin, T, e0, e1))
            return
        self._gemm_launch(planes, w, bias, resid, out, M, N, Cin, T, act, count, rps, nxt, post, post_act, pad)

    def _gemm_launch(self, planes, w, bias, resid, out, M, N, Cin, T, act, count, rps, nxt, post, post_act, pad):
        rc = _lib.lib().svdd_trunk_gemm(planes[0].data_ptr(), planes[1].data_ptr() if len(planes) > 1 else None, w.data_ptr(),
                                        _ptr(bias), _ptr(resid), _ptr(out), M, N, Cin, T, Cin, N, act, _ptr(count), rps,
                                        nxt[0].data_ptr() if nxt else None, nxt[1].data_ptr() if nxt and len(nxt) > 1 else None,
                                        _ptr(post[0]) if post else None, _ptr(post[1]) if post else None, post_act, pad, _stream())
        _lib.check(rc, "svdd_trunk_gemm")

    def _act(self, x, bn, act, rows, C, rps, pad, planes, count):
        rc = _lib.lib().svdd_trunk_act_split(x.data_ptr(), _ptr(bn[0]) if bn else None, _ptr(bn[1]) if bn else None, act, rows, C, rps,
                                             pad, planes[0].data_ptr(), planes[1].data_ptr() if len(planes) > 1 else None,
                                             _ptr(count), _stream())
        _lib.check(rc, "svdd_trunk_act_split")

    def _ln(self, x, ln, rows, C, planes, count, rps):
        rc = _lib.lib().svdd_trunk_layernorm_split(x.data_ptr(), ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2]), rows, C,
                                                   planes[0].data_ptr(), planes[1].data_ptr() if len(planes) > 1 else None,
                                                   _ptr(count), rps, _stream())
        _lib.check(rc, "svdd_trunk_layernorm_split")

    def _workspace(self, n, L, dev):
        """ONE workspace per (L, device), sized for the largest batch seen so far and reused for smaller ones (an SVDD-MC decode
        scores n = B parents once and then n = B * M candidates at every step: keyed by n, the GB-sized buffers and the parents'
        carried state were freed and reallocated at every switch, with both copies alive during the swap). A larger batch drops
        the old buffers BEFORE allocating the new ones."""
        key = (L, str(dev))
        ws = self._ws.get(key)
        if ws is None or ws["cap"] < n:
            ws = None
            self._ws = {}                                          # free first: never two GB-sized workspaces at once
            fmax, pmax = self._workspace_sizes(n, L)
            for S in (2, 3, 4):                                   # room for S parts of the batch side by side (forward_tokens)
                fh, ph = self._workspace_sizes((n + S - 1) // S, L)
                fmax, pmax = max(fmax, S * (fh + 4096)), max(pmax, S * (ph + 4 * 4096))
            # two sets of operand planes: a GEMM reads one and writes the next GEMM's operands into the other
            pmax += _Planes.FRONT
            ws = {"f": [torch.empty(fmax, dtype=torch.float32, device=dev) for _ in range(4)],
                  "p": [_Planes(pmax, self.parts, dev, self.plane_dtype), _Planes(pmax, self.parts, dev, self.plane_dtype)], "cap": n}
            self._ws = {key: ws}
        return ws

    def _workspace_sizes(self, n, L):
        """(elements of an fp32 buffer, elements of an operand plane behind its front guard) that n sequences of length L need."""
        rows0 = n * (L + 2)
        Tf = _level_len(L, len(self.levels))                      # tokens left for the transformer tower
        # the largest activation is at the first level (rows halve per level, channels at most double overall)
        fmax = max(rows0 * self.levels[0]["C"], max((n * ((L >> i) + 4 * WIN_K + 1)) * lv["C"] for i, lv in enumerate(self.levels)),
                   n * Tf * max(self.pw_out, 2 * self.C, 2 * self.tf[0]["nq"] + self.tf[0]["nv"] if self.tf else 0)) + 1024
        pmax = max((GUARD + rows0 + TAIL) * max(64, self.levels[0]["C"]),
                   max((GUARD + n * ((L >> max(i - 1, 0)) + 4 * WIN_K + 1) + TAIL) * lv["C"] for i, lv in enumerate(self.levels)),
                   (GUARD + Tf * n + TAIL) * max(self.pw_out, 2 * self.C))
        return fmax, pmax

    def _rel_k(self, d, length, dev):
        key = (id(d), length)
        r = self._relk.get(key)
        if r is None:
            pos = _positional_features(length, d["nfeat"], dev)
            r = (pos @ d["rel_w"].t()).view(2 * length - 1, d["heads"], d["dk"]).transpose(0, 1).contiguous()   # [h, 2L-1, dk]
            self._relk[key] = r
        return r

    # ------------------------------------------------------------------ forward
    def _convs(self, ws, i, src, rows, rps, pad, cnt):
        """The three GEMMs of conv-tower level i on the operand planes `src` ([rows, C_in]: plane set 0 or a buffer of its own;
        level 0: the unfolded stem operand): block output -> f[2], pooling logits -> f[3]."""
        f, P = ws["f"], self.parts
        lv = self.levels[i]
        C = lv["C"]
        planes = lambda which: ws["p"][which].view(rows, C)[:P]                    # noqa: E731
        w, b, cin, T = (self.stem_w, self.stem_b, 64, 1) if i == 0 else (lv["a_w"], lv["a_b"], lv["a_cin"], 5)
        self._gemm(src, w, b, None, f[1], rows, C, cin, T, ACT_NONE, cnt, rps, nxt=planes(1), post=lv["b_bn"], post_act=ACT_GELU, pad=pad)
        self._gemm(planes(1), lv["b_w"], lv["b_b"], f[1], f[2], rows, C, C, 1, ACT_NONE, cnt, rps, nxt=planes(0), post=None,
                   post_act=ACT_NONE, pad=pad)
        self._gemm(planes(0), lv["pool_w"], None, None, f[3], rows, C, C, 1, ACT_NONE, cnt, rps)

    def _unfold(self, ws, tok, rows, count, win=None):
        n, L = tok.shape
        ph = ws["p"][0].view(rows, 64)[:self.parts]
        lib = _lib.lib()
        if win is None:
            _lib.check(lib.svdd_trunk_stem_unfold(tok.data_ptr(), n, L, ph[0].data_ptr(), _ptr(count), _stream()), "svdd_trunk_stem_unfold")
        else:
            _lib.check(lib.svdd_trunk_stem_unfold_win(tok.data_ptr(), n, L, self.share_slots, win[0].data_ptr(), win[1].data_ptr(), win[2].data_ptr(),
                                                      ph[0].data_ptr(), _ptr(count), _stream()), "svdd_trunk_stem_unfold_win")
        if self.parts == 2:
            ph[1][: rows * 64].zero_()                           # the one-hot operand is exact: its lo plane is zero
        return ph

    def _parent_levels(self, ws, parent_tok, pp, depth):
        """Whole sequences: the pooled operand planes of the first `depth` levels of `parent_tok` -> pp[1 .. depth]."""
        B, L = parent_tok.shape
        P, f, lib = self.parts, ws["f"], _lib.lib()
        Lc = L
        for d in range(depth):
            rps = Lc + 2
            rows = B * rps
            src = self._unfold(ws, parent_tok, rows, None) if d == 0 else pp[d].view(rows, self.levels[d]["a_cin"])[:P]
            self._convs(ws, d, src, rows, rps, 2, None)
            nx = self.levels[d + 1]
            tg = pp[d + 1].view(B * ((Lc + 1) // 2 + 2), nx["a_cin"])[:P]
            rc = lib.svdd_trunk_attn_pool(f[2].data_ptr(), f[3].data_ptr(), B, Lc, self.levels[d]["C"], None, None, tg[0].data_ptr(),
                                          tg[1].data_ptr() if P == 2 else None, _ptr(nx["a_bn"][0]), _ptr(nx["a_bn"][1]), ACT_GELU, _stream())
            _lib.check(rc, "svdd_trunk_attn_pool")
            Lc = (Lc + 1) // 2

    def _window_levels(self, ws, tok, count, parent_tok, pidx, div, pp, depth, win, whole):
        """The first `depth` levels of `tok` on the windows of rows where a row differs from row pidx[c] // div of parent_tok,
        whose pooled planes are pp[1 .. depth]. whole[d] (or None): where the whole-sequence operand planes of level d + 1 go
        (pooled window rows + the parent's rows elsewhere); the last level's must be given."""
        n, L = tok.shape
        B = parent_tok.shape[0]
        K, P, f, lib = self.share_slots, self.parts, ws["f"], _lib.lib()
        w0, wlen, seg = win
        rc = lib.svdd_trunk_windows(tok.data_ptr(), parent_tok.data_ptr(), pidx.data_ptr(), div, n, L, 7, depth, K, _ptr(count),
                                    w0.data_ptr(), wlen.data_ptr(), seg.data_ptr(), _stream())
        _lib.check(rc, "svdd_trunk_windows")
        cs = torch.cumsum(seg, 1, dtype=torch.int32)
        off = cs - seg
        Lc = L
        for d in range(depth):
            rows = n * (Lc + (4 * K if d else 0))                 # the bound the grids are sized for; live rows: cs[d, n K - 1]
            lv, nx = self.levels[d], self.levels[d + 1]
            src = self._unfold(ws, tok, rows, count, (w0[0], wlen[0], off[0])) if d == 0 else ws["p"][0].view(rows, lv["a_cin"])[:P]
            self._convs(ws, d, src, rows, 1, 0, cs[d, n * K - 1:])
            last = d + 1 == depth
            Lo = (Lc + 1) // 2
            ppl = pp[d + 1].view(B * (Lo + 2), nx["a_cin"])[:P]
            outs = []
            if whole[d] is not None:
                outs.append((whole[d], (None, None, None)))
            if not last:                                          # the compact segments of the next level (plane set 0: its GEMMs are done with it)
                outs.append((ws["p"][0].view(n * (Lo + 4 * K), nx["a_cin"])[:P],
                             (w0[d + 1].data_ptr(), wlen[d + 1].data_ptr(), off[d + 1].data_ptr())))
            for tg, nxt_win in outs:
                rc = lib.svdd_trunk_attn_pool_win(f[2].data_ptr(), f[3].data_ptr(), n, Lc, lv["C"], 2 if d else 0, K, w0[d].data_ptr(),
                                                  wlen[d].data_ptr(), off[d].data_ptr(), pidx.data_ptr(), div, ppl[0].data_ptr(),
                                                  ppl[1].data_ptr() if P == 2 else None, _ptr(count), tg[0].data_ptr(),
                                                  tg[1].data_ptr() if P == 2 else None, _ptr(nx["a_bn"][0]), _ptr(nx["a_bn"][1]), ACT_GELU,
                                                  *nxt_win, _stream())
                _lib.check(rc, "svdd_trunk_attn_pool_win")
            Lc = Lo
        return cs[:, n * K - 1]

    def _parents(self, ws, shared, n, depth):
        """The parents' first `depth` levels of this call (svdd_trunk.hip, "first level shared"), as pooled operand planes:
        updated from the parents of the previous call on the windows that changed (x_t differs from x_{t-1} at the positions
        the last step unmasked) — whole sequences when there is no previous call of this shape. -> the state dict."""
        parent_tok, pidx, div = shared
        B, L = parent_tok.shape
        assert parent_tok.dtype == torch.uint8 and parent_tok.is_contiguous()
        assert pidx.dtype == torch.int32 and pidx.is_contiguous() and pidx.numel() >= n and div >= 1
        dev, P, K = parent_tok.device, self.parts, self.share_slots
        key = ("pp", B, L, depth, K)
        st = ws.get(key)
        if st is None:
            mk = lambda: [None] + [_Planes(_Planes.FRONT + (B * (_level_len(L, d) + 2) + TAIL) * self.levels[d]["a_cin"], P, dev, self.plane_dtype)  # noqa: E731
                                   for d in range(1, depth + 1)]
            st = ws[key] = {"pp": [mk(), mk()], "cur": 0, "x": None, "ids": torch.arange(B, dtype=torch.int32, device=dev),
                            "win": {}, "pwin": torch.empty((3, depth, B * K), dtype=torch.int32, device=dev)}
        lens = [_level_len(L, d) for d in range(depth + 1)]
        if st["x"] is None or not self.share_parent_steps:
            self._parent_levels(ws, parent_tok, st["pp"][st["cur"]], depth)
            self.last_parent_rows = None
        else:                                                      # from the previous parents' planes, on the windows that changed
            old, new = st["pp"][st["cur"]], st["pp"][1 - st["cur"]]
            whole = [new[d + 1].view(B * (lens[d + 1] + 2), self.levels[d + 1]["a_cin"])[:P] for d in range(depth)]
            self.last_parent_rows = self._window_levels(ws, parent_tok, None, st["x"], st["ids"], 1, old, depth, st["pwin"], whole)
            st["cur"] = 1 - st["cur"]
        if st["x"] is None:
            st["x"] = torch.empty_like(parent_tok)
        st["x"].copy_(parent_tok)
        return st

    def _shared_levels(self, ws, st, tok, count, shared, depth, half):
        """The first `depth` levels of candidates `tok` that differ from their parents at a few positions: up to share_slots
        windows of rows per candidate and level, the parents' planes (st, _parents) elsewhere. Leaves the operand planes of
        level `depth` in plane set 0 — the same bits as the whole-sequence path."""
        parent_tok, pidx, div = shared
        n, L = tok.shape
        K, P = self.share_slots, self.parts
        win = st["win"].get((half, n))
        if win is None:
            win = st["win"][half, n] = torch.empty((3, depth, n * K), dtype=torch.int32, device=tok.device)
        whole = [None] * (depth - 1) + [ws["p"][0].view(n * (_level_len(L, depth) + 2), self.levels[depth]["a_cin"])[:P]]
        return self._window_levels(ws, tok, count, parent_tok, pidx, div, st["pp"][st["cur"]], depth, win, whole)

    def _share_depth(self, L):
        """Levels that can be shared: every level but the last shared one is pooled into compact segments and needs an even length."""
        if L > 256 or L % 2:
            return 0
        d = 1
        while d < min(self.share_levels, len(self.levels) - 1) and (L >> (d - 1)) % 2 == 0:
            d += 1
        return min(d, self.share_levels)

    @torch.no_grad()
    def forward_tokens(self, tok, count=None, shared=None):
        """See _forward_tokens. The plane format is a host-side switch of the library (svdd_set_option): set for the span of the
        call (every launch of the call is enqueued inside it), restored on the way out."""
        with _OPTION_LOCK:                       # (two value nets of different precision on two host threads must not interleave)
            prev_planes = _lib.set_option(6, 1 if self.f32 else 0)
            prev_conc = _lib.current_option(4, 51)
            try:
                return self._forward_tokens(tok, count, shared)
            finally:                             # what the caller had set, not constants
                _lib.set_option(6, prev_planes)
                _lib.set_option(4, prev_conc)

    def _forward_tokens(self, tok, count=None, shared=None):
        """tok [n, L] u8 -> scores [n, n_tasks, 1]; count: int32 device scalar = live rows (rows beyond it are undefined).
        shared = (parent_tok [B, L] u8, parent_idx int32 [n], div): row c of tok is a candidate of row parent_idx[c] // div of
        parent_tok and differs from it at a few positions; the first levels are then computed on those windows only (exact)."""
        assert tok.is_cuda and tok.dtype == torch.uint8 and tok.is_contiguous()
        n, L = tok.shape
        dev = tok.device
        ws = self._workspace(n, L, dev)
        depth = self._share_depth(L) if shared is not None and self.share_level0 and shared[0].shape[0] <= n else 0
        st = self._parents(ws, shared, n, depth) if depth else None
        T = _level_len(L, len(self.levels))
        zs = torch.empty((n * T, self.pw_out), dtype=torch.float32, device=dev)
        for d in self.tf:
            self._rel_k(d, T, dev)                                # cached tensors: made before the streams fork
        # Two halves of the candidates on two streams: rows are independent, and one dependent chain of kernels leaves the tail
        # of every GEMM round idle (the tower's 7680-row GEMMs at a C4 step are 180 to 384 tiles of 256 x 256 on 256 CUs: 0.7 to
        # 1.5 rounds) and nothing running under the pooling / LayerNorm / attention kernels; two chains fill each other's gaps.
        # Each half works in its own region of every workspace buffer. Same kernels per row: same bits.
        S = max(1, min(self.tower_streams, 4))
        nA = (n + S - 1) // S                                     # rows of the largest part
        reg_f = (ws["f"][0].numel() // S) & ~4095
        reg_p = ((ws["p"][0].buf[0].numel() - _Planes.FRONT) // S) & ~4095
        need_f, need_p = self._workspace_sizes(nA, L)
        if S > 1 and not (n * T >= 2048 and need_f <= reg_f and need_p + 2 * 4096 <= reg_p):
            S = 1
        self.last_streams = S
        _lib.set_option(4, 50 + (S if self.gemm_conc_hint else 1))   # the GEMMs' tile-height choice prices a launch against CUs / S
        if S == 1:
            self.last_window_rows = self._candidates(ws, st, tok, count, shared, depth, zs, 0)
        else:
            # rows k, k + S, k + 2 S, ...: the parts stay balanced whatever the (device-side) live count is
            cnts = [None if count is None else torch.div(count + (S - 1 - k), S, rounding_mode="floor").to(torch.int32) for k in range(S)]
            toks = [tok[k::S].contiguous() for k in range(S)]
            pids = [None if shared is None else shared[1][:n][k::S].contiguous() for k in range(S)]
            zoff = [0]
            for k in range(S):
                zoff.append(zoff[-1] + toks[k].shape[0] * T)
            wss = [ws]
            for k in range(1, S):
                w = dict(ws)
                w["f"] = [b[k * reg_f:] for b in ws["f"]]
                w["p"] = [_PlanesAt(pl, k * reg_p) for pl in ws["p"]]
                for pl in w["p"]:
                    for b in pl.buf:
                        b[_Planes.FRONT - 2 * 4096: _Planes.FRONT].zero_()  # the rows in front of this part's first sequence
                wss.append(w)
            main = torch.cuda.current_stream()
            from . import ops
            self._side = [ops.side_stream(dev, k) for k in range(S)]   # process-wide, chosen so that any two run concurrently (ops.side_stream)
            stats = []
            for k in range(S):
                sd = self._side[k]
                sd.wait_stream(main)
                with torch.cuda.stream(sd):
                    sh = None if shared is None else (shared[0], pids[k], shared[2])
                    stats.append(self._candidates(wss[k], st, toks[k], cnts[k], sh, depth, zs[zoff[k]: zoff[k + 1]], k))
            for k in range(S):
                main.wait_stream(self._side[k])
            self.last_window_rows = None if stats[0] is None else sum(stats[1:], stats[0])
            sc = self._head(zs, n, T)
            s = torch.empty_like(sc)
            for k in range(S):
                s[k::S] = sc[zoff[k] // T: zoff[k + 1] // T]
            return s[:, :, None]
        s = self._head(zs, n, T)
        return s[:, :, None]

    def _head(self, zs, n, T):
        """ConvHead: 1 x 1 convolution to n_tasks + mean over the T tokens, [n T, pw_out] -> [n, n_tasks]. A row-wise reduction, not
        a library GEMM: hipBLASLt's result for a row changed in the last bit with the row's position in the batch (the compacted /
        interleaved orders of the two-stream path gave 1e-7 differences against one chain), and a candidate's score must not
        depend on where the compaction put it."""
        z = zs.view(n, T, -1)
        cols = [(z * self.head_w[:, j]).sum(dim=(1, 2)) for j in range(self.head_w.shape[1])]
        return torch.stack(cols, dim=1) / T + self.head_b

    def _candidates(self, ws, st, tok, count, shared, depth, zs, half):
        """Conv tower (the first `depth` levels on windows), transformer tower and pointwise block of the rows `tok` -> zs."""
        n, L = tok.shape
        f, P = ws["f"], self.parts
        lib = _lib.lib()
        side = 0                                                  # the plane set the NEXT GEMM reads

        def planes(rows, C, which):
            return ws["p"][which].view(rows, C)[:P]

        # ---- conv tower. Every GEMM / pooling epilogue writes the operand planes of the GEMM that follows it
        # (BatchNorm + GELU + hi / lo split fused; round 3a ran a separate element-wise pass per GEMM: 9 ms of 73).
        Lc, rps = L, L + 2
        cur = 0                                                   # index of the fp32 buffer that holds x
        stats = None
        if depth:
            stats = self._shared_levels(ws, st, tok, count, shared, depth, half)
            Lc = _level_len(L, depth)
            rps = Lc + 2
        else:
            rows = n * rps
            ph = self._unfold(ws, tok, rows, count)
            lv0 = self.levels[0]
            C = lv0["C"]
            self._gemm(ph, self.stem_w, self.stem_b, None, f[cur], rows, C, 64, 1, ACT_NONE, count, rps,
                       nxt=planes(rows, C, 1 - side), post=lv0["b_bn"], post_act=ACT_GELU, pad=2)
            side = 1 - side
        for i, lv in enumerate(self.levels):
            if i < depth:
                continue
            rows = n * rps
            if i > 0:                                             # k = 5 block: z = conv5(gelu(bn(x))); planes of x come from the pool
                z = f[(cur + 1) % 4]
                self._gemm(planes(rows, lv["a_cin"], side), lv["a_w"], lv["a_b"], None, z, rows, lv["a_cout"], lv["a_cin"], 5,
                           ACT_NONE, count, rps, nxt=planes(rows, lv["C"], 1 - side), post=lv["b_bn"], post_act=ACT_GELU, pad=2)
                side = 1 - side
                cur = (cur + 1) % 4
            C = lv["C"]
            y = f[(cur + 1) % 4]                                  # 1x1 residual block: y = conv1(gelu(bn(x))) + x ; planes of y for the pool
            self._gemm(planes(rows, C, side), lv["b_w"], lv["b_b"], f[cur], y, rows, C, C, 1, ACT_NONE, count, rps,
                       nxt=planes(rows, C, 1 - side), post=None, post_act=ACT_NONE, pad=2)
            side = 1 - side
            lg = f[(cur + 2) % 4]                                 # attention pooling: logits = W_pool y
            self._gemm(planes(rows, C, side), lv["pool_w"], None, None, lg, rows, C, C, 1, ACT_NONE, count, rps)
            Lo = (Lc + 1) // 2
            last = i + 1 == len(self.levels)
            xn = f[(cur + 3) % 4]
            if last:                                              # the transformer tower takes the fp32 rows
                args = (xn.data_ptr(), _ptr(count), None, None, None, None, ACT_NONE)
            else:                                                 # the next level's k = 5 block takes gelu(bn(x)) as planes only
                nx = self.levels[i + 1]
                pn = planes(n * (Lo + 2), nx["a_cin"], 1 - side)
                args = (None, _ptr(count), pn[0].data_ptr(), pn[1].data_ptr() if P == 2 else None, _ptr(nx["a_bn"][0]),
                        _ptr(nx["a_bn"][1]), ACT_GELU)
                side = 1 - side
            _lib.check(lib.svdd_trunk_attn_pool(y.data_ptr(), lg.data_ptr(), n, Lc, C, *args, _stream()), "svdd_trunk_attn_pool")
            cur = (cur + 3) % 4
            Lc = Lo
            rps = Lc + 2
        # ---- transformer tower on the Lc tokens left (2 for L = 200), pointwise block
        C = self.C
        T = Lc
        x = f[cur][: n * rps * C].view(n, rps, C)[:, :T].reshape(n * T, C).contiguous()
        self._tower(ws, x, zs, n, T, count)
        return stats

    def _tower(self, ws, x, zs, n, T, count):
        """Transformer tower + pointwise block on rows x [n T, C] (fp32) -> zs [n T, pw_out]."""
        f, P, lib, dev, C = ws["f"], self.parts, _lib.lib(), x.device, self.C
        rows = n * T
        cur = 0

        def planes(rows, C, which):
            return ws["p"][which].view(rows, C)[:P]

        for d in self.tf:
            h, dk, dv, nq, nv = d["heads"], d["dk"], d["dv"], d["nq"], d["nv"]
            pl = planes(rows, C, 0)
            self._ln(x, d["ln1"], rows, C, pl, count, T)
            nqkv = 2 * nq + nv
            qkv = f[(cur + 1) % 4][: rows * nqkv].view(rows, nqkv)
            self._gemm(pl, d["qkv_w"], None, None, qkv, rows, nqkv, C, 1, ACT_NONE, count, T)
            pl = planes(rows, h * dv, 1)
            if T <= 4:                                            # one launch: logits, softmax, weighted sum, hi / lo split
                rc = lib.svdd_trunk_attn_small(qkv.data_ptr(), self._rel_k(d, T, dev).data_ptr(), d["content_bias"].data_ptr(),
                                               d["pos_bias"].data_ptr(), n, T, h, dk, dv, pl[0].data_ptr(),
                                               pl[1].data_ptr() if P == 2 else None, _ptr(count), _stream())
                _lib.check(rc, "svdd_trunk_attn_small")
            else:
                q = qkv[:, :nq].view(n, T, h, dk).transpose(1, 2) * dk ** -0.5
                k = qkv[:, nq:2 * nq].view(n, T, h, dk).transpose(1, 2)
                v = qkv[:, 2 * nq:].view(n, T, h, dv).transpose(1, 2)
                rel_logits = _relative_shift(torch.einsum("bhid,hjd->bhij", q + d["pos_bias"], self._rel_k(d, T, dev)))
                logits = torch.matmul(q + d["content_bias"], k.transpose(-1, -2)) + rel_logits
                o = torch.matmul(torch.softmax(logits, dim=-1), v).transpose(1, 2).reshape(rows, h * dv).contiguous()
                self._act(o, None, ACT_NONE, rows, h * dv, T, 0, pl, count)
            x2 = f[(cur + 2) % 4][: rows * C].view(rows, C)
            self._gemm(pl, d["out_w"], d["out_b"], x, x2, rows, C, h * dv, 1, ACT_NONE, count, T)
            pl = planes(rows, C, 0)
            self._ln(x2, d["ln2"], rows, C, pl, count, T)
            # FFN: the hidden layer only ever exists as the operand planes of the second GEMM
            hid = planes(rows, 2 * C, 1)
            self._gemm(pl, d["f1_w"], d["f1_b"], None, None, rows, 2 * C, C, 1, ACT_RELU, count, T, nxt=hid)
            x3 = f[(cur + 3) % 4][: rows * C].view(rows, C)
            self._gemm(hid, d["f2_w"], d["f2_b"], x2, x3, rows, C, 2 * C, 1, ACT_NONE, count, T)
            x = x3
            cur = (cur + 3) % 4
        # ---- pointwise block (no residual, no pool) + trunk GELU
        pl = planes(rows, C, 0)
        self._act(x, self.pw_bn, ACT_GELU, rows, C, T, 0, pl, count)
        self._gemm(pl, self.pw_w, self.pw_b, None, zs, rows, self.pw_out, C, 1, ACT_GELU, count, T)

    def forward(self, onehot):
        """Value-function interface: one-hot fp32 [n, L, 4] (or the reward-model layout [n, 4, L]) -> [n, n_tasks, 1]."""
        if onehot.shape[1] == 4 and onehot.shape[2] != 4:
            onehot = onehot.transpose(1, 2)
        tok = torch.where(onehot.sum(dim=2) == 0, 4, onehot.argmax(dim=2)).to(torch.uint8).contiguous()
        return self.forward_tokens(tok)

    def tokens_ok(self, L):
        return True

"""Inference-time re-formulations of the nets for MI355X.

The sampler treats the nets as opaque callables; these wrappers compute the SAME functions as
`backbone.CNNModel` and `value_nets.ConvGRUTrunk + ConvHead` (same weights, fp32) but laid out
for the hardware:

  * activations stay channels-last ([n, L, C] in memory) end to end, so MIOpen's NHWC implicit-GEMM
    convolutions run without the NCHW<->NHWC `batched_transpose` kernels PyTorch otherwise inserts
    around every Conv1d, and LayerNorm normalises the contiguous last dimension;
  * eval-mode BatchNorm is folded into the preceding convolution (rocprof r01_v0: MIOpen's
    BatchNormFwdInfer took 0.72 ms per call, 12.5 % of the step);
  * the bidirectional GRU — ~2900 MIOpen launches and 11.7 ms per call at n=2560, L=200 — is ONE
    launch of the hand-written MFMA kernel `svdd_gru_bidir_f32` (csrc/svdd_nets.hip).

Numerics: fp32 throughout; results differ from the plain modules only by fp32 re-association
(BN folding, GEMM tiling) and the GRU's hardware exp/rcp — observed <= 1e-5 on the scores,
well inside the 1e-4 soft-value tolerance of the north star (tests/test_fused_gpu.py).
"""
import ctypes

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .backbone import CNNModel
from .value_nets import ConvGRUTrunk, ConvHead


def _cl(w):
    """Conv1d weight [O,I,k] -> Conv2d weight [O,I,1,k] in channels_last memory."""
    return w.detach().unsqueeze(2).contiguous(memory_format=torch.channels_last)


def pack_gru(gru):
    """nn.GRU(64, 64, bidirectional) -> (wpack [2,4,64,96], bpack [2,4,64]) for svdd_gru_bidir_f32.

    Lane (j = lane & 15, g = lane >> 4) of wave w holds, for hidden unit u = 16 w + j and input
    k = 16 g + s (s = 0..15): W_ir, W_hr, W_iz, W_hz, W_in, W_hn rows — the B operands of
    v_mfma_f32_16x16x4_f32 with the k axis permuted so that each lane's A operand is 16 contiguous floats."""
    H = gru.hidden_size
    assert H == 64 and gru.input_size == 64 and gru.bidirectional and gru.num_layers == 1 and gru.bias
    packs, biases = [], []
    for sfx in ("", "_reverse"):
        w_ih = getattr(gru, "weight_ih_l0" + sfx).detach().float()     # [3H, I] rows r,z,n
        w_hh = getattr(gru, "weight_hh_l0" + sfx).detach().float()
        b_ih = getattr(gru, "bias_ih_l0" + sfx).detach().float()
        b_hh = getattr(gru, "bias_hh_l0" + sfx).detach().float()
        mats = [w_ih[0:H], w_hh[0:H], w_ih[H:2 * H], w_hh[H:2 * H], w_ih[2 * H:], w_hh[2 * H:]]   # each [u, k]
        # [6, u=(w,j), k=(g,s)] -> [w, g, j, 6, s] -> lanes = g*16 + j
        t = torch.stack(mats).view(6, 4, 16, 4, 16).permute(1, 3, 2, 0, 4).reshape(4, 64, 96)
        packs.append(t)
        biases.append(torch.stack([b_ih[0:H] + b_hh[0:H], b_ih[H:2 * H] + b_hh[H:2 * H], b_ih[2 * H:], b_hh[2 * H:]]))
    return torch.stack(packs).contiguous(), torch.stack(biases).contiguous()


def gru_bidir(x_nlc, wpack, bpack, count=None, out=None):
    """x [n, L, 64] fp32 (contiguous) -> [2, n, L, 64] per-direction hidden states (HIP kernel). count: int32 device
    scalar with the number of valid rows of a compacted batch (None: n)."""
    assert x_nlc.is_cuda and x_nlc.dtype == torch.float32 and x_nlc.is_contiguous() and x_nlc.shape[2] == 64
    n, L, _ = x_nlc.shape
    if out is None:
        out = torch.empty((2, n, L, 64), dtype=torch.float32, device=x_nlc.device)
    rc = _lib.lib().svdd_gru_bidir_f32(x_nlc.data_ptr(), wpack.data_ptr(), bpack.data_ptr(), out.data_ptr(), n, L,
                                       _ptr(count), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_gru_bidir_f32")
    return out


def _ptr(t):
    return t.data_ptr() if t is not None else None


def pack_gru_bwd(gru):
    """nn.GRU(64, 64, bidirectional) -> wpack_bwd [2, 4, 64, 96] for svdd_gru_bidir_bwd_f32: lane (j, g) of wave w holds, for
    output column c = 16 w + j and k' = 48 g + s (k' = gate * 64 + unit, gates r, z, n): [0:48) W_hh[k'][c], [48:96) W_ih[k'][c]."""
    H = gru.hidden_size
    assert H == 64 and gru.input_size == 64 and gru.bidirectional and gru.num_layers == 1 and gru.bias
    packs = []
    for sfx in ("", "_reverse"):
        w_ih = getattr(gru, "weight_ih_l0" + sfx).detach().float()     # [3H (k'), 64 (c)]
        w_hh = getattr(gru, "weight_hh_l0" + sfx).detach().float()
        both = torch.stack([w_hh, w_ih])                                # [2, k' = (g, s), c = (w, j)]
        packs.append(both.view(2, 4, 48, 4, 16).permute(3, 1, 4, 0, 2).reshape(4, 64, 96))   # [w][g][j][2][s] -> lanes = 16 g + j
    return torch.stack(packs).contiguous()


class GruBidirFunction(torch.autograd.Function):
    """The bidirectional GRU on the hand-written kernels WITH a gradient to its input (csrc/svdd_gru_train.hip): forward =
    svdd_gru_bidir_train_f32 (the inference kernel's bits + saved gates), backward = svdd_gru_bidir_bwd_f32 (BPTT, d/dx
    only: the weights are frozen in every decode path). x [n, L, 64] -> [2, n, L, 64] per-direction hidden states."""

    @staticmethod
    def forward(ctx, x, wpack, bpack, wpack_bwd):
        assert x.is_cuda and x.dtype == torch.float32 and x.shape[2] == 64
        x = x.contiguous()
        n, L, _ = x.shape
        out = torch.empty((2, n, L, 64), dtype=torch.float32, device=x.device)
        save = torch.empty((2, n, L, 4, 64), dtype=torch.float32, device=x.device)
        rc = _lib.lib().svdd_gru_bidir_train_f32(x.data_ptr(), wpack.data_ptr(), bpack.data_ptr(), out.data_ptr(), save.data_ptr(),
                                                 n, L, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "svdd_gru_bidir_train_f32")
        ctx.save_for_backward(out, save, wpack_bwd)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        out, save, wpack_bwd = ctx.saved_tensors
        _, n, L, _ = out.shape
        g = grad_out.contiguous().float()
        dx = torch.empty((2, n, L, 64), dtype=torch.float32, device=out.device)
        rc = _lib.lib().svdd_gru_bidir_bwd_f32(g.data_ptr(), out.data_ptr(), save.data_ptr(), wpack_bwd.data_ptr(), dx.data_ptr(),
                                               n, L, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _lib.check(rc, "svdd_gru_bidir_bwd_f32")
        return dx[0] + dx[1], None, None, None


class DilatedConvFunction(torch.autograd.Function):
    """One 'same'-padded stride-1 Conv1d (odd kernel, any dilation, no bias) on channels-last rows [n, L, cin] through
    svdd_conv1d_cl_f32 in BOTH directions: its backward-data pass is the same convolution with the taps flipped and the
    channel axes swapped (wpack_t = pack_conv(W.flip(2).transpose(0, 1))). Weights are frozen (no weight gradient): the
    DPS baseline only differentiates with respect to the input (reference diffusion_gosai.py:1321-1330)."""

    @staticmethod
    def forward(ctx, x, wpack, wpack_t, cout, cin, taps, dilation):
        ctx.cfg = (cin, taps, dilation)
        ctx.save_for_backward(wpack_t)
        return conv1d_cl(x.contiguous(), wpack, cout, taps, dilation)

    @staticmethod
    def backward(ctx, g):
        cin, taps, dilation = ctx.cfg
        (wpack_t,) = ctx.saved_tensors
        return conv1d_cl(g.contiguous().float(), wpack_t, cin, taps, dilation), None, None, None, None, None, None


class BackboneLayersFunction(torch.autograd.Function):
    """The residual layers of the dilated-CNN backbone on channels-last rows (reference models/dnaconv.py:212-247 forward2:
    feat' = relu(conv_i(LayerNorm_i(feat + time_bias_i)) + b_i) + feat, i = 0 .. n - 1) with a hand-written backward to the
    INPUT: per layer and direction one convolution (svdd_conv1d_cl_f32; backward-data = the same kernel on flipped /
    transposed taps) and ONE element-wise pass (svdd_bb_layer_fwd_f32 / svdd_bb_layer_bwd_f32) instead of PyTorch's add,
    LayerNorm, ReLU, add and their four backward kernels. Saves every layer's input rows and ReLU mask. Weights are frozen
    (no weight gradients): the DPS baseline differentiates with respect to the input only (diffusion_gosai.py:1321-1330)."""

    keep_masks, last_masks = False, None

    @staticmethod
    def forward(ctx, feat, tb, gamma, beta, bias, eps, packs):
        """feat [B, L, C] ; tb [n, B, C] time biases ; gamma, beta, bias [n, C] ; packs: n x (wpack, wpack_t, dilation)."""
        feat = feat.contiguous().float()
        B, L, C = feat.shape
        n = len(packs)
        lib, st = _lib.lib(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rows = B * L
        feats = torch.empty((n + 1, B, L, C), dtype=torch.float32, device=feat.device)
        masks = torch.empty((n, B, L, C), dtype=torch.uint8, device=feat.device)
        hn = torch.empty_like(feat)
        feats[0].copy_(feat)
        _lib.check(lib.svdd_bb_layer_fwd_f32(None, None, feats[0].data_ptr(), tb[0].data_ptr(), gamma[0].data_ptr(), beta[0].data_ptr(),
                                             eps, None, None, hn.data_ptr(), rows, L, C, st), "svdd_bb_layer_fwd_f32")
        for i, (wp, _, d) in enumerate(packs):
            y = conv1d_cl(hn, wp, C, 9, d)
            last = i + 1 == n
            _lib.check(lib.svdd_bb_layer_fwd_f32(y.data_ptr(), bias[i].data_ptr(), feats[i].data_ptr(),
                                                 None if last else tb[i + 1].data_ptr(), None if last else gamma[i + 1].data_ptr(),
                                                 None if last else beta[i + 1].data_ptr(), eps, feats[i + 1].data_ptr(),
                                                 masks[i].data_ptr(), None if last else hn.data_ptr(), rows, L, C, st),
                       "svdd_bb_layer_fwd_f32")
        ctx.save_for_backward(feats, masks, tb, gamma)
        ctx.cfg = (eps, packs)
        if BackboneLayersFunction.keep_masks:                     # tests: the ReLU decisions of this forward pass
            BackboneLayersFunction.last_masks = masks
        return feats[n].clone()

    @staticmethod
    def backward(ctx, G):
        feats, masks, tb, gamma = ctx.saved_tensors
        eps, packs = ctx.cfg
        n, B, L, C = masks.shape
        lib, st = _lib.lib(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rows = B * L
        G = G.contiguous().float()
        gt = G * masks[n - 1]                                      # gradient at the last layer's pre-activation
        for i in range(n - 1, -1, -1):
            g_hn = conv1d_cl(gt, packs[i][1], C, 9, packs[i][2])
            G_new = torch.empty_like(G)
            gt = torch.empty_like(G) if i else None
            _lib.check(lib.svdd_bb_layer_bwd_f32(g_hn.data_ptr(), feats[i].data_ptr(), tb[i].data_ptr(), gamma[i].data_ptr(), eps,
                                                 G.data_ptr(), masks[i - 1].data_ptr() if i else None, G_new.data_ptr(),
                                                 gt.data_ptr() if i else None, rows, L, C, st), "svdd_bb_layer_bwd_f32")
            G = G_new
        return G, None, None, None, None, None, None


def pack_conv(weight):
    """Conv1d weight [cout, cin, taps] -> [taps][cin/32][cout][32] for svdd_conv1d_cl_f32."""
    co, ci, T = weight.shape
    assert ci % 32 == 0
    return weight.detach().float().permute(2, 1, 0).reshape(T, ci // 32, 32, co).permute(0, 1, 3, 2).contiguous()


def conv1d_cl(x_nlc, wpack, cout, taps, dilation, bias=None, f_prev=None, act=-1, ln=None):
    """x [n, L, cin] fp32 contiguous -> y [n, L, cout], HIP kernel svdd_conv1d_cl_f32.
    act -1: raw conv (no bias); 0: relu(conv + bias) + f_prev; 1: relu(conv + bias + f_prev); 2: conv + bias + f_prev.
    ln = (tb | None, gamma, beta): also returns hn = LayerNorm(y + tb) * gamma + beta -> (y, hn)."""
    assert x_nlc.is_cuda and x_nlc.dtype == torch.float32 and x_nlc.is_contiguous()
    n, L, cin = x_nlc.shape
    y = torch.empty((n, L, cout), dtype=torch.float32, device=x_nlc.device)
    hn = torch.empty_like(y) if ln is not None else None
    tb, gamma, beta = ln if ln is not None else (None, None, None)
    rc = _lib.lib().svdd_conv1d_cl_f32(x_nlc.data_ptr(), wpack.data_ptr(), y.data_ptr(), n, L, cin, cout, taps, dilation,
                                       _ptr(bias), _ptr(f_prev), int(act), _ptr(tb), _ptr(gamma), _ptr(beta), _ptr(hn),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_conv1d_cl_f32")
    return y if ln is None else (y, hn)


ACT_RELU_THEN_ADD, ACT_ADD_THEN_RELU, ACT_NONE = 0, 1, 2


def pack_tower(stem_weight, layer_weights):
    """Weight tiles [2 + 10*nlayers][64][32] in the execution order of svdd_conv_tower_f32.
    stem_weight [64,4,15]: K index k = 4*tap + channel (60 real + 4 zero), two 32-wide chunks;
    each layer weight [64,64,5] (BatchNorm already folded): for chunk c in (0,1), for tap t in 0..4: W[:, 32c:32c+32, t]."""
    co = stem_weight.shape[0]
    k = stem_weight.detach().float().permute(0, 2, 1).reshape(co, 60)                  # [co][4*t + c]
    k = torch.cat([k, k.new_zeros(co, 4)], dim=1)
    tiles = [k[:, :32], k[:, 32:]]
    for w in layer_weights:
        w = w.detach().float()
        for c in range(2):
            for t in range(5):
                tiles.append(w[:, 32 * c: 32 * c + 32, t])
    return torch.stack([t.contiguous() for t in tiles]).contiguous()


def conv_tower(onehot, tiles, bias, residual_mask, count=None):
    """onehot [n, L, 4] fp32 -> [n, L, 64]: stem + (len(bias) - 1) residual conv layers in ONE launch
    (HIP kernel svdd_conv_tower_f32, activations resident in LDS)."""
    assert onehot.is_cuda and onehot.dtype == torch.float32 and onehot.is_contiguous() and onehot.shape[2] == 4
    n, L, _ = onehot.shape
    out = torch.empty((n, L, 64), dtype=torch.float32, device=onehot.device)
    rc = _lib.lib().svdd_conv_tower_f32(onehot.data_ptr(), tiles.data_ptr(), bias.data_ptr(), out.data_ptr(), n, L,
                                        bias.shape[0] - 1, int(residual_mask), _ptr(count),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_conv_tower_f32")
    return out


def pack_tail(w1, b1, gamma, beta):
    """(dense1 weight W1 [128, 64], bias b1, LayerNorm gamma/beta [64]) -> (w1pack [64 lanes][128], b1' [128]) for
    svdd_value_tail_f32. The LayerNorm affine is folded into the linear map: W1' = W1 diag(gamma), b1' = b1 + W1 beta.
    Lane (j = lane & 15, g = lane >> 4) holds W1'[16 ct + j][ch(g, s)] at index 16 ct + s, with
    ch(g, s) = 16 (s // 4) + 4 g + s % 4 (the channel order in which the kernel's lanes read a row)."""
    w64 = w1.detach().double()
    wf = (w64 * gamma.detach().double()[None, :]).float()
    bf = (b1.detach().double() + w64 @ beta.detach().double()).float().contiguous()
    w = wf.view(8, 16, 4, 4, 4)                                   # [ct][j][i = s // 4][g][c = s % 4]
    return w.permute(3, 1, 0, 2, 4).reshape(64, 128).contiguous(), bf


def value_tail(h, w1pack, b1f, w_eff, b_eff, count=None, out=None):
    """h [2, n, L, 64] (GRU output, both directions) -> scores [n, n_tasks]: direction sum + LayerNorm + dense1 + ReLU +
    collapsed (dense2, head) + mean over length in one pass (HIP kernel svdd_value_tail_f32); (w1pack, b1f) from pack_tail."""
    assert h.is_cuda and h.dtype == torch.float32 and h.is_contiguous() and h.shape[0] == 2 and h.shape[3] == 64
    _, n, L, _ = h.shape
    T = w_eff.shape[1]
    if out is None:
        out = torch.empty((n, T), dtype=torch.float32, device=h.device)
    assert out.is_contiguous() and out.shape == (n, T)
    rc = _lib.lib().svdd_value_tail_f32(h[0].data_ptr(), h[1].data_ptr(), w1pack.data_ptr(), b1f.data_ptr(),
                                        w_eff.data_ptr(), b_eff.data_ptr(), out.data_ptr(), n, L, T, _ptr(count),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_value_tail_f32")
    return out


def pack_backbone(cnn):
    """CNNModel (hidden_dim 128, alphabet 5) -> the operand images of svdd_backbone_cnn_f32 (include/svdd_hip.h):
    dict(table0, tiles, vec, w2, dil). Time biases are those of sigma == 0 (diffusion_gosai.py:334-335)."""
    H = cnn.args.hidden_dim
    assert H == 128 and cnn.alphabet_size == 5 and cnn.linear.kernel_size[0] == 9
    nl = len(cnn.convs)
    with torch.no_grad():
        dev = cnn.linear.weight.device
        table0 = cnn.linear.weight.detach().float().permute(2, 1, 0).contiguous()       # [t][c][co]
        tiles = []
        for conv in cnn.convs:
            assert conv.kernel_size[0] == 9 and conv.in_channels == H and conv.out_channels == H
            w = conv.weight.detach().float()                                              # [co][ci][t]
            tiles.append(w.view(H, 4, 32, 9).permute(1, 3, 0, 2))                        # [c][t][co][k]
        wf1 = cnn.final_conv[0].weight.detach().float()[:, :, 0]                          # [co][ci]
        tiles = torch.cat([torch.stack(tiles).reshape(-1), wf1.view(H, 4, 32).permute(1, 0, 2).reshape(-1)]).contiguous()
        tb = cnn._time_biases(torch.zeros(1, device=dev))
        vec = torch.zeros(nl + 2, 4, H, device=dev)
        vec[0, 0] = cnn.linear.bias
        for i in range(nl):
            vec[1 + i, 0] = cnn.convs[i].bias
            vec[1 + i, 1] = tb[i].reshape(H)
            vec[1 + i, 2] = cnn.norms[i].weight
            vec[1 + i, 3] = cnn.norms[i].bias
        vec[nl + 1, 0] = cnn.final_conv[0].bias
        w2 = torch.cat([cnn.final_conv[2].weight.detach().float()[:, :, 0].reshape(-1),
                        cnn.final_conv[2].bias.detach().float()]).contiguous()
    return dict(table0=table0, tiles=tiles, vec=vec.contiguous(), w2=w2, dil=[c.dilation[0] for c in cnn.convs])


_BB_SPLIT_WS = {}          # device -> the scratch tensor registered with svdd_backbone_set_workspace (kept alive here)
BB_SPLIT_MAX_SEQ = 128     # the small-batch form splits a sequence over 2 workgroups up to 128 sequences, over 4 up to 64


def _backbone_split_workspace(dev):
    """Caller-owned scratch of the small-batch backbone (svdd_backbone_cnn_f32 on 2 / 4 workgroups per sequence, same bits):
    the double-buffered LayerNorm images of up to 128 sequences + arrival counters (27 MB), registered once per process."""
    key = str(dev)
    nbytes = BB_SPLIT_MAX_SEQ * (2 * 208 * 128 * 4 + 4) + 4
    if key not in _BB_SPLIT_WS:
        _BB_SPLIT_WS[key] = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    if _BB_SPLIT_WS.get("registered") != key:          # the library holds ONE workspace pointer: follow the device in use
        _lib.check(_lib.lib().svdd_backbone_set_workspace(_BB_SPLIT_WS[key].data_ptr(), nbytes), "svdd_backbone_set_workspace")
        _BB_SPLIT_WS["registered"] = key
    return _BB_SPLIT_WS[key]


def pack_backbone_grad(cnn):
    """CNNModel -> the operand images of svdd_backbone_cnn_grad_f32 (the input gradient of the whole backbone in one launch, DPS):
    dict(tiles_bwd, gamma). The backward-data pass of a 'same'-padded stride-1 dilated convolution is the same convolution with the
    taps flipped and the channel axes swapped, so the tiles are pack_backbone's layout of W'[ci][co][t] = W[co][ci][8 - t] — in
    PROCESSING order: W_f1^T (the first 1x1 of final_conv) as [4][128][32], then layer nl - 1 down to 0 as [4][9][128][32]."""
    H = cnn.args.hidden_dim
    assert H == 128 and cnn.alphabet_size == 5 and cnn.linear.kernel_size[0] == 9
    with torch.no_grad():
        wf1t = cnn.final_conv[0].weight.detach().float()[:, :, 0].t().contiguous()        # [ci][co]: output channel of the transpose first
        parts = [wf1t.view(H, 4, 32).permute(1, 0, 2).reshape(-1)]
        for conv in reversed(list(cnn.convs)):
            wt = conv.weight.detach().float().flip(2).transpose(0, 1).contiguous()         # [ci][co][t']
            parts.append(wt.view(H, 4, 32, 9).permute(1, 3, 0, 2).reshape(-1))            # [c][t][ci][k]
        gamma = torch.stack([n.weight.detach().float() for n in cnn.norms]).contiguous()
    return dict(tiles_bwd=torch.cat(parts).contiguous(), gamma=gamma)


def backbone_cnn_save(tokens, pk):
    """The one-launch forward (the inference kernel's bits) that also leaves what the gradient kernel needs, in its lane-private
    layout (svdd_backbone_cnn_save_f32): tokens [n, L] u8, 104 < L <= 208 -> (logits [n, L, 5], (xhat, rstd, mask))."""
    assert tokens.is_cuda and tokens.dtype == torch.uint8 and tokens.is_contiguous()
    n, L = tokens.shape
    nl = len(pk["dil"])
    dev = tokens.device
    out = torch.empty((n, L, 5), dtype=torch.float32, device=dev)
    xhat = torch.empty((n, nl, 56, 512), dtype=torch.float32, device=dev)
    rstd = torch.empty((n, nl, 208), dtype=torch.float32, device=dev)
    mask = torch.empty((n, nl + 2, 512), dtype=torch.int64, device=dev)
    dil = (ctypes.c_int * nl)(*pk["dil"])
    rc = _lib.lib().svdd_backbone_cnn_save_f32(tokens.data_ptr(), pk["table0"].data_ptr(), pk["tiles"].data_ptr(), pk["vec"].data_ptr(),
                                               pk["w2"].data_ptr(), out.data_ptr(), n, L, nl, dil, xhat.data_ptr(), rstd.data_ptr(),
                                               mask.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_backbone_cnn_save_f32")
    return out, (xhat, rstd, mask)


def backbone_cnn_grad(dlogits, pk, pkg, saved):
    """d loss / d onehot(x) [n, L, 5] from d loss / d logits [n, L, 5] through the whole backbone, one launch
    (svdd_backbone_cnn_grad_f32); `saved` from backbone_cnn_save of the same tokens."""
    xhat, rstd, mask = saved
    n, L, _ = dlogits.shape
    nl = len(pk["dil"])
    g = dlogits.contiguous().float()
    dx = torch.empty((n, L, 5), dtype=torch.float32, device=g.device)
    dil = (ctypes.c_int * nl)(*pk["dil"])
    rc = _lib.lib().svdd_backbone_cnn_grad_f32(g.data_ptr(), pkg["tiles_bwd"].data_ptr(), pkg["gamma"].data_ptr(), pk["w2"].data_ptr(),
                                               pk["table0"].data_ptr(), xhat.data_ptr(), rstd.data_ptr(), mask.data_ptr(), dx.data_ptr(),
                                               n, L, nl, dil, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_backbone_cnn_grad_f32")
    return dx


def decode_backbone_masks(mask, L):
    """The ReLU decisions backbone_cnn_save left (int64 [n, nl + 2, 512], lane-private bits) as bool [n, nl + 2, L, 128] (tests)."""
    n, K, _ = mask.shape
    tid = torch.arange(512, device=mask.device)
    w, lane = tid >> 6, tid & 63
    cg, rh, j, g = w & 3, w >> 2, lane & 15, lane >> 4
    out = torch.zeros((n, K, 208, 128), dtype=torch.bool, device=mask.device)
    for r in range(7):
        for ct in range(2):
            for e in range(4):
                row = 16 * (rh + 2 * r) + 4 * g + e
                col = 32 * cg + j + 16 * ct
                ok = row < 208
                bit = ((mask >> ((r * 2 + ct) * 4 + e)) & 1).bool()                     # [n, K, 512]
                out[:, :, row[ok], col[ok]] = bit[:, :, ok]
    return out[:, :, :L]


class BackboneOneLaunchFunction(torch.autograd.Function):
    """CNNModel.forward2 on a HARD one-hot input (the DPS case: x_onehot = one_hot(x_t), diffusion_gosai.py:1308) as one launch
    each way: forward = the inference kernel's bits + saved x-hat / rstd / ReLU decisions, backward = the input gradient through all
    20 layers (svdd_backbone_cnn_grad_f32). Weights frozen: the only gradient is the one with respect to `x_onehot`."""

    @staticmethod
    def forward(ctx, x_onehot, tokens, pk, pkg):
        logits, saved = backbone_cnn_save(tokens, pk)
        ctx.pk, ctx.pkg, ctx.saved = pk, pkg, saved
        return logits

    @staticmethod
    def backward(ctx, g):
        dx = backbone_cnn_grad(g, ctx.pk, ctx.pkg, ctx.saved)
        ctx.saved = None
        return dx, None, None, None


def check_backbone_split():
    """Raises if a group barrier of a small-batch backbone launch (several workgroups per sequence) timed out since the last
    check: its logits were computed from a partly exchanged image. Reads one int from the device (synchronises) — only after a
    split launch may have happened; the samplers call it at the end of every decode (Diffusion._decode_scope)."""
    if not _BB_SPLIT_WS.pop("used", False):
        return
    err = ctypes.c_int(0)
    _lib.check(_lib.lib().svdd_backbone_split_status(ctypes.byref(err)), "svdd_backbone_split_status")
    if err.value:
        raise _lib.SvddError("svdd_backbone_cnn_f32: a workgroup of a small-batch (several workgroups per sequence) launch waited in vain "
                             "for its partners — something else occupied the CUs (another process or stream on this GPU). The logits "
                             "of that launch are invalid. Run with svdd_set_option(SVDD_OPT_BACKBONE_SPLIT, 1) when the GPU is shared.")


def backbone_cnn(tokens, pk, count=None, out=None, row_idx=None, scatter=False):
    """tokens [n, L] uint8 -> raw logits fp32 [n, L, 5]: the whole backbone forward in ONE launch
    (HIP kernel svdd_backbone_cnn_f32)."""
    assert tokens.is_cuda and tokens.dtype == torch.uint8 and tokens.is_contiguous()
    n, L = tokens.shape
    if 104 < L <= 208 and count is None and row_idx is None and (n <= BB_SPLIT_MAX_SEQ or 0 < n % 256 <= BB_SPLIT_MAX_SEQ):
        _backbone_split_workspace(tokens.device)       # small batches, and the tail round of a batch that is not a multiple of the CUs
        _BB_SPLIT_WS["used"] = True                    # -> check_backbone_split() at the end of the decode
    if out is None:
        out = torch.empty((n, L, 5), dtype=torch.float32, device=tokens.device)
    dil = (ctypes.c_int * len(pk["dil"]))(*pk["dil"])
    rc = _lib.lib().svdd_backbone_cnn_f32(tokens.data_ptr(), pk["table0"].data_ptr(), pk["tiles"].data_ptr(),
                                          pk["vec"].data_ptr(), pk["w2"].data_ptr(), out.data_ptr(), n, L,
                                          len(pk["dil"]), dil, _ptr(count), _ptr(row_idx), int(scatter),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_backbone_cnn_f32")
    return out


LP_DTYPES = {"f16x3": (torch.float16, 2), "bf16x3": (torch.bfloat16, 2), "f16": (torch.float16, 1), "bf16": (torch.bfloat16, 1)}


def _pow2_floor(v):
    import math
    return 2.0 ** math.floor(math.log2(v))


def _split16(w, dtype, parts):
    """fp32 tensor -> [..., parts] 16-bit: hi = rn16(w) and (parts == 2) lo = rn16(w - hi)."""
    hi = w.to(dtype)
    if parts == 1:
        return hi.unsqueeze(-1)
    lo = (w - hi.float()).to(dtype)
    return torch.stack([hi, lo], dim=-1)


def pack_backbone_lp(cnn, precision):
    """CNNModel -> the operand images of svdd_backbone_cnn_lp (include/svdd_hip.h) for one of the split-precision modes
    "f16x3", "bf16x3", "f16", "bf16": dict(table0, tiles, vec, lscale, w2, dil, prec).
    Weight tile (layer, chunk c, tap t): [4 cg][64 lanes = 16 g + j][2 ct][parts][8 e] =
    s_w * W[32 cg + 2 j + ct][32 c + 8 g + e][t]. f16 operands are scaled by powers of two (exact): the weights of a
    stage so that max |W| lands in [1024, 2048), the LayerNorm output by sa <= 16 chosen from its bound
    sqrt(127) max|gamma| + max|beta| so that it cannot overflow; bf16 needs no scaling."""
    dtype, parts = LP_DTYPES[precision]
    pk = pack_backbone(cnn)
    H, nl = 128, len(cnn.convs)
    f16 = dtype == torch.float16
    tiles, lscale = [], []
    with torch.no_grad():
        def tile_set(w, taps):                      # w [co][ci][taps] fp32 (already scaled) -> [c][t][cg][g][j][ct][parts][e]
            v = _split16(w.reshape(4, 16, 2, 4, 4, 8, taps), dtype, parts)       # [cg][j][ct][c][g][e][t][parts]
            return v.permute(3, 6, 0, 4, 1, 2, 7, 5).contiguous()
        for i, conv in enumerate(cnn.convs):
            w = conv.weight.detach().float()
            sw = _pow2_floor(2047.0 / float(w.abs().max())) if f16 else 1.0
            bound = 11.27 * float(cnn.norms[i].weight.abs().max()) + float(cnn.norms[i].bias.abs().max())
            sa = min(16.0, _pow2_floor(16384.0 / max(bound, 1e-30))) if f16 else 1.0
            tiles.append(tile_set(w * sw, 9).reshape(-1))
            lscale.append((sa, 1.0 / (sa * sw)))
        wf1 = cnn.final_conv[0].weight.detach().float()                           # [co][ci][1]
        sw = _pow2_floor(2047.0 / float(wf1.abs().max())) if f16 else 1.0
        tiles.append(tile_set(wf1 * sw, 1).reshape(-1))
        lscale.append((1.0, 1.0 / sw))
        pk["tiles"] = torch.cat(tiles).contiguous()
        pk["lscale"] = torch.tensor(lscale, dtype=torch.float32, device=pk["vec"].device).contiguous()
    pk["prec"] = _lib.PRECISIONS[precision]
    return pk


def backbone_cnn_lp(tokens, pk, count=None, out=None, row_idx=None, scatter=False):
    """tokens [n, L] uint8 -> raw logits fp32 [n, L, 5] on the 16-bit matrix cores (HIP kernel svdd_backbone_cnn_lp);
    pk from pack_backbone_lp."""
    assert tokens.is_cuda and tokens.dtype == torch.uint8 and tokens.is_contiguous()
    n, L = tokens.shape
    if out is None:
        out = torch.empty((n, L, 5), dtype=torch.float32, device=tokens.device)
    dil = (ctypes.c_int * len(pk["dil"]))(*pk["dil"])
    rc = _lib.lib().svdd_backbone_cnn_lp(tokens.data_ptr(), pk["table0"].data_ptr(), pk["tiles"].data_ptr(),
                                         pk["vec"].data_ptr(), pk["lscale"].data_ptr(), pk["w2"].data_ptr(),
                                         out.data_ptr(), n, L, len(pk["dil"]), dil, pk["prec"], _ptr(count), _ptr(row_idx),
                                         int(scatter), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_backbone_cnn_lp")
    return out


def pack_tower_lp(stem_weight, layer_weights, precision):
    """(tiles, inv) for svdd_conv_tower_lp: the tiles of pack_tower re-ordered per MFMA lane and split into 16-bit hi / lo,
    [2 + 10*nlayers][2 cp][64 lanes = 16 g + j][2 ct][P][8 e] = s_w * W[32 cp + 2 j + ct][8 g + e]; inv[stage] = 1 / s_w
    (f16: s_w a power of two that puts max |W| of the stage into [1024, 2048); bf16: 1)."""
    dtype, parts = LP_DTYPES[precision]
    f16 = dtype == torch.float16
    t32 = pack_tower(stem_weight, layer_weights)                         # [2 + 10 nl][64 co][32 k]
    nl = len(layer_weights)
    stage = torch.tensor([0, 0] + [1 + i for i in range(nl) for _ in range(10)], device=t32.device)
    inv = []
    for st in range(nl + 1):
        m = float(t32[stage == st].abs().max())
        inv.append(1.0 / _pow2_floor(2047.0 / m) if f16 else 1.0)
    invt = torch.tensor(inv, dtype=torch.float32, device=t32.device)
    scaled = t32 / invt[stage][:, None, None]
    v = _split16(scaled.reshape(-1, 2, 16, 2, 4, 8), dtype, parts)       # [it][cp][j][ct][g][e][parts]
    return v.permute(0, 1, 4, 2, 3, 6, 5).contiguous().reshape(-1), invt.contiguous()


def pack_gru_lp(gru, precision):
    """nn.GRU(64, 64, bidirectional) -> (wpack, bpack, inv) for svdd_gru_bidir_lp: wpack [2][4 w][64 lanes][6][2 c][P][8] =
    s_w * W_m[16 w + j][32 c + 8 g + e], m = ir, hr, iz, hz, in, hn; bpack as pack_gru; inv [2] = 1 / s_w per direction."""
    dtype, parts = LP_DTYPES[precision]
    f16 = dtype == torch.float16
    H = gru.hidden_size
    assert H == 64 and gru.input_size == 64 and gru.bidirectional and gru.num_layers == 1 and gru.bias
    _, bpack = pack_gru(gru)
    packs, inv = [], []
    for sfx in ("", "_reverse"):
        w_ih = getattr(gru, "weight_ih_l0" + sfx).detach().float()
        w_hh = getattr(gru, "weight_hh_l0" + sfx).detach().float()
        mats = torch.stack([w_ih[0:H], w_hh[0:H], w_ih[H:2 * H], w_hh[H:2 * H], w_ih[2 * H:], w_hh[2 * H:]])   # [6][u][k]
        sw = _pow2_floor(2047.0 / float(mats.abs().max())) if f16 else 1.0
        v = _split16((mats * sw).reshape(6, 4, 16, 2, 4, 8), dtype, parts)      # [m][w][j][c][g][e][parts]
        packs.append(v.permute(1, 4, 2, 0, 3, 6, 5).contiguous())               # [w][g][j][m][c][parts][e]
        inv.append(1.0 / sw)
    return (torch.stack(packs).contiguous().reshape(-1), bpack,
            torch.tensor(inv, dtype=torch.float32, device=bpack.device))


def pack_tail_lp(w1, b1, gamma, beta, precision):
    """-> (w1pack 16-bit, b1' fp32 [128], inv float) for svdd_value_tail_lp: the LayerNorm affine folded as in pack_tail,
    w1pack [64 lanes = 16 g + j][8 ct][2 c][P][8] = s_w * W1'[16 ct + j][32 c + 8 g + e]."""
    dtype, parts = LP_DTYPES[precision]
    w64 = w1.detach().double()
    wf = (w64 * gamma.detach().double()[None, :]).float()
    bf = (b1.detach().double() + w64 @ beta.detach().double()).float().contiguous()
    sw = _pow2_floor(2047.0 / float(wf.abs().max())) if dtype == torch.float16 else 1.0
    v = _split16((wf * sw).reshape(8, 16, 2, 4, 8), dtype, parts)                # [ct][j][c][g][e][parts]
    return v.permute(3, 1, 0, 2, 5, 4).contiguous().reshape(-1), bf, 1.0 / sw


def tiles_parts(tiles, bias):
    """1 or 2: whether a pack_tower_lp image holds hi only or (hi, lo)."""
    return tiles.numel() // ((2 + 10 * (bias.shape[0] - 1)) * 4 * 64 * 8)


def conv_tower_lp(tok, tiles, bias, inv, residual_mask, prec, count=None):
    """tokens [n, L] u8 -> tower output [n, L, P, 64] in 16-bit planes (hi, lo; value = hi + lo), the form the
    split-precision GRU consumes (HIP kernel svdd_conv_tower_lp)."""
    assert tok.is_cuda and tok.dtype == torch.uint8 and tok.is_contiguous()
    n, L = tok.shape
    out = torch.empty((n, L, tiles_parts(tiles, bias), 64), dtype=tiles.dtype, device=tok.device)
    rc = _lib.lib().svdd_conv_tower_lp(tok.data_ptr(), tiles.data_ptr(), bias.data_ptr(), inv.data_ptr(), out.data_ptr(),
                                       n, L, bias.shape[0] - 1, int(residual_mask), _ptr(count), prec,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_conv_tower_lp")
    return out


def conv_tower_windows_lp(cand, win, parent_out, tiles, bias, inv, residual_mask, prec, live_idx=None, count=None, out=None):
    """Tower output of the candidates cand [B, M, L] u8 on their row windows `win`, the rest copied from parent_out
    [B, L, 64] (HIP kernel svdd_conv_tower_windows_lp). live_idx / count (int32 device tensors): process only the listed
    candidates, compacted into the first `count` rows of the output."""
    B, M, L = cand.shape
    n = B * M
    if out is None:
        out = torch.empty((n, L, parent_out.shape[2], 64), dtype=tiles.dtype, device=cand.device)
    rc = _lib.lib().svdd_conv_tower_windows_lp(cand.data_ptr(), tiles.data_ptr(), bias.data_ptr(), inv.data_ptr(),
                                               win.data_ptr(), parent_out.data_ptr(), out.data_ptr(), n, L, M,
                                               bias.shape[0] - 1, int(residual_mask), _ptr(live_idx), _ptr(count), prec,
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_conv_tower_windows_lp")
    return out


def gru_bidir_lp(x_nlc, wpack, bpack, inv, prec, count=None, out=None):
    """x -> [2, n, L, 64] per-direction hidden states (fp32) on the 16-bit matrix cores (svdd_gru_bidir_lp).
    x: [n, L, 64] fp32, or the split-precision tower's output [n, L, P, 64] in 16-bit planes (hi, lo)."""
    assert x_nlc.is_cuda and x_nlc.is_contiguous() and x_nlc.shape[-1] == 64
    split = x_nlc.dim() == 4
    assert split or x_nlc.dtype == torch.float32
    n, L = x_nlc.shape[0], x_nlc.shape[1]
    if out is None:
        out = torch.empty((2, n, L, 64), dtype=torch.float32, device=x_nlc.device)
    rc = _lib.lib().svdd_gru_bidir_lp(None if split else x_nlc.data_ptr(), x_nlc.data_ptr() if split else None,
                                      wpack.data_ptr(), bpack.data_ptr(), inv.data_ptr(), out.data_ptr(),
                                      n, L, _ptr(count), prec, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_gru_bidir_lp")
    return out


def value_tail_lp(h, w1pack, b1f, w_eff, b_eff, inv, prec, count=None, out=None):
    """h [2, n, L, 64] -> scores [n, n_tasks] (svdd_value_tail_lp)."""
    _, n, L, _ = h.shape
    T = w_eff.shape[1]
    if out is None:
        out = torch.empty((n, T), dtype=torch.float32, device=h.device)
    assert out.is_contiguous() and out.shape == (n, T) and out.dtype == torch.float32
    rc = _lib.lib().svdd_value_tail_lp(h[0].data_ptr(), h[1].data_ptr(), w1pack.data_ptr(), b1f.data_ptr(),
                                       w_eff.data_ptr(), b_eff.data_ptr(), float(inv), out.data_ptr(), n, L, T,
                                       _ptr(count), prec, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_value_tail_lp")
    return out


TOWER_WINDOW_MARGIN = 27     # +-17 rows receptive field of the 5-layer tower + 10 rows of window-edge error


def candidate_windows(cand, x, margin=TOWER_WINDOW_MARGIN, flags=None):
    """cand [B, M, L] u8, x [B, L] u8 -> int32 [B*M, 2]: the 16-aligned row window (w0, w1) around the positions where a
    candidate differs from its parent, (0, 0) for an exact copy (HIP kernel svdd_candidate_windows)."""
    assert cand.is_cuda and cand.dtype == torch.uint8 and cand.is_contiguous() and x.dtype == torch.uint8 and x.is_contiguous()
    B, M, L = cand.shape
    win = torch.empty((B * M, 2), dtype=torch.int32, device=cand.device)
    rc = _lib.lib().svdd_candidate_windows(cand.data_ptr(), x.data_ptr(), B, L, M, margin, win.data_ptr(), _ptr(flags),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_candidate_windows")
    return win


def conv_tower_windows(onehot, win, parent_out, M, tiles, bias, residual_mask, live_idx=None, count=None, out=None):
    """Tower output [n, L, 64] of the candidates' one-hot [n = B*M, L, 4], computing only the row windows `win` and
    copying the rest from the parents' tower output [B, L, 64] (HIP kernel svdd_conv_tower_windows_f32).
    out: a [k, L, 64] buffer for a launch over PART of a compacted list — live_idx is then a slice of the list, count the
    device-side length of that part, and the launch takes at most k entries."""
    assert onehot.is_cuda and onehot.dtype == torch.float32 and onehot.is_contiguous() and onehot.shape[2] == 4
    n, L, _ = onehot.shape
    assert parent_out.is_contiguous() and parent_out.shape == (n // M, L, 64) and win.shape == (n, 2)
    if out is None:
        out = torch.empty((n, L, 64), dtype=torch.float32, device=onehot.device)
    else:
        assert live_idx is not None and count is not None and out.is_contiguous() and out.shape[1:] == (L, 64)
        n = out.shape[0]
    rc = _lib.lib().svdd_conv_tower_windows_f32(onehot.data_ptr(), tiles.data_ptr(), bias.data_ptr(), win.data_ptr(),
                                                parent_out.data_ptr(), out.data_ptr(), n, L, M, bias.shape[0] - 1,
                                                int(residual_mask), _ptr(live_idx), _ptr(count),
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_conv_tower_windows_f32")
    return out


def epilogue_ln(y, bias=None, f_prev=None, tb=None, gamma=None, beta=None, act=ACT_RELU_THEN_ADD, want_norm=True,
                want_sum=True):
    """Fused conv epilogue on channels-last rows (HIP kernel svdd_epilogue_ln_f32):
    f = relu(y + bias) + f_prev | relu(y + bias + f_prev) | y + bias + f_prev ; hn = LayerNorm(f + tb)*gamma + beta.
    y: any tensor whose memory is [rows, C] contiguous (e.g. a channels_last conv2d output). -> (f | None, hn | None),
    both with y's shape/strides."""
    C = bias.numel() if bias is not None else gamma.numel()
    rows = y.numel() // C
    f = torch.empty_like(y) if want_sum else None
    hn = torch.empty_like(y) if want_norm else None
    assert y.dtype == torch.float32 and (f if f is not None else hn).stride() == y.stride()
    rc = _lib.lib().svdd_epilogue_ln_f32(y.data_ptr(), _ptr(bias), _ptr(f_prev), _ptr(tb), _ptr(gamma), _ptr(beta),
                                         _ptr(f), _ptr(hn), rows, C, int(act),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "svdd_epilogue_ln_f32")
    return f, hn


class FusedValueNet(nn.Module):
    """`head(embedding(onehot))` of a ConvGRUTrunk + ConvHead pair in one module:
    forward(onehot fp32 [n, L, 4]) -> scores [n, n_tasks, 1]   (the reference call at diffusion_gosai.py:1208-1209).
    Also accepts the reward-model layout [n, 4, L] (Enformer.py:1422-1423)."""

    def __init__(self, embedding: ConvGRUTrunk, head: ConvHead):
        super().__init__()
        blocks = embedding.conv_tower.blocks
        self.in_channels = embedding.in_channels
        self.stem_w = nn.Parameter(_cl(blocks[0].conv.weight), requires_grad=False)
        self.stem_b = nn.Parameter(blocks[0].conv.bias.detach().clone(), requires_grad=False)
        self.stem_pad = blocks[0].conv.kernel_size[0] // 2
        ws, bs, wpacks, folded_ws, self.pads, self.residual = [], [], [], [], [], []
        for blk in blocks[1:]:
            w, b = blk.conv.weight.detach(), blk.conv.bias.detach()
            bn = blk.norm.layer
            if isinstance(bn, nn.BatchNorm1d):                       # fold eval-mode BN into the conv
                s = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
                w = w * s[:, None, None]
                b = (b - bn.running_mean) * s + bn.bias.detach()
            assert not blk.residual or isinstance(blk.channel_transform.layer, nn.Identity)
            folded_ws.append(w)
            ws.append(nn.Parameter(_cl(w), requires_grad=False))
            bs.append(nn.Parameter(b.clone(), requires_grad=False))
            wpacks.append(nn.Parameter(pack_conv(w) if tuple(w.shape) == (64, 64, 5) and blk.conv.dilation[0] == 1
                                       else torch.zeros(0), requires_grad=False))
            self.pads.append(blk.conv.kernel_size[0] // 2 * blk.conv.dilation[0])
            self.residual.append(blk.residual)
        self.ws, self.bs, self.wpacks = nn.ParameterList(ws), nn.ParameterList(bs), nn.ParameterList(wpacks)
        self.use_hip_conv = False
        # the whole tower as one LDS-resident kernel when it has the reference shape: stem 4->64 x 15, then 64->64 x 5
        stem = blocks[0].conv
        self.tower_ok = (tuple(stem.weight.shape) == (64, 4, 15) and 1 <= len(ws) <= 8 and
                         all(tuple(w.shape) == (64, 64, 5) for w in folded_ws) and
                         all(blk.conv.dilation[0] == 1 for blk in blocks[1:]))
        if self.tower_ok:
            self.tw_tiles = nn.Parameter(pack_tower(stem.weight, folded_ws), requires_grad=False)
            self.tw_bias = nn.Parameter(torch.stack([stem.bias.detach()] + [b.detach() for b in bs]).contiguous(),
                                        requires_grad=False)
            self.tw_resmask = sum(1 << k for k, r in enumerate(self.residual) if r)
        self.use_fused_tower = True
        gt = embedding.gru_tower
        wpack, bpack = pack_gru(gt.gru)
        self.register_buffer("wpack", wpack)
        self.register_buffer("bpack", bpack)
        # FFN (LayerNorm -> Linear 64->128 -> ReLU -> Linear 128->64, Enformer.py:2010-2047) and the head
        # (1x1 conv 64->n_tasks + mean over length, :2131-2173). The last two maps are linear with nothing in
        # between, so they collapse into one 128->n_tasks map; the mean over length commutes with it.
        d1, d2 = gt.ffn.dense1, gt.ffn.dense2
        self.ln_w = nn.Parameter(d1.norm.layer.weight.detach().clone(), requires_grad=False)
        self.ln_b = nn.Parameter(d1.norm.layer.bias.detach().clone(), requires_grad=False)
        self.w1 = nn.Parameter(d1.linear.weight.detach().clone(), requires_grad=False)          # [128, 64]
        self.b1 = nn.Parameter(d1.linear.bias.detach().clone(), requires_grad=False)
        hw = head.channel_transform.conv.layer
        wh, bh = hw.weight.detach()[:, :, 0].double(), hw.bias.detach().double()                  # [T,64], [T]
        w2, b2 = d2.linear.weight.detach().double(), d2.linear.bias.detach().double()            # [64,128], [64]
        self.w_eff = nn.Parameter((wh @ w2).float().t().contiguous(), requires_grad=False)       # [128, T]
        self.b_eff = nn.Parameter((wh @ b2 + bh).float(), requires_grad=False)                   # [T]
        self.tail_ok = tuple(self.w1.shape) == (128, 64) and self.w_eff.shape[1] <= 4
        if self.tail_ok:
            wp, bf = pack_tail(self.w1, self.b1, self.ln_w, self.ln_b)
            self.w1pack = nn.Parameter(wp, requires_grad=False)
            self.b1f = nn.Parameter(bf, requires_grad=False)
        self.use_fused_tail = True
        self.share_parent_tower = True
        self.split_gru_rounds = True         # candidate_scores_compact, late steps: the live candidates as two parts so that the GRU's second round
                                             # hides under the first part's tower (same bits; _windows_gru_tail_split)
        self.sort_live_by_window = True      # candidate_scores_compact: live candidates ordered by window size, largest first (A/B knob; same bits)
        # "f32" (exact, default) or one of LP_DTYPES: the conv tower, GRU and tail on the 16-bit matrix cores
        # (csrc/svdd_lp_*.hip). Needs the reference-shaped net (tower_ok, tail_ok).
        self.precision = "f32"
        self._gru_mod = (gt.gru,)                                   # tuple: keeps the module out of this one's registry
        self._stem_w_raw = stem.weight.detach()
        self._folded_ws = [w.detach() for w in folded_ws]
        self._lp = {}
        self._grad_packs = None
        self.gru_off_chain = True            # mean_score_input_grad: W_i x and W_i^T da as whole-chip launches beside the GRU's serial chains (A/B knob)
        self._ln_eps = d1.norm.layer.eps

    def lp_ok(self, L):
        return self.precision != "f32" and self.tower_ok and self.tail_ok and L <= 208

    def _lp_pack(self):
        """Operand images of the split-precision kernels for self.precision (packed on first use)."""
        pk = self._lp.get(self.precision)
        if pk is None:
            dev = self.tw_bias.device
            tiles, tinv = pack_tower_lp(self._stem_w_raw.to(dev), [w.to(dev) for w in self._folded_ws], self.precision)
            gw, gb, ginv = pack_gru_lp(self._gru_mod[0], self.precision)
            tw, tb1, tail_inv = pack_tail_lp(self.w1, self.b1, self.ln_w, self.ln_b, self.precision)
            pk = dict(prec=_lib.PRECISIONS[self.precision], tiles=tiles.to(dev), tinv=tinv.to(dev), gw=gw.to(dev),
                      gb=gb.to(dev), ginv=ginv.to(dev), tw=tw.to(dev), tb1=tb1.to(dev), tail_inv=tail_inv)
            self._lp[self.precision] = pk
        return pk

    def forward_tokens(self, tok, count=None, out=None):
        """Scores [n, n_tasks, 1] of the token rows tok [n, L] u8 (4 = MASK) through the hand-written kernels, in the
        module's precision. count: int32 device scalar = number of valid rows of a compacted batch (rows beyond it are
        neither computed nor defined). out: fp32 [n, n_tasks] buffer for the scores (a caller that runs parts of a batch on
        several streams owns the result buffer)."""
        from . import ops
        if self.precision != "f32":
            pk = self._lp_pack()
            seq = conv_tower_lp(tok.contiguous(), pk["tiles"], self.tw_bias, pk["tinv"], self.tw_resmask, pk["prec"], count)
            return self._after_tower_lp(seq, pk, count, out)
        seq = conv_tower(ops.transform_samples(tok.contiguous()), self.tw_tiles, self.tw_bias, self.tw_resmask, count)
        return self._after_tower(seq, tok.shape[0], tok.shape[1], count, out)

    def grad_ok(self, L):
        """True when forward_grad applies: the reference-shaped net in fp32 at a length the static 64 -> 64 conv kernel has."""
        return self.tower_ok and self.tail_ok and L in (200, 50) and all(p.numel() for p in self.wpacks)   # (fp32 whatever self.precision)

    def forward_grad(self, x):
        """Scores [n, n_tasks, 1] of a RELAXED input x [n, L, 4] (fp32, e.g. softmax probabilities) WITH autograd to x — the reward
        call of the DPS baseline (reference diffusion_gosai.py:1326-1329: reward_model(softmax(E[x0 | x_t]))) without MIOpen: the
        stem as unfold + matmul, every 64 -> 64 x 5 convolution on svdd_conv1d_cl_f32 in both directions (DilatedConvFunction;
        eval-mode BatchNorm folded), the GRU on the hand-written forward + BPTT kernels (GruBidirFunction), the element-wise ops
        and the tail left to torch autograd on channels-last rows. MIOpen served these small convolutions' backward passes as
        im2col + one GEMM per sample (2.7 ms of a 8.3 ms DPS step at B = 256). Weights frozen: input gradient only."""
        n, L, _ = x.shape
        if self._grad_packs is None:
            dev = self.tw_bias.device
            w_stem = self._stem_w_raw.to(dev).float().permute(2, 1, 0).reshape(-1, self._stem_w_raw.shape[0]).contiguous()   # [4 t + c][co]
            packs_t = [pack_conv(w.to(dev).float().flip(2).transpose(0, 1).contiguous()) for w in self._folded_ws]
            self._grad_packs = (w_stem, packs_t, pack_gru_bwd(self._gru_mod[0]).to(dev))
        w_stem, packs_t, gru_bwd = self._grad_packs
        T = self._stem_w_raw.shape[2]
        xp = F.pad(x, (0, 0, T // 2, T // 2))
        cols = torch.cat([xp[:, k:k + L] for k in range(T)], dim=2)                          # [n, L, 4 T], tap-major
        f = F.relu(cols @ w_stem + self.stem_b)
        for wp, wpt, b, res in zip(self.wpacks, packs_t, self.bs, self.residual):
            c = DilatedConvFunction.apply(f, wp, wpt, 64, 64, 5, 1) + b
            f = F.relu(c + f) if res else F.relu(c)
        y2 = GruBidirFunction.apply(f, self.wpack, self.bpack, gru_bwd)
        hn = F.layer_norm(y2[0] + y2[1], (64,), self.ln_w, self.ln_b, self._ln_eps)
        z = F.relu(F.linear(hn, self.w1, self.b1))
        return ((z @ self.w_eff).mean(dim=1) + self.b_eff)[:, :, None]

    def mean_score_input_grad(self, x):
        """d mean_n(score_n[task 0]) / d x for a RELAXED input x [n, L, 4] (fp32) — the gradient the DPS baseline needs of its reward
        call (reference diffusion_gosai.py:1326-1329: reward_model(softmax(E)).mean().backward()) — WITHOUT autograd: 16 launches of
        hand-written kernels (stem, 5 x conv + fused epilogue, GRU with saved gates | tail forward + backward in one pass, BPTT,
        direction sum + ReLU gate, 5 x transposed conv + gate, stem transpose). forward_grad is the autograd form of the same function
        (tests/test_fused_gpu.py compares the two). Only where grad_ok(L)."""
        x = x.contiguous().float()
        n, L, _ = x.shape
        lib, st = _lib.lib(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self._grad_packs is None:
            self.forward_grad(torch.zeros(1, L, 4, device=x.device))                 # packs the transposed weights
        w_stem, packs_t, gru_bwd = self._grad_packs
        dev = x.device
        f = torch.empty((n, L, 64), dtype=torch.float32, device=dev)
        _lib.check(lib.svdd_reward_stem_f32(x.data_ptr(), w_stem.data_ptr(), self.stem_b.data_ptr(), f.data_ptr(), n, L,
                                            self._stem_w_raw.shape[2], st), "svdd_reward_stem_f32")
        fs = [f]
        for wp, b, res in zip(self.wpacks, self.bs, self.residual):
            fs.append(conv1d_cl(fs[-1], wp, 64, 5, 1, bias=b, f_prev=fs[-1] if res else None, act=ACT_ADD_THEN_RELU))
        out = torch.empty((2, n, L, 64), dtype=torch.float32, device=dev)
        save = torch.empty((2, n, L, 4, 64), dtype=torch.float32, device=dev)
        gates = torch.empty((2, n * L, 192), dtype=torch.float32, device=dev)         # b + W_i x, then reused for the gate derivatives
        gout = torch.empty_like(out)
        w_eff0 = self.w_eff[:, 0].contiguous()
        dxg = torch.empty((2, n, L, 64), dtype=torch.float32, device=dev)             # work buffers of the tower's backward pass
        if self.gru_off_chain:
            # the non-recurrent halves (W_i x forwards, W_i^T da backwards) as whole-chip launches beside the serial chains
            _lib.check(lib.svdd_gru_bidir_train2_f32(fs[-1].data_ptr(), self.wpack.data_ptr(), self.bpack.data_ptr(), gates.data_ptr(),
                                                     out.data_ptr(), save.data_ptr(), n, L, st), "svdd_gru_bidir_train2_f32")
        else:
            _lib.check(lib.svdd_gru_bidir_train_f32(fs[-1].data_ptr(), self.wpack.data_ptr(), self.bpack.data_ptr(), out.data_ptr(),
                                                    save.data_ptr(), n, L, st), "svdd_gru_bidir_train_f32")
        _lib.check(lib.svdd_reward_tail_grad_f32(out[0].data_ptr(), out[1].data_ptr(), self.w1.data_ptr(), self.b1.data_ptr(),
                                                 self.ln_w.data_ptr(), self.ln_b.data_ptr(), w_eff0.data_ptr(), float(self._ln_eps), n, L,
                                                 gout[0].data_ptr(), gout[1].data_ptr(), st), "svdd_reward_tail_grad_f32")
        g = dxg[1]                                                                   # the gradient at the last layer's pre-activation
        if self.gru_off_chain:
            _lib.check(lib.svdd_gru_bidir_bwd2_f32(gout.data_ptr(), out.data_ptr(), save.data_ptr(), gru_bwd.data_ptr(), gates.data_ptr(),
                                                   fs[-1].data_ptr(), g.data_ptr(), n, L, st), "svdd_gru_bidir_bwd2_f32")
        else:
            dxd = torch.empty_like(out)
            _lib.check(lib.svdd_gru_bidir_bwd_f32(gout.data_ptr(), out.data_ptr(), save.data_ptr(), gru_bwd.data_ptr(), dxd.data_ptr(), n, L, st),
                       "svdd_gru_bidir_bwd_f32")
            _lib.check(lib.svdd_sum_gate_f32(dxd[0].data_ptr(), dxd[1].data_ptr(), fs[-1].data_ptr(), g.data_ptr(), n * L * 64, st),
                       "svdd_sum_gate_f32")
        bufs = [gout[1], dxg[0]]
        for k in range(len(packs_t) - 1, -1, -1):                                    # block k + 1: fs[k + 1] = relu(conv_k(fs[k]) + b (+ fs[k]))
            y = bufs[k & 1]
            _lib.check(lib.svdd_conv1d_cl_gated_f32(g.data_ptr(), packs_t[k].data_ptr(), y.data_ptr(), n, L, 64, 64, 5, 1,
                                                    g.data_ptr() if self.residual[k] else None, fs[k].data_ptr(), st),
                       "svdd_conv1d_cl_gated_f32")
            g = y
        dx = torch.empty((n, L, 4), dtype=torch.float32, device=dev)
        _lib.check(lib.svdd_reward_stem_bwd_f32(g.data_ptr(), w_stem.data_ptr(), dx.data_ptr(), n, L, self._stem_w_raw.shape[2], st),
                   "svdd_reward_stem_bwd_f32")
        return dx

    def kernels_ok(self, L):
        """True when the whole net runs on the hand-written kernels (tower, GRU, tail) for sequences of length L."""
        return self.use_fused_tower and self.tower_ok and self.use_fused_tail and self.tail_ok and L <= 208

    def candidate_scores_compact(self, onehot, cand, x, ws):
        """Exact work-skipping for SVDD-MC: the scores of the LIVE candidates only — a candidate
        that unmasked nothing is a copy of its parent and has the parent's score. Fills the workspace ws (flags,
        live_idx, slot, count: int32 device tensors) and returns the compacted scores [n, n_tasks] (first count[0] rows
        valid, in live_idx order). Same kernels, same bits per candidate as forward_candidates; no host round trip."""
        from . import ops
        B, M, L = cand.shape
        win = candidate_windows(cand, x, flags=ws.flags)
        # the live candidates by DESCENDING window size: a windowed-tower workgroup takes as many row tiles as its window has, and
        # with the long ones dispatched first the launch does not end on a few CUs that began a 13-tile window last (fp32 tower
        # 487 -> 395 us per launch on a C2 decode's states, tools/tower_order_probe.py). A row's result does not depend on its
        # place in the compacted batch: same bits, same tokens.
        split = self._gru_split(B * M) if (self.precision == "f32" and getattr(ws, "late", False)) else 0
        if split:
            ops.compact_by_key(ws.flags, ws.live_idx, ws.slot, ws.count3, split=split)
        else:
            (ops.compact_by_key if self.sort_live_by_window else ops.compact_flags)(ws.flags, ws.live_idx, ws.slot, ws.count)
        if getattr(ws, "n_win_rows", None) is not None:                     # Diffusion.skip_stats: rows the tower computes this step
            ws.n_win_rows += (win[:, 1] - win[:, 0]).sum()
        # The parents' tower output is carried from step to step: the next parent IS the selected candidate, whose tower
        # output this step computes (bit-identical to a full tower pass on it) — ws.advance_parent copies it over after
        # the select. Only the first step runs the tower on the parents.
        lp = self.precision != "f32"
        if ws.parent_out is None or ws.parent_out_lp != self.precision:
            # (the parents of the first step are B copies of the all-MASK row: one row through the tower, broadcast — a row's output
            #  does not depend on the batch around it; Diffusion.dedup_prior)
            xp = x[:1].contiguous() if getattr(ws, "prior_rows_identical", False) else x
            if lp:
                pk = self._lp_pack()
                po = conv_tower_lp(xp, pk["tiles"], self.tw_bias, pk["tinv"], self.tw_resmask, pk["prec"])
            else:
                po = conv_tower(ops.transform_samples(xp), self.tw_tiles, self.tw_bias, self.tw_resmask)
            ws.parent_out = po if xp is x else po.expand(B, *po.shape[1:]).contiguous()
            ws.prior_rows_identical = False
            ws.parent_out_lp = self.precision
        if lp:
            pk = self._lp_pack()
            ws.seq = conv_tower_windows_lp(cand, win, ws.parent_out, pk["tiles"], self.tw_bias, pk["tinv"], self.tw_resmask,
                                           pk["prec"], live_idx=ws.live_idx, count=ws.count)
            return self._after_tower_lp(ws.seq, pk, ws.count)[:, :, 0]
        if split:
            return self._windows_gru_tail_split(onehot, win, ws, B * M, L, M, split)
        ws.seq = conv_tower_windows(onehot, win, ws.parent_out, M, self.tw_tiles, self.tw_bias, self.tw_resmask,
                                    live_idx=ws.live_idx, count=ws.count)
        return self._after_tower(ws.seq, B * M, L, ws.count)[:, :, 0]

    GRU_ROUND_ROWS = None        # rows of ONE round of the GRU's (tile of 16 sequences, direction) units on this chip's CUs; None: from
                                 # the device (gru_round_rows: 8 x CUs = 2048 on MI355X's 256), an int: override (tests, A/B)

    def gru_round_rows(self):
        """One workgroup = 16 sequences in one direction, one workgroup per CU per round: a round covers CUs / 2 tiles = 8 x CUs rows."""
        if self.GRU_ROUND_ROWS is not None:
            return int(self.GRU_ROUND_ROWS)
        r = self.__dict__.get("_round_rows")
        if r is None:
            r = self.__dict__["_round_rows"] = 8 * _lib.device_info()[1]     # asked once per net (one process drives one GPU)
        return r

    def _gru_split(self, n):
        """Rows of the first part when n compacted candidates are to run as two parts (0: do not split): only where the second
        part is at most 5/16 of a round (n = B * M = 2560 at config 2 on 256 CUs: 2048 + 512)."""
        if not (self.split_gru_rounds and self.use_fused_tail and self.tail_ok) or torch.cuda.is_current_stream_capturing():
            return 0
        r = self.gru_round_rows()
        return r if r < n <= r + (5 * r) // 16 else 0

    def _windows_gru_tail_split(self, onehot, win, ws, n, L, M, split):
        """The late steps of a decode have more live candidates than ONE round of the GRU's (tile, direction) units on the chip's
        CUs (2048 rows on 256): the launch then takes two rounds, the second with 3/4 of the chip idle (0.67 instead of 0.36 ms).
        Here the compacted list runs as two parts: B = the entries from `split` on (the smallest windows, the list being sorted),
        A = the first `split`:   tower(B) -> [GRU(B), tail(B) on a side stream || tower(A)] -> GRU(A) -> tail(A).
        B's GRU hides under A's tower (a GRU workgroup and a tower workgroup share a CU: 33 + 57 KB of LDS, 108 + 104 VGPRs).
        Same kernels on the same rows: same bits per candidate (tools/gru_split_probe.py: 1270 -> 1060 us per late step)."""
        dev = onehot.device
        # persistent buffers + the side stream: kept on the NET (one set per (n, L, split, device)), not on the per-decode workspace —
        # every decode used to allocate ~400 MB and a new stream; decodes on one stream follow each other, so the set is never shared
        key = (n, L, split, str(dev))
        cache = self.__dict__.setdefault("_split_bufs", {})
        sb = cache.get(key)
        if sb is None:
            cache.clear()
            from . import ops
            sb = cache[key] = dict(n=n, L=L, side=ops.side_stream(dev, 0), ev_b=torch.cuda.Event(), ev_done=torch.cuda.Event(),
                                   seq=torch.empty((n, L, 64), device=dev), h_a=torch.empty((2, split, L, 64), device=dev),
                                   h_b=torch.empty((2, n - split, L, 64), device=dev), sc=torch.empty((n, self.w_eff.shape[1]), device=dev))
        ws.split_bufs = sb
        seq, sc, side = sb["seq"], sb["sc"], sb["side"]
        c_a, c_b = ws.count3[1:2], ws.count3[2:3]
        main = torch.cuda.current_stream()
        conv_tower_windows(onehot, win, ws.parent_out, M, self.tw_tiles, self.tw_bias, self.tw_resmask,
                           live_idx=ws.live_idx[split:], count=c_b, out=seq[split:])
        sb["ev_b"].record(main)
        with torch.cuda.stream(side):
            side.wait_event(sb["ev_b"])
            gru_bidir(seq[split:], self.wpack, self.bpack, c_b, out=sb["h_b"])
            value_tail(sb["h_b"], self.w1pack, self.b1f, self.w_eff, self.b_eff, c_b, out=sc[split:])
            sb["ev_done"].record(side)
        conv_tower_windows(onehot, win, ws.parent_out, M, self.tw_tiles, self.tw_bias, self.tw_resmask,
                           live_idx=ws.live_idx, count=c_a, out=seq[:split])
        gru_bidir(seq[:split], self.wpack, self.bpack, c_a, out=sb["h_a"])
        value_tail(sb["h_a"], self.w1pack, self.b1f, self.w_eff, self.b_eff, c_a, out=sc[:split])
        main.wait_event(sb["ev_done"])
        ws.seq = seq
        return sc[:, 0]

    def _after_tower_lp(self, seq, pk, count=None, out=None):
        h = gru_bidir_lp(seq, pk["gw"], pk["gb"], pk["ginv"], pk["prec"], count=count)
        return value_tail_lp(h, pk["tw"], pk["tb1"], self.w_eff, self.b_eff, pk["tail_inv"], pk["prec"], count=count, out=out)[:, :, None]

    def forward(self, x):
        if x.shape[1] == self.in_channels and x.shape[2] != self.in_channels:
            x = x.transpose(1, 2)                                   # reward-model layout [n,4,L] -> [n,L,4]
        n, L, C = x.shape
        if self.lp_ok(L) and x.is_cuda:
            # the 16-bit tower takes tokens: the engine's inputs are exact one-hot rows (MASK = zero row)
            tok = torch.where(x.sum(dim=2) == 0, 4, x.argmax(dim=2)).to(torch.uint8)
            return self.forward_tokens(tok)
        if self.use_fused_tower and self.tower_ok and L <= 208 and x.is_cuda:
            seq = conv_tower(x.contiguous(), self.tw_tiles, self.tw_bias, self.tw_resmask)
            return self._after_tower(seq, n, L)
        f = x.contiguous().view(n, 1, L, C).permute(0, 3, 1, 2)     # [n,4,1,L] view with channels_last strides
        f, _ = epilogue_ln(F.conv2d(f, self.stem_w, None, padding=(0, self.stem_pad)), self.stem_b, want_norm=False)
        # measured (tools/conv_microbench.py): at 64->64 x 5 taps MIOpen's igemm (216 us) still beats our kernel
        # (246 us), so the hand-written conv is opt-in for the tower
        hip_conv = self.use_hip_conv and L in (200, 50) and n * L >= 192 * 200
        for w, b, wp, pad, res in zip(self.ws, self.bs, self.wpacks, self.pads, self.residual):
            if hip_conv and wp.numel():                               # conv + bias + residual + ReLU in one kernel
                seq = f.permute(0, 2, 3, 1).reshape(n, L, 64)
                f = conv1d_cl(seq, wp, 64, 5, 1, bias=b, f_prev=seq if res else None, act=ACT_ADD_THEN_RELU)
                f = f.view(n, 1, L, 64).permute(0, 3, 1, 2)
            else:
                y = F.conv2d(f, w, None, padding=(0, pad))
                f, _ = epilogue_ln(y, b, f if res else None, act=ACT_ADD_THEN_RELU, want_norm=False)   # relu(conv + b + f)
        seq = f.permute(0, 2, 3, 1).reshape(n, L, f.shape[1])       # [n,L,64] — a view, memory is already NLC
        return self._after_tower(seq.contiguous(), n, L)

    def candidates_ok(self, L, M):
        """True when forward_candidates can share the parents' tower work (reference-shaped tower, one sequence per
        tile)."""
        return (self.use_fused_tower and self.tower_ok and self.share_parent_tower and M > 1 and 104 < L <= 208 and
                self.tw_bias.shape[0] == 6)

    def forward_candidates(self, onehot, cand, x):
        """Scores of the B*M candidates (onehot [B*M, L, 4], row b*M + m; cand [B, M, L] u8) of the parents x [B, L] u8.
        Same result as forward(onehot), bit for bit; the conv tower is evaluated once per parent and, per candidate,
        only on the row window around the positions it changed (svdd_conv_tower_windows_f32)."""
        from . import ops
        B, M, L = cand.shape
        if self.lp_ok(L):
            pk = self._lp_pack()
            parent_out = conv_tower_lp(x, pk["tiles"], self.tw_bias, pk["tinv"], self.tw_resmask, pk["prec"])
            win = candidate_windows(cand, x)
            seq = conv_tower_windows_lp(cand, win, parent_out, pk["tiles"], self.tw_bias, pk["tinv"], self.tw_resmask, pk["prec"])
            return self._after_tower_lp(seq, pk)
        parent_out = conv_tower(ops.transform_samples(x), self.tw_tiles, self.tw_bias, self.tw_resmask)
        win = candidate_windows(cand, x)
        seq = conv_tower_windows(onehot, win, parent_out, M, self.tw_tiles, self.tw_bias, self.tw_resmask)
        return self._after_tower(seq, B * M, L)

    def _after_tower(self, seq, n, L, count=None, out=None):
        h = gru_bidir(seq, self.wpack, self.bpack, count)
        if self.use_fused_tail and self.tail_ok:
            return value_tail(h, self.w1pack, self.b1f, self.w_eff, self.b_eff, count, out=out)[:, :, None]
        assert count is None and out is None, "compacted batches need the fused tail kernel"
        # LayerNorm(h_fwd + h_bwd) in one pass (the direction sum of Enformer.py:1617 + dense1.norm)
        _, hn = epilogue_ln(h[0], None, h[1], None, self.ln_w, self.ln_b, act=ACT_NONE, want_sum=False)
        z = F.relu(F.linear(hn, self.w1, self.b1))                  # [n,L,128]
        return ((z @ self.w_eff).mean(dim=1) + self.b_eff)[:, :, None]


class FusedBackbone(nn.Module):
    """CNNModel.forward for the sampler's zero-sigma case, channels-last. forward(tokens [B,L]) ->
    raw logits fp32 [B, L, 5] (contiguous, layout BLV)."""

    def __init__(self, cnn: CNNModel):
        super().__init__()
        self.H = cnn.args.hidden_dim
        self.first_w = nn.Parameter(_cl(cnn.linear.weight), requires_grad=False)
        self.first_b = nn.Parameter(cnn.linear.bias.detach().clone(), requires_grad=False)
        self.ws = nn.ParameterList([nn.Parameter(_cl(c.weight), requires_grad=False) for c in cnn.convs])
        # hand-written fp32-MFMA conv (csrc/svdd_nets.hip) for the 128->128, 9-tap layers
        self.wpacks = nn.ParameterList([nn.Parameter(pack_conv(c.weight), requires_grad=False) if self.H == 128 and
                                        c.kernel_size[0] == 9 else nn.Parameter(torch.zeros(0), requires_grad=False)
                                        for c in cnn.convs])
        self.use_hip_conv = True
        # In-kernel epilogue (bias/ReLU/residual + the next layer's LayerNorm, row-wise through LDS): saves one HBM
        # round trip of the activations but serialises a memory-bound tail behind the MFMA loop of a 1-workgroup-
        # per-CU kernel; measured 3.29 ms vs 2.93 ms per backbone forward with the separate 19-us epilogue kernel
        # (which runs at 5.5 TB/s chip-wide). Kept for testing, off by default.
        self.fuse_conv_epilogue = False
        self.bs = nn.ParameterList([nn.Parameter(c.bias.detach().clone(), requires_grad=False) for c in cnn.convs])
        self.dil = [c.dilation[0] for c in cnn.convs]
        self.norms = cnn.norms
        with torch.no_grad():                                       # time biases at sigma == 0 (diffusion_gosai.py:334-335)
            dev = cnn.linear.weight.device
            tb = cnn._time_biases(torch.zeros(1, device=dev))
        self.tb = nn.ParameterList([nn.Parameter(t.reshape(1, 1, 1, self.H).clone(), requires_grad=False) for t in tb])
        self.f1_w = nn.Parameter(_cl(cnn.final_conv[0].weight), requires_grad=False)
        self.f1_b = nn.Parameter(cnn.final_conv[0].bias.detach().clone(), requires_grad=False)
        self.f2_w = nn.Parameter(_cl(cnn.final_conv[2].weight), requires_grad=False)
        self.f2_b = nn.Parameter(cnn.final_conv[2].bias.detach().clone(), requires_grad=False)
        self.register_buffer("eye", torch.eye(cnn.alphabet_size), persistent=False)
        # the whole forward in ONE launch (svdd_backbone_cnn_f32): residual stream in registers, activations in LDS
        self.one_launch = self.H == 128 and cnn.alphabet_size == 5 and all(c.kernel_size[0] == 9 for c in cnn.convs)
        self.use_one_launch = True
        self.min_tiles_one_launch = 0
        self.precision = "f32"          # or one of LP_DTYPES: the one-launch kernel on the 16-bit matrix cores
        self._cnn = (cnn,)
        self._lp = {}
        self._grad_pack = None
        if self.one_launch:
            pk = pack_backbone(cnn)
            self.ol_dil = pk.pop("dil")
            for k, v in pk.items():
                self.register_buffer("ol_" + k, v, persistent=False)

    def ol_pack(self):
        return dict(table0=self.ol_table0, tiles=self.ol_tiles, vec=self.ol_vec, w2=self.ol_w2, dil=self.ol_dil)

    def grad_ok(self, L):
        """True when the differentiable pass (DPS) can run as one launch each way: the one-launch kernel, one sequence per tile."""
        return self.one_launch and self.use_one_launch and 104 < L <= 208      # (always the fp32 kernels, whatever self.precision)

    def grad_pack(self):
        """Operand images of svdd_backbone_cnn_grad_f32 (packed on first use)."""
        if self._grad_pack is None:
            self._grad_pack = {k: v.to(self.ol_tiles.device) for k, v in pack_backbone_grad(self._cnn[0]).items()}
        return self._grad_pack

    def forward_with_grad(self, x_onehot, tokens):
        """Raw logits [n, L, 5] with autograd to `x_onehot` [n, L, 5], which must be the hard one-hot of `tokens` [n, L] u8."""
        return BackboneOneLaunchFunction.apply(x_onehot, tokens.contiguous(), self.ol_pack(), self.grad_pack())

    def kernel_ok(self, L):
        """True when a forward of length-L sequences is the one-launch kernel (the work-skipping paths need it: they
        hand it compacted batches whose size only the device knows)."""
        return self.one_launch and self.use_one_launch and L <= 208 and self.min_tiles_one_launch == 0

    def forward_rows(self, tok, count=None, out=None, row_idx=None, scatter=False):
        """One-launch kernel on a compacted batch (see fused.backbone_cnn): tok [n, L] u8; count / row_idx int32 device
        tensors; out: logits buffer to write into."""
        if self.precision != "f32":
            pk = self._lp.get(self.precision)
            if pk is None:
                pk = self._lp[self.precision] = pack_backbone_lp(self._cnn[0], self.precision)
            return backbone_cnn_lp(tok, pk, count, out, row_idx, scatter)
        return backbone_cnn(tok, dict(table0=self.ol_table0, tiles=self.ol_tiles, vec=self.ol_vec, w2=self.ol_w2,
                                      dil=self.ol_dil), count, out, row_idx, scatter)

    def forward(self, seq, sigma=None):
        B, L = seq.shape
        if self.precision != "f32" and self.one_launch and L <= 208 and seq.is_cuda:
            pk = self._lp.get(self.precision)
            if pk is None:
                pk = self._lp[self.precision] = pack_backbone_lp(self._cnn[0], self.precision)
            tok = seq if seq.dtype == torch.uint8 else seq.to(torch.uint8)
            return backbone_cnn_lp(tok.contiguous(), pk)
        # One workgroup per tile of whole sequences, ~2.2 ms per workgroup whatever the batch. Below ~192 tiles the
        # layer-wise path (MIOpen + our conv kernels, 41 launches) finishes sooner (0.9 ms at B = 32), but it is a
        # different fp32 summation order and MIOpen's algorithm choice is not reproducible run to run, so by default
        # EVERY batch size takes the one-launch kernel: a row's logits are then the same bits whatever batch, tile or
        # compacted sub-batch it is evaluated in (the exact work-skipping paths rely on that). Raise
        # `min_tiles_one_launch` to 192 to trade that for small-batch latency.
        if (self.one_launch and self.use_one_launch and L <= 208 and seq.is_cuda and
                (B + 208 // L - 1) // (208 // L) >= self.min_tiles_one_launch):
            tok = seq if seq.dtype == torch.uint8 else seq.to(torch.uint8)
            return backbone_cnn(tok.contiguous(), dict(table0=self.ol_table0, tiles=self.ol_tiles, vec=self.ol_vec,
                                                       w2=self.ol_w2, dil=self.ol_dil))
        onehot = self.eye[seq.long()]                               # [B,L,5]
        f = onehot.view(B, 1, L, onehot.shape[2]).permute(0, 3, 1, 2)
        n = len(self.ws)
        # f_0 = relu(conv(onehot) + b) ; hn_0 = LN(f_0 + tb_0)                          (dnaconv.py:184,188-194)
        f, hn = epilogue_ln(F.conv2d(f, self.first_w, None, padding=(0, 4)), self.first_b, None,
                            self.tb[0], self.norms[0].weight, self.norms[0].bias)
        # one workgroup per 224-row tile of whole sequences: worth it once the tiles fill most of the 256 CUs
        hip_conv = self.use_hip_conv and self.H == 128 and L in (200, 50) and B * L >= 192 * 200
        for i, (w, b) in enumerate(zip(self.ws, self.bs)):
            d = self.dil[i]
            last = i + 1 == n
            if hip_conv and self.fuse_conv_epilogue and self.wpacks[i].numel():
                # f_{i+1} = relu(conv + b) + f_i inside the conv kernel; then hn_{i+1} = LN(f_{i+1} + tb_{i+1})
                xin = hn.permute(0, 2, 3, 1).reshape(B, L, self.H)
                fp = f.permute(0, 2, 3, 1).reshape(B, L, self.H)
                if last:
                    f = conv1d_cl(xin, self.wpacks[i], self.H, 9, d, bias=b, f_prev=fp, act=ACT_RELU_THEN_ADD)
                else:
                    f, hn = conv1d_cl(xin, self.wpacks[i], self.H, 9, d, bias=b, f_prev=fp, act=ACT_RELU_THEN_ADD,
                                      ln=(self.tb[i + 1], self.norms[i + 1].weight, self.norms[i + 1].bias))
                    hn = hn.view(B, 1, L, self.H).permute(0, 3, 1, 2)
                f = f.view(B, 1, L, self.H).permute(0, 3, 1, 2)
                continue
            if hip_conv and self.wpacks[i].numel():
                y = conv1d_cl(hn.permute(0, 2, 3, 1).reshape(B, L, self.H), self.wpacks[i], self.H, 9, d)
                y = y.view(B, 1, L, self.H).permute(0, 3, 1, 2)
            else:
                y = F.conv2d(hn, w, None, padding=(0, 4 * d), dilation=(1, d))
            # f_{i+1} = relu(y + b) + f_i ; hn_{i+1} = LN(f_{i+1} + tb_{i+1})           (:195-197, then :188-194)
            f, hn = epilogue_ln(y, b, f, None if last else self.tb[i + 1],
                                None if last else self.norms[i + 1].weight, None if last else self.norms[i + 1].bias,
                                want_norm=not last)
        f = F.conv2d(F.relu(F.conv2d(f, self.f1_w, self.f1_b)), self.f2_w, self.f2_b)   # [B,5,1,L] channels_last
        return f.permute(0, 2, 3, 1).reshape(B, L, f.shape[1])      # [B,L,5] contiguous view

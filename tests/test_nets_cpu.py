"""The PyTorch modules that stand in for the reference's nets (backbone, ConvGRU value/reward net)
reproduce the reference's outputs from the reference's own weights (fixture nets_tiny.npz, made
by tests/golden/make_golden.py). CPU, fp32; same ATen kernels => tight tolerance."""
import numpy as np
import torch

from svdd_amd.backbone import CNNModel
from svdd_amd.config import ModelConfig
from svdd_amd.value_nets import ConvGRUTrunk, ConvHead, RewardModel, load_reference_state_dict


def _sd(g, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix + ".")}


def tiny_nets(g):
    bb = CNNModel(ModelConfig(hidden_dim=16, num_cnn_stacks=1)).eval()
    bb.load_state_dict(_sd(g, "backbone"), strict=True)
    emb = ConvGRUTrunk(stem_in_channels=4, stem_channels=8, stem_kernel_size=15, n_conv=3, channel_init=8,
                       kernel_size=5, dropout=0.1).eval()
    load_reference_state_dict(emb, _sd(g, "embedding"))
    head = ConvHead(1, 8).eval()
    load_reference_state_dict(head, _sd(g, "head"))
    return bb, emb, head


def test_backbone_matches_reference(golden):
    g = golden("nets_tiny.npz")
    bb, _, _ = tiny_nets(g)
    x = torch.from_numpy(g["probe_x"])
    with torch.no_grad():
        out = bb(x, torch.zeros(x.shape[0]))
        out_u8 = bb(x.to(torch.uint8), None, zero_sigma=True)
    assert out.stride() == (5 * x.shape[1], 1, x.shape[1])      # [B,5,L] memory image, like the reference
    assert np.abs(out.numpy() - g["probe_logits"]).max() <= 1e-6
    assert torch.equal(out, out_u8)


def test_value_and_reward_match_reference(golden):
    g = golden("nets_tiny.npz")
    _, emb, head = tiny_nets(g)
    x = torch.from_numpy(g["probe_x"])
    oh = torch.nn.functional.one_hot(x * (x != 4), 4) * (x != 4)[..., None]
    with torch.no_grad():
        v = head(emb(oh.float()))
        r = RewardModel(emb, head)(oh.float().transpose(1, 2))
    assert v.shape == (x.shape[0], 1, 1)
    assert np.abs(v.numpy() - g["probe_value"]).max() <= 1e-6
    assert np.abs(r.numpy() - g["probe_reward"]).max() <= 1e-6


def test_fullsize_nets_equal_the_reference_classes_at_seed_44(golden):
    """g12_fullsize_probe.npz: the REFERENCE's CNNModel / ConvGRUTrunk / ConvHead at full size, random-initialised at
    torch.manual_seed(44) in synthetic.build's order. The mirrors must create the same parameters in the same order
    (every tensor's sum is compared) and compute the same functions."""
    from svdd_amd import synthetic
    g = golden("g12_fullsize_probe.npz")
    model, emb, head, _ = synthetic.build("dna", "cpu", seed=int(g["seed"]))
    for name, mod in (("backbone", model.backbone), ("embedding", emb), ("head", head)):
        sums = np.array([float(p.double().sum()) for p in mod.state_dict().values()])
        assert sums.shape == g[name + "_param_sums"].shape, name        # same state_dict keys (strict-load compatible)
        assert np.array_equal(sums, g[name + "_param_sums"]), name
    x = torch.from_numpy(g["x"].astype(np.int64))
    with torch.no_grad():
        logits = model.backbone(x, torch.zeros(4))
        oh = (torch.nn.functional.one_hot(x.clamp(max=3), 4) * (x != 4)[..., None]).float()
        value = head(emb(oh)).reshape(-1)
    assert np.abs(logits.numpy() - g["logits"]).max() <= 2e-6
    assert np.abs(value.numpy() - g["value"]).max() <= 1e-6


def test_flop_formulas():
    assert CNNModel.flops_per_position() == 5_943_808


def test_enformer_shaped_value_trunk():
    """Config-4 value function shape (decode.py:78-80): self-contained Enformer-shaped trunk + ConvHead; parity
    unpinned (enformer_pytorch is absent offline) — interface, sizes and the relative-shift indexing are checked."""
    from svdd_amd.enformer_value import EnformerTrunk, _relative_shift, exponential_linspace_int
    from svdd_amd.value_nets import ConvHead
    assert exponential_linspace_int(768, 1536, 6, 128) == [768, 896, 1024, 1152, 1280, 1536]
    L = 5
    r = torch.arange(2 * L - 1).float().repeat(L, 1)[None, None]
    want = torch.tensor([[(j - i) + (L - 1) for j in range(L)] for i in range(L)]).float()
    assert torch.equal(_relative_shift(r)[0, 0], want)            # logits indexed by (key - query) distance
    torch.manual_seed(0)
    trunk = EnformerTrunk(n_conv=4, channels=384, n_transformers=2, n_heads=2, key_len=16).eval()
    head = ConvHead(1, 768).eval()
    x = torch.zeros(3, 200, 4)
    x.scatter_(2, torch.randint(0, 4, (3, 200, 1)), 1.0)
    with torch.no_grad():
        y = trunk(x)
        s = head(y)
        s2 = head(trunk(x[1:2]))
    assert y.shape == (3, 768, 13) and s.shape == (3, 1, 1) and torch.isfinite(s).all()
    assert torch.allclose(s[1:2], s2, atol=1e-5)                   # rows independent (eval-mode BN)
    with torch.device("meta"):
        full = EnformerTrunk()
    n_params = sum(p.numel() for p in full.parameters())
    assert 225e6 < n_params < 235e6                                # SURVEY: ~230 M parameters


def test_dit_backbone_structure():
    """DiT backbone (dead code in the reference; pinned by g16 in the test below). Checks the reference's parameter names, the
    adaLN-zero init (zero logits at init, like the reference's zero-initialised final layer), and the attention
    block against an explicit softmax formulation."""
    from svdd_amd.config import dit_config
    from svdd_amd.diffusion import Diffusion
    from svdd_amd import dit as D
    torch.manual_seed(0)
    d = Diffusion(dit_config(length=50, hidden_size=64, cond_dim=32, n_blocks=2, n_heads=4, dropout=0.0)).eval()
    names = set(d.backbone.state_dict())
    for k in ("vocab_embed.embedding", "sigma_map.mlp.0.weight", "blocks.0.norm1.weight", "blocks.0.attn_qkv.weight",
              "blocks.0.attn_out.weight", "blocks.0.mlp.2.bias", "blocks.1.adaLN_modulation.weight",
              "output_layer.norm_final.weight", "output_layer.linear.weight", "output_layer.adaLN_modulation.bias"):
        assert k in names, k
    x = torch.randint(0, 5, (3, 50))
    with torch.no_grad():
        out = d.backbone(x, torch.zeros(3))
        assert out.shape == (3, 50, 5) and out.is_contiguous() and float(out.abs().max()) == 0.0
        for p in d.backbone.parameters():                      # un-zero the adaLN / output layers
            if float(p.abs().max()) == 0.0:
                p.normal_(0, 0.05)
        out = d.backbone(x, torch.zeros(3))
        # explicit evaluation of block 0's attention path
        blk = d.backbone.blocks[0]
        h0 = d.backbone.vocab_embed(x)
        c = torch.nn.functional.silu(d.backbone.sigma_map(torch.zeros(3)))
        sa, sc, ga, _, _, _ = blk.adaLN_modulation(c)[:, None].chunk(6, dim=2)
        h = blk.norm1(h0) * (1 + sc) + sa
        q, k, v = blk.attn_qkv(h).view(3, 50, 3, 4, 16).permute(2, 0, 3, 1, 4)
        q, k = D._rotary(q, k, d.backbone.rotary_emb.inv_freq)
        att = torch.softmax(q @ k.transpose(-1, -2) / 4.0, -1) @ v
        ref = h0 + ga * blk.attn_out(att.transpose(1, 2).reshape(3, 50, 64))
        mid = h0 + ga * blk.attn_out(torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(3, 50, 64))
    assert torch.allclose(ref, mid, atol=1e-5) and torch.isfinite(out).all() and float(out.abs().max()) > 0
    # rotary: position 0 is unrotated, norms are preserved
    qq = torch.randn(1, 2, 7, 16)
    r, _ = D._rotary(qq, qq, D.Rotary(16).inv_freq)
    assert torch.allclose(r[:, :, 0], qq[:, :, 0]) and torch.allclose(r.norm(dim=-1), qq.norm(dim=-1), atol=1e-5)


def test_dit_equals_reference_fixture():
    """g16: the reference's own models/dit.py (`/root/reference/models/dit.py:214-369`) run on CPU by make_golden.py, with
    flash_attn's two entry points (:115 rotary, :272 varlen attention) replaced by a plain matmul-softmax / rotate-half
    stand-in — pinned up to that stand-in. The reference state_dict must load with strict=True (parameter AND buffer names),
    and the logits must agree for non-zero sigma (time conditioning) and for the zero sigma the sampler feeds."""
    import os
    import numpy as np
    from svdd_amd import dit as D
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g16_dit.npz"))
    hs, cd, nb, nh, L = (int(v) for v in g["hp"])
    m = D.DIT(D.DiTModelConfig(hidden_size=hs, cond_dim=cd, n_blocks=nb, n_heads=nh, dropout=0.0, length=L), vocab_size=5).eval()
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("dit.")}
    m.load_state_dict(sd, strict=True)
    idx = torch.from_numpy(g["indices"])
    with torch.no_grad():
        out = m(idx, torch.from_numpy(g["sigma"]))
        out0 = m(idx, torch.zeros(idx.shape[0]))
    assert out.shape == g["logits"].shape
    assert float(np.abs(g["logits"]).max()) > 0.1
    assert float((out - torch.from_numpy(g["logits"])).abs().max()) <= 2e-5
    assert float((out0 - torch.from_numpy(g["logits_sigma0"])).abs().max()) <= 2e-5


def _g17_modules():
    """Builds this repo's EnformerTrunk + ConvHead at g17's size with the fixture's (seed-regenerated) reference weights,
    mapped from the reference's parameter names (`Enformer.py` wrappers keep their layer in `.layer`; its transformer tower
    keeps the blocks in `.blocks`; its FeedForwardBlock `:2010-2047` = dense1 (LayerNorm + Linear + ReLU) -> dense2 (Linear),
    `ffn.dense` is an unused leftover)."""
    import os
    import sys
    import numpy as np
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gdir)
    from seeded_weights import draw
    from svdd_amd.enformer_value import EnformerTrunk
    from svdd_amd.value_nets import ConvHead
    g = np.load(os.path.join(gdir, "g17_enformer_trunk.npz"))
    n_conv, ch, n_tf, heads, key_len = (int(v) for v in g["hp"])
    names = [str(n) for n in g["names"]]
    ref = draw(names, [[d for d in sh if d >= 0] for sh in g["shapes"]], int(g["seed"]))
    trunk = EnformerTrunk(n_conv=n_conv, channels=ch, n_transformers=n_tf, n_heads=heads, key_len=key_len).eval()
    head = ConvHead(1, 2 * ch).eval()

    from svdd_amd.enformer_value import load_reference_state_dict
    load_reference_state_dict(trunk, {k[6:]: v for k, v in ref.items() if k.startswith("trunk.")})
    sd_h = {k[5:]: v for k, v in ref.items() if k.startswith("head.")}
    head.load_state_dict(sd_h, strict=True)
    tok = torch.from_numpy(g["tokens"]).long()
    x = torch.nn.functional.one_hot(tok.clamp(max=3), 4).float() * (tok < 4)[..., None]
    return trunk, head, x, g


def test_enformer_trunk_equals_reference_wiring_fixture():
    """g17: the reference's own EnformerTrunk / ConvHead classes (`/root/reference/Enformer.py:1271-1334, 1807-2047, 2131-2173`)
    run on CPU by make_golden.py with enformer_pytorch's five symbols bound to this repo's restatements: pins the reference's
    wiring (conv tower channel schedule, NACDR order, residuals, attention pooling placement, transformer / feed-forward
    block, pointwise block, head) — "pinned up to the enformer_pytorch stand-in"; the attention math itself stays
    unpinned."""
    import numpy as np
    trunk, head, x, g = _g17_modules()
    with torch.no_grad():
        y = trunk(x)
        v = head(y)
    assert y.shape == g["trunk_out"].shape and float(np.abs(g["trunk_out"]).max()) > 0.05
    assert float((y - torch.from_numpy(g["trunk_out"])).abs().max()) <= 2e-5
    assert float((v - torch.from_numpy(g["value"])).abs().max()) <= 2e-5


def test_trunk_fp32_weight_packing_is_the_documented_fragment_order():
    """fused_trunk.pack_gemm_weight_f32 (the fp32-plane form of svdd_trunk_gemm, include/svdd_hip.h SVDD_OPT_TRUNK_PLANES_F32):
    [KB = c * T + t][N/128][8 n-tiles][2 pieces][64 lanes = 16 g + j][4 e] = W[128 nb + 16 nt + j][32 c + 8 g + 4 p + e][t] —
    what the kernel's LDS-DMA copies verbatim into MFMA fragment order. Checked element by element, k = 1 and k = 5."""
    from svdd_amd.fused_trunk import pack_gemm_weight_f32
    g = torch.Generator().manual_seed(0)
    for N, Cin, T in ((256, 64, 5), (128, 96, 1)):
        w = torch.randn(N, Cin, T, generator=g)
        pk = pack_gemm_weight_f32(w if T > 1 else w[:, :, 0]).view(Cin // 32, T, N // 128, 8, 2, 4, 16, 4)   # [c][t][nb][nt][p][g][j][e]
        assert pk.dtype == torch.float32 and pk.numel() == w.numel()
        idx = torch.randint(0, 10 ** 9, (200, 8), generator=g)
        for c, t, nb, nt, p, gg, j, e in idx.tolist():
            c, t, nb, nt, p, gg, j, e = c % (Cin // 32), t % T, nb % (N // 128), nt % 8, p % 2, gg % 4, j % 16, e % 4
            assert pk[c, t, nb, nt, p, gg, j, e] == w[128 * nb + 16 * nt + j, 32 * c + 8 * gg + 4 * p + e, t]

"""DiT masked-diffusion backbone (reference `models/dit.py:324-370`) as a PyTorch-ROCm module.

In the reference snapshot this backbone is dead code (`models/__init__.py:1` comments the import out and
it needs the CUDA-only `flash_attn`). It is pinned against the reference's own `models/dit.py` run on CPU with a
plain matmul-softmax / rotate-half STAND-IN for flash_attn's two entry points (`tests/golden/make_golden.py`
`install_flash_attn_standin`, fixture `g16_dit.npz`, `tests/test_nets_cpu.py::test_dit_equals_reference_fixture`):
**pinned up to that stand-in**. Same layer structure, parameter AND buffer names as the reference `DIT` (its
state_dict loads with strict=True), with
  * attention through `F.scaled_dot_product_attention` (ROCm's fused kernels; the GEMMs are where MFMA is used),
  * rotary embeddings applied to q and k in the non-interleaved ("rotate-half") convention of
    `flash_attn.layers.rotary.apply_rotary_emb_qkv_` (`models/dit.py:111-115`),
  * adaLN-zero conditioning on the (zeroed, when `time_conditioning=False`) noise level (`:214-290, 303-321`).
`forward(indices[B,L], sigma[B]) -> logits fp32 [B, L, vocab]` (contiguous: layout BLV for the sampler kernels).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F
from torch import nn


@dataclass
class DiTModelConfig:
    """configs_gosai/model/small.yaml keys the reference DIT reads."""
    hidden_size: int = 768
    cond_dim: int = 128
    n_blocks: int = 12
    n_heads: int = 12
    dropout: float = 0.1
    scale_by_sigma: bool = True
    length: int = 200


class LayerNormW(nn.Module):
    """LayerNorm without bias: F.layer_norm(x) * weight (reference models/dit.py:124-132)."""

    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.dim = dim

    def forward(self, x):
        return F.layer_norm(x.float(), [self.dim]) * self.weight


class TimestepEmbedder(nn.Module):
    """Sinusoidal features of sigma -> MLP (reference :148-189)."""

    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size))
        self.frequency_embedding_size = frequency_embedding_size

    def forward(self, t):
        half = self.frequency_embedding_size // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None]
        return self.mlp(torch.cat([torch.cos(args), torch.sin(args)], dim=-1))


class EmbeddingLayer(nn.Module):
    def __init__(self, dim, vocab_dim):
        super().__init__()
        self.embedding = nn.Parameter(torch.empty((vocab_dim, dim)))
        nn.init.kaiming_uniform_(self.embedding, a=math.sqrt(5))

    def forward(self, x):
        return self.embedding[x]


class Rotary(nn.Module):
    """Holds the reference's `rotary_emb.inv_freq` buffer (models/dit.py:81-85) so that checkpoints load strictly."""

    def __init__(self, dim, base=10000.0):
        super().__init__()
        self.register_buffer("inv_freq", 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim)))


def _rotary(q, k, inv_freq):
    """q, k: [B, H, L, D]. Non-interleaved rotary: (x1, x2) halves -> (x1 cos - x2 sin, x1 sin + x2 cos)."""
    d, length = q.shape[-1], q.shape[-2]
    ang = torch.arange(length, device=q.device, dtype=torch.float32)[:, None] * inv_freq[None].float()
    cos, sin = ang.cos().to(q.dtype), ang.sin().to(q.dtype)

    def rot(x):
        x1, x2 = x[..., : d // 2], x[..., d // 2:]
        return torch.cat([x1 * cos - x2 * sin, x1 * sin + x2 * cos], dim=-1)

    return rot(q), rot(k)


class DDiTBlock(nn.Module):
    def __init__(self, dim, n_heads, cond_dim, mlp_ratio=4, dropout=0.1):
        super().__init__()
        self.n_heads = n_heads
        self.norm1 = LayerNormW(dim)
        self.attn_qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.attn_out = nn.Linear(dim, dim, bias=False)
        self.norm2 = LayerNormW(dim)
        self.mlp = nn.Sequential(nn.Linear(dim, mlp_ratio * dim), nn.GELU(approximate="tanh"),
                                 nn.Linear(mlp_ratio * dim, dim))
        self.dropout = nn.Dropout(dropout)
        self.adaLN_modulation = nn.Linear(cond_dim, 6 * dim)
        nn.init.zeros_(self.adaLN_modulation.weight)
        nn.init.zeros_(self.adaLN_modulation.bias)

    def forward(self, x, c, inv_freq):
        B, L, D = x.shape
        shift_a, scale_a, gate_a, shift_m, scale_m, gate_m = self.adaLN_modulation(c)[:, None].chunk(6, dim=2)
        h = self.norm1(x) * (1 + scale_a) + shift_a
        q, k, v = self.attn_qkv(h).view(B, L, 3, self.n_heads, D // self.n_heads).permute(2, 0, 3, 1, 4)
        q, k = _rotary(q, k, inv_freq)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, L, D)
        x = x + gate_a * self.dropout(self.attn_out(a))
        h = self.norm2(x) * (1 + scale_m) + shift_m
        return x + gate_m * self.dropout(self.mlp(h))


class DDitFinalLayer(nn.Module):
    def __init__(self, hidden_size, out_channels, cond_dim):
        super().__init__()
        self.norm_final = LayerNormW(hidden_size)
        self.linear = nn.Linear(hidden_size, out_channels)
        self.adaLN_modulation = nn.Linear(cond_dim, 2 * hidden_size)
        for m in (self.linear, self.adaLN_modulation):
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)

    def forward(self, x, c):
        shift, scale = self.adaLN_modulation(c)[:, None].chunk(2, dim=2)
        return self.linear(self.norm_final(x) * (1 + scale) + shift)


class DIT(nn.Module):
    def __init__(self, model_config, vocab_size):
        super().__init__()
        m = model_config
        self.vocab_size = vocab_size
        self.vocab_embed = EmbeddingLayer(m.hidden_size, vocab_size)
        self.sigma_map = TimestepEmbedder(m.cond_dim)
        self.rotary_emb = Rotary(m.hidden_size // m.n_heads)
        self.blocks = nn.ModuleList([DDiTBlock(m.hidden_size, m.n_heads, m.cond_dim, dropout=m.dropout)
                                     for _ in range(m.n_blocks)])
        self.output_layer = DDitFinalLayer(m.hidden_size, vocab_size, m.cond_dim)
        self.scale_by_sigma = m.scale_by_sigma
        self.autocast_bf16 = False      # the reference runs the blocks under bf16 autocast (:364); fp32 here by default

    def forward(self, indices, sigma):
        x = self.vocab_embed(indices.long())
        c = F.silu(self.sigma_map(sigma))
        with torch.autocast(device_type=x.device.type, dtype=torch.bfloat16, enabled=self.autocast_bf16):
            for blk in self.blocks:
                x = blk(x, c, self.rotary_emb.inv_freq)
            x = self.output_layer(x, c)
        return x.float()

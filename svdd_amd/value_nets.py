"""Value-function / reward networks of the reference's decode configs, as PyTorch-ROCm modules.

To the SVDD hot path these are opaque callables:
    head(embedding(onehot fp32 [n,L,4]))  -> [n,1,1]         (SVDD-MC,  diffusion_gosai.py:1208-1209)
    reward_model(onehot fp32 [n,4,L])     -> [n,n_tasks,1]   (SVDD-PM / TDS, :1430, :1269)
Any `nn.Module` with those signatures works. This file provides the architecture the reference
instantiates for the RNA/“rna_saluki” tasks and that BASELINE.json's synthetic configs use:
`ConvGRUTrunk` + `ConvHead` (reference Enformer.py:32-49, 1337-1426, 1571-1751, 2131-2173).

Attribute paths AND creation order of every parameter equal the reference's — including the two modules the
reference registers but never calls (`Stem.norm`, Enformer.py:1790, and `FeedForwardBlock.dense`, :2031-2033, both
"saluki" leftovers) — so that a reference checkpoint loads with a plain strict `load_state_dict`, and a given
`torch.manual_seed` yields the very same random-init network as the reference's classes (the unused Linear consumes
RNG draws before the head is created; tests/golden/g12_fullsize_probe.npz pins this at full size).
"""
import torch
import torch.nn.functional as F
from torch import nn


class _Wrapped(nn.Module):
    """A single layer kept under `.layer`, matching the reference's Norm/ChannelTransform wrappers."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, x):
        return self.layer(x)


class Stem(nn.Module):
    """Conv1d('same') + ReLU on the one-hot input (reference Enformer.py:1754-1804)."""

    def __init__(self, in_channels, out_channels, kernel_size):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, padding="same")
        self.norm = _Wrapped(nn.LayerNorm(out_channels))     # registered, never applied (reference :1790,1801)

    def forward(self, x):
        return F.relu(self.conv(x))


class ConvBlock(nn.Module):
    """order "CDNRA": conv -> dropout -> batch-norm -> (+ input) -> ReLU (reference Enformer.py:2176-2292)."""

    def __init__(self, in_channels, out_channels, kernel_size, dilation=1, norm=True, residual=False, dropout=0.0):
        super().__init__()
        self.norm = _Wrapped(nn.BatchNorm1d(out_channels) if norm else nn.Identity())
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, padding="same", dilation=dilation)
        self.dropout = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        self.residual = residual
        if residual:
            self.channel_transform = _Wrapped(nn.Identity() if in_channels == out_channels
                                              else nn.Conv1d(in_channels, out_channels, 1, padding="same"))

    def forward(self, x):
        y = self.norm(self.dropout(self.conv(x)))
        if self.residual:
            y = y + self.channel_transform(x)
        return F.relu(y)


class ConvTower(nn.Module):
    def __init__(self, stem_in_channels, stem_channels, stem_kernel_size, n_blocks, channel_init, channel_mult,
                 kernel_size, norm, residual, dropout):
        super().__init__()
        self.blocks = nn.ModuleList([Stem(stem_in_channels, stem_channels, stem_kernel_size)])
        cin, cout = stem_channels, channel_init
        for _ in range(1, n_blocks):
            self.blocks.append(ConvBlock(cin, cout, kernel_size, norm=norm, residual=residual, dropout=dropout))
            cin, cout = cout, int(cout * channel_mult)
        self.out_channels = cin

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x


class LinearBlock(nn.Module):
    """(LayerNorm) -> Linear -> dropout -> (ReLU) (reference Enformer.py:2050-2099)."""

    def __init__(self, in_len, out_len, norm, act, dropout):
        super().__init__()
        self.norm = _Wrapped(nn.LayerNorm(in_len) if norm else nn.Identity())
        self.linear = nn.Linear(in_len, out_len)
        self.dropout = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        self.act = act

    def forward(self, x):
        y = self.dropout(self.linear(self.norm(x)))
        return F.relu(y) if self.act else y


class FeedForwardBlock(nn.Module):
    def __init__(self, in_len, dropout):
        super().__init__()
        self.dense1 = LinearBlock(in_len, in_len * 2, norm=True, act=True, dropout=dropout)
        self.dense2 = LinearBlock(in_len * 2, in_len, norm=False, act=False, dropout=dropout)
        self.dense = LinearBlock(in_len, in_len, norm=True, act=True, dropout=dropout)   # registered, never applied (:2031-2046)

    def forward(self, x):
        return self.dense2(self.dense1(x))


class GRUBlock(nn.Module):
    """Bidirectional GRU, forward+backward halves summed, then the feed-forward block
    (reference Enformer.py:1571-1630)."""

    def __init__(self, in_channels, n_layers=1, dropout=0.0):
        super().__init__()
        self.gru = nn.GRU(input_size=in_channels, hidden_size=in_channels, bidirectional=True, batch_first=True,
                          num_layers=n_layers, dropout=dropout if n_layers > 1 else 0)
        self.ffn = FeedForwardBlock(in_channels, dropout)

    _hip_gru = None                            # set by Diffusion.compute_gradient_DPS: x [n, L, C] -> [2, n, L, C] with a gradient

    def forward(self, x):                     # [n, C, L]
        if self._hip_gru is not None and x.is_cuda:
            y2 = self._hip_gru(x.transpose(1, 2))          # hand-written forward + BPTT kernels (csrc/svdd_gru_train.hip)
            y = y2[0] + y2[1]
        else:
            y = self.gru(x.transpose(1, 2))[0]    # [n, L, 2C]
            h = self.gru.hidden_size
            y = y[:, :, :h] + y[:, :, h:]
        return self.ffn(y).transpose(1, 2)


class ConvGRUTrunk(nn.Module):
    """Conv tower + bidirectional GRU (reference Enformer.py:1337-1426). Defaults are the values the
    reference hard-codes for the RNA value function (Enformer.py:32-49)."""

    def __init__(self, stem_in_channels=4, stem_channels=64, stem_kernel_size=15, n_conv=6, channel_init=64,
                 channel_mult=1, kernel_size=5, conv_norm=True, residual=True, n_gru=1, dropout=0.1):
        super().__init__()
        self.conv_tower = ConvTower(stem_in_channels, stem_channels, stem_kernel_size, n_conv, channel_init,
                                    channel_mult, kernel_size, conv_norm, residual, dropout)
        self.gru_tower = GRUBlock(self.conv_tower.out_channels, n_layers=n_gru, dropout=dropout)
        self.in_channels = stem_in_channels

    def forward(self, x):
        if x.shape[1] != self.in_channels:    # accepts [n,L,4] (value-fn call) or [n,4,L] (reward call), :1422-1423
            x = x.transpose(1, 2)
        return self.gru_tower(self.conv_tower(x))

    @staticmethod
    def flops_per_position(stem_in=4, C=64, stem_k=15, n_conv=6, k=5):
        conv = stem_in * C * stem_k + (n_conv - 1) * C * C * k
        gru = 2 * 3 * (C * C + C * C)          # 2 directions x 3 gates x (W_ih + W_hh)
        ffn = C * 2 * C + 2 * C * C
        return 2 * (conv + gru + ffn + C)       # + 1x1 head


class ConvHead(nn.Module):
    """1x1 conv to n_tasks channels, then adaptive average pool over length -> [n, n_tasks, 1]
    (reference Enformer.py:2131-2173 with act_func=None, norm=False, pool_func='avg')."""

    def __init__(self, n_tasks=1, in_channels=64):
        super().__init__()
        self.channel_transform = nn.Module()
        self.channel_transform.conv = _Wrapped(nn.Conv1d(in_channels, n_tasks, kernel_size=1, padding="same"))

    def forward(self, x):
        return F.adaptive_avg_pool1d(self.channel_transform.conv(x), 1)


class RewardModel(nn.Module):
    """`reward_model(x[n,4,L]) -> [n,n_tasks,1]`: trunk + head (reference OriBaseModel, Enformer.py:1105-1127)."""

    def __init__(self, embedding, head):
        super().__init__()
        self.embedding = embedding
        self.head = head

    def forward(self, x):
        return self.head(self.embedding(x))


def load_reference_state_dict(module, state_dict):
    """Loads a reference checkpoint (strict: every key of the reference's module exists here, used or not)."""
    return module.load_state_dict(state_dict, strict=True)

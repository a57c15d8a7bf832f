"""Random-init networks of the shapes BASELINE.json's configs name (there are no checkpoints
offline): `torch.manual_seed(44)` — the reference CLI's default seed, decode.py:181 — then the
backbone and value/reward nets with PyTorch default init (SURVEY.md §8d "Synthetic inputs")."""
import torch

from .config import dna_config, rna_config
from .diffusion import Diffusion
from .value_nets import ConvGRUTrunk, ConvHead, RewardModel


def build(task="dna", device="cuda", seed=44, hidden_dim=128, num_cnn_stacks=4, value_channels=64, n_conv=6,
          value="convgru", enformer_kwargs=None):
    """-> (Diffusion, embedding, head, reward_model) in eval mode on `device`.

    task "dna": L=200 (configs_gosai) ; "rna": L=50 (configs_gosai_rna). The value function is the
    ConvGRU trunk + ConvHead the reference builds at Enformer.py:32-49; the reward model is a second,
    independently initialised net of the same shape (stand-in for the gReLU oracle, Enformer.py:103-131)."""
    torch.manual_seed(seed)
    cfg = (dna_config if task == "dna" else rna_config)(hidden_dim=hidden_dim, num_cnn_stacks=num_cnn_stacks)
    model = Diffusion(cfg)
    if value == "enformer":      # BASELINE config 4: decode.py:78-80 EnformerTrunk(7 conv, 1536 ch, 11 transformers) + ConvHead(1, 3072)
        from .enformer_value import EnformerTrunk
        kw = dict(n_conv=7, channels=1536, n_transformers=11, n_heads=8, key_len=64)
        kw.update(enformer_kwargs or {})
        embedding = EnformerTrunk(**kw)
        head = ConvHead(1, 2 * kw["channels"])
    else:
        embedding = ConvGRUTrunk(stem_in_channels=4, stem_channels=value_channels, stem_kernel_size=15, n_conv=n_conv,
                                 channel_init=value_channels, kernel_size=5, dropout=0.1)
        head = ConvHead(1, value_channels)
    reward = RewardModel(ConvGRUTrunk(stem_in_channels=4, stem_channels=value_channels, stem_kernel_size=15,
                                      n_conv=n_conv, channel_init=value_channels, kernel_size=5, dropout=0.1),
                         ConvHead(1, value_channels))
    for m in (model, embedding, head, reward):
        m.to(device).eval()
        for p in m.parameters():
            p.requires_grad_(False)
    return model, embedding, head, reward

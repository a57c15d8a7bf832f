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


def test_flop_formulas():
    assert CNNModel.flops_per_position() == 5_943_808

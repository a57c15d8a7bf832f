"""Deterministic weights for fixtures whose state_dict would be megabytes (g17): every tensor is re-drawn from a seeded CPU
torch.Generator in the recorded (name, shape) order, so the fixture stores names + shapes + outputs only and the test
regenerates bit-identical weights (same torch build on the GPU box). Used by make_golden.py (writer) and the tests (reader)."""
import math

import torch


def draw(names, shapes, seed):
    g = torch.Generator().manual_seed(int(seed))
    out = {}
    for name, shape in zip(names, shapes):
        shape = tuple(int(v) for v in shape)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros(shape, dtype=torch.int64)
        elif name.endswith("running_var"):
            out[name] = torch.rand(shape, generator=g) + 0.5
        elif len(shape) >= 2:
            fan_in = 1
            for v in shape[1:]:
                fan_in *= v
            out[name] = torch.randn(shape, generator=g) / math.sqrt(max(fan_in, 1))
        elif name.endswith("weight"):                       # norm gains
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:                                               # biases, running means
            out[name] = 0.1 * torch.randn(shape, generator=g)
    return out

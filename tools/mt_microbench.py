"""K8 (svdd_mt19937_uniform_f32: torch's CPU mt19937 stream on the device) at the config-2 step size, and the replay-mode
decode against the Philox decode. Usage: python tools/mt_microbench.py [--decodes 2]"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from svdd_amd import _lib, ops, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--decodes", type=int, default=2)
args = ap.parse_args()
DEV = "cuda:0"
torch.manual_seed(0)
wp = ops.mt_state_from_torch(torch.get_rng_state())
state = torch.from_numpy(wp.astype(np.uint32).view(np.int32).copy()).to(DEV)
for n in (2_560_000, 256_000, 5_120_000):
    out = torch.empty(n, device=DEV)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        _lib.lib().svdd_mt19937_uniform_f32(state.data_ptr(), out.data_ptr(), n, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.lib().svdd_mt19937_uniform_f32(state.data_ptr(), out.data_ptr(), n, s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"mt19937 n={n}: {ms * 1e3:.1f} us = {ms * 1e6 / n:.3f} ns/output")

model, emb, head, _ = synthetic.build("dna", DEV)
B, L, M, S = 256, 200, 10, 128
res = {}
for mode, how in (("philox", None), ("replay", "device"), ("replay", "host")):
    model.rng_mode = mode
    if how:
        model.replay_rng = how
    torch.manual_seed(0)
    with torch.no_grad():
        model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.decodes):
            x0 = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
        torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.decodes
    res[(mode, how)] = el
    print(f"C2 decode rng={mode}{'/' + how if how else ''}: {el * 1e3:.1f} ms = {B / el:.1f} seq/s")
# the double-buffered side-stream generation against the host replay at the full config-2 size: same tokens, same generator state after
same = []
for seed in range(4):
    outs = []
    for how in ("device", "host"):
        model.rng_mode, model.replay_rng = "replay", how
        torch.manual_seed(100 + seed)
        with torch.no_grad():
            x = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
        torch.cuda.synchronize()
        outs.append((x.cpu(), torch.rand(4)))
    same.append(bool(torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])))
print("device replay == host replay on 4 seeded C2 decodes (tokens and generator state):", same)
print(f"replay(device) / philox = {res[('replay', 'device')] / res[('philox', None)]:.3f} ; replay(host) / philox = {res[('replay', 'host')] / res[('philox', None)]:.3f}")

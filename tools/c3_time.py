"""C3 (BASELINE configs[2]: RNA SVDD-PM, B=256, L=50, M=10, 128 steps) wall clock + the backbone's share, fp32 and f16x3.
Usage: python tools/c3_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, synthetic
model, _, _, rew = synthetic.build("rna", "cuda:0")
model.rng_mode, model.philox_seed = "philox", 0
run = lambda: model.controlled_sample_tweedie(rew, num_steps=128, eval_sp_size=256, sample_M=10, options="True")
for prec in ("f32", "f16x3"):
    model.precision = prec
    run(); torch.cuda.synchronize()
    t = time.perf_counter(); run(); run(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 2
    _lib.profile_enable(True); run(); torch.cuda.synchronize(); _lib.profile_enable(False)
    bb = _lib.profile_collect(6)
    for k in (0, 1, 3, 5, 7): _lib.profile_collect(k)
    print(f"{prec}: {dt * 1e3:.1f} ms/decode = {256 / dt:.1f} seq/s ; backbone {bb[0]:.1f} ms in {bb[1]} launches")

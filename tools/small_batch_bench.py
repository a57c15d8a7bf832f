"""Small batches: one backbone forward and a whole SVDD-MC decode at B = 4 / 32 / 64 / 128 with the
backbone on one workgroup per sequence (SVDD_OPT_BACKBONE_SPLIT = 1) against the automatic split (4 workgroups per
sequence up to 64 sequences, 2 up to 128). Usage: python tools/small_batch_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, synthetic

DEV = "cuda:0"
model, emb, head, _ = synthetic.build("dna", DEV)
model.rng_mode, model.philox_seed = "philox", 0
lib = _lib.lib()


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for B, M in ((4, 2), (4, 10), (32, 10), (64, 10), (128, 10), (256, 10)):
    x = torch.randint(0, 5, (B, 200), device=DEV, dtype=torch.uint8)
    res = {}
    for name, opt in (("one workgroup / sequence", 1), ("split (auto)", 0)):
        _lib.set_option(7, opt)
        with torch.no_grad():
            fwd = timed(lambda: model._backbone_logits(x), 20)
            dec = timed(lambda: model.controlled_sample(emb, head, num_steps=128, eval_sp_size=B, sample_M=M), 2)
        res[name] = (fwd, dec)
        print(f"B={B:4d} M={M:2d} {name:26s}: backbone forward {fwd:6.3f} ms   decode {dec:7.1f} ms = {B / dec * 1e3:7.1f} seq/s")
    a, b = res["one workgroup / sequence"], res["split (auto)"]
    print(f"          speed-up: forward x{a[0] / b[0]:.2f}  decode x{a[1] / b[1]:.2f}")
_lib.set_option(7, 0)

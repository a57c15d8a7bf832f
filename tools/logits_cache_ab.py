"""A/B: the per-row logits cache at L = 200 (one workgroup per sequence: skipping the ~17 % unchanged rows frees CUs but a workgroup
still takes its 2.1 ms — unless the freed power raises the clock). Usage: python tools/logits_cache_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, synthetic
model, emb, head, _ = synthetic.build("dna", "cuda:0")
model.rng_mode, model.philox_seed = "philox", 0
run = lambda: model.controlled_sample(emb, head, num_steps=128, eval_sp_size=256, sample_M=10)
for rep in range(2):
    for mode in ("auto", "on"):
        model.logits_cache = mode
        run(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(4):
            x = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 4
        _lib.profile_enable(True); run(); torch.cuda.synchronize(); _lib.profile_enable(False)
        bb = _lib.profile_collect(6)
        for k in (0, 1, 3, 5, 7): _lib.profile_collect(k)
        print(f"logits_cache={mode}: {dt * 1e3:.1f} ms/decode = {256 / dt:.1f} seq/s ; backbone {bb[0]:.1f} ms in {bb[1]} launches ({bb[0] / bb[1] * 1e3:.1f} us each)")

"""A/B on one box: the C2 decode with the late steps' live candidates as two parts (FusedValueNet.split_gru_rounds) vs one part.
Usage: python tools/gru_split_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic
model, emb, head, _ = synthetic.build("dna", "cuda:0")
model.rng_mode, model.philox_seed = "philox", 0
fn = model.value_callable(emb, head)
run = lambda: model.controlled_sample(emb, head, num_steps=128, eval_sp_size=256, sample_M=10)
for rep in range(3):
    for on in (False, True):
        fn.split_gru_rounds = on
        run(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print(f"split_gru_rounds={on}: {dt * 1e3:.2f} ms/decode = {256 / dt:.1f} seq/s")

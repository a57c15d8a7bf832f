"""C2 decode: when do the late steps run as two parts? A/B of Diffusion.late_steps_from: "auto" (round 6: the last step's live count, read
back asynchronously, within 3 % of one GRU round), the fixed 0.8 of rounds 4-5, 0.0 (always) and split off. Same tokens required.
Usage: python tools/late_steps_ab.py"""
import hashlib
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic
model, emb, head, _ = synthetic.build("dna", "cuda:0")
model.rng_mode, model.philox_seed = "philox", 0
fn = model.value_callable(emb, head)
run = lambda: model.controlled_sample(emb, head, num_steps=128, eval_sp_size=256, sample_M=10)   # noqa: E731
dig = set()
for rep in range(2):
    for name, on, frm in (("auto", True, "auto"), ("0.8", True, 0.8), ("always", True, 0.0), ("off", False, 0.8)):
        fn.split_gru_rounds, model.late_steps_from = on, frm
        run(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            x = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        dig.add(hashlib.sha1(x.to(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16])
        print(f"late_steps_from = {name:7s}: {dt * 1e3:.1f} ms/decode = {256 / dt:.1f} seq/s", flush=True)
assert len(dig) == 1, dig
print("same tokens in every form")

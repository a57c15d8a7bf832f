"""C3 (BASELINE configs[2]: RNA SVDD-PM, B=256, L=50, M=10, 128 steps): the skipping loop's live candidates as ONE part per step against
the two-part step of round 6 (Diffusion.pm_two_part: whole backbone rounds, then the remainder with the first part's x0-hat + reward
net on a side stream). Same tokens required; wall clock per decode (mean of 3 after a warm-up) and the backbone's summed launch time.
Usage: python tools/c3_two_part_ab.py [precision ...]"""
import hashlib
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, synthetic
model, _, _, rew = synthetic.build("rna", "cuda:0")
model.rng_mode, model.philox_seed = "philox", 0
run = lambda: model.controlled_sample_tweedie(rew, num_steps=128, eval_sp_size=256, sample_M=10, options="True")   # noqa: E731
for prec in (sys.argv[1:] or ["f32", "f16x3"]):
    model.precision = prec
    dig = {}
    for two in (False, True, False, True):
        model.pm_two_part = two
        run(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            x = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        _lib.profile_enable(True); run(); torch.cuda.synchronize(); _lib.profile_enable(False)
        pr = {k: _lib.profile_collect(k) for k in (0, 1, 3, 5, 6, 7)}
        dig[two] = hashlib.sha1(x.to(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]
        print(f"{prec} two_part={two}: {dt * 1e3:.1f} ms/decode = {256 / dt:.1f} seq/s ; backbone {pr[6][0]:.1f} ms / {pr[6][1]} launches ; "
              f"tower {pr[5][0]:.1f} gru {pr[3][0]:.1f} tail {pr[7][0]:.1f} ms ; x0 {dig[two]}", flush=True)
    assert dig[True] == dig[False], dig
print("same tokens in both forms")

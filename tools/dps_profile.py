"""DPS baseline (BASELINE configs[4], reference decode_DPS.py / diffusion_gosai.py:1286-1330) wall-clock per step at a shard size.
Usage: python tools/dps_profile.py [B] [S] [--single-forward]   (run under rocprofv3 --kernel-trace --stats for the kernel split)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(pos[0]) if len(pos) > 0 else 256
S = int(pos[1]) if len(pos) > 1 else 8
dna, emb, head, rew = synthetic.build("dna", "cuda:0")
dna.rng_mode = "philox"
dna.dps_single_forward = "--single-forward" in sys.argv   # opt-in: q_xs from the differentiable pass (one backbone forward per step)
fn = lambda: dna.controlled_sample_DPS(rew, 10.0, num_steps=S, eval_sp_size=B)   # noqa: E731
fn(); torch.cuda.synchronize()
t = time.perf_counter()
out = fn()
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"DPS{' (single forward)' if dna.dps_single_forward else ''} B={B} L=200 S={S}: {dt * 1e3:.1f} ms/decode = {dt / S * 1e3:.2f} ms/step ; at S=128: {B / (dt / S * 128):.1f} seq/s")

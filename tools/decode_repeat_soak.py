"""Whole decodes repeated with the same Philox key must give the same tokens every time, in every precision mode (a race in any of
the one-launch kernels shows up as a run that differs). Usage: python tools/decode_repeat_soak.py [repeats] [modes]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32", "f16x3", "bf16x3", "bf16"]
dev = "cuda:0"
model, emb, head, _ = synthetic.build("dna", dev)
model.rng_mode = "philox"                              # keyed by (seed, row, step): the same key must give the same decode
for mode in modes:
    model.precision = mode
    digests = set()
    for i in range(n):
        model.philox_seed = 12345
        x = model.controlled_sample(emb, head, eval_sp_size=256, sample_M=10)
        digests.add(hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:16])
    print(f"{mode}: {n} decodes of B = 256, L = 200, M = 10, 128 steps -> {len(digests)} distinct result(s) {sorted(digests)}")
model.precision = "f32"

"""The no-autograd DPS decode (round 6) repeated with the same Philox key must give the same tokens every time (a race in any of its 22
launches per step would show as a changing digest), and the C3 decode on the short-tile backbone bodies likewise.
Usage: python tools/dps_repeat_soak.py [repeats]"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dig = lambda x: hashlib.sha1(x.to(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]   # noqa: E731
model, emb, head, rew = synthetic.build("dna", "cuda:0")
model.rng_mode, model.philox_seed = "philox", 7
d = {dig(model.controlled_sample_DPS(rew, 10.0, num_steps=128, eval_sp_size=256)) for _ in range(R)}
print("DPS B=256 128 steps:", sorted(d)); assert len(d) == 1
rna, _, _, rrew = synthetic.build("rna", "cuda:0")
rna.rng_mode, rna.philox_seed = "philox", 7
d = {dig(rna.controlled_sample_tweedie(rrew, num_steps=128, eval_sp_size=256, sample_M=10, options="True")) for _ in range(R)}
print("C3 SVDD-PM B=256 L=50:", sorted(d)); assert len(d) == 1
for B in (37, 300, 1100):                       # ragged batches of short sequences: mixed tile plans, 4- and 2-slot bodies
    d = {dig(rna.decode_sample(num_steps=16, eval_sp_size=B)) for _ in range(R)}
    print(f"un-guided L=50 B={B}:", sorted(d)); assert len(d) == 1
print("ok")

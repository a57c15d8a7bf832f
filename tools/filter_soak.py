"""Soak of K1's exact-arithmetic filter (svdd_amd/csrc/svdd_kernels.hip): >= 1e9 categorical draws A/B'd between the
filtered fast path and the forced-exact path on the GPU, on logits of several scales, both layouts, masked fractions and
move chances taken from the real 128-step schedule. Prints one JSON record (mismatches must be 0) with the rate of
draws the filter hands to the exact path. Usage: python tools/filter_soak.py [target_draws=1.05e9] > profiles/rNN_filter_soak.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, noise_schedule, ops
from svdd_amd.config import dna_config

target = float(sys.argv[1]) if len(sys.argv) > 1 else 1.05e9
dev = "cuda:0"
B, L, M = 8192, 200, 20
tab = noise_schedule.move_chance_table(noise_schedule.get_noise(dna_config()), 128, 1e-5)[0].numpy()
stats = torch.zeros(2, dtype=torch.int64, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
cand = [torch.empty(B, M, L, dtype=torch.uint8, device=dev) for _ in range(2)]
onehot = [torch.empty(B * M, L, 4, device=dev) for _ in range(2)]
draws = mismatched_tokens = mismatched_launches = launches = 0
per_scale = {}
t0 = time.time()
while draws < target:
    scale = (0.3, 1.0, 3.0, 10.0, 25.0)[launches % 5]
    step = (launches * 37) % 128
    frac_masked = max(0.02, 1.0 - step / 128.0)
    if launches % 2 == 0:
        logits = torch.randn(B, L, 5, device=dev, generator=gen) * scale
    else:
        logits = (torch.randn(B, 5, L, device=dev, generator=gen) * scale).transpose(1, 2)        # BVL view
    if launches % 7 == 3:
        logits[:64, :, :4] = 0.25                                                                   # equal q: near-ties between categories
    x = torch.where(torch.rand(B, L, device=dev, generator=gen) < frac_masked, 4,
                    torch.randint(0, 4, (B, L), device=dev, generator=gen)).to(torch.uint8)
    rng = ops.Rng(seed=0xC0FFEE + launches, step=step, row_offset=launches * B)
    for k, force in enumerate((False, True)):
        _lib.set_force_exact(force)
        if not force:
            _lib.check(_lib.lib().svdd_k1_stats(stats.data_ptr()), "svdd_k1_stats")
        ops.propose(logits, x, float(tab[step, 2]), float(tab[step, 1]), M, rng, cand=cand[k], onehot=onehot[k])
        _lib.check(_lib.lib().svdd_k1_stats(None), "svdd_k1_stats")
    _lib.set_force_exact(False)
    bad = int((cand[0] != cand[1]).sum()) + int((onehot[0] != onehot[1]).sum())
    mismatched_tokens += bad
    mismatched_launches += bad > 0
    launches += 1
    prev = draws
    draws, ex = int(stats[0]), int(stats[1])
    d = per_scale.setdefault(scale, [0, 0, 0])
    d[0] += draws - prev; d[1] += ex - sum(v[1] for v in per_scale.values()); d[2] += 1
torch.cuda.synchronize()
masked_draws, exact_draws = int(stats[0]), int(stats[1])
print(json.dumps({"what": "K1 filter A/B soak: filtered fast path vs forced-exact path, Philox, B=8192 L=200 M=20 per launch",
                  "launches": launches, "draws_at_masked_positions": masked_draws, "draws_sent_to_exact_path": exact_draws,
                  "ambiguity_rate": exact_draws / max(masked_draws, 1), "mismatched_tokens": mismatched_tokens,
                  "mismatched_launches": mismatched_launches, "seconds": round(time.time() - t0, 1),
                  "per_logit_scale": {str(k): {"draws": v[0], "to_exact_path": v[1], "rate": v[1] / max(v[0], 1), "launches": v[2]}
                                      for k, v in sorted(per_scale.items())},
                  "note": "scale 25 drives |max logit| past the fast path's 60 guard on purpose (those positions go to the exact path wholesale)",
                  "layouts": ["BLV", "BVL"]}))
assert mismatched_tokens == 0

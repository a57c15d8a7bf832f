"""Row-tile fraction the windowed tower computes at every step of a real SVDD-MC decode (config 2, synthetic nets)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic, fused
dev = "cuda:0"
model, emb, head, _ = synthetic.build("dna", dev)
model.rng_mode, model.philox_seed = "philox", 0
stats = []
orig = fused.candidate_windows
def spy(cand, x, margin=fused.TOWER_WINDOW_MARGIN, flags=None):
    win = orig(cand, x, margin, flags=flags)
    stats.append((float(((win[:, 1] - win[:, 0]) // 16).float().mean()) / 13, float((x == 4).float().mean()),
                  float((cand != x[:, None, :]).float().sum(2).mean())))
    return win
fused.candidate_windows = spy
model.controlled_sample(emb, head, num_steps=128, eval_sp_size=256, sample_M=10)
for i in range(0, 128, 8):
    f, m, c = stats[i]
    print(f"step {i:3d}: masked {m:5.1%}  changes/candidate {c:5.2f}  row tiles computed {f:5.1%}")
print("mean row-tile fraction over the decode: %.1f%%" % (100 * sum(s[0] for s in stats) / len(stats)))

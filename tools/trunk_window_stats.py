"""How many rows of the shared conv-tower levels a real config-4 decode computes: SVDD-MC, B = 256, M = 20, L = 200,
128 steps, Enformer-shaped value trunk, bf16x3 — compact rows per shared level and live candidates, summed over the decode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic
from svdd_amd.fused_trunk import FusedEnformerValueNet

dev = "cuda:0"
B, M, L, S = 256, 20, 200, 128
model, emb, head, _ = synthetic.build("dna", dev, value="enformer")
model.rng_mode, model.philox_seed, model.precision = "philox", 5, (sys.argv[1] if len(sys.argv) > 1 else "bf16x3")   # precision: bf16x3 (default) | f32 | bf16
fn = model.value_callable(emb, head)
assert isinstance(fn, FusedEnformerValueNet)
acc = {"rows": None, "live": torch.zeros(1, dtype=torch.int64, device=dev), "prow": None, "calls": 0}
orig = fn.forward_tokens


def counted(tok, count=None, shared=None):
    out = orig(tok, count=count, shared=shared)
    if shared is not None and fn.last_window_rows is not None:
        r = fn.last_window_rows.long()
        acc["rows"] = r if acc["rows"] is None else acc["rows"] + r
        acc["live"] += count.long()
        if fn.last_parent_rows is not None:
            p = fn.last_parent_rows.long()
            acc["prow"] = p if acc["prow"] is None else acc["prow"] + p
        acc["calls"] += 1
    return out


fn.forward_tokens = counted
with torch.no_grad():
    x = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
torch.cuda.synchronize()
live = int(acc["live"])
lens = [200, 100, 50, 25]
print(f"C4 decode, {acc['calls']} steps: live candidates {live} of {B * M * S} ({live / (B * M * S):.1%})")
for d, r in enumerate(acc["rows"].tolist()):
    print(f"  level {d}: compact rows {r} = {r / (live * lens[d]):.1%} of the live candidates' rows (+ 4 context rows per window from level 1 on)")
if acc["prow"] is not None:
    for d, r in enumerate(acc["prow"].tolist()):
        print(f"  parents, level {d}: compact rows {r} = {r / (B * (acc['calls'] - 1) * lens[d]):.1%} of the parents' rows")

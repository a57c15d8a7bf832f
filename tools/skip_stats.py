"""How much of an SVDD decode is exactly redundant (SURVEY.md section 7 "Exact work-skipping"): per diffusion step, the
fraction of candidates that are copies of their parent x_t (nothing unmasked: diffusion_gosai.py:1203 leaves the row
untouched), the fraction of rows whose selected candidate is such a copy (x_{t-1} == x_t, so the next backbone forward
would reproduce the same logits because time_conditioning is False, :334-335), and duplicate candidates of one parent.
Usage: python tools/skip_stats.py [mc|pm] [B] [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import ops, synthetic

mode = sys.argv[1] if len(sys.argv) > 1 else "mc"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
M = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = "cuda:0"
task = "dna" if mode == "mc" else "rna"
model, emb, head, reward = synthetic.build(task, dev)
model.rng_mode, model.philox_seed = "philox", 0
L, S = model.config.model.length, 128
sched = model._schedule(S, 1e-5)[0]
x = torch.full((B, L), 4, dtype=torch.uint8, device=dev)
rows = []
with torch.no_grad():
    for i in range(S):
        logits = model._backbone_logits(x)
        cand, onehot, _ = ops.propose(logits, x, sched[i, 2], sched[i, 1], M, model._rng(i, M, B, L, logits))
        if mode == "mc":
            scores = model._value_scores(emb, head, onehot, B, M, cand, x)
        else:
            scores = model._tweedie_scores(cand, reward, "True", "dna")
        xn = model._select(scores, cand, i)
        same = (cand == x[:, None, :]).all(dim=2)                       # [B, M] candidate == parent
        dup = 0
        c = cand.cpu()
        for b in range(B):
            dup += M - len({bytes(c[b, m].numpy()) for m in range(M)} | {bytes(x[b].cpu().numpy())}) + 1
        rows.append((float((x == 4).float().mean()), float(same.float().mean()), float((xn == x).all(dim=1).float().mean()),
                     dup / (B * M), float((cand != x[:, None, :]).float().sum(2).mean())))
        x = xn
print(f"{mode} B={B} L={L} M={M} S={S}")
print("step  masked  cand==parent  row unchanged  cand dup-of-(parent|earlier cand)  changes/cand")
for i in list(range(0, S, 8)) + [S - 1]:
    m, s, r, d, c = rows[i]
    print(f"{i:4d}  {m:6.1%}  {s:11.1%}  {r:12.1%}  {d:12.1%}  {c:6.2f}")
n = len(rows)
print("mean over the decode: cand==parent %.1f%%  row unchanged %.1f%%  redundant candidates %.1f%%" %
      (100 * sum(r[1] for r in rows) / n, 100 * sum(r[2] for r in rows) / n, 100 * sum(r[3] for r in rows) / n))
live = [B * M * (1 - r[3]) for r in rows]
print("value-net rows per step after dedup: mean %.0f  max %.0f  steps with <= 2048 rows: %d / %d" %
      (sum(live) / n, max(live), sum(1 for v in live if v <= 2048), n))

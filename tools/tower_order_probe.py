"""Does the ORDER in which the live candidates' windows are handed to the windowed tower matter? The workgroups of
svdd_conv_tower_windows_* take 4 .. 13 row tiles each, two share a CU, and the dispatcher hands them out in grid order: a long
window that starts last leaves the rest of the chip idle. On the states of a real C2 decode (Philox, seed 0): the windowed tower
(fp32 and f16x3) on the compacted live candidates in index order (what the decode does) against the same list sorted by window
size, largest first. Same outputs per candidate either way (a row's result does not depend on its place in the batch).
Usage (GPU box): python tools/tower_order_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import fused, ops, synthetic

dev = "cuda:0"
B, M, L, S = 256, 10, 200, 128
model, emb, head, _ = synthetic.build("dna", dev)
model.rng_mode, model.philox_seed = "philox", 0
model.state_trace = []
model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
torch.cuda.synchronize()
states, model.state_trace = model.state_trace, None
fv = model.value_callable(emb, head)
sched = model._schedule(S, 1e-5)[0]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


i32 = dict(dtype=torch.int32, device=dev)
tot = {}
print("step  live  tiles/cand |  f32 index  f32 sorted | f16x3 index f16x3 sorted   (us per launch)")
for i in (4, 16, 32, 48, 64, 80, 96, 112, 124):
    x = states[i]
    logits = model._backbone_logits(x)
    cand = torch.empty((B, M, L), dtype=torch.uint8, device=dev)
    onehot = torch.empty((B * M, L, 4), device=dev)
    ops.propose(logits, x, sched[i, 2], sched[i, 1], M, ops.Rng(seed=0, row_offset=0, step=i), cand=cand, onehot=onehot)
    flags, live_idx, slot = (torch.empty(B * M, **i32) for _ in range(3))
    count = torch.zeros(1, **i32)
    win = fused.candidate_windows(cand, x, flags=flags)
    ops.compact_flags(flags, live_idx, slot, count)
    k = int(count)
    nt = ((win[:, 1] - win[:, 0]) // 16)
    order = torch.argsort(nt[live_idx[:k].long()], descending=True, stable=True)
    sorted_idx = live_idx.clone()
    sorted_idx[:k] = live_idx[:k][order]
    parent = fused.conv_tower(ops.transform_samples(x), fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
    pk = None
    row = [i, k, float(nt[live_idx[:k].long()].float().mean())]
    ref = fused.conv_tower_windows(onehot, win, parent, M, fv.tw_tiles, fv.tw_bias, fv.tw_resmask, live_idx=live_idx, count=count)
    alt = fused.conv_tower_windows(onehot, win, parent, M, fv.tw_tiles, fv.tw_bias, fv.tw_resmask, live_idx=sorted_idx, count=count)
    assert torch.equal(ref[:k][order], alt[:k])
    for idx in (live_idx, sorted_idx):
        row.append(timeit(lambda: fused.conv_tower_windows(onehot, win, parent, M, fv.tw_tiles, fv.tw_bias, fv.tw_resmask,
                                                           live_idx=idx, count=count)))
    fv.precision = "f16x3"
    pk = fv._lp_pack()
    parent_lp = fused.conv_tower_lp(x, pk["tiles"], fv.tw_bias, pk["tinv"], fv.tw_resmask, pk["prec"])
    for idx in (live_idx, sorted_idx):
        row.append(timeit(lambda: fused.conv_tower_windows_lp(cand, win, parent_lp, pk["tiles"], fv.tw_bias, pk["tinv"], fv.tw_resmask,
                                                              pk["prec"], live_idx=idx, count=count)))
    fv.precision = "f32"
    print("%4d %5d %8.2f    | %9.1f %10.1f  | %10.1f %11.1f" % tuple(row))
    for j, key in enumerate(("f32 index", "f32 sorted", "f16x3 index", "f16x3 sorted")):
        tot[key] = tot.get(key, 0.0) + row[3 + j]
print("sum over the sampled steps:", {k: round(v, 1) for k, v in tot.items()})

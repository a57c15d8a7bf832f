"""Does decoding the two halves of a batch on two HIP streams pay? (Rows are independent and Philox is keyed by the global
row, so the halves are exactly the whole.) fp32 kernels saturate the matrix pipes and gain nothing (round 1);
the split-precision kernels are latency-bound, so one half's value net can run under the other half's backbone."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import ops, synthetic

dev = "cuda:0"
precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
nsplit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model, emb, head, _ = synthetic.build("dna", dev)
model.rng_mode, model.philox_seed, model.precision = "philox", 5, precision
B, L, M, S = 256, 200, 10, 128
sched = model._schedule(S, 1e-5)[0]
fn = model.value_callable(emb, head)
from svdd_amd import _lib
_lib.set_option(7, 1)        # halves of 128 rows would each take the small-batch backbone (2 workgroups per sequence that wait for each other): not with two launches in flight


def half_decode(rows, row0):
    """generator: one diffusion step per next(); finally yields the decoded tokens"""
    x = torch.full((rows, L), 4, dtype=torch.uint8, device=dev)
    cand = torch.empty((rows, M, L), dtype=torch.uint8, device=dev)
    onehot = torch.empty((rows * M, L, 4), dtype=torch.float32, device=dev)
    ws = model._SkipWorkspace(rows, M, dev)
    ws.parent_score.copy_(fn.forward_tokens(x).reshape(rows))
    for i in range(S):
        model.row_offset = row0
        logits = model._backbone_logits(x)
        ops.propose(logits, x, sched[i, 2], sched[i, 1], M, model._rng(i, M, rows, L, logits), cand=cand, onehot=onehot)
        sc = fn.candidate_scores_compact(onehot, cand, x, ws).reshape(-1)
        x = model._select_compact(sc, ws, cand, i)
        yield None
    model.row_offset = row0
    yield model._noise_removal(x)


def split_decode():
    streams = [ops.side_stream(dev, k) for k in range(nsplit)]      # probed: on hardware queues of their own (round 6)
    rows = B // nsplit
    gens = []
    for k, st in enumerate(streams):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            gens.append(half_decode(rows, k * rows))
    out = [None] * nsplit
    for _ in range(S + 1):
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                out[k] = next(gens[k])
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    model.row_offset = 0
    return torch.cat(out)


with torch.no_grad():
    ref = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
    got = split_decode()
    torch.cuda.synchronize()
    print("tokens equal to the single-stream decode:", bool(torch.equal(ref, got)))
    for name, f in (("one stream", lambda: model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)),
                    (f"{nsplit} streams x {B // nsplit} rows", split_decode)):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{precision} {name:28s} {dt * 1e3:8.1f} ms/decode  {B / dt:8.1f} seq/s")

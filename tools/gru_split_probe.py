"""Late steps of a C2 decode have more than 2048 live candidates: the GRU's (tile, direction) units then exceed the 256 CUs and the
launch takes two rounds (0.67 instead of 0.36 ms) with 3/4 of the chip idle in the second. Probe: split the compacted candidates
into A = the first 2040 (255 units) and B = the rest, and run  tower(B) -> [GRU(B) on a side stream || tower(A)] -> GRU(A) -> tails,
so that B's GRU hides under A's tower. Same kernels, same outputs per candidate. Usage: python tools/gru_split_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib, fused, ops, synthetic

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
lib = _lib.lib()
side = torch.cuda.Stream()
i32 = dict(dtype=torch.int32, device=dev)
SPLIT = 2040


def tower(onehot, win, parent, live_idx, count, out, nlaunch):
    rc = lib.svdd_conv_tower_windows_f32(onehot.data_ptr(), fv.tw_tiles.data_ptr(), fv.tw_bias.data_ptr(), win.data_ptr(), parent.data_ptr(),
                                         out.data_ptr(), nlaunch, L, M, 5, int(fv.tw_resmask), live_idx.data_ptr(), count.data_ptr(),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


print("step  live |  tower+GRU+tail now | split pipeline   (us)")
for i in (100, 108, 112, 116, 120, 124, 127):
    x = states[i]
    logits = model._backbone_logits(x)
    cand = torch.empty((B, M, L), dtype=torch.uint8, device=dev)
    onehot = torch.empty((B * M, L, 4), device=dev)
    ops.propose(logits, x, sched[i, 2], sched[i, 1], M, ops.Rng(seed=0, row_offset=0, step=i), cand=cand, onehot=onehot)
    flags, live_idx, slot = (torch.empty(B * M, **i32) for _ in range(3))
    count = torch.zeros(1, **i32)
    win = fused.candidate_windows(cand, x, flags=flags)
    ops.compact_by_key(flags, live_idx, slot, count)
    k = int(count)
    parent = fused.conv_tower(ops.transform_samples(x), fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
    n = B * M
    seq = torch.empty((n, L, 64), device=dev)

    def now():
        tower(onehot, win, parent, live_idx, count, seq, n)
        h = fused.gru_bidir(seq, fv.wpack, fv.bpack, count)
        return fused.value_tail(h, fv.w1pack, fv.b1f, fv.w_eff, fv.b_eff, count)
    ref = now()[:k].clone()
    if k <= SPLIT:
        print("%4d %5d | %12.1f | (no second round)" % (i, k, timeit(now)))
        continue
    cA = torch.tensor([SPLIT], **i32)
    cB = torch.tensor([k - SPLIT], **i32)
    nB = n - SPLIT
    hA = torch.empty((2, SPLIT, L, 64), device=dev)
    hB = torch.empty((2, nB, L, 64), device=dev)
    ev1, ev2 = torch.cuda.Event(), torch.cuda.Event()

    def split():
        main = torch.cuda.current_stream()
        tower(onehot, win, parent, live_idx[SPLIT:], cB, seq[SPLIT:], nB)
        ev1.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev1)
            fused.gru_bidir(seq[SPLIT:], fv.wpack, fv.bpack, cB, out=hB)
            sB = fused.value_tail(hB, fv.w1pack, fv.b1f, fv.w_eff, fv.b_eff, cB)
            ev2.record(side)
        tower(onehot, win, parent, live_idx, cA, seq, SPLIT)
        fused.gru_bidir(seq[:SPLIT], fv.wpack, fv.bpack, cA, out=hA)
        sA = fused.value_tail(hA, fv.w1pack, fv.b1f, fv.w_eff, fv.b_eff, cA)
        main.wait_event(ev2)
        return sA, sB
    sA, sB = split()
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([sA[:SPLIT], sB[:k - SPLIT]]), ref), "split pipeline changed a score"
    print("%4d %5d | %12.1f | %12.1f" % (i, k, timeit(now), timeit(split)))

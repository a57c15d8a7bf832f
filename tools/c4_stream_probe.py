"""Does the C4 decode's time depend on WHICH streams of torch's pool the trunk's two chains run on? (round 6: bench.py's C4 leg measured
66 seq/s inside the full default run and 81 alone.) K pool streams are taken before the trunk takes its two; with the selection rule of
svdd_amd.ops.side_stream off probe of svdd_amd.ops.side_stream off (SVDD_SIDE_STREAM_PROBE=0) every fourth K puts one chain on the hardware queue of the decode's own
stream: 65 instead of 80 seq/s. With the probe (default: a candidate stream is kept only if a marker on it is not held up by a sleeping
kernel on the current stream or on a stream already kept) every K gives good streams.
Usage: python tools/c4_stream_probe.py [K ...]      (run it twice: SVDD_SIDE_STREAM_PROBE=0 and default)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import ops, synthetic
dev = "cuda:0"
model, emb, head, _ = synthetic.build("dna", dev, value="enformer")
model.rng_mode, model.philox_seed, model.precision = "philox", 0, "bf16x3"
keep = []
run = lambda S: model.controlled_sample(emb, head, num_steps=S, eval_sp_size=256, sample_M=20)   # noqa: E731
run(8); torch.cuda.synchronize()
print("SVDD_SIDE_STREAM_PROBE =", os.environ.get("SVDD_SIDE_STREAM_PROBE", "1 (default)"))
for K in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6, 7, 8]:
    while len(keep) < K:
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            torch.zeros(1, device=dev)
        keep.append(s)
    ops._SIDE_STREAMS.clear()                 # the engine's side streams are taken from torch's pool AFTER the K dummies
    run(8); torch.cuda.synchronize()
    t = time.perf_counter(); run(128); torch.cuda.synchronize()
    el = time.perf_counter() - t
    idx = [s_.stream_id >> 5 for s_ in ops._SIDE_STREAMS.get(0, [])]
    print(f"{K} pool streams taken before the engine's: {256 / el:.1f} seq/s   (engine side streams: pool indices {idx})", flush=True)

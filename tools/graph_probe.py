"""Can a whole SVDD-MC decode be captured into ONE HIP graph, and does it pay? (SURVEY.md section 8f.1 "HIP-graph the 128-step
loop".) Philox mode + exact work-skipping: no host round trip inside the loop, every launch goes to torch's current stream.
Prints eager vs replay wall time per decode at a few batch sizes and checks that the replay reproduces the eager tokens."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic

dev = "cuda:0"
precision = sys.argv[1] if len(sys.argv) > 1 else "f32"
model, emb, head, _ = synthetic.build("dna", dev)
model.rng_mode, model.philox_seed, model.precision = "philox", 5, precision
M, S = 10, 128
for B in (4, 32, 256):
    run = lambda: model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)   # noqa: E731
    ref = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ref = run()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 3
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = run()
        g.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        t0 = time.perf_counter()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 3
        print(f"{precision} B={B:4d}: eager {t_eager * 1e3:8.1f} ms/decode ({t_eager / S * 1e3:.3f} ms/step)   graph replay {t_graph * 1e3:8.1f} ms "
              f"({t_graph / S * 1e3:.3f} ms/step)   tokens equal: {same}")
    except Exception as e:                                                  # noqa: BLE001
        print(f"{precision} B={B}: eager {t_eager * 1e3:.1f} ms/decode ; capture failed: {type(e).__name__}: {str(e)[:300]}")

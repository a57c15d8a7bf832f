"""Does the ConvGRU value net's chain (tower -> GRU -> tail) on the candidates of one step gain from two half batches on
two streams (as the config-4 trunk does)? forward_tokens on n rows: one chain vs even / odd rows on two streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic

dev = "cuda:0"
model, emb, head, _ = synthetic.build("dna", dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for prec in ("f32", "f16x3", "bf16"):
    model.precision = prec
    fn = model.value_callable(emb, head)
    for n in (2560, 2048, 1536):
        tok = torch.randint(0, 5, (n, 200), device=dev, dtype=torch.uint8)
        parts = [tok[0::2].contiguous(), tok[1::2].contiguous()]

        def one():
            return fn.forward_tokens(tok)

        def two():
            main = torch.cuda.current_stream()
            out = []
            for k, st in enumerate(streams):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    out.append(fn.forward_tokens(parts[k]))
            for st in streams:
                main.wait_stream(st)
            return out

        res = {}
        for name, f in (("one chain", one), ("two streams", two)):
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                r = f()
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t) / 20 * 1e6
        a, b = one().reshape(-1), two()
        same = torch.equal(a[0::2], b[0].reshape(-1)) and torch.equal(a[1::2], b[1].reshape(-1))
        print(f"{prec:6s} n={n}: one chain {res['one chain']:7.1f} us   two streams {res['two streams']:7.1f} us   same bits: {same}")

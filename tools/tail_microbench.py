"""Per-launch time of the GRU and value-tail kernels inside FusedValueNet (HIP events bound to the dispatches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic, _lib
from svdd_amd.fused import FusedValueNet
dev = "cuda:0"
model, emb, head, _ = synthetic.build("dna", dev)
fv = FusedValueNet(emb, head).to(dev).eval()
for n, L in [(2560, 200), (5120, 200), (2048, 200)]:
    oh = torch.zeros(n, L, 4, device=dev); oh.scatter_(2, torch.randint(0, 4, (n, L, 1), device=dev), 1.0)
    for _ in range(3): fv(oh)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(10): fv(oh)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    parts = {name: _lib.profile_collect(k) for name, k in (("tower", 5), ("gru", 3), ("tail", 7))}
    print(f"n={n} L={L}: " + "  ".join(f"{k} {t / c * 1e3:7.1f} us" for k, (t, c) in parts.items()))

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic, _lib
from svdd_amd.fused import FusedValueNet, conv_tower
dev = "cuda:0"
model, emb, head, _ = synthetic.build("dna", dev)
fv = FusedValueNet(emb, head).to(dev).eval()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for n, L in [(2560, 200), (5120, 200), (2560, 50)]:
    oh = torch.zeros(n, L, 4, device=dev); oh.scatter_(2, torch.randint(0, 4, (n, L, 1), device=dev), 1.0)
    t = timeit(lambda: conv_tower(oh, fv.tw_tiles, fv.tw_bias, fv.tw_resmask))
    fl = 2.0 * n * L * (4 * 64 * 15 + 5 * 64 * 64 * 5)
    fv.use_fused_tower = True; t_all = timeit(lambda: fv(oh))
    fv.use_fused_tower = False; t_old = timeit(lambda: fv(oh))
    print(f"n={n} L={L}: tower kernel {t:8.1f} us ({fl/t/1e6:6.1f} TFLOP/s) ; value net fused-tower {t_all:8.1f} us vs layerwise {t_old:8.1f} us")

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic
from svdd_amd.fused import FusedValueNet, FusedBackbone, gru_bidir
dev = "cuda:0"
model, emb, head, _ = synthetic.build("dna", dev)
fv = FusedValueNet(emb, head).to(dev).eval()
fb = FusedBackbone(model.backbone).to(dev).eval()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    for n in (96, 2560):
        oh = torch.zeros(n, 200, 4, device=dev); oh.scatter_(2, torch.randint(0, 4, (n, 200, 1), device=dev), 1.0)
        with torch.no_grad():
            fv(oh); a = fv(oh); b = fv(oh)
            x = torch.randn(n, 200, 64, device=dev)
            g1 = gru_bidir(x, fv.wpack, fv.bpack); g2 = gru_bidir(x, fv.wpack, fv.bpack)
            print(f"det={det} n={n} value repeat diff {(a-b).abs().max().item():.3e}  gru repeat diff {(g1-g2).abs().max().item():.3e}  value ms {timeit(lambda: fv(oh)):.3f}")
    x = torch.randint(0, 5, (256, 200), device=dev).to(torch.uint8)
    with torch.no_grad():
        fb(x); a = fb(x); b = fb(x)
        print(f"det={det} backbone repeat diff {(a-b).abs().max().item():.3e}  ms {timeit(lambda: fb(x)):.3f}")

"""Enformer-shaped value trunk (BASELINE config 4) forward time at n sequences, with and without MIOpen's find mode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd.enformer_value import EnformerTrunk
from svdd_amd.value_nets import ConvHead
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
trunk = EnformerTrunk().to(dev).eval()
head = ConvHead(1, 3072).to(dev).eval()
x = torch.zeros(n, 200, 4, device=dev); x.scatter_(2, torch.randint(0, 4, (n, 200, 1), device=dev), 1.0)
def timeit(fn, k=3):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
fl = EnformerTrunk.flops_per_sequence() * n
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    with torch.no_grad():
        t = timeit(lambda: head(trunk(x)))
    print(f"n={n} cudnn.benchmark={bench}: {t:8.2f} ms  ({fl / t / 1e9:6.1f} TFLOP/s)")

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic
model, emb, head, _ = synthetic.build("dna", "cpu")
x = torch.randint(0, 5, (256, 200))
oh = torch.randn(256, 200, 4)
print("cpu count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    with torch.no_grad():
        model.backbone(x, torch.zeros(256)); t = time.perf_counter(); model.backbone(x, torch.zeros(256)); tb = time.perf_counter() - t
        head(emb(oh)); t = time.perf_counter(); head(emb(oh)); tv = time.perf_counter() - t
    print(th, "threads: backbone %.3f s  value(B=256) %.3f s" % (tb, tv))

"""A few launches of the fused conv-tower kernel at n=2560, L=200 (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic
from svdd_amd.fused import FusedValueNet, conv_tower
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
model, emb, head, _ = synthetic.build("dna", dev)
fv = FusedValueNet(emb, head).to(dev).eval()
oh = torch.zeros(n, 200, 4, device=dev); oh.scatter_(2, torch.randint(0, 4, (n, 200, 1), device=dev), 1.0)
for _ in range(5):
    conv_tower(oh, fv.tw_tiles, fv.tw_bias, fv.tw_resmask)
torch.cuda.synchronize()

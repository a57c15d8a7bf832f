import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd.fused import conv1d_cl, pack_conv
dil = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(256, 200, 128, device="cuda"); w = torch.randn(128, 128, 9, device="cuda") * 0.05
wp = pack_conv(w)
for _ in range(5):
    y = conv1d_cl(x, wp, 128, 9, dil)
torch.cuda.synchronize()

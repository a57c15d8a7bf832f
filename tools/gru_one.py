import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd.fused import gru_bidir, pack_gru
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
gru = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).to("cuda").eval()
wp, bp = pack_gru(gru)
x = torch.randn(n, 200, 64, device="cuda")
for _ in range(4): gru_bidir(x, wp, bp)
torch.cuda.synchronize()

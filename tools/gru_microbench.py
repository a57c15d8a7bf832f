import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib
from svdd_amd.fused import gru_bidir, pack_gru
dev = "cuda:0"
gru = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).to(dev).eval()
wp, bp = pack_gru(gru)
for n, L in [(2560, 200), (5120, 200), (2560, 50)]:
    x = torch.randn(n, L, 64, device=dev)
    for mode in (0, 2):
        _lib.lib().svdd_gru_set_mode(mode)
        for _ in range(3): gru_bidir(x, wp, bp)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(10): gru_bidir(x, wp, bp)
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        tot, k = _lib.profile_collect(3)
        print(f"n={n} L={L} mode={mode}: {tot/k*1e3:.1f} us  ({2*n*L*2*(3*64*128)/(tot/k*1e-3)/1e12:.1f} TFLOP/s)")
_lib.lib().svdd_gru_set_mode(0)

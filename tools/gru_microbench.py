import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib
from svdd_amd.fused import gru_bidir, pack_gru
dev = "cuda:0"
gru = torch.nn.GRU(64, 64, bidirectional=True, batch_first=True).to(dev).eval()
wp, bp = pack_gru(gru)
for n, L in [(2560, 200), (2048, 200), (1984, 200), (1280, 200), (5120, 200)]:
    x = torch.randn(n, L, 64, device=dev)
    for mode in (1, 4):                              # 1: one wave does both halves ; 4 (= default): producer / consumer waves
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
# a compacted batch: tensors laid out for 2560 rows, the valid-row count on the device
x = torch.randn(2560, 200, 64, device=dev)
for live in (2560, 2100, 2048, 1984, 1500):
    cnt = torch.tensor([live], dtype=torch.int32, device=dev)
    for _ in range(3): gru_bidir(x, wp, bp, count=cnt)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(10): gru_bidir(x, wp, bp, count=cnt)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    tot, k = _lib.profile_collect(3)
    print(f"n=2560 buffers, count={live}: {tot/k*1e3:.1f} us")

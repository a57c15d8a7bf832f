"""Time the hand-written fp32-MFMA dilated conv against MIOpen at the backbone / value-net shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from svdd_amd.fused import conv1d_cl, pack_conv
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for (n, L, cin, cout, T, dil) in [(256, 200, 128, 128, 9, 1), (256, 200, 128, 128, 9, 4), (256, 200, 128, 128, 9, 16),
                                  (256, 200, 128, 128, 9, 64), (2560, 200, 64, 64, 5, 1), (256, 50, 128, 128, 9, 1), (2560, 50, 64, 64, 5, 1)]:
    x = torch.randn(n, L, cin, device=dev); w = torch.randn(cout, cin, T, device=dev) * 0.05
    wp = pack_conv(w)
    xcl = x.view(n, 1, L, cin).permute(0, 3, 1, 2)
    wcl = w.unsqueeze(2).contiguous(memory_format=torch.channels_last)
    t_mine = timeit(lambda: conv1d_cl(x, wp, cout, T, dil))
    t_mi = timeit(lambda: F.conv2d(xcl, wcl, None, padding=(0, (T // 2) * dil), dilation=(1, dil)))
    fl = 2.0 * n * L * cin * cout * T
    print(f"n={n:5d} L={L:3d} {cin:3d}->{cout:3d} T={T} dil={dil:2d}: ours {t_mine:8.1f} us ({fl/t_mine/1e6:6.1f} TF nominal)   MIOpen {t_mi:8.1f} us ({fl/t_mi/1e6:6.1f} TF)")

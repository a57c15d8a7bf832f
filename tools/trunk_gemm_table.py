"""Per-GEMM table of one forward of the config-4 value trunk (3840 candidates, one chain of kernels): shape, launches, time, TFLOP/s
(fp32-equivalent), 256-row tiles and rounds of the chip. Usage: python tools/trunk_gemm_table.py [f32|bf16x3|bf16]"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from svdd_amd import synthetic
from svdd_amd.fused_trunk import FusedEnformerValueNet
model, emb, head, _ = synthetic.build("dna", "cuda:0", value="enformer")
n, L = 3840, 200
tok = torch.randint(0, 5, (n, L), device="cuda:0", dtype=torch.uint8)
fn = FusedEnformerValueNet(emb, head, sys.argv[1] if len(sys.argv) > 1 else "f32")
fn.tower_streams = 1
fn.forward_tokens(tok); torch.cuda.synchronize()
fn.timing = []
fn.forward_tokens(tok); torch.cuda.synchronize()
import collections
agg = collections.OrderedDict()
for Mr, N, C, T, e0, e1 in fn.timing:
    k = (Mr, N, C, T)
    ms = e0.elapsed_time(e1)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += ms
tot = sum(v[1] for v in agg.values())
for (Mr, N, C, T), (cnt, ms) in agg.items():
    fl = 2.0 * Mr * N * C * T * cnt
    tiles = ((Mr + 255) // 256) * ((N + 255) // 256)
    print(f"M={Mr:7d} N={N:5d} K={C*T:6d} x{cnt:3d}: {ms:8.2f} ms {fl/ms/1e9:7.1f} TFLOP/s  tiles={tiles:5d} rounds={tiles/256:5.2f}  {100*ms/tot:4.1f}%")
print("total", tot)

"""The Enformer-shaped value trunk at fp32 (svdd_trunk.hip with fp32 planes) against the PyTorch modules and bf16x3:
one forward on n candidates, GEMM time split, and a C4-shard decode (B = 256, M = 20, 128 steps) at each precision.
Usage: python tools/trunk_f32_bench.py [--decode 1]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import synthetic
from svdd_amd.fused_trunk import FusedEnformerValueNet

ap = argparse.ArgumentParser()
ap.add_argument("--decode", type=int, default=1)
ap.add_argument("--n", type=int, default=3840)
ap.add_argument("--torch", type=int, default=1, help="0: skip the decode on the PyTorch modules (its first decode spends minutes in MIOpen's kernel search)")
args = ap.parse_args()
DEV = "cuda:0"
model, emb, head, _ = synthetic.build("dna", DEV, value="enformer")
L = 200
tok = torch.randint(0, 5, (args.n, L), device=DEV, dtype=torch.uint8)
flops = float(emb.flops_per_sequence(L)) * args.n
for prec in ("f32", "bf16x3"):
    fn = FusedEnformerValueNet(emb, head, prec)
    for streams in (1, 2):
        fn.tower_streams = streams
        fn.forward_tokens(tok)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn.forward_tokens(tok)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"{prec} forward n={args.n} streams={streams}: {ms:.1f} ms = {flops / ms / 1e9:.1f} TFLOP/s (algorithmic)")
    fn.tower_streams = 1
    fn.timing = []
    fn.forward_tokens(tok)
    torch.cuda.synchronize()
    issued = sum(2.0 * Mr * N * C * T for Mr, N, C, T, _, _ in fn.timing)
    gms = sum(e0.elapsed_time(e1) for _, _, _, _, e0, e1 in fn.timing)
    print(f"{prec}: {len(fn.timing)} GEMM launches, {gms:.1f} ms, issued {issued / gms / 1e9:.1f} TFLOP/s (x3 passes not counted)")
    fn.timing = None
    del fn
    torch.cuda.empty_cache()
if args.decode:
    B, M, S = 256, 20, 128
    model.rng_mode, model.philox_seed = "philox", 0
    for prec, fused in (("f32", True), ("bf16x3", True)) + ((("f32", False),) if args.torch else ()):
        model.precision, model.fuse_trunk_f32 = prec, fused
        model.clear_fused()
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        x = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
        torch.cuda.synchronize()
        warm = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(args.decode):
            x = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / args.decode
        print(f"C4 shard decode precision={prec} {'hand-written trunk' if fused else 'PyTorch modules'}: {el:.2f} s = {B / el:.2f} seq/s (first decode {warm:.1f} s)")

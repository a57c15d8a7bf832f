"""A/B of svdd_trunk_gemm's tile-height policy inside whole config-4 shard decodes (B = 256, M = 20, 128 steps), same process:
256-row tiles only (option 41), by cost (40) without / with the concurrency hint. Usage: python tools/trunk_tile_ab.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from svdd_amd import _lib, synthetic
model, emb, head, _ = synthetic.build("dna", "cuda:0", value="enformer")
model.rng_mode, model.philox_seed = "philox", 0
lib = _lib.lib()
B, M, S = 256, 20, 128
for prec in ("f32", "bf16x3"):
    model.precision = prec
    fn = model.value_callable(emb, head)
    model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M); torch.cuda.synchronize()
    for name, bm, hint in (("256-row tiles only", 41, True), ("by cost, no concurrency hint", 40, False), ("by cost, with hint", 40, True), ("256-row tiles only", 41, True), ("by cost, with hint", 40, True)):
        _lib.set_option(4, bm); fn.gemm_conc_hint = hint
        t0 = time.perf_counter()
        model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"{prec:7s} {name:32s}: {el:6.2f} s = {B / el:6.2f} seq/s")
_lib.set_option(4, 40)

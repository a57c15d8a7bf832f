"""Wall-clock of the BASELINE.json configs that fit one MI355X (per-GPU shard sizes), Philox RNG, random-init nets.
Prints decoded sequences/s for each. Usage: python tools/config_sweep.py [--quick] [--precision f32|f16x3|...] [--no-skip]
--no-skip switches the exact work-skipping off (Diffusion.skip_unchanged); the hit rates are printed when it is on."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from svdd_amd import synthetic

quick = "--quick" in sys.argv
skip = "--no-skip" not in sys.argv
precision = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "f32"
dev = "cuda:0"
print(f"# precision {precision}, exact work-skipping {'on' if skip else 'off'}")

def run(name, fn, B, reps=2):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    assert out.shape[0] == B and int(out.max()) <= 3
    extra = ""
    for m in (dna_models if "dna_models" in globals() else []):
        if m.skip_stats:
            st = m.skip_stats
            extra = (f"  [live candidates {st['live_candidates'] / st['candidates']:.1%}, changed rows "
                     f"{st['changed_row_steps'] / st['row_steps']:.1%}]")
            m.skip_stats.clear()
    print(f"{name:78s} {dt*1e3:9.1f} ms/decode  {B/dt:9.1f} seq/s{extra}", flush=True)

S = 16 if quick else 128
dna, emb, head, rew = synthetic.build("dna", dev)
dna.rng_mode, dna.precision, dna.skip_unchanged, dna.skip_stats = "philox", precision, skip, ({} if skip else None)
dna_models = [dna]
run(f"C2 DNA SVDD-MC  B=256 L=200 M=10 S={S}", lambda: dna.controlled_sample(emb, head, num_steps=S, eval_sp_size=256, sample_M=10), 256)
run(f"C4-shard DNA SVDD-MC B=256 L=200 M=20 S={S} (ConvGRU value net)", lambda: dna.controlled_sample(emb, head, num_steps=S, eval_sp_size=256, sample_M=20), 256)
run(f"un-guided decode_sample B=256 L=200 S={S}", lambda: dna.decode_sample(num_steps=S, eval_sp_size=256), 256)
run(f"C5-shard DNA TDS B=256 L=200 S={S} (per-shard population)", lambda: dna.controlled_sample_TDS(rew, 0.5, num_steps=S, eval_sp_size=256), 256)
run(f"C5 DNA TDS B=2048 L=200 S={S} (one population on one GPU)", lambda: dna.controlled_sample_TDS(rew, 0.5, num_steps=S, eval_sp_size=2048), 2048, reps=1)
rna, emb_r, head_r, rew_r = synthetic.build("rna", dev)
rna.rng_mode, rna.precision, rna.skip_unchanged, rna.skip_stats = "philox", precision, skip, ({} if skip else None)
dna_models.append(rna)
run(f"C3 RNA SVDD-PM (tweedie) B=256 L=50 M=10 S={S}", lambda: rna.controlled_sample_tweedie(rew_r, num_steps=S, eval_sp_size=256, sample_M=10, options="True"), 256)
run(f"RNA SVDD-MC B=256 L=50 M=10 S={S}", lambda: rna.controlled_sample(emb_r, head_r, num_steps=S, eval_sp_size=256, sample_M=10), 256)
S2 = 4 if quick else 16
run(f"DPS DNA B=64 L=200 S={S2} (autograd baseline)", lambda: dna.controlled_sample_DPS(rew, 10.0, num_steps=S2, eval_sp_size=64), 64, reps=1)
run(f"C5 DPS DNA B=256 L=200 S={S} (gradient guidance, scale 10)", lambda: dna.controlled_sample_DPS(rew, 10.0, num_steps=S, eval_sp_size=256), 256, reps=1)

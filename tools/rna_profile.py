import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from svdd_amd import synthetic, _lib
for prec in ("f32", "f16x3"):
    rna, emb, head, rew = synthetic.build("rna", "cuda:0")
    rna.rng_mode, rna.precision = "philox", prec
    f = lambda: rna.controlled_sample(emb, head, num_steps=128, eval_sp_size=256, sample_M=10)
    f(); torch.cuda.synchronize()
    _lib.profile_enable(True)
    t = time.perf_counter(); f(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    _lib.profile_enable(False)
    names = {0: "propose", 1: "select", 3: "gru", 5: "tower", 6: "backbone", 7: "tail"}
    print(prec, f"{dt*1e3:.1f} ms/decode", {names[k]: round(_lib.profile_collect(k)[0], 1) for k in names})
    g = lambda: rna.controlled_sample_tweedie(rew, num_steps=128, eval_sp_size=256, sample_M=10, options="True")
    g(); torch.cuda.synchronize()
    _lib.profile_enable(True)
    t = time.perf_counter(); g(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    _lib.profile_enable(False)
    print(prec, "PM", f"{dt*1e3:.1f} ms/decode", {names[k]: round(_lib.profile_collect(k)[0], 1) for k in names})

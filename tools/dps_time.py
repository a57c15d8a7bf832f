import sys, time, torch
sys.path.insert(0, "/root/repo")
from svdd_amd import synthetic, _lib
model, emb, head, rew = synthetic.build("dna", "cuda:0")
model.rng_mode, model.philox_seed = "philox", 0
for one in (True, False):
    model.dps_one_launch = one
    model.controlled_sample_DPS(rew, 10.0, num_steps=8, eval_sp_size=256); torch.cuda.synchronize()
    t = time.perf_counter()
    x = model.controlled_sample_DPS(rew, 10.0, num_steps=32, eval_sp_size=256); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 32
    _lib.profile_enable(True)
    model.controlled_sample_DPS(rew, 10.0, num_steps=8, eval_sp_size=256); torch.cuda.synchronize()
    _lib.profile_enable(False)
    pr = {k: _lib.profile_collect(k) for k in (2, 3, 6, 10)}
    print("one_launch", one, "ms/step %.3f -> %.1f seq/s at 128 steps" % (dt * 1e3, 256 / (dt * 128)), {k: (round(v[0] / max(v[1], 1) * 1e3, 1), v[1]) for k, v in pr.items()})

"""-m gpu: VALUES of the decode harness (`BaseModel.controlled_decode*`, reference Enformer.py:399-477) and of the
un-guided `_sample` / `_presample` paths (reference diffusion_gosai.py:820-886, Enformer.py:135-160).

The 5-tuple (samples, value_func_preds, reward_model_preds, top_k_values, baseline_preds) is recomputed independently
from the engine's primitives: batch k of a harness call is the sampler run with Philox seed base + k (guided batches
first, then gen_batch_num * sample_M baseline batches); predictions are the plain modules on the batch's one-hot;
baseline_preds are the first gen_batch_num baseline batches; top-k is the best len / sample_M of ALL baseline rewards.
And against the REFERENCE's own harness methods (g22_harness.npz, recorded by tests/golden/make_golden.py g22): the same
5-tuples in replay mode, including the tweedie variant's flat `samples` list and its baseline-as-top-k."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def small():
    from svdd_amd import synthetic
    from svdd_amd.config import SamplingConfig
    model, emb, head, reward = synthetic.build("rna", DEV)
    model.config.sampling = SamplingConfig(steps=6)
    return model, emb, head, reward


def _onehot(tok):
    return (torch.nn.functional.one_hot(tok.clamp(max=3), 4) * (tok != 4)[..., None]).float()


@pytest.mark.parametrize("method", ["mc", "tweedie"])
def test_controlled_decode_values(small, method):
    from svdd_amd.harness import BaseModel, batch_seed
    model, emb, head, reward = small
    Bsz, M, G = 4, 3, 2
    model.rng_mode, model.philox_seed = "philox", 40
    hm = BaseModel(emb, head, model, reward, batch_size=Bsz, task="rna")
    with torch.no_grad():
        if method == "mc":
            samples, vpred, rpred, topk, base = hm.controlled_decode(gen_batch_num=G, sample_M=M)
        else:
            samples, vpred, rpred, topk, base = hm.controlled_decode_tweedie(gen_batch_num=G, sample_M=M, options="True")
    assert model.philox_seed == 40                                   # restored
    # --- independent recomputation
    exp_samples, exp_v, exp_r = [], [], []
    with torch.no_grad():
        for k in range(G):
            model.philox_seed = batch_seed(40, k)
            if method == "mc":
                b = model.controlled_sample(emb, head, eval_sp_size=Bsz, sample_M=M)
            else:
                b = model.controlled_sample_tweedie(reward, eval_sp_size=Bsz, sample_M=M, options="True", task="rna")
            exp_samples.append(b)
            exp_v.append(head(emb(_onehot(b))).reshape(Bsz))
            exp_r.append(reward(_onehot(b).transpose(1, 2))[:, 0].reshape(Bsz))
        all_base = []
        for i in range(G * M):
            model.philox_seed = batch_seed(40, G + i)
            b = model.decode_sample(eval_sp_size=Bsz)
            all_base.append(reward(_onehot(b).transpose(1, 2))[:, 0].reshape(Bsz))
    model.philox_seed = 40
    if method == "tweedie":            # the reference's tweedie harness `extend`s: a flat list of rows (Enformer.py:766; g22)
        assert len(samples) == G * Bsz and samples[0].shape == (model.config.model.length,)
        samples = [torch.stack(samples[k * Bsz:(k + 1) * Bsz]) for k in range(G)]
    assert len(samples) == G
    for a, b in zip(samples, exp_samples):
        assert a.dtype == torch.int64 and torch.equal(a, b)
    assert not torch.equal(samples[0], samples[1])                    # batches are not copies of each other
    assert torch.allclose(vpred.reshape(-1), torch.cat(exp_v), atol=1e-6)
    assert torch.allclose(rpred.reshape(-1), torch.cat(exp_r), atol=1e-6)
    assert torch.allclose(base.reshape(-1), torch.cat(all_base[:G]), atol=1e-6)          # slice i < gen_batch_num
    if method == "tweedie":            # ... and returns cat(baseline_preds) in the top-k slot (Enformer.py:802; g22)
        assert torch.equal(topk, base)
        return
    allv = torch.cat(all_base)
    k = int(len(allv) / M)
    exp_topk = torch.sort(allv, descending=True).values[:k]
    assert topk.shape == (k,) and torch.allclose(topk, exp_topk, atol=1e-6)
    assert bool((topk[:-1] >= topk[1:]).all())                         # descending


@pytest.mark.parametrize("kind", ["mc", "pm", "pmh", "tds"])
def test_harness_5_tuple_equals_the_references_own_harness(golden, kind):
    """g22: the reference's `BaseModel.controlled_decode` / `controlled_decode_tweedie` (options "True" and the default-like
    bool True -> heuristic branch) / `controlled_decode_TDS` (Enformer.py:399-477, 719-813, 479-557) run by the reference
    itself with the tiny fixture nets. This harness in replay mode must return the same 5-tuple: the same decoded samples in
    the same container shape (token-exact: guided batches and all gen_batch_num * sample_M baseline batches consume the global
    RNG stream in the reference's order), predictions / top-k / baseline within 1e-4."""
    from svdd_amd.harness import BaseModel
    from svdd_amd.value_nets import ConvGRUTrunk, ConvHead, RewardModel
    from tests import e2e_parity
    g = golden("g22_harness.npz")
    L, S, B, G, M = (int(g[k]) for k in ("L", "S", "B", "G", "M"))
    model, emb, head = e2e_parity.tiny_engine(golden("nets_tiny.npz"), L, S, DEV)
    emb_r = ConvGRUTrunk(stem_in_channels=4, stem_channels=8, stem_kernel_size=15, n_conv=3, channel_init=8, kernel_size=5,
                         dropout=0.1)
    emb_r.load_state_dict(e2e_parity._sd(g, "reward_embedding"), strict=True)
    head_r = ConvHead(1, 8)
    head_r.load_state_dict(e2e_parity._sd(g, "reward_head"), strict=True)
    reward = RewardModel(emb_r, head_r).to(DEV).eval()
    model.rng_mode = "replay"
    hm = BaseModel(emb, head, model, reward, batch_size=B, task="dna")
    torch.manual_seed(int(g["seed"]))
    np.random.seed(int(g["np_seed"]))
    with torch.no_grad():
        if kind == "mc":
            out = hm.controlled_decode(gen_batch_num=G, sample_M=M)
        elif kind == "tds":
            out = hm.controlled_decode_TDS(gen_batch_num=G, sample_M=M, alpha=float(g["alpha"]))
        else:
            out = hm.controlled_decode_tweedie(gen_batch_num=G, sample_M=M, options="True" if kind == "pm" else True)
    samples, vpred, rpred, topk, base = out
    assert len(samples) == int(g[kind + "_samples_len"])
    assert tuple(samples[0].shape) == tuple(g[kind + "_samples_item_shape"])
    assert samples[0].dtype == torch.int64
    assert np.array_equal(torch.stack(list(samples)).cpu().numpy(), g[kind + "_samples"])
    for name, t in (("value_func_preds", vpred), ("reward_model_preds", rpred), ("top_k", topk), ("baseline_preds", base)):
        ref = g[kind + "_" + name]
        assert tuple(t.shape) == tuple(g[kind + "_" + name + "_shape"]), name
        assert np.abs(t.cpu().numpy() - ref).max() <= 1e-4, name


class _Replay(torch.nn.Module):
    def __init__(self, logits_seq):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))
        self.logits_seq, self.k = logits_seq, 0

    def forward(self, x, sigma):
        lg = self.logits_seq[self.k % len(self.logits_seq)]
        self.k += 1
        return torch.from_numpy(np.ascontiguousarray(np.swapaxes(lg, 1, 2))).to(x.device).transpose(1, 2)


def _replay_model(g):
    from svdd_amd.config import Config, ModelConfig, SamplingConfig
    from svdd_amd.diffusion import Diffusion
    cfg = Config(model=ModelConfig(hidden_dim=16, num_cnn_stacks=1, length=int(g["L"])),
                 sampling=SamplingConfig(steps=int(g["S"])))
    return Diffusion(cfg, backbone=_Replay(g["logits"])).to(DEV).eval()


def test_sample_returns_the_references_intermediate_states(golden):
    """`_sample` (reference :820-886) = decode_sample + the S - 1 intermediate states: on the reference's recorded
    un-guided run (g10: logits of every step, mt19937 seed) both the states and x_0 are the reference's."""
    g = golden("g10_decode_sample.npz")
    S, B = int(g["S"]), int(g["B"])
    d = _replay_model(g)
    torch.manual_seed(int(g["seed"]))
    x0, mid = d._sample(eval_sp_size=B)
    assert len(mid) == S - 1 and all(m.dtype == torch.int64 for m in mid)
    for i, m in enumerate(mid):
        assert np.array_equal(m.cpu().numpy(), g["xs"][i + 1].astype(np.int64)), i
    assert np.array_equal(x0.cpu().numpy(), g["x0"])


def test_presample_builds_the_per_step_eval_sets(golden, small):
    """BaseModel.__init__'s pre-sampling (reference Enformer.py:135-160): val_batch_num un-guided `_sample` runs; the
    one-hot states of step j of all runs are concatenated into eval_time_step_batches[j], each paired with the reward of
    the run's final sample."""
    from svdd_amd.harness import BaseModel
    g = golden("g10_decode_sample.npz")
    S, B, L = int(g["S"]), int(g["B"]), int(g["L"])
    _, emb, head, reward = small
    d = _replay_model(g)
    torch.manual_seed(int(g["seed"]))
    hm = BaseModel(emb, head, d, reward, batch_size=B, task="rna", val_batch_num=1)
    assert len(hm.eval_time_step_batches) == S and len(hm.eval_time_step_targets) == S
    with torch.no_grad():
        target = reward(_onehot(torch.from_numpy(g["x0"]).to(DEV)).transpose(1, 2))[:, 0]
    for jstep in range(S):
        states = g["xs"][jstep + 1].astype(np.int64) if jstep < S - 1 else g["x0"]
        exp = _onehot(torch.from_numpy(states).to(DEV)).long()
        assert torch.equal(hm.eval_time_step_batches[jstep], exp), jstep
        assert torch.allclose(hm.eval_time_step_targets[jstep], target, atol=1e-6)


def test_sharded_decode_equals_unsharded_with_the_fused_nets():
    """SURVEY.md section 8e on one GPU: a config-2-shaped decode of 2 b rows as ONE batch equals the same rows decoded as
    two shards of b rows through distributed.sharded_sample (Philox keyed by the global row) — token-exact, with the
    real fused nets (one-launch backbone, parent-sharing tower, GRU, tail), in fp32 and in a split-precision mode."""
    from svdd_amd import distributed, synthetic
    model, emb, head, _ = synthetic.build("dna", DEV)
    model.rng_mode, model.philox_seed = "philox", 123
    total, M, S = 96, 10, 128
    for precision in ("f32", "f16x3"):
        model.precision = precision
        sampler = lambda **kw: model.controlled_sample(emb, head, num_steps=S, sample_M=M, **kw)   # noqa: E731
        whole = distributed.sharded_sample(model, total, sampler, rank=0, world=1)
        parts = [distributed.sharded_sample(model, total, sampler, rank=r, world=2) for r in range(2)]
        ragged = [distributed.sharded_sample(model, total, sampler, rank=r, world=5) for r in range(5)]
        assert whole.shape == (total, 200) and int(whole.max()) <= 3
        assert torch.equal(torch.cat(parts), whole)
        assert torch.equal(torch.cat(ragged), whole)                  # 96 rows over 5 ranks: 20,19,19,19,19
    model.precision = "f32"
    assert model.row_offset == 0


@pytest.mark.parametrize("replay_rng", ["device", "host"])
def test_sharded_replay_decode_equals_the_unsharded_replay_decode(golden, replay_rng):
    """SURVEY.md section 8e, parity mode: "each rank replays the global mt19937 stream and slices its rows". Every rank — seeded
    like the reference's single process — generates the WHOLE batch's uniforms of a step and K1 reads the rows of its shard
    (svdd_rng.uniforms_rows / row_offset): the shards of 2 and of 5 (ragged) ranks, concatenated, are token for token the
    unsharded replay decode — which for g13's seed and size is the REFERENCE's own run — and every rank's generator ends in the
    state the unsharded decode leaves. SVDD-MC with the work-skipping loop, SVDD-PM, the un-guided decode."""
    from svdd_amd import distributed, synthetic
    model, emb, head, _ = synthetic.build("dna", DEV)
    model.rng_mode, model.replay_rng = "replay", replay_rng
    g = golden("g13_traj_mc_full_m10.npz")                           # the reference's run: B = 4, M = 10, 32 steps, seed 2
    B, M, S, seed = int(g["B"]), int(g["M"]), int(g["S"]), int(g["seed"])
    sampler = lambda **kw: model.controlled_sample(emb, head, num_steps=S, sample_M=M, **kw)   # noqa: E731
    torch.manual_seed(seed)
    whole = sampler(eval_sp_size=B)
    after = torch.rand(8)
    assert np.array_equal(whole.cpu().numpy(), g["x0"])              # = the reference's x_0
    for world in (2, 3):
        parts = []
        for r in range(world):
            torch.manual_seed(seed)
            parts.append(distributed.sharded_sample(model, B, sampler, rank=r, world=world))
            assert torch.equal(torch.rand(8), after)                 # the rank consumed the whole batch's stream
        assert torch.equal(torch.cat(parts), whole), world
    # a larger, ragged case on the RNA nets: MC (skipping loop), PM, un-guided
    rna, emb_r, head_r, reward_r = synthetic.build("rna", DEV)
    rna.rng_mode, rna.replay_rng = "replay", replay_rng
    runs = {"mc": lambda **kw: rna.controlled_sample(emb_r, head_r, num_steps=10, sample_M=4, **kw),
            "pm": lambda **kw: rna.controlled_sample_tweedie(reward_r, num_steps=8, sample_M=3, options="True", **kw),
            "plain": lambda **kw: rna.decode_sample(num_steps=10, **kw)}
    for name, run in runs.items():
        torch.manual_seed(77)
        whole = run(eval_sp_size=23)
        parts = []
        for r in range(4):
            torch.manual_seed(77)
            parts.append(distributed.sharded_sample(rna, 23, run, rank=r, world=4))
        assert torch.equal(torch.cat(parts), whole), name
    assert model.row_offset == 0 and model._shard is None


_TDS_RANKS = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
from svdd_amd import distributed, synthetic
rank, world, local = distributed.init_from_env("gloo")      # two ranks share the box's one GPU: gloo through the host
dev = "cuda:0"
model, emb, head, reward = synthetic.build("dna", dev)
model.rng_mode, model.philox_seed = "philox", 9
total, S, alpha = {total}, 24, 0.5
sampler = lambda **kw: model.controlled_sample_TDS(reward, alpha, num_steps=S, **kw)
np.random.seed(3)
out = distributed.sharded_sample(model, total, sampler)      # the per-step all-gather + whole-batch resample
assert out.shape == (total, 200) and model._shard is None
if rank == 0:
    np.random.seed(3)
    whole = sampler(eval_sp_size=total)                      # the same decode as one batch on one GPU
    assert torch.equal(out, whole), int((out != whole).sum())
    assert len(torch.unique(whole, dim=0)) < total           # resampling did duplicate particles: rows were coupled
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("total", [32, 27])
def test_tds_sharded_over_two_ranks_equals_unsharded(tmp_path, total):
    """BASELINE.json configs[4] sharded: SMC/TDS is the one sampler whose step couples the rows of the batch. Two
    processes (gloo; both on this box's GPU) decode a population of `total` particles through sharded_sample, exchanging
    proposals + rewards + uniforms once per step; the gathered result equals the one-process decode token for token."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "tds_ranks.py"
    script.write_text(_TDS_RANKS.format(root=root, total=total))
    port = 33500 + os.getpid() % 2000 + total
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1", SVDD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert res.stdout.count("ok") == 2


_TDS_REPLAY_RANKS = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
from svdd_amd import distributed, synthetic
rank, world, local = distributed.init_from_env("gloo")      # two ranks share the box's one GPU: gloo through the host
g = dict(np.load({fixture!r}))
model, emb, head, reward = synthetic.build("dna", "cuda:0")
model.rng_mode = "replay"                                    # every rank replays the WHOLE population's stream, K1 reads its rows
S, B, alpha = int(g["S"]), int(g["B"]), float(g["alpha"])
sampler = lambda **kw: model.controlled_sample_TDS(reward, alpha, num_steps=S, **kw)
torch.manual_seed(int(g["seed"]))                            # both ranks: the reference process's seeds
np.random.seed(int(g["np_seed"]))
out = distributed.sharded_sample(model, B, sampler)         # per step: one all-gather + the whole-population resample on each rank
assert out.shape == (B, 200) and model._shard is None
same = float((out.cpu().numpy() == g["x0"]).all(axis=1).mean())
print("rank", rank, "rows identical to the reference's x_0:", same)
assert same == 1.0
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_tds_two_ranks_in_replay_mode_reproduce_the_references_run(tmp_path):
    """The strongest multi-rank statement that can be checked on one GPU: the reference's own SMC / TDS run at the configs[4]
    shard size (g23: 256 particles, 128 steps) decoded by TWO ranks of 128 particles each in parity mode — each rank replays the
    whole population's mt19937 stream (K1 reads its rows), the per-step exchange assembles the population, every rank resamples it
    identically — and the gathered x_0 is the REFERENCE's x_0, row for row."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "tds_replay_ranks.py"
    script.write_text(_TDS_REPLAY_RANKS.format(root=root, fixture=os.path.join(root, "tests", "golden", "g23_traj_tds_c5.npz")))
    port = 35500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1", SVDD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert res.stdout.count("ok") == 2


_RCCL_PROBE = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from svdd_amd import distributed
rank, world, local = distributed.init_from_env()            # backend "nccl" = RCCL (what bench.py --gpus N uses)
assert world == 2 or os.environ.get("SVDD_PROBE_ONE_RANK")
if world == 1:
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
dev = "cuda:0"
assert dist.get_backend() == "nccl"
dist.all_reduce(torch.zeros(1, device=dev)); dist.barrier(device_ids=[0])
t = torch.arange(256 * 200, device=dev).remainder(4).to(torch.uint8).view(256, 200)
out = t.new_empty((dist.get_world_size() * 256, 200))
dist.all_gather_into_tensor(out, t)                          # distributed.gather_tokens's collective, uint8
assert torch.equal(out[:256], t)
tm = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(tm, op=dist.ReduceOp.MAX)                    # bench.py: max over ranks of the elapsed time
mine = torch.tensor([0.25, 0.5], device=dev, dtype=torch.float64)
allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
dist.all_gather(allr, mine)                                  # bench.py: per_rank
assert float(tm.item()) == 1.5 and float(allr[0][1]) == 0.5
dist.barrier(device_ids=[0]); dist.destroy_process_group()
print("RCCL_PROBE_OK")
'''


def test_rccl_collectives_of_the_bench_on_one_rank(tmp_path):
    """bench.py --gpus N and distributed.gather_tokens over the REAL backend (RCCL), as far as a one-GPU box allows: a
    one-rank RCCL process group running exactly the collectives the N > 1 bench issues (float32 warm-up all-reduce, barrier
    with device_ids, uint8 all_gather_into_tensor of the tokens, float64 MAX all-reduce, float64 all_gather). The N-rank
    logic itself is covered with gloo (tests/test_host_cpu.py, test_tds_sharded_over_two_ranks_equals_unsharded)."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rccl_probe.py"
    script.write_text(_RCCL_PROBE)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               SVDD_PROBE_ONE_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SVDD_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_PROBE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("rng", ["philox", "replay"])
def test_bench_real_two_rank_branch_equals_the_unsharded_decode(rng):
    """bench.py's REAL multi-rank branch (not --dry-run): `python bench.py --gpus 2 ...` starts two ranks that share this box's one GPU
    (gloo through the host, chosen by the launcher), each decodes its 64 rows (Philox keyed by the global row / the whole batch's
    mt19937 stream replayed per rank), max-over-ranks timing, per-rank times, the one all-gather, and one split-precision leg. The
    digest of the gathered batch in the line must equal the digest of the SAME decode done unsharded in this process (128 rows)."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    from svdd_amd import synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    B, S = 64, 16
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", str(B),
           "--diffusion-steps", str(S), "--cpu-steps", "0", "--c4-steps", "0", "--c3-steps", "0", "--c5-steps", "0",
           "--alt-precision", "f16x3", "--alt-steps", "1", "--rng", rng]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    assert lines[0] == res.stdout.splitlines()[-1] and len(lines[0].encode()) < 6144      # the driver's record keeps the last ~8 KB of stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["backend"] == "gloo"
    assert line["config"]["global_batch"] == 2 * B and line["config"]["rng"] == rng
    assert line["value"] > 0 and len(line["per_rank"]["decode_ms"]) == 2 and len(line["per_rank"]["allgather_ms"]) == 2
    assert line["alt"]["f16x3"]["value"] > 0 and line["roofline"]["frac"] > 0 and "cpu_baseline" in line
    # the same decode unsharded, here
    model, emb, head, _ = synthetic.build("dna", "cuda:0")
    model.rng_mode, model.philox_seed, model.row_offset = rng, 0, 0
    digest = {}
    for mode in ("f32", "f16x3"):
        model.precision = mode
        torch.manual_seed(0)
        x = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=2 * B, sample_M=10)
        digest[mode] = hashlib.sha1(x.to(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]
    assert line["x0_sha1"] == digest["f32"], (line["x0_sha1"], digest)
    assert line["alt"]["f16x3"]["x0_sha1"] == digest["f16x3"], (line["alt"]["f16x3"]["x0_sha1"], digest)


def test_bench_real_eight_rank_branch_equals_the_unsharded_decode():
    """The first real `--gpus 8` run must not be the first execution of any branch: `python bench.py --gpus 8`
    from a bare shell starts EIGHT ranks (sharing this box's one GPU; gloo through the host, chosen by the launcher because there are
    fewer GPUs than ranks), each decodes its 8 rows keyed by the global row, max-over-ranks timing, eight per-rank times, one
    all-gather. ranks_seen == 8, the gathered 64 rows' digest equals the unsharded 64-row decode's, and the line is the compact one."""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    from svdd_amd import synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    B, S, W = 8, 4, 8
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(W), "--steps", "1", "--warmup", "0", "--batch", str(B),
           "--diffusion-steps", str(S), "--cpu-steps", "0", "--c4-steps", "0", "--c3-steps", "0", "--c5-steps", "0", "--alt-precision", ""]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    last = res.stdout.splitlines()[-1]
    assert len(last.encode()) < 6144
    line = json.loads(last)
    assert line["n_gpus"] == W and line["ranks_seen"] == W and line["backend"] == "gloo" and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == W * B and line["config"]["sharding"] == f"rows x{W}, 1 all-gather"
    assert line["value"] > 0 and len(line["per_rank"]["decode_ms"]) == W and len(line["per_rank"]["allgather_ms"]) == W
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"] is None and "alt" not in line
    model, emb, head, _ = synthetic.build("dna", "cuda:0")
    model.rng_mode, model.philox_seed, model.row_offset = "philox", 0, 0
    x = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=W * B, sample_M=10)
    assert line["x0_sha1"] == hashlib.sha1(x.to(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]


def test_gru_round_rows_follow_the_cu_count(monkeypatch):
    """FusedValueNet's two-part late steps split at ONE round of GRU units on the chip: 8 rows per CU (a workgroup = 16 sequences in
    one direction), read from the device — not the literal 2048 of a 256-CU part."""
    from svdd_amd import _lib, synthetic
    from svdd_amd.fused import FusedValueNet
    model, emb, head, _ = synthetic.build("dna", "cuda:0")
    fn = model.value_callable(emb, head)
    assert isinstance(fn, FusedValueNet)
    ncu = _lib.device_info()[1]
    assert fn.gru_round_rows() == 8 * ncu
    r = 8 * ncu
    assert fn._gru_split(r) == 0 and fn._gru_split(r + 1) == r and fn._gru_split(r + (5 * r) // 16) == r and fn._gru_split(r + (5 * r) // 16 + 1) == 0
    if ncu == 256:
        assert fn._gru_split(2560) == 2048                                    # config 2: B * M = 2560 = 2048 + 512
    fn.__dict__.pop("_round_rows", None)
    monkeypatch.setattr(_lib, "device_info", lambda: ("gfx950", 304))        # a 304-CU part
    assert fn.gru_round_rows() == 2432 and fn._gru_split(2560) == 2432 and fn._gru_split(2432) == 0
    fn.__dict__.pop("_round_rows", None)
    monkeypatch.setattr(_lib, "device_info", lambda: ("gfx950", 128))
    assert fn.gru_round_rows() == 1024 and fn._gru_split(2560) == 0 and fn._gru_split(1280) == 1024
    fn.__dict__.pop("_round_rows", None)
    fn.split_gru_rounds = False
    assert fn._gru_split(2560) == 0

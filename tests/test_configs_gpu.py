"""-m gpu: the BASELINE.json configs and reference branches that earlier rounds left unexercised in their STATED form.

  * C4 (configs[3]): SVDD-MC with the FULL 230 M-parameter Enformer-shaped value trunk, M = 20, L = 200, on a slice of a
    C4 shard, against the oracle's replay of the recorded nets (reference decode.py:78-80, diffusion_gosai.py:1174-1228);
  * C5 (configs[4]), DPS half: controlled_sample_DPS with the full-size nets at L = 200, and the reference's own DPS run
    (fixture g11, diffusion_gosai.py:1286-1330) replayed through the WHOLE step on the GPU: gradient -> guided q_xs ->
    draw -> next state;
  * SVDD-PM's heuristic branch — what decode_tweedie.py actually runs, because `options == "True"` compares the default
    bool with a string (diffusion_gosai.py:1414,1420-1424) — against the reference run g14;
  * one reference step at M = 10 and M = 20 (g15) through K1 + K2;
  * select_mode = "multinomial" through a whole decode; the `dps` CLI."""
import numpy as np
import pytest
import torch

from oracle import svdd_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


def bvl_view(a):
    return dev(np.ascontiguousarray(np.swapaxes(a, 1, 2))).transpose(1, 2)


def _trace_np(model):
    tr = [(lg.cpu().numpy(), None if sc is None else sc.cpu().numpy()) for lg, sc in model.trace]
    model.trace = None
    return tr


# ------------------------------------------------------------------------------------------ g15: M = 10 / 20
@pytest.mark.parametrize("M", [10, 20])
def test_reference_step_at_M10_M20_through_the_kernels(golden, M):
    from svdd_amd import ops
    g = golden(f"g15_step_mc_m{M}.npz")
    x = g["x"].astype(np.uint8)
    B, L = x.shape
    torch.manual_seed(int(g["seed"]))
    uni = torch.rand(M, B, 5, L).to(DEV)                            # the reference's M x rand_like(q_xs) stream
    cand, onehot, q = ops.propose(bvl_view(g["logits"]), dev(x), float(g["dm"]), float(g["mcs"]), M, ops.Rng(uniforms=uni),
                                  want_q=True)
    assert np.array_equal(cand.cpu().numpy(), g["cand"])
    assert np.allclose(q.cpu().numpy(), g["q_xs"], rtol=2e-6, atol=0)
    x_next, soft, idx = ops.select(dev(g["scores"]), cand)
    assert np.array_equal(idx.cpu().numpy(), g["idx"])              # exact ties, 1-ulp plateaus, near-uniform rows
    assert np.abs(soft.cpu().numpy() - g["soft"]).max() <= 1e-6
    assert np.array_equal(x_next.cpu().numpy(), g["x_next"])


# ------------------------------------------------------------------------------------------ g14: PM heuristic branch
class _ReplayBackbone(torch.nn.Module):
    def __init__(self, logits_seq, x_seq):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))
        self.logits_seq, self.x_seq, self.k = logits_seq, x_seq, 0

    def forward(self, x, sigma):
        assert np.array_equal(x.cpu().numpy(), self.x_seq[self.k].astype(np.int64)), f"x at call {self.k}"
        lg = self.logits_seq[self.k]
        self.k += 1
        return bvl_view(lg)


def test_tweedie_heuristic_branch_replays_reference_run(golden):
    """Recorded logits / scores through controlled_sample_tweedie with the DEFAULT options: one backbone call per step, the
    reward model is shown transform_samples(candidate)^T (MASK rows zero), x_0 exact."""
    from svdd_amd.config import Config, ModelConfig, SamplingConfig
    from svdd_amd.diffusion import Diffusion
    g = golden("g14_traj_pm_heuristic.npz")
    S, B, L, M = int(g["S"]), int(g["B"]), int(g["L"]), int(g["M"])
    cfg = Config(model=ModelConfig(hidden_dim=16, num_cnn_stacks=1, length=L), sampling=SamplingConfig(steps=S))
    bb = _ReplayBackbone(g["logits"], g["xs"])
    d = Diffusion(cfg, backbone=bb).to(DEV).eval()
    calls = {"n": 0}

    def reward(oh):
        i = calls["n"]
        calls["n"] += 1
        want = orc.transform_samples(g["cand"][i].reshape(B * M, L), transposed=True)
        assert np.array_equal(oh.cpu().numpy(), want)
        return dev(g["scores"][i].reshape(B * M)).view(B * M, 1, 1)

    torch.manual_seed(int(g["seed"]))
    x0 = d.controlled_sample_tweedie(reward, eval_sp_size=B, sample_M=M)
    assert bb.k == S + 1 and calls["n"] == S
    assert np.array_equal(x0.cpu().numpy(), g["x0"])


def test_tweedie_heuristic_branch_real_tiny_nets_follow_the_reference(golden):
    """Same run with the reference's real (tiny) nets on the GPU, free-running in replay mode."""
    from tests import e2e_parity
    from svdd_amd.value_nets import RewardModel
    g = golden("g14_traj_pm_heuristic.npz")
    S, B, L, M = int(g["S"]), int(g["B"]), int(g["L"]), int(g["M"])
    model, emb, head = e2e_parity.tiny_engine(golden("nets_tiny.npz"), L, S, DEV)
    reward = RewardModel(emb, head).to(DEV).eval()
    model.rng_mode, model.trace, model.state_trace = "replay", [], []
    torch.manual_seed(int(g["seed"]))
    x0 = model.controlled_sample_tweedie(reward, eval_sp_size=B, sample_M=M)
    xs = np.stack([x.cpu().numpy() for x in model.state_trace])
    tr = _trace_np(model)
    model.state_trace = None
    assert np.array_equal(xs, g["xs"])
    for i in range(S):
        assert np.abs(tr[i][0] - g["logits"][i]).max() <= TOL
        assert np.abs(tr[i][1] - g["scores"][i]).max() <= TOL
    assert np.array_equal(x0.cpu().numpy(), g["x0"])


# ------------------------------------------------------------------------------------------ g11: a whole DPS step
def test_dps_whole_step_replays_reference_run(golden):
    """The reference's controlled_sample_DPS run (g11) on the GPU, step by step: backbone -> SUBS -> q, autograd through
    forward2 and the reward net -> guidance -> guided q_xs within 1e-4 of the reference's; the draw with the reference's
    uniforms gives the reference's next state exactly; then the free-running decode (replay RNG) gives its x_0."""
    from svdd_amd import ops
    from tests import e2e_parity
    from svdd_amd.value_nets import RewardModel
    g = golden("g11_traj_dps.npz")
    S, B, L, scale = int(g["S"]), int(g["B"]), int(g["L"]), float(g["scale"])
    assert int(g["q_is_bvl"]) == 1
    model, emb, head = e2e_parity.tiny_engine(golden("nets_tiny.npz"), L, S, DEV)
    reward = RewardModel(emb, head).to(DEV).eval()
    sched = model._schedule(S, 1e-5)[0]
    for i in range(S):
        x = dev(g["xs"][i])
        q = model._dps_guided_q(x, sched[i, 1], sched[i, 2], reward, scale)
        assert np.allclose(q.cpu().numpy(), g["q"][i], rtol=1e-4, atol=1e-4), np.abs(q.cpu().numpy() - g["q"][i]).max()
        u = dev(np.ascontiguousarray(np.swapaxes(g["u"][i], 1, 2))[None])                 # [1, B, 5, L]: the stream order
        cand, _ = ops.sample_categorical(q, x, 1, ops.Rng(uniforms=u, uniforms_layout=ops.LAYOUT_BVL))
        if i + 1 < S:
            assert np.array_equal(cand.cpu().numpy()[:, 0], g["xs"][i + 1]), f"step {i}"
    model.rng_mode = "replay"
    torch.manual_seed(int(g["seed"]))
    x0 = model.controlled_sample_DPS(reward, scale, eval_sp_size=B)
    assert np.array_equal(x0.cpu().numpy(), g["x0"])


@pytest.fixture(scope="module")
def full_nets():
    from svdd_amd import synthetic
    return synthetic.build("dna", DEV)


def test_dps_whole_step_full_size_reference_run(golden, full_nets):
    """g20: the reference's controlled_sample_DPS with its FULL-SIZE nets and reward model (L = 200, B = 3, 4 steps, guidance
    scale 300: factors up to 1.06. The gradient of a 20-layer fp32 backward differs between MIOpen and the CPU's MKL-DNN by
    ~2e-7 absolute, 1e-3 of its size, so q = q0 * exp(scale * gradient) is comparable at 1e-4 only while scale * 2e-7 << 1e-4;
    at scale 1e4 the same run differs by 2e-3 in q) on the GPU, step by step as for g11: the guided q_xs — exp(scale * gradient) of an
    autograd pass through forward2 and the reward net (MIOpen's fused GRU backward) — within 1e-4 relative of the reference's
    CPU autograd, the draw with the reference's uniforms gives its next state, the free-running decode its x_0."""
    from svdd_amd import ops
    g = golden("g20_traj_dps_full.npz")
    S, B, L, scale = int(g["S"]), int(g["B"]), int(g["L"]), float(g["scale"])
    model, emb, head, reward = full_nets
    for name, mod in (("backbone", model.backbone), ("reward_embedding", reward.embedding), ("reward_head", reward.head)):
        sums = np.array([float(p.double().sum()) for p in mod.state_dict().values()])
        assert np.allclose(sums, g[name + "_param_sums"], rtol=0, atol=1e-6), name
    sched = model._schedule(S, 1e-5)[0]
    worst = 0.0
    for i in range(S):
        x = dev(g["xs"][i])
        q = model._dps_guided_q(x, sched[i, 1], sched[i, 2], reward, scale)
        ref = g["q"][i]
        worst = max(worst, float(np.abs(q.cpu().numpy() - ref).max() / np.abs(ref).max()))
        assert np.allclose(q.cpu().numpy(), ref, rtol=1e-4, atol=1e-4), np.abs(q.cpu().numpy() - ref).max()
        u = dev(np.ascontiguousarray(np.swapaxes(g["u"][i], 1, 2))[None])
        cand, _ = ops.sample_categorical(q, x, 1, ops.Rng(uniforms=u, uniforms_layout=ops.LAYOUT_BVL))
        if i + 1 < S:
            assert np.array_equal(cand.cpu().numpy()[:, 0], g["xs"][i + 1]), f"step {i}"
    model.rng_mode = "replay"
    torch.manual_seed(int(g["seed"]))
    x0 = model.controlled_sample_DPS(reward, scale, num_steps=S, eval_sp_size=B)
    assert np.array_equal(x0.cpu().numpy(), g["x0"])
    print("g20 dps: max |q - q_ref| / max|q_ref| =", worst)


def test_dps_at_the_c5_shard_batch_against_the_reference_run(golden, full_nets):
    """g26: BASELINE configs[4]'s gradient-guidance baseline as the REFERENCE ran it at the per-GPU shard size (controlled_sample_DPS,
    diffusion_gosai.py:980-1019, 1286-1330: B = 256, L = 200, 128 steps, full-size nets + reward model, guidance scale 25600 = g20's
    300 at a batch-mean reward) against the one-launch pair of round 5 (forward-with-saved-statistics + svdd_backbone_cnn_grad_f32,
    reward call on the hand-written convolution / GRU kernels). Teacher-forced on all 128 recorded states: the guided q_xs of the
    stored rows within 1e-4, every row's sum of q within 1e-4 relative, the categorical draw from the replayed mt19937 stream gives
    the reference's next state (a row may differ only at a near-tie: a handful of 32,768 row-steps); then the free-running decode."""
    from svdd_amd import ops
    g = golden("g26_traj_dps_c5.npz")
    S, B, L, scale = int(g["S"]), int(g["B"]), int(g["L"]), float(g["scale"])
    model, emb, head, reward = full_nets
    for name, mod in (("backbone", model.backbone), ("reward_embedding", reward.embedding), ("reward_head", reward.head)):
        sums = np.array([float(p.double().sum()) for p in mod.state_dict().values()])
        assert np.allclose(sums, g[name + "_param_sums"], rtol=0, atol=1e-6), name
    assert model.dps_one_launch and model._dps_one_launch(torch.zeros(1, L, 5, device=DEV)) is not None      # the new path is the one tested
    sched = model._schedule(S, 1e-5)[0]
    kept = {int(s): k for k, s in enumerate(g["q_steps"])}
    nq = int(g["q_rows"])
    keep_mode = model.rng_mode
    model.rng_mode = "replay"
    rows_same, worst_q, worst_sum = 0, 0.0, 0.0
    try:
        torch.manual_seed(int(g["seed"]))
        x_last = None
        for i in range(S):
            x = dev(g["xs"][i])
            q = model._dps_guided_q(x, sched[i, 1], sched[i, 2], reward, scale)
            if i in kept:
                ref = g["q"][kept[i]]
                mine = q[:nq].cpu().numpy()
                worst_q = max(worst_q, float(np.abs(mine - ref).max() / np.abs(ref).max()))
                assert np.allclose(mine, ref, rtol=1e-4, atol=1e-4), (i, np.abs(mine - ref).max())
            qs = q.double().sum(dim=(1, 2)).cpu().numpy()
            worst_sum = max(worst_sum, float(np.abs(qs - g["qsum"][i]).max() / np.abs(g["qsum"][i]).max()))
            cand, _ = ops.sample_categorical(q, x, 1, model._rng(i, 1, B, L, q))     # the reference's rand_like(q_xs) of this step
            nxt = cand.view(B, L)
            if i + 1 < S:
                rows_same += int((nxt.cpu().numpy() == g["xs"][i + 1]).all(axis=1).sum())
            else:
                x_last = nxt
        x0_tf = model._noise_removal(x_last.contiguous()).cpu().numpy()
        tf_rows = int((x0_tf == g["x0"]).all(axis=1).sum())
        torch.manual_seed(int(g["seed"]))
        x0 = model.controlled_sample_DPS(reward, scale, num_steps=S, eval_sp_size=B).cpu().numpy()
    finally:
        model.rng_mode = keep_mode
    free_rows = int((x0 == g["x0"]).all(axis=1).sum())
    print("g26 dps c5: q rel err", worst_q, "row-sum rel err", worst_sum, "next states identical", rows_same, "of", (S - 1) * B,
          "x0 rows (teacher-forced last step)", tf_rows, "x0 rows (free-running)", free_rows)
    assert worst_sum <= 1e-4, worst_sum
    # recorded (round 5): all 32,512 next states identical, x_0 exact both ways, q 5.5e-5, row sums 1.1e-5. One near-tie of the
    # categorical race may flip with a kernel change; a handful of rows may not.
    assert rows_same >= (S - 1) * B - 2, rows_same
    assert tf_rows >= B - 1 and free_rows >= B - 2, (tf_rows, free_rows)


def test_dps_at_config5_shape_full_size_nets(full_nets):
    """BASELINE configs[4], DPS half, at a shard slice: B = 32, L = 200, the full-size backbone (autograd through
    forward2) and reward net. Valid tokens; zero guidance is the un-guided ancestral decode (same Philox draws); guidance
    changes the outcome; gradients are finite."""
    model, _, _, reward = full_nets
    model.rng_mode, model.philox_seed = "philox", 17
    B, S = 32, 6
    try:
        a = model.controlled_sample_DPS(reward, 50.0, num_steps=S, eval_sp_size=B)
        c = model.controlled_sample_DPS(reward, 0.0, num_steps=S, eval_sp_size=B)
        d = model.decode_sample(num_steps=S, eval_sp_size=B)
        x = torch.full((B, 200), 4, dtype=torch.uint8, device=DEV)
        sched = model._schedule(S, 1e-5)[0]
        q = model._dps_guided_q(x, sched[0, 1], sched[0, 2], reward, 50.0)
        model.dps_single_forward = True            # opt-in: q_xs from the differentiable pass's log-probs (one backbone forward per step)
        q1 = model._dps_guided_q(x, sched[0, 1], sched[0, 2], reward, 0.0)
        e = model.controlled_sample_DPS(reward, 0.0, num_steps=S, eval_sp_size=B)
        model.dps_single_forward = False
        q2 = model._dps_guided_q(x, sched[0, 1], sched[0, 2], reward, 0.0)
    finally:
        model.rng_mode, model.dps_single_forward = "replay", False
    assert float((q1 - q2).abs().max()) <= 1e-5 * float(q2.abs().max())
    assert float((e != d).float().mean()) <= 1e-3
    assert a.shape == (B, 200) and a.dtype == torch.int64 and int(a.max()) <= 3 and int(a.min()) >= 0
    assert torch.equal(c, d)
    assert bool(torch.isfinite(q).all()) and float(q.min()) >= 0.0
    assert q.shape == (B, 200, 5)


def test_dps_in_a_split_precision_mode_takes_its_gradient_from_the_fp32_pair(full_nets):
    """precision = "f16x3": the sampling forward (q_xs) runs in the mode's arithmetic, the gradient comes from the fp32 one-launch pair
    (the differentiable pass has always been fp32). Zero guidance is then bit-for-bit the mode's own un-guided decode; guidance
    changes the outcome; the gradient kernel really ran."""
    from svdd_amd import _lib
    model, _, _, reward = full_nets
    keep = (model.rng_mode, model.philox_seed, model.precision)
    model.rng_mode, model.philox_seed, model.precision = "philox", 23, "f16x3"
    B, S = 16, 5
    try:
        _lib.profile_collect(10)
        _lib.profile_enable(True)
        a = model.controlled_sample_DPS(reward, 0.0, num_steps=S, eval_sp_size=B)
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        grad_launches = _lib.profile_collect(10)[1]
        for k in (0, 1, 2, 3, 5, 6, 7):
            _lib.profile_collect(k)
        d = model.decode_sample(num_steps=S, eval_sp_size=B)
        c = model.controlled_sample_DPS(reward, 2000.0, num_steps=S, eval_sp_size=B)
    finally:
        model.rng_mode, model.philox_seed, model.precision = keep
    assert grad_launches == S
    assert torch.equal(a, d)
    assert int(c.max()) <= 3 and not torch.equal(c, d)


# ------------------------------------------------------------------------------------------ C4 in its stated form
@pytest.mark.parametrize("skip_generic", [False, True])
def test_c4_full_enformer_trunk_slice_vs_oracle(skip_generic):
    """BASELINE configs[3]: SVDD-MC, M = 20, L = 200, value function = the FULL Enformer-shaped trunk (7 conv blocks to
    1536 channels, 11 transformer blocks, ~230 M parameters; reference decode.py:78-80, Enformer.py:1271-1334) on 8 rows of
    a 256-row shard (row_offset keys Philox by the global row). The oracle recomputes every propose / select / finalize
    from the recorded logits and scores: tokens exact."""
    from svdd_amd import synthetic
    model, emb, head, _ = synthetic.build("dna", DEV, value="enformer")
    n_params = sum(p.numel() for p in emb.parameters()) + sum(p.numel() for p in head.parameters())
    assert n_params > 2.0e8, n_params
    B, L, M, S = 8, 200, 20, 3
    sched = model._schedule(S, 1e-5)[0]
    # round 4: at the reference's precision (fp32) the trunk runs on the hand-written fp32 kernels (svdd_trunk.hip, fp32 planes +
    # v_mfma_f32_16x16x4_f32), not on the PyTorch modules: scores within 1e-4 of the module on the same weights
    from svdd_amd.fused_trunk import FusedEnformerValueNet
    fn = model.value_callable(emb, head)
    assert model.precision == "f32" and isinstance(fn, FusedEnformerValueNet) and fn.precision == "f32"
    tok = torch.randint(0, 5, (24, L), device=DEV, dtype=torch.uint8)
    onehot = (torch.nn.functional.one_hot(tok.long().clamp(max=3), 4) * (tok != 4)[..., None]).float()
    with torch.no_grad():
        ref, got = head(emb(onehot)).reshape(-1), fn.forward_tokens(tok).reshape(-1)
    assert float((ref - got).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    model.rng_mode, model.philox_seed, model.row_offset, model.skip_generic, model.trace = "philox", 4, 512, skip_generic, []
    x_gpu = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M).cpu().numpy()
    trace = _trace_np(model)
    assert trace[0][1].shape == (B, M) and np.isfinite(trace[0][1]).all()
    x_orc = orc.replay_controlled_sample(trace, sched, B, L, M, seed=4, row_offset=512)
    assert np.array_equal(x_gpu, x_orc)
    assert int(x_gpu.max()) <= 3


# ------------------------------------------------------------------------------------------ multinomial select, whole decode
@pytest.mark.parametrize("task,L", [("rna", 50), ("dna", 200)])
def test_multinomial_select_whole_decode_vs_oracle(task, L):
    """select_mode = "multinomial" (the reference's commented-out diffusion_gosai.py:1223) through a whole decode, plain and
    work-skipping paths, against the oracle's replay with the same Philox stream."""
    from svdd_amd import synthetic
    model, emb, head, _ = synthetic.build(task, DEV)
    B, M, S = 6, 5, 10
    sched = model._schedule(S, 1e-5)[0]
    outs = []
    for skip in (True, False):
        model.rng_mode, model.philox_seed, model.select_mode, model.skip_unchanged, model.trace = "philox", 99, "multinomial", skip, []
        x_gpu = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M).cpu().numpy()
        trace = _trace_np(model)
        x_orc = orc.replay_controlled_sample(trace, sched, B, L, M, seed=99, mode=1)
        assert np.array_equal(x_gpu, x_orc)
        outs.append(x_gpu)
    assert np.array_equal(outs[0], outs[1])
    model.select_mode, model.skip_unchanged, model.rng_mode = "argmax", True, "replay"
    amax = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
    assert amax.shape == (B, L)


# ------------------------------------------------------------------------------------------ CLI: decode_DPS.py
def test_cli_dps_writes_reference_npz(tmp_path):
    """decode_DPS.py contract (reference decode_DPS.py:112-119): ./log/{task}-{reward_name}_DPS.npz, keys decoding / baseline."""
    from svdd_amd import cli
    from svdd_amd.config import SamplingConfig
    import svdd_amd.synthetic as syn
    orig = syn.build

    def small_build(task, device, seed=44, **kw):
        m = orig(task, device, seed=seed)
        m[0].config.sampling = SamplingConfig(steps=5)
        return m

    syn.build = small_build
    try:
        path, out = cli.main("dps", ["--task", "rna", "--batch_size", "4", "--sample_M", "2", "--val_batch_num", "2",
                                     "--out_dir", str(tmp_path), "--rng", "philox", "--guidance_scale", "10"])
    finally:
        syn.build = orig
    assert path.endswith("rna-MRL_DPS.npz")
    z = np.load(path)
    assert set(z.files) == {"decoding", "baseline"} and z["decoding"].shape == (8,) and z["baseline"].shape == (8,)
    assert np.isfinite(z["decoding"]).all() and np.isfinite(z["baseline"]).all()

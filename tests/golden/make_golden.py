"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE on CPU.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The reference is imported from where it lies (tests/golden/_ref_import.py); only inputs and
outputs are stored. Fixtures:

  g1_sample_categorical.npz   _sample_categorical                 diffusion_gosai.py:30-34
  g2_subs.npz                 Diffusion._subs_parameterization    :286-304
  g3_schedule.npz             noise/move-chance prologue           :1036-1038,1176-1187
  g4_transform.npz            transform_samples                    :1462-1470
  g5_step_mc.npz              one _ddpm_update_finetune_controlled :1174-1228 (adversarial score ties)
  g5_step_mc_bvl.npz          same, backbone output laid out [B,5,L] like the reference CNN's
  g6_traj_mc_c1.npz           controlled_sample, C1 (B=4,L=200,M=2,S=128), tiny nets, per-step record
  g6_traj_mc_s16.npz          controlled_sample, S=16, B=3, L=50, M=5
  g7_traj_pm.npz              controlled_sample_tweedie, S=8, B=3, L=50, M=3
  g8_traj_tds.npz             controlled_sample_TDS, S=8, B=6, L=50 (np.random.seed)
  g9_rng.npz                  torch / numpy mt19937 streams
  g10_decode_sample.npz       decode_sample (un-guided), S=16, B=3, L=50
  g11_traj_dps.npz            controlled_sample_DPS, S=6, B=3, L=50: guided q_xs, uniforms, gradients per step
  nets_tiny.npz               state_dicts of the tiny nets used above (for net-parity tests)
  g13_traj_mc_full_c1.npz     controlled_sample with the FULL-SIZE nets of g12 (seed 44): BASELINE configs[0] (B=4, L=200, M=2,
                              S=128), every step's state, raw logits, candidates and value scores
  g13_traj_mc_full_m10.npz    same nets, B=4, L=200, M=10, S=32
  g14_traj_pm_heuristic.npz   controlled_sample_tweedie with the DEFAULT options=True (a bool: `options == "True"` is False,
                              :1414), i.e. the heuristic branch :1420-1424, tiny nets, S=8, B=3, L=50, M=3
  g15_step_mc_m10.npz / _m20  one _ddpm_update_finetune_controlled step at the widths the reference is run with (M=10
                              default, M=20 in BASELINE configs[3]) with adversarial ties: pins softmax/argmax over M
  g16_dit.npz                 the reference's models/dit.py DIT (hidden 32, 2 blocks, 2 heads, cond 16; zero-initialised adaLN /
                              output maps re-drawn so that every path carries signal) on B=3, L=24 tokens with non-zero sigma:
                              state_dict, indices, sigma, logits. flash_attn (CUDA-only, absent) is replaced by a plain
                              matmul-softmax / rotate-half STAND-IN of its two entry points: pinned up to that stand-in
  g17_enformer_trunk.npz      the reference's Enformer.py EnformerTrunk + ConvHead (n_conv=3, channels=384, 2 transformer blocks, 2
                              heads, key_len=16) on B=3, L=40 one-hot rows. enformer_pytorch (absent, unpinned) is replaced by a
                              STAND-IN = this repo's restatement of its five symbols (svdd_amd/enformer_value.py), so the fixture
                              pins the reference's OWN wiring (conv tower, block order NACDR, residuals, transformer block,
                              feed-forward, pointwise, head) and not the attention math: pinned up to that stand-in
  g18_traj_mc_full_rna.npz    controlled_sample with the FULL-SIZE nets at L = 50 (the RNA configs' length: several sequences share a
                              208-row kernel tile), B=6, M=5, S=24: every step's state, logits, candidates, scores (as g13)
  g18_traj_pm_full_rna.npz    controlled_sample_tweedie(options="True") — BASELINE configs[2]'s sampler — with the FULL-SIZE nets and a
                              full-size ConvGRU reward model (the fifth / sixth module synthetic.build("rna") creates), B=4, M=4, S=16:
                              states, logits, candidates, candidate logits, the x0-hat one-hots the reward model saw, scores
  g19_traj_tds_full.npz       controlled_sample_TDS (BASELINE configs[4]'s SMC baseline) with the FULL-SIZE nets and reward model at L = 200,
                              B=8, S=10, alpha=0.5: states, proposals, the three backbone outputs of a step, both reward vectors,
                              numpy's uniforms
  g20_traj_dps_full.npz       controlled_sample_DPS (configs[4]'s gradient-guidance baseline) with the FULL-SIZE nets and reward model, L = 200,
                              B=3, S=4, guidance scale 300 (factors up to 1.06): per step the guided q_xs, the uniforms, the gradient (as g11)
  -- round 4: reference runs AT the BASELINE batch sizes (lean: states, delta-coded candidates, scores; `python make_golden.py g21|g22|g23|g24|g25`,
     8 torch threads, 2 - 10 min each on the build container's 8 cores) --
  g21_traj_mc_c2.npz          controlled_sample at BASELINE configs[1] (the headline): B=256, L=200, M=10, S=128, full-size seed-44 nets
  g21_traj_pm_c3.npz          controlled_sample_tweedie(options="True") at configs[2]: B=256, L=50, M=10, S=128 (+ the x0-hat rows, reward scores)
  g22_harness.npz             Enformer.BaseModel.controlled_decode / _tweedie (both options) / _TDS THEMSELVES (object.__new__, stubbed gReLU
                              loader), tiny nets, gen_batch_num=2, sample_M=3: the 5-tuples
  g23_traj_tds_c5.npz         controlled_sample_TDS at configs[4]'s per-GPU shard: 256 particles, L=200, S=128 (proposals, x0-hat rows, rewards,
                              np.random.choice's ancestor indices)
  g24_decode_sample_c2.npz    decode_sample (un-guided) at B=256, L=200, S=128
  g25_traj_mc_m20.npz         controlled_sample with M=20 (configs[3]'s sampler shape, ConvGRU value net) at B=256, L=200, S=48
  -- round 5 (`python make_golden.py g26`, 8 torch threads, ~10 min) --
  g26_traj_dps_c5.npz         controlled_sample_DPS at configs[4]'s per-GPU shard: B=256, L=200, S=128, guidance scale 25600 (= g20's 300 at a
                              batch-mean reward): every state, x_0, the guided q_xs of 32 rows at 5 steps, per-row sums of q at every step
  g12_fullsize_probe.npz      FULL-SIZE reference nets (CNNModel hidden 128 x 4 stacks; ConvGRUTrunk 64 ch, n_conv 6 +
                              ConvHead) built at torch.manual_seed(44) in the order svdd_amd/synthetic.py builds them,
                              evaluated on 4 probe rows: logits, value scores, a checksum of every parameter tensor
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")
from _ref_import import import_reference, make_cfg  # noqa: E402

dg, En = import_reference()
torch.set_num_threads(1)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def tiny_diffusion(length, steps, seed=44):
    torch.manual_seed(seed)
    return dg.Diffusion(make_cfg(length=length, hidden_dim=16, num_cnn_stacks=1, steps=steps)).eval()


def tiny_value(seed=45):
    torch.manual_seed(seed)
    emb = En.ConvGRUTrunk(stem_in_channels=4, stem_channels=8, stem_kernel_size=15, n_conv=3, channel_init=8,
                          channel_mult=1, kernel_size=5, act_func="relu", conv_norm=True, pool_func=None,
                          pool_size=None, residual=True, crop_len=0, n_gru=1, dropout=0.1, gru_norm=True).eval()
    head = En.ConvHead(n_tasks=1, in_channels=8, act_func=None, pool_func="avg", norm=False).eval()
    # non-trivial BatchNorm statistics so eval-mode BN is exercised
    g = torch.Generator().manual_seed(seed)
    for m in emb.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    return emb, head


def sd_np(prefix, module):
    return {f"{prefix}.{k}": v.detach().numpy() for k, v in module.state_dict().items()}


# ----------------------------------------------------------------------------- G9
def g9():
    out = {}
    for seed, n in [(0, 140), (44, 5000), (123456789, 256000)]:
        torch.manual_seed(seed)
        r = torch.rand(n)
        out[f"torch_s{seed}_n{n}_head"] = r[:256]
        out[f"torch_s{seed}_n{n}_tail"] = r[-256:]
        out[f"torch_s{seed}_n{n}_sum"] = np.float64(r.double().sum().item())
    # rand_like on a [B,L,5] tensor consumes the same row-major stream
    torch.manual_seed(7)
    a = torch.rand_like(torch.empty(3, 50, 5))
    torch.manual_seed(7)
    b = torch.rand(3 * 50 * 5).view(3, 50, 5)
    assert torch.equal(a, b)
    out["torch_s7_randlike_3_50_5"] = a
    for seed, n in [(0, 10), (44, 1000)]:
        np.random.seed(seed)
        out[f"numpy_s{seed}_n{n}"] = np.random.random_sample(n)
    save("g9_rng.npz", **out)


# ----------------------------------------------------------------------------- G1
def g1():
    torch.manual_seed(1)
    B, L = 4, 200
    p = torch.softmax(torch.randn(B, L, 4) * 2.0, -1) * 0.0078
    q = torch.cat([p, torch.full((B, L, 1), 0.91)], -1)
    # rows with the unmasked pattern (all zero except own token + MASK entry)
    un = torch.rand(B, L) < 0.4
    tok = torch.randint(0, 4, (B, L))
    qz = torch.zeros_like(q)
    qz.scatter_(2, tok[..., None], 0.0078)
    qz[..., 4] = 0.91
    q = torch.where(un[..., None], qz, q).contiguous()
    torch.manual_seed(11)
    u = torch.rand_like(q)
    torch.manual_seed(11)
    tokens = dg._sample_categorical(q)
    save("g1_sample_categorical.npz", q=q, u=u, tokens=tokens, seed=11)


# ----------------------------------------------------------------------------- G2 / G4
def g2_g4(d):
    torch.manual_seed(2)
    B, L = 4, 200
    logits = torch.randn(B, L, 5) * 3.0
    logits[0, :10] = 0.0                     # exact ties
    logits[1, :10, :4] = torch.tensor([1.0, 1.0, 0.5, 1.0])
    xt = torch.where(torch.rand(B, L) < 0.5, torch.randint(0, 4, (B, L)), torch.full((B, L), 4))
    logp = d._subs_parameterization(logits.clone(), xt)
    save("g2_subs.npz", logits=logits, xt=xt, logp=logp)
    oh = d.transform_samples(xt)
    save("g4_transform.npz", tokens=xt, onehot=oh)


def sched_rows(d, S, eps=1e-5):
    """[S, 6] fp32 rows (t_i, sigma_t, sigma_s, mct, mcs, mct - mcs) exactly as the per-step prologue computes them."""
    timesteps = torch.linspace(1, eps, S + 1)
    dt = (1 - eps) / S
    rows = []
    for i in range(S):
        t = timesteps[i] * torch.ones(2, 1)
        sigma_t, _ = d.noise(t)
        sigma_s, _ = d.noise(t - dt)
        mct = 1 - torch.exp(-sigma_t.squeeze(-1))
        mcs = 1 - torch.exp(-sigma_s.squeeze(-1))
        rows.append([t[0, 0].item(), sigma_t[0, 0].item(), sigma_s[0, 0].item(), mct[0].item(), mcs[0].item(),
                     (mct - mcs)[0].item()])
    return np.asarray(rows, dtype=np.float32)


# ----------------------------------------------------------------------------- G3
def g3(d):
    out = {}
    for S in (128, 16, 8):
        eps = 1e-5
        timesteps = torch.linspace(1, eps, S + 1)
        dt = (1 - eps) / S
        rows = []
        for i in range(S):
            t = timesteps[i] * torch.ones(2, 1)
            sigma_t, _ = d.noise(t)
            sigma_s, _ = d.noise(t - dt)
            mct = 1 - torch.exp(-sigma_t.squeeze(-1))
            mcs = 1 - torch.exp(-sigma_s.squeeze(-1))
            rows.append([t[0, 0].item(), sigma_t[0, 0].item(), sigma_s[0, 0].item(), mct[0].item(), mcs[0].item(),
                         (mct - mcs)[0].item()])
        out[f"S{S}"] = np.asarray(rows, dtype=np.float32)
        out[f"S{S}_t_last"] = np.float32(timesteps[-1].item())
        out[f"S{S}_sigma_last"] = np.float32(d.noise(timesteps[-1] * torch.ones(1, 1))[0].item())
    save("g3_schedule.npz", **out)


# ------------------------------------------------------------- recording wrappers
class RecBackbone(torch.nn.Module):
    """Wraps the reference backbone; records (x, raw logits) of every call."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner
        self.calls = []

    def forward(self, x, sigma, *a, **k):
        out = self.inner(x, sigma, *a, **k)
        self.calls.append((x.clone(), out.detach().clone()))
        return out


class FixedBackbone(torch.nn.Module):
    def __init__(self, logits):
        super().__init__()
        self.logits = logits

    def forward(self, x, sigma):
        return self.logits.clone()


class RecCallable:
    def __init__(self, fn):
        self.fn = fn
        self.inputs, self.outputs = [], []

    def __call__(self, x):
        self.inputs.append(x.detach().clone())
        y = self.fn(x)
        self.outputs.append(y.detach().clone())
        return y

    def eval(self):
        return self


# ----------------------------------------------------------------------------- G5
def g5(d, bvl=False):
    """bvl=False: backbone output contiguous [B,L,5] (stream order [m][b][l][v]);
    bvl=True: backbone output is a permuted view of a [B,5,L] buffer, as the reference's CNNModel
    returns (models/dnaconv.py:201) -> rand_like fills in memory order [m][b][v][l]."""
    torch.manual_seed(6 if bvl else 5)
    B, L, M = 8, 200, 6
    logits = torch.randn(B, L, 5) * 1.5
    x = torch.where(torch.rand(B, L) < 0.45, torch.randint(0, 4, (B, L)), torch.full((B, L), 4))
    x[7] = 4
    scores = torch.randn(B, M) * 0.3
    scores[1] = 0.25                                  # all tied -> index 0
    scores[2, 4] = scores[2].max() + 0.5              # clear winner
    scores[3, 2] = scores[3, 5] = scores[3].max() + 1.0   # exact two-way tie -> lowest index
    scores[4] = torch.tensor([0.1, 0.1 + 2 ** -27, 0.1, 0.1 + 2 ** -27, 0.1, 0.1])  # 1-ulp near tie
    scores[5] = scores[5] * 50.0                      # large spread
    scores[6] = -30.0 + torch.arange(M) * 1e-3        # offset
    inner = d.backbone
    if bvl:
        logits = logits.permute(0, 2, 1).contiguous().permute(0, 2, 1)
        assert logits.stride() == (5 * L, 1, L)
    d.backbone = FixedBackbone(logits)
    emb = RecCallable(lambda t: t)
    calls = {"n": 0}

    def head_fn(t):
        m = calls["n"]
        calls["n"] += 1
        return scores[:, m].clone().view(B, 1, 1)

    head = RecCallable(head_fn)
    S, eps = 128, 1e-5
    i = 40
    t = torch.linspace(1, eps, S + 1)[i] * torch.ones(B, 1)
    dt = (1 - eps) / S
    torch.manual_seed(55)
    uniforms = torch.rand(M, B, 5, L) if bvl else torch.rand(M, B, L, 5)   # stream (= memory) order
    torch.manual_seed(55)
    x_next, x_in, q_xs, copy_flag = d._ddpm_update_finetune_controlled(x, t, dt, emb, head, repeats=M)
    assert q_xs.stride() == logits.stride()
    cand = torch.stack([oh.argmax(-1) * (oh.sum(-1) > 0) + 4 * (oh.sum(-1) == 0) for oh in emb.inputs], 1)
    soft = torch.softmax(scores, 1)
    idx = torch.argmax(soft, 1)
    sigma_t, _ = d.noise(t)
    sigma_s, _ = d.noise(t - dt)
    mct = (1 - torch.exp(-sigma_t.squeeze(-1)))[0]
    mcs = (1 - torch.exp(-sigma_s.squeeze(-1)))[0]
    d.backbone = inner
    save("g5_step_mc_bvl.npz" if bvl else "g5_step_mc.npz", logits=logits, x=x, scores=scores, uniforms=uniforms, seed=55, mct=mct, mcs=mcs,
         dm=(mct - mcs), q_xs=q_xs, copy_flag=copy_flag, cand=cand.to(torch.uint8),
         onehot=torch.stack(emb.inputs, 1), soft=soft, idx=idx, x_next=x_next)


# ----------------------------------------------------------------------------- G6
def traj_mc(name, L, S, B, M, seed):
    d = tiny_diffusion(L, S)
    emb_m, head_m = tiny_value()
    rec = RecBackbone(d.backbone)
    d.backbone = rec
    emb, head = RecCallable(emb_m), RecCallable(head_m)
    torch.manual_seed(seed)
    x0 = d.controlled_sample(emb, head, eval_sp_size=B, sample_M=M)
    xs = torch.stack([c[0] for c in rec.calls])                 # [S+1,B,L]
    logits = torch.stack([c[1] for c in rec.calls])             # [S+1,B,L,5]
    onehots = torch.stack(emb.inputs).view(S, M, B, L, 4)
    scores = torch.stack([o.squeeze() for o in head.outputs]).view(S, M, B).permute(0, 2, 1).contiguous()
    d.backbone = rec.inner
    save(name, xs=xs.to(torch.uint8), logits=logits, scores=scores,
         cand=(onehots.argmax(-1) * (onehots.sum(-1) > 0) + 4 * (onehots.sum(-1) == 0)).permute(0, 2, 1, 3).to(torch.uint8),
         x0=x0, seed=seed, B=B, L=L, M=M, S=S)
    return d, emb_m, head_m


# ----------------------------------------------------------------------------- G7
class RewardWrap(torch.nn.Module):
    """reward_model(x[B,4,L]) -> [B, n_tasks, 1] built from the reference's own OriBaseModel."""

    def __init__(self, emb, head):
        super().__init__()
        self.m = En.OriBaseModel(embedding=emb, head=head)

    def forward(self, x):
        return self.m(x)


def traj_pm(seed=3):
    L, S, B, M = 50, 8, 3, 3
    d = tiny_diffusion(L, S)
    emb_m, head_m = tiny_value()
    reward = RewardWrap(emb_m, head_m).eval()
    rec = RecBackbone(d.backbone)
    d.backbone = rec
    rr = RecCallable(reward)
    torch.manual_seed(seed)
    x0 = d.controlled_sample_tweedie(rr, eval_sp_size=B, sample_M=M, options="True", task="dna")
    # per step: 1 + M backbone calls; final: 1
    calls = rec.calls
    xs, logits, cx, cl = [], [], [], []
    k = 0
    for _ in range(S):
        xs.append(calls[k][0]); logits.append(calls[k][1]); k += 1
        cx.append(torch.stack([calls[k + m][0] for m in range(M)], 1))
        cl.append(torch.stack([calls[k + m][1] for m in range(M)], 1))
        k += M
    xs.append(calls[k][0]); logits.append(calls[k][1])
    scores = torch.stack([o[:, 0].squeeze() for o in rr.outputs]).view(S, M, B).permute(0, 2, 1).contiguous()
    x0hat_oh = torch.stack(rr.inputs).view(S, M, B, 4, L).permute(0, 2, 1, 3, 4).contiguous()
    d.backbone = rec.inner
    save("g7_traj_pm.npz", xs=torch.stack(xs).to(torch.uint8), logits=torch.stack(logits),
         cand=torch.stack(cx).to(torch.uint8), cand_logits=torch.stack(cl), x0hat_onehot_t=x0hat_oh,
         scores=scores, x0=x0, seed=seed, B=B, L=L, M=M, S=S)


# ----------------------------------------------------------------------------- G8
def traj_tds(seed=4, np_seed=9, alpha=0.5, full=False):
    if full:                                            # g19: the reference's full-size classes, seed 44 (synthetic.build's order)
        L, S, B = 200, 10, 8
        d, emb_v, head_v = full_nets(length=L, steps=S)
        emb_m, head_m = full_reward()
    else:
        L, S, B = 50, 8, 6
        d = tiny_diffusion(L, S)
        emb_m, head_m = tiny_value()
    reward = RewardWrap(emb_m, head_m).eval()
    rec = RecBackbone(d.backbone)
    d.backbone = rec
    rr = RecCallable(reward)
    torch.manual_seed(seed)
    np.random.seed(np_seed)
    x0 = d.controlled_sample_TDS(rr, alpha, eval_sp_size=B)
    calls = rec.calls
    xs, logits, samples, s_logits, den_logits = [], [], [], [], []
    for i in range(S):
        xs.append(calls[3 * i][0]); logits.append(calls[3 * i][1])
        samples.append(calls[3 * i + 1][0]); s_logits.append(calls[3 * i + 1][1])
        den_logits.append(calls[3 * i + 2][1])
    xs.append(calls[3 * S][0]); logits.append(calls[3 * S][1])
    num = torch.stack([rr.outputs[2 * i][:, 0][:, 0] for i in range(S)])
    den = torch.stack([rr.outputs[2 * i + 1][:, 0][:, 0] for i in range(S)])
    np.random.seed(np_seed)
    choice_u = np.random.random_sample(S * B).reshape(S, B)
    d.backbone = rec.inner
    extra = {}
    if full:
        extra = {n_ + "_param_sums": np.array([float(p.double().sum()) for p in mod.state_dict().values()])
                 for n_, mod in (("backbone", d.backbone), ("embedding", emb_v), ("head", head_v), ("reward_embedding", emb_m),
                                 ("reward_head", head_m))}
        extra["sched"] = sched_rows(d, S)
    save("g19_traj_tds_full.npz" if full else "g8_traj_tds.npz", xs=torch.stack(xs).to(torch.uint8), logits=torch.stack(logits),
         samples=torch.stack(samples).to(torch.uint8), sample_logits=torch.stack(s_logits),
         den_logits=torch.stack(den_logits), num=num, den=den, choice_u=choice_u, x0=x0, alpha=alpha,
         seed=seed, np_seed=np_seed, B=B, L=L, S=S, **extra)


# ----------------------------------------------------------------------------- G10
def g10(seed=6):
    L, S, B = 50, 16, 3
    d = tiny_diffusion(L, S)
    rec = RecBackbone(d.backbone)
    d.backbone = rec
    torch.manual_seed(seed)
    x0 = d.decode_sample(eval_sp_size=B)
    save("g10_decode_sample.npz", xs=torch.stack([c[0] for c in rec.calls]).to(torch.uint8),
         logits=torch.stack([c[1] for c in rec.calls]), x0=x0, seed=seed, B=B, L=L, S=S)
    d.backbone = rec.inner


# ----------------------------------------------------------------------------- G11
def traj_dps(seed=12, scale=50.0, full=False):
    """controlled_sample_DPS (diffusion_gosai.py:980-1019, 1286-1330), S=6, B=3, L=50: per step the guided
    q_xs handed to _sample_categorical, the uniforms it drew and the result. full: g20, the full-size classes at L = 200."""
    if full:
        L, S, B = 200, 4, 3
        d, emb_v, head_v = full_nets(length=L, steps=S)
        emb_m, head_m = full_reward()
    else:
        L, S, B = 50, 6, 3
        d = tiny_diffusion(L, S)
        emb_m, head_m = tiny_value()
    reward = RewardWrap(emb_m, head_m).eval()
    rec = {"q": [], "u": [], "x": [], "grad": []}
    orig_sc, orig_rl, orig_grad = dg._sample_categorical, torch.rand_like, d.compute_gradient_DPS

    def rl(t, *a, **k):
        r = orig_rl(t, *a, **k)
        rec["u"].append(r.detach().clone())
        return r

    def sc(q):
        rec["q"].append(q.detach().clone())
        return orig_sc(q)

    def grad(x_onehot, x, reward_model, sigma_s, copy_flag):
        g = orig_grad(x_onehot, x, reward_model, sigma_s, copy_flag)
        rec["x"].append(x.clone())
        rec["grad"].append(g.clone())
        return g

    dg._sample_categorical, torch.rand_like, d.compute_gradient_DPS = sc, rl, grad
    try:
        torch.manual_seed(seed)
        x0 = d.controlled_sample_DPS(reward, scale, eval_sp_size=B)
    finally:
        dg._sample_categorical, torch.rand_like, d.compute_gradient_DPS = orig_sc, orig_rl, orig_grad
    q = torch.stack(rec["q"])
    print("DPS q_xs strides", rec["q"][0].stride(), "u strides", rec["u"][0].stride())
    extra = {}
    if full:
        extra = {n_ + "_param_sums": np.array([float(p.double().sum()) for p in mod.state_dict().values()])
                 for n_, mod in (("backbone", d.backbone), ("embedding", emb_v), ("head", head_v), ("reward_embedding", emb_m),
                                 ("reward_head", head_m))}
    save("g20_traj_dps_full.npz" if full else "g11_traj_dps.npz", xs=torch.stack(rec["x"]).to(torch.uint8), q=q,
         q_is_bvl=int(rec["q"][0].stride()[1] == 1), u=torch.stack(rec["u"]), grad=torch.stack(rec["grad"]), x0=x0, seed=seed,
         scale=scale, B=B, L=L, S=S, **extra)


def nets():
    d200 = tiny_diffusion(200, 128)
    emb_m, head_m = tiny_value()
    arrs = {}
    arrs.update(sd_np("backbone", d200.backbone))
    arrs.update(sd_np("embedding", emb_m))
    arrs.update(sd_np("head", head_m))
    # one forward of each for net-parity tests
    torch.manual_seed(8)
    x = torch.where(torch.rand(3, 200) < 0.5, torch.randint(0, 4, (3, 200)), torch.full((3, 200), 4))
    with torch.no_grad():
        arrs["probe_x"] = x.numpy()
        arrs["probe_logits"] = d200.backbone(x, torch.zeros(3)).numpy()
        oh = d200.transform_samples(x).float()
        arrs["probe_value"] = head_m(emb_m(oh)).numpy()
        arrs["probe_reward"] = RewardWrap(emb_m, head_m).eval()(oh.transpose(1, 2)).numpy()
    save("nets_tiny.npz", **arrs)


def g12_fullsize_probe(seed=44):
    """The nets of BASELINE.json's configs at full size, random-initialised exactly like svdd_amd.synthetic.build
    (torch.manual_seed(44) — decode.py:181's default — then Diffusion, value trunk, value head, in that order), run by the
    REFERENCE's own modules. Pins the full-size architecture (20 dilated layers, n_conv = 6, GRU, FFN, head) and the
    parameter creation order of the build's mirrors, which the tiny fixtures cannot."""
    torch.manual_seed(seed)
    d = dg.Diffusion(make_cfg(length=200, hidden_dim=128, num_cnn_stacks=4, steps=128)).eval()
    emb = En.ConvGRUTrunk(stem_in_channels=4, stem_channels=64, stem_kernel_size=15, n_conv=6, channel_init=64,
                          channel_mult=1, kernel_size=5, act_func="relu", conv_norm=True, pool_func=None,
                          pool_size=None, residual=True, crop_len=0, n_gru=1, dropout=0.1, gru_norm=True).eval()
    head = En.ConvHead(n_tasks=1, in_channels=64, act_func=None, pool_func="avg", norm=False).eval()
    torch.manual_seed(8)
    x = torch.where(torch.rand(4, 200) < 0.6, torch.full((4, 200), 4), torch.randint(0, 4, (4, 200)))
    x[3] = 4                                                     # the all-MASK prior row
    arrs = {"x": x.numpy().astype(np.uint8), "seed": seed}
    with torch.no_grad():
        arrs["logits"] = d.backbone(x, torch.zeros(4)).numpy()                      # raw backbone output [4, 200, 5]
        arrs["logp"] = d.forward(x, torch.zeros(4)).numpy()                         # after _subs_parameterization
        oh = d.transform_samples(x).float()
        arrs["value"] = head(emb(oh)).reshape(-1).numpy()
    for name, mod in (("backbone", d.backbone), ("embedding", emb), ("head", head)):
        arrs[name + "_param_sums"] = np.array([float(p.double().sum()) for p in mod.state_dict().values()])
    save("g12_fullsize_probe.npz", **arrs)


def full_nets(seed=44, length=200, steps=128):
    """The reference's own classes at full size, initialised exactly like g12 / svdd_amd.synthetic.build."""
    torch.manual_seed(seed)
    d = dg.Diffusion(make_cfg(length=length, hidden_dim=128, num_cnn_stacks=4, steps=steps)).eval()
    emb = En.ConvGRUTrunk(stem_in_channels=4, stem_channels=64, stem_kernel_size=15, n_conv=6, channel_init=64,
                          channel_mult=1, kernel_size=5, act_func="relu", conv_norm=True, pool_func=None,
                          pool_size=None, residual=True, crop_len=0, n_gru=1, dropout=0.1, gru_norm=True).eval()
    head = En.ConvHead(n_tasks=1, in_channels=64, act_func=None, pool_func="avg", norm=False).eval()
    return d, emb, head


def full_reward():
    """The reward model synthetic.build creates right after (backbone, embedding, head): a second ConvGRU trunk + head drawn
    from the same RNG stream (stand-in for the gReLU oracle, Enformer.py:103-131)."""
    emb = En.ConvGRUTrunk(stem_in_channels=4, stem_channels=64, stem_kernel_size=15, n_conv=6, channel_init=64,
                          channel_mult=1, kernel_size=5, act_func="relu", conv_norm=True, pool_func=None,
                          pool_size=None, residual=True, crop_len=0, n_gru=1, dropout=0.1, gru_norm=True).eval()
    head = En.ConvHead(n_tasks=1, in_channels=64, act_func=None, pool_func="avg", norm=False).eval()
    return emb, head


def g18_traj_pm_full_rna(seed=21):
    """BASELINE configs[2]'s sampler (controlled_sample_tweedie, options="True": diffusion_gosai.py:1105-1145, 1373-1460) run by
    the reference with FULL-SIZE nets at L = 50: per step one backbone call on x_t and M on the candidates, the reward model on
    the x0-hat one-hots. Weights are not stored: seed 44 in synthetic.build("rna")'s order reproduces them (parameter sums kept)."""
    L, S, B, M = 50, 16, 4, 4
    d, emb_v, head_v = full_nets(length=L, steps=S)
    emb_r, head_r = full_reward()
    reward = RewardWrap(emb_r, head_r).eval()
    rec = RecBackbone(d.backbone)
    d.backbone = rec
    rr = RecCallable(reward)
    torch.manual_seed(seed)
    x0 = d.controlled_sample_tweedie(rr, eval_sp_size=B, sample_M=M, options="True", task="dna")
    calls = rec.calls
    xs, logits, cx, cl = [], [], [], []
    k = 0
    for _ in range(S):
        xs.append(calls[k][0]); logits.append(calls[k][1]); k += 1
        cx.append(torch.stack([calls[k + m][0] for m in range(M)], 1))
        cl.append(torch.stack([calls[k + m][1] for m in range(M)], 1))
        k += M
    xs.append(calls[k][0]); logits.append(calls[k][1])
    assert k + 1 == len(calls)
    scores = torch.stack([o[:, 0].squeeze() for o in rr.outputs]).view(S, M, B).permute(0, 2, 1).contiguous()
    x0hat_oh = torch.stack(rr.inputs).view(S, M, B, 4, L).permute(0, 2, 1, 3, 4).contiguous()
    d.backbone = rec.inner
    arrs = {n_ + "_param_sums": np.array([float(p.double().sum()) for p in mod.state_dict().values()])
            for n_, mod in (("backbone", d.backbone), ("embedding", emb_v), ("head", head_v), ("reward_embedding", emb_r),
                            ("reward_head", head_r))}
    save("g18_traj_pm_full_rna.npz", xs=torch.stack(xs).to(torch.uint8), logits=torch.stack(logits),
         cand=torch.stack(cx).to(torch.uint8), cand_logits=torch.stack(cl), x0hat_onehot_t=x0hat_oh.to(torch.uint8),
         scores=scores, x0=x0, seed=seed, net_seed=44, B=B, L=L, M=M, S=S, sched=sched_rows(d, S), **arrs)


def g13_traj_mc_full(name, S, B, M, seed, L=200):
    """A whole reference controlled_sample run (diffusion_gosai.py:1021-1061, 1174-1228) with the FULL-SIZE random-init
    nets (the ones bench.py times), recording every step: the state x_t, the raw backbone output, the M candidates and
    their value scores. The -m gpu tests feed these states to the hand-written net kernels (teacher forcing) and compare
    every step's logits / scores; the weights are not stored: seed 44 in synthetic.build's order reproduces them (g12)."""
    d, emb_m, head_m = full_nets(steps=S, length=L)
    rec = RecBackbone(d.backbone)
    d.backbone = rec
    emb, head = RecCallable(emb_m), RecCallable(head_m)
    torch.manual_seed(seed)
    x0 = d.controlled_sample(emb, head, eval_sp_size=B, sample_M=M)
    xs = torch.stack([c[0] for c in rec.calls])
    logits = torch.stack([c[1] for c in rec.calls])
    onehots = torch.stack(emb.inputs).view(S, M, B, L, 4)
    scores = torch.stack([o.squeeze() for o in head.outputs]).view(S, M, B).permute(0, 2, 1).contiguous()
    d.backbone = rec.inner
    arrs = {n_ + "_param_sums": np.array([float(p.double().sum()) for p in mod.state_dict().values()])
            for n_, mod in (("backbone", d.backbone), ("embedding", emb_m), ("head", head_m))}
    save(name, xs=xs.to(torch.uint8), logits=logits, scores=scores,
         cand=(onehots.argmax(-1) * (onehots.sum(-1) > 0) + 4 * (onehots.sum(-1) == 0)).permute(0, 2, 1, 3).to(torch.uint8),
         x0=x0, seed=seed, net_seed=44, B=B, L=L, M=M, S=S, sched=sched_rows(d, S), **arrs)


def g14_traj_pm_heuristic(seed=13):
    """controlled_sample_tweedie called the way decode_tweedie.py calls it — options left at its default, the bool True —
    which the string compare at :1414 sends down the heuristic branch :1420-1424 (reward of the raw x_t with MASK rows
    zero; no candidate backbone forward)."""
    L, S, B, M = 50, 8, 3, 3
    d = tiny_diffusion(L, S)
    emb_m, head_m = tiny_value()
    reward = RewardWrap(emb_m, head_m).eval()
    rec = RecBackbone(d.backbone)
    d.backbone = rec
    rr = RecCallable(reward)
    torch.manual_seed(seed)
    x0 = d.controlled_sample_tweedie(rr, eval_sp_size=B, sample_M=M)
    assert len(rec.calls) == S + 1                     # one backbone call per step: no Tweedie forward in this branch
    xs = torch.stack([c[0] for c in rec.calls])
    logits = torch.stack([c[1] for c in rec.calls])
    scores = torch.stack([o[:, 0].squeeze() for o in rr.outputs]).view(S, M, B).permute(0, 2, 1).contiguous()
    oh = torch.stack(rr.inputs).view(S, M, B, 4, L)    # what the reward model saw: [B,4,L] one-hot of the candidate
    cand = (oh.argmax(3) * (oh.sum(3) > 0) + 4 * (oh.sum(3) == 0)).permute(0, 2, 1, 3).to(torch.uint8)
    d.backbone = rec.inner
    save("g14_traj_pm_heuristic.npz", xs=xs.to(torch.uint8), logits=logits, cand=cand, scores=scores, x0=x0, seed=seed,
         B=B, L=L, M=M, S=S)


def g15_step(d, M):
    """G5 at the sample widths the reference actually runs (M = 10: decode.py's default; M = 20: BASELINE configs[3]),
    reference memory layout ([B,5,L] buffer behind the logits). Scores hold exact ties, 1-ulp near-ties spread over all
    M columns and |s| < 0.5 rows where distinct scores can collapse after exp / sum."""
    torch.manual_seed(150 + M)
    B, L = 12, 200
    logits = (torch.randn(B, L, 5) * 1.5).permute(0, 2, 1).contiguous().permute(0, 2, 1)
    x = torch.where(torch.rand(B, L) < 0.45, torch.randint(0, 4, (B, L)), torch.full((B, L), 4))
    x[7] = 4
    scores = torch.randn(B, M) * 0.3
    scores[1] = 0.25                                                   # all tied -> index 0
    scores[2, M - 2] = scores[2].max() + 0.5                            # clear winner near the end
    scores[3, 2] = scores[3, M - 1] = scores[3].max() + 1.0             # exact two-way tie -> lowest index
    base = torch.full((M,), 0.1)
    base[1::2] = 0.1 + 2 ** -27                                        # alternating 1-ulp pattern
    scores[4] = base
    scores[5] = scores[5] * 50.0                                       # large spread
    scores[6] = -30.0 + torch.arange(M) * 1e-3                         # offset
    scores[8] = torch.randn(M) * 1e-7                                  # near-uniform: what random-init value nets give
    scores[9] = 0.3 + torch.randint(-1, 2, (M,)).float() * 2 ** -25    # +-1 ulp jitter around one value
    scores[10] = torch.nextafter(torch.full((M,), 0.4), torch.tensor(1.0))
    scores[10, M // 2:] = 0.4                                          # two plateaus one ulp apart, larger one first
    scores[11] = torch.flip(scores[10], dims=[0])                      # ... larger one last
    inner = d.backbone
    d.backbone = FixedBackbone(logits)
    emb = RecCallable(lambda t: t)
    calls = {"n": 0}

    def head_fn(t):
        m = calls["n"]
        calls["n"] += 1
        return scores[:, m].clone().view(B, 1, 1)

    head = RecCallable(head_fn)
    S, eps = 128, 1e-5
    t = torch.linspace(1, eps, S + 1)[70] * torch.ones(B, 1)
    dt = (1 - eps) / S
    torch.manual_seed(56)          # the M x rand_like(q_xs) draws are torch.rand(M, B, 5, L) of this seed (pinned by g9 / g5)
    x_next, x_in, q_xs, copy_flag = d._ddpm_update_finetune_controlled(x, t, dt, emb, head, repeats=M)
    cand = torch.stack([oh.argmax(-1) * (oh.sum(-1) > 0) + 4 * (oh.sum(-1) == 0) for oh in emb.inputs], 1)
    soft = torch.softmax(scores, 1)
    idx = torch.argmax(soft, 1)
    sigma_t, _ = d.noise(t)
    sigma_s, _ = d.noise(t - dt)
    mct = (1 - torch.exp(-sigma_t.squeeze(-1)))[0]
    mcs = (1 - torch.exp(-sigma_s.squeeze(-1)))[0]
    d.backbone = inner
    save(f"g15_step_mc_m{M}.npz", logits=logits, x=x, scores=scores, seed=56, mct=mct, mcs=mcs,
         dm=(mct - mcs), q_xs=q_xs, cand=cand.to(torch.uint8), soft=soft, idx=idx, x_next=x_next)


# ----------------------------------------------------------------------------- G16 (DiT)
def install_flash_attn_standin():
    """models/dit.py:4-5 imports the CUDA-only flash_attn and uses two of its entry points (:115, :272). Stand-ins with the
    published semantics of flash-attn 2.x, in plain fp32 torch ops (no fused kernel):
      layers.rotary.apply_rotary_emb_qkv_(qkv[b,s,3,h,d], cos[s,d/2], sin[s,d/2])  non-interleaved rotary on q and k, in place
      flash_attn_interface.flash_attn_varlen_qkvpacked_func(qkv[(b s),3,h,d], cu_seqlens, max_seqlen, p, causal=False)
                                                                                 softmax(q k^T / sqrt(d)) v per sequence"""
    import importlib.machinery
    import types

    def mod(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m
        return m

    def apply_rotary_emb_qkv_(qkv, cos, sin):
        half = cos.shape[-1]
        c, s_ = cos[None, :, None, :], sin[None, :, None, :]
        for i in (0, 1):
            x1, x2 = qkv[:, :, i, :, :half].clone(), qkv[:, :, i, :, half:2 * half].clone()
            qkv[:, :, i, :, :half] = x1 * c - x2 * s_
            qkv[:, :, i, :, half:2 * half] = x1 * s_ + x2 * c
        return qkv

    def flash_attn_varlen_qkvpacked_func(qkv, cu_seqlens, max_seqlen, dropout_p=0.0, softmax_scale=None, causal=False):
        assert dropout_p == 0.0 and not causal
        out = torch.empty_like(qkv[:, 0])
        scale = softmax_scale if softmax_scale is not None else qkv.shape[-1] ** -0.5
        cu = cu_seqlens.tolist()
        for a, b in zip(cu[:-1], cu[1:]):
            q, k, v = (qkv[a:b, i].transpose(0, 1) for i in range(3))            # [h, s, d]
            out[a:b] = (torch.softmax(q @ k.transpose(1, 2) * scale, dim=-1) @ v).transpose(0, 1)
        return out

    fa = mod("flash_attn")
    fa.layers = mod("flash_attn.layers")
    fa.layers.rotary = mod("flash_attn.layers.rotary")
    fa.layers.rotary.apply_rotary_emb_qkv_ = apply_rotary_emb_qkv_
    fa.flash_attn_interface = mod("flash_attn.flash_attn_interface")
    fa.flash_attn_interface.flash_attn_varlen_qkvpacked_func = flash_attn_varlen_qkvpacked_func
    if "omegaconf" not in sys.modules:
        try:
            import omegaconf  # noqa: F401
        except ImportError:
            mod("omegaconf").OmegaConf = type("OmegaConf", (), {"create": staticmethod(lambda d: d)})


def g16_dit(seed=61):
    install_flash_attn_standin()
    import importlib
    dit = importlib.import_module("models.dit")            # /root/reference/models/dit.py, imported where it lies
    from _ref_import import Cfg
    hp = dict(hidden_size=32, cond_dim=16, n_blocks=2, n_heads=2, dropout=0.0, scale_by_sigma=True, length=24)
    torch.manual_seed(seed)
    m = dit.DIT(Cfg(model=hp), vocab_size=5).eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                                  # adaLN-zero / zero output map: re-draw, or the logits are all 0
        for mod_ in list(m.blocks) + [m.output_layer]:
            mod_.adaLN_modulation.weight.copy_(torch.randn(mod_.adaLN_modulation.weight.shape, generator=g) * 0.2)
            mod_.adaLN_modulation.bias.copy_(torch.randn(mod_.adaLN_modulation.bias.shape, generator=g) * 0.2)
        m.output_layer.linear.weight.copy_(torch.randn(m.output_layer.linear.weight.shape, generator=g) * 0.3)
        m.output_layer.linear.bias.copy_(torch.randn(5, generator=g) * 0.1)
    idx = torch.randint(0, 5, (3, hp["length"]), generator=g)
    sigma = torch.tensor([0.0, 0.37, 2.5])
    with torch.no_grad():
        logits = m(idx, sigma).float()
        logits0 = m(idx, torch.zeros(3)).float()           # what the sampler feeds with time_conditioning off
    assert float(logits.abs().max()) > 0.1
    save("g16_dit.npz", indices=idx.to(torch.uint8), sigma=sigma, logits=logits, logits_sigma0=logits0,
         hp=np.array([hp["hidden_size"], hp["cond_dim"], hp["n_blocks"], hp["n_heads"], hp["length"]]), **sd_np("dit", m))


# ----------------------------------------------------------------------------- G17 (Enformer-shaped trunk)
def g17_enformer_trunk(seed=71):
    """Enformer.py:8-9 imports GELU, AttentionPool, relative_shift, Attention, exponential_linspace_int from enformer_pytorch
    (absent offline, version unpinned). The reference module was imported with those names stubbed to None; here they are
    bound to adapters over svdd_amd.enformer_value's restatements and the reference's EnformerTrunk / ConvHead are run as
    they are."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from svdd_amd import enformer_value as ev

    class Attention(ev.RelPosAttention):
        def __init__(self, dim, heads, dim_key, dim_value, dropout=0.0, pos_dropout=0.0, num_rel_pos_features=None,
                     use_tf_gamma=False):
            assert not use_tf_gamma
            super().__init__(dim, heads, dim_key, dim_value, num_rel_pos_features)

    En.GELU, En.AttentionPool, En.Attention = ev.EnformerGELU, ev.AttentionPool, Attention
    En.relative_shift, En.exponential_linspace_int = ev._relative_shift, ev.exponential_linspace_int
    from seeded_weights import draw
    torch.manual_seed(seed)
    trunk = En.EnformerTrunk(n_conv=3, channels=384, n_transformers=2, n_heads=2, key_len=16).eval()
    head = En.ConvHead(n_tasks=1, in_channels=768, act_func=None, pool_func="avg", norm=False).eval()
    # 3.6 M parameters: not stored. Every tensor is re-drawn from a seeded generator in state_dict order (seeded_weights.py);
    # the test regenerates the same tensors from the recorded names / shapes.
    both = torch.nn.ModuleDict({"trunk": trunk, "head": head})
    sd = both.state_dict()
    names, shapes = list(sd), [tuple(v.shape) for v in sd.values()]
    both.load_state_dict(draw(names, shapes, seed), strict=True)
    g = torch.Generator().manual_seed(seed + 1)
    tok = torch.randint(0, 5, (3, 40), generator=g)
    x = torch.nn.functional.one_hot(tok.clamp(max=3), 4).float() * (tok < 4)[..., None]      # MASK rows all-zero (a8)
    with torch.no_grad():
        y = trunk(x)
        v = head(y)
    assert y.shape == (3, 768, 5) and v.shape == (3, 1, 1) and float(y.abs().max()) > 0.05
    maxr = max(len(sh) for sh in shapes)
    save("g17_enformer_trunk.npz", tokens=tok.to(torch.uint8), trunk_out=y, value=v, seed=seed,
         hp=np.array([3, 384, 2, 2, 16]), names=np.array(names),
         shapes=np.array([list(sh) + [-1] * (maxr - len(sh)) for sh in shapes], dtype=np.int64))


# ------------------------------------------------------------------ g21: the headline configs themselves (round 4)
def _delta(cand, parent):
    """cand u8 [B, M, L] against its parent x_t [B, L]: 255 where the candidate kept the parent's token (most positions)."""
    return torch.where(cand == parent[:, None, :], torch.full_like(cand, 255), cand)


class LeanRecBackbone(torch.nn.Module):
    """Records every call's input tokens as u8 and, for the call numbers in `keep`, the raw output."""

    def __init__(self, inner, keep=()):
        super().__init__()
        self.inner, self.keep = inner, set(keep)
        self.xs, self.logits = [], {}

    def forward(self, x, sigma, *a, **k):
        out = self.inner(x, sigma, *a, **k)
        n = len(self.xs)
        self.xs.append(x.to(torch.uint8))
        if n in self.keep:
            self.logits[n] = out.detach().clone()
        return out


def g21_traj_mc_c2(seed=0, B=256, L=200, M=10, S=128):
    """BASELINE.json configs[1] — the headline — run by the reference itself: controlled_sample (diffusion_gosai.py:1021-1061,
    1174-1228) with the full-size seed-44 nets at B = 256, L = 200, M = 10, 128 steps (≈ 5 min on the 8 build cores). Kept small:
    every step's state (u8), candidates (delta-coded against the state), value scores and selections; raw backbone outputs of
    three calls only (first, middle, the noise-removal call). Weights are not stored (seed 44 in synthetic.build's order; sums kept)."""
    d, emb_m, head_m = full_nets(steps=S, length=L)
    rec = LeanRecBackbone(d.backbone, keep=(0, S // 2, S))
    d.backbone = rec
    cands, scores = [], []

    def emb(x):                                                       # x: one-hot float [B, L, 4] of one candidate set
        s = x.sum(-1)
        cands.append(torch.where(s > 0, x.argmax(-1), torch.full_like(x.argmax(-1), 4)).to(torch.uint8))
        return emb_m(x)

    def head(h):
        y = head_m(h)
        scores.append(y.detach().squeeze().clone())
        return y

    torch.manual_seed(seed)
    x0 = d.controlled_sample(emb, head, eval_sp_size=B, sample_M=M)
    d.backbone = rec.inner
    xs = torch.stack(rec.xs)                                          # [S + 1, B, L]
    cand = torch.stack(cands).view(S, M, B, L).permute(0, 2, 1, 3).contiguous()
    sc = torch.stack(scores).view(S, M, B).permute(0, 2, 1).contiguous()
    idx = torch.softmax(sc, dim=2).argmax(dim=2).to(torch.uint8)      # :1219-1225
    assert torch.equal(torch.gather(cand, 2, idx.long()[:, :, None, None].expand(S, B, 1, L))[:, :, 0], xs[1:])
    arrs = {n_ + "_param_sums": np.array([float(p.double().sum()) for p in mod.state_dict().values()])
            for n_, mod in (("backbone", d.backbone), ("embedding", emb_m), ("head", head_m))}
    kept = sorted(rec.logits)
    save("g21_traj_mc_c2.npz", xs=xs, cand_delta=torch.stack([_delta(cand[s], xs[s]) for s in range(S)]), scores=sc, idx=idx,
         x0=x0.to(torch.uint8), logits_steps=np.array(kept), logits=torch.stack([rec.logits[k] for k in kept]),
         seed=seed, net_seed=44, B=B, L=L, M=M, S=S, threads=torch.get_num_threads(), sched=sched_rows(d, S), **arrs)


def g21_traj_pm_c3(seed=0, B=256, L=50, M=10, S=128):
    """BASELINE.json configs[2] run by the reference: controlled_sample_tweedie(options="True") (diffusion_gosai.py:1105-1145,
    1373-1460), full-size seed-44 nets + reward model at B = 256, L = 50, M = 10, 128 steps. Per step: state, candidates
    (delta-coded), the x0-hat rows handed to the reward model (as tokens), reward scores, selections."""
    d, emb_v, head_v = full_nets(length=L, steps=S)
    emb_r, head_r = full_reward()
    reward = RewardWrap(emb_r, head_r).eval()
    rec = LeanRecBackbone(d.backbone, keep=(0, S * (1 + M)))
    d.backbone = rec
    x0hats, scores = [], []

    class Rew:
        def __call__(self, x):                                        # [B, 4, L] one-hot float
            x0hats.append(x.argmax(1).to(torch.uint8))
            assert bool((x.sum(1) == 1).all())
            y = reward(x)
            scores.append(y[:, 0].detach().squeeze().clone())
            return y

        def eval(self):
            return self

    torch.manual_seed(seed)
    x0 = d.controlled_sample_tweedie(Rew(), eval_sp_size=B, sample_M=M, options="True", task="dna")
    d.backbone = rec.inner
    calls = rec.xs
    assert len(calls) == S * (1 + M) + 1
    xs = torch.stack([calls[s * (1 + M)] for s in range(S)] + [calls[-1]])
    cand = torch.stack([torch.stack([calls[s * (1 + M) + 1 + m] for m in range(M)], 1) for s in range(S)])   # [S, B, M, L]
    x0hat = torch.stack(x0hats).view(S, M, B, L).permute(0, 2, 1, 3).contiguous()
    sc = torch.stack(scores).view(S, M, B).permute(0, 2, 1).contiguous()
    idx = torch.softmax(sc, dim=2).argmax(dim=2).to(torch.uint8)
    assert torch.equal(torch.gather(cand, 2, idx.long()[:, :, None, None].expand(S, B, 1, L))[:, :, 0], xs[1:])
    arrs = {n_ + "_param_sums": np.array([float(p.double().sum()) for p in mod.state_dict().values()])
            for n_, mod in (("backbone", d.backbone), ("embedding", emb_v), ("head", head_v), ("reward_embedding", emb_r),
                            ("reward_head", head_r))}
    kept = sorted(rec.logits)
    save("g21_traj_pm_c3.npz", xs=xs, cand_delta=torch.stack([_delta(cand[s], xs[s]) for s in range(S)]),
         x0hat_delta=torch.where(x0hat == cand, torch.full_like(x0hat, 255), x0hat), scores=sc, idx=idx, x0=x0.to(torch.uint8),
         logits_calls=np.array(kept), logits=torch.stack([rec.logits[k] for k in kept]),
         seed=seed, net_seed=44, B=B, L=L, M=M, S=S, threads=torch.get_num_threads(), sched=sched_rows(d, S), **arrs)


def g23_traj_tds_c5(seed=0, np_seed=1, alpha=0.5, B=256, L=200, S=128):
    """BASELINE.json configs[4]'s SMC / TDS baseline at its per-GPU shard size, run by the reference: controlled_sample_TDS
    (diffusion_gosai.py:938-978, 1230-1284) with the full-size seed-44 nets + reward model, 256 particles, 128 steps (3 backbone
    forwards + 2 reward calls per step; ~8 min on the 8 build cores). Lean: states, proposals (delta-coded), the x0-hat rows handed
    to the reward model for the numerator (as tokens; the denominator's are the previous numerator's resampled), both reward
    vectors, the ancestor indices np.random.choice drew, x_0."""
    d, emb_v, head_v = full_nets(length=L, steps=S)
    emb_r, head_r = full_reward()
    reward = RewardWrap(emb_r, head_r).eval()
    rec = LeanRecBackbone(d.backbone, keep=(0, 3 * S))
    d.backbone = rec
    xh, outs, choices = [], [], []

    class Rew:
        def __call__(self, x):                                        # [B, 4, L] one-hot float
            assert bool((x.sum(1) == 1).all())
            xh.append(x.argmax(1).to(torch.uint8))
            y = reward(x)
            outs.append(y[:, 0][:, 0].detach().clone())
            return y

        def eval(self):
            return self

    real_choice = np.random.choice

    def choice(*a, **k):
        r = real_choice(*a, **k)
        choices.append(np.asarray(r).copy())
        return r

    torch.manual_seed(seed)
    np.random.seed(np_seed)
    np.random.choice = choice
    try:
        x0 = d.controlled_sample_TDS(Rew(), alpha, eval_sp_size=B)
    finally:
        np.random.choice = real_choice
    d.backbone = rec.inner
    calls = rec.xs
    assert len(calls) == 3 * S + 1 and len(outs) == 2 * S and len(choices) == S
    xs = torch.stack([calls[3 * i] for i in range(S)] + [calls[3 * S]])
    samples = torch.stack([calls[3 * i + 1] for i in range(S)])
    x0hat_num = torch.stack([xh[2 * i] for i in range(S)])
    num = torch.stack([outs[2 * i] for i in range(S)])
    den = torch.stack([outs[2 * i + 1] for i in range(S)])
    idx = np.stack(choices).astype(np.int32)
    assert torch.equal(torch.stack([samples[i][torch.from_numpy(idx[i]).long()] for i in range(S)]), xs[1:])
    np.random.seed(np_seed)
    choice_u = np.random.random_sample(S * B).reshape(S, B)
    arrs = {n_ + "_param_sums": np.array([float(p.double().sum()) for p in mod.state_dict().values()])
            for n_, mod in (("backbone", d.backbone), ("embedding", emb_v), ("head", head_v), ("reward_embedding", emb_r),
                            ("reward_head", head_r))}
    kept = sorted(rec.logits)
    save("g23_traj_tds_c5.npz", xs=xs, sample_delta=torch.where(samples == xs[:-1], torch.full_like(samples, 255), samples),
         x0hat_num_delta=torch.where(x0hat_num == samples, torch.full_like(samples, 255), x0hat_num), num=num, den=den, idx=idx,
         choice_u=choice_u, x0=x0.to(torch.uint8), logits_calls=np.array(kept), logits=torch.stack([rec.logits[k] for k in kept]),
         alpha=alpha, seed=seed, np_seed=np_seed, net_seed=44, B=B, L=L, S=S, threads=torch.get_num_threads(), sched=sched_rows(d, S), **arrs)


def g26_traj_dps_c5(seed=0, scale=25600.0, B=256, L=200, S=128, keep_q=(0, 32, 64, 96, 127), q_rows=32):
    """BASELINE.json configs[4]'s gradient-guidance baseline AT its per-GPU shard size, run by the reference: controlled_sample_DPS
    (diffusion_gosai.py:980-1019, 1286-1330) with the full-size seed-44 nets + reward model, B = 256, L = 200, 128 steps (a forward and
    a forward + backward through the backbone and the reward net per step; ~10 min on the 8 build cores). The reward is the MEAN over
    the batch (:1329), so a sample's gradient is 1 / B of its own: guidance scale 25600 = 300 * 256 / 3 gives the factors of g20 (up to
    ~1.06) and keeps scale * (fp32 autograd noise) ~ 3e-5, comparable at 1e-4. Lean: every state x_t, x_0, the guided q_xs of the first
    `q_rows` rows at the steps `keep_q`, the per-step max of q (a checksum over all rows); the uniforms are NOT stored — the test replays
    the mt19937 stream (one rand_like(q) per step, in q's memory order [b][v][l])."""
    d, emb_v, head_v = full_nets(length=L, steps=S)
    emb_m, head_m = full_reward()
    reward = RewardWrap(emb_m, head_m).eval()
    rec = {"q": {}, "x": [], "qmax": [], "qsum": []}
    orig_sc, orig_grad = dg._sample_categorical, d.compute_gradient_DPS

    def sc(q):
        i = len(rec["qmax"])
        if i in keep_q:
            rec["q"][i] = q[:q_rows].detach().clone()
        rec["qmax"].append(float(q.max()))
        rec["qsum"].append(q.detach().double().sum(dim=(1, 2)).float().clone())   # [B]: a per-row checksum of the guided weights (detached:
                                                                                  # q carries the autograd graph of a whole backbone forward)
        assert q.stride()[1] == 1                                             # [b][v][l] memory order (the CNN's permuted output)
        return orig_sc(q)

    def grad(x_onehot, x, reward_model, sigma_s, copy_flag):
        rec["x"].append(x.to(torch.uint8).clone())
        return orig_grad(x_onehot, x, reward_model, sigma_s, copy_flag)

    dg._sample_categorical, d.compute_gradient_DPS = sc, grad
    try:
        torch.manual_seed(seed)
        x0 = d.controlled_sample_DPS(reward, scale, eval_sp_size=B)
    finally:
        dg._sample_categorical, d.compute_gradient_DPS = orig_sc, orig_grad
    assert len(rec["x"]) == S and len(rec["qmax"]) == S
    arrs = {n_ + "_param_sums": np.array([float(p.double().sum()) for p in mod.state_dict().values()])
            for n_, mod in (("backbone", d.backbone), ("reward_embedding", emb_m), ("reward_head", head_m))}
    save("g26_traj_dps_c5.npz", xs=torch.stack(rec["x"]), x0=x0.to(torch.uint8), q_steps=np.array(sorted(rec["q"])),
         q=torch.stack([rec["q"][i] for i in sorted(rec["q"])]), q_rows=q_rows, qmax=np.array(rec["qmax"], np.float32),
         qsum=torch.stack(rec["qsum"]), seed=seed, net_seed=44, scale=scale, B=B, L=L, S=S, threads=torch.get_num_threads(),
         sched=sched_rows(d, S), **arrs)


def g24_decode_sample_c2(seed=0, B=256, L=200, S=128):
    """The un-guided ancestral decode (`decode_sample`, diffusion_gosai.py:888-936, 1147-1172: what the harness's baseline loop
    runs gen_batch_num * sample_M times) by the reference at the headline batch: B = 256, L = 200, 128 steps, full-size seed-44
    backbone. Every state + x_0 (+ the raw logits of the first and the noise-removal call)."""
    d, _, _ = full_nets(steps=S, length=L)
    rec = LeanRecBackbone(d.backbone, keep=(0, S))
    d.backbone = rec
    torch.manual_seed(seed)
    x0 = d.decode_sample(eval_sp_size=B)
    d.backbone = rec.inner
    assert len(rec.xs) == S + 1
    kept = sorted(rec.logits)
    save("g24_decode_sample_c2.npz", xs=torch.stack(rec.xs), x0=x0.to(torch.uint8), logits_steps=np.array(kept),
         logits=torch.stack([rec.logits[k] for k in kept]), seed=seed, net_seed=44, B=B, L=L, S=S, threads=torch.get_num_threads(),
         backbone_param_sums=np.array([float(p.double().sum()) for p in d.backbone.state_dict().values()]), sched=sched_rows(d, S))


def g25_traj_mc_m20():
    """BASELINE configs[3]'s sampler shape — SVDD-MC with M = 20 candidates per row — at the shard batch (B = 256, L = 200) by the
    reference, with the ConvGRU value net (the Enformer-shaped trunk of that config cannot be imported: enformer_pytorch is absent),
    48 steps of a 48-step schedule: K1 / K2 at M = 20 (the 32-lane candidate groups of the select kernel) on 245,760 reference
    candidates."""
    d, emb_m, head_m = full_nets(steps=48, length=200)
    S, B, L, M, seed = 48, 256, 200, 20, 3
    rec = LeanRecBackbone(d.backbone, keep=(0, S))
    d.backbone = rec
    cands, scores = [], []

    def emb(x):
        s_ = x.sum(-1)
        cands.append(torch.where(s_ > 0, x.argmax(-1), torch.full_like(x.argmax(-1), 4)).to(torch.uint8))
        return emb_m(x)

    def head(h):
        y = head_m(h)
        scores.append(y.detach().squeeze().clone())
        return y

    torch.manual_seed(seed)
    x0 = d.controlled_sample(emb, head, eval_sp_size=B, sample_M=M)
    d.backbone = rec.inner
    xs = torch.stack(rec.xs)
    cand = torch.stack(cands).view(S, M, B, L).permute(0, 2, 1, 3).contiguous()
    sc = torch.stack(scores).view(S, M, B).permute(0, 2, 1).contiguous()
    idx = torch.softmax(sc, dim=2).argmax(dim=2).to(torch.uint8)
    assert torch.equal(torch.gather(cand, 2, idx.long()[:, :, None, None].expand(S, B, 1, L))[:, :, 0], xs[1:])
    arrs = {n_ + "_param_sums": np.array([float(p.double().sum()) for p in mod.state_dict().values()])
            for n_, mod in (("backbone", d.backbone), ("embedding", emb_m), ("head", head_m))}
    kept = sorted(rec.logits)
    save("g25_traj_mc_m20.npz", xs=xs, cand_delta=torch.stack([_delta(cand[s_], xs[s_]) for s_ in range(S)]), scores=sc, idx=idx,
         x0=x0.to(torch.uint8), logits_steps=np.array(kept), logits=torch.stack([rec.logits[k] for k in kept]),
         seed=seed, net_seed=44, B=B, L=L, M=M, S=S, threads=torch.get_num_threads(), sched=sched_rows(d, S), **arrs)


def g21():
    # full-size batches: all build cores (the small fixtures above are generated single-threaded; g21 records the thread count)
    torch.set_num_threads(int(os.environ.get("SVDD_GOLDEN_THREADS", "8")))
    g21_traj_mc_c2()
    g21_traj_pm_c3()


# ------------------------------------------------------------------ g22: the reference's own harness (round 4)
def g22_harness(seed=17, L=50, S=16, B=3, G=2, M=3, alpha=0.5):
    """`Enformer.BaseModel.controlled_decode` (Enformer.py:399-477), `controlled_decode_tweedie` (:719-813) and
    `controlled_decode_TDS` (:479-557) THEMSELVES, with the tiny fixture nets (nets_tiny.npz's backbone / value net; a second
    tiny ConvGRU as the reward oracle, weights stored here). `BaseModel.__init__` hard-wires checkpoint paths, Hydra and
    `.cuda()` (:76-131), so the object is made with object.__new__ + the attributes the methods read; the gReLU checkpoint
    loader (`LightningModel.load_from_checkpoint`, :429-432) returns the fixture reward net and its `.cuda()` is a no-op on
    this CPU-only build. Records the 5-tuples: the RNG order (guided batches, then gen_batch_num * sample_M baseline batches),
    the `samples` container shape (a list of batches for MC / TDS, a flat list of rows for tweedie: `samples.extend`, :766)
    and `top_k_values` (a real top-k for MC / TDS, `cat(baseline_preds)` for tweedie, :802)."""
    d = tiny_diffusion(L, S)
    emb_m, head_m = tiny_value()
    emb_r, head_r = tiny_value(seed=46)
    reward = RewardWrap(emb_r, head_r).eval()
    reward.cuda = lambda *a, **k: reward
    En.LightningModel.load_from_checkpoint = staticmethod(lambda *a, **k: reward)
    bm = object.__new__(En.BaseModel)
    torch.nn.Module.__init__(bm)
    bm.task, bm.n_tasks, bm.saluki_body = "dna", 1, 0
    bm.embedding, bm.head, bm.ref_model, bm.reward_model = emb_m, head_m, d, reward
    bm.NUM_SAMPLES_PER_BATCH = B
    bm.eval()
    arrs = {}
    arrs.update(sd_np("reward_embedding", emb_r))
    arrs.update(sd_np("reward_head", head_r))
    for kind, call in (("mc", lambda: bm.controlled_decode(G, M)),
                       ("pm", lambda: bm.controlled_decode_tweedie(G, M, "True")),
                       ("pmh", lambda: bm.controlled_decode_tweedie(G, M, True)),
                       ("tds", lambda: bm.controlled_decode_TDS(G, M, alpha))):
        torch.manual_seed(seed)
        np.random.seed(seed + 1)
        samples, vf, rm, topk, base = call()
        arrs[kind + "_samples_len"] = len(samples)
        arrs[kind + "_samples_item_shape"] = np.array(samples[0].shape)
        arrs[kind + "_samples"] = torch.stack([s_ for s_ in samples]).to(torch.uint8)
        arrs[kind + "_value_func_preds"] = vf
        arrs[kind + "_reward_model_preds"] = rm
        arrs[kind + "_top_k"] = topk
        arrs[kind + "_baseline_preds"] = base
        for k_ in ("value_func_preds", "reward_model_preds", "top_k", "baseline_preds"):
            arrs[kind + "_" + k_ + "_shape"] = np.array(arrs[kind + "_" + k_].shape)
    save("g22_harness.npz", seed=seed, np_seed=seed + 1, L=L, S=S, B=B, G=G, M=M, alpha=alpha, **arrs)


def g18():
    g13_traj_mc_full("g18_traj_mc_full_rna.npz", S=24, B=6, M=5, seed=5, L=50)
    g18_traj_pm_full_rna()
    traj_tds(seed=31, np_seed=32, full=True)            # g19
    traj_dps(seed=41, scale=300.0, full=True)           # g20 (scale * autograd noise ~ 3e-5: q stays comparable at 1e-4)


def new_round3():
    g13_traj_mc_full("g13_traj_mc_full_c1.npz", S=128, B=4, M=2, seed=0)
    g13_traj_mc_full("g13_traj_mc_full_m10.npz", S=32, B=4, M=10, seed=2)
    g14_traj_pm_heuristic()
    d = tiny_diffusion(200, 128)
    g15_step(d, 10)
    g15_step(d, 20)
    g16_dit()
    g17_enformer_trunk()
    g18()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g12":
        g12_fullsize_probe()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g18":
        g18()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g26":
        torch.set_num_threads(int(os.environ.get("SVDD_GOLDEN_THREADS", "8")))
        g26_traj_dps_c5()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("g24", "g25"):
        torch.set_num_threads(int(os.environ.get("SVDD_GOLDEN_THREADS", "8")))
        {"g24": g24_decode_sample_c2, "g25": g25_traj_mc_m20}[sys.argv[1]]()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g23":
        torch.set_num_threads(int(os.environ.get("SVDD_GOLDEN_THREADS", "8")))
        g23_traj_tds_c5()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g22":
        g22_harness()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("g21", "g21_c2", "g21_c3"):
        torch.set_num_threads(int(os.environ.get("SVDD_GOLDEN_THREADS", "8")))
        {"g21": g21, "g21_c2": g21_traj_mc_c2, "g21_c3": g21_traj_pm_c3}[sys.argv[1]]()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g17":
        g17_enformer_trunk()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g16":
        g16_dit()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "r3":
        new_round3()
        sys.exit(0)
    d = tiny_diffusion(200, 128)
    g9()
    g1()
    g2_g4(d)
    g3(d)
    g5(d)
    g5(d, bvl=True)
    traj_mc("g6_traj_mc_c1.npz", L=200, S=128, B=4, M=2, seed=0)
    traj_mc("g6_traj_mc_s16.npz", L=50, S=16, B=3, M=5, seed=1)
    traj_pm()
    traj_tds()
    g10()
    traj_dps()
    nets()
    g12_fullsize_probe()
    new_round3()

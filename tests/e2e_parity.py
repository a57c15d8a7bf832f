"""End-to-end comparison of a GPU decode with real nets against a reference trajectory recorded on the CPU
(hard part (ii)): the reference's tiny nets (tests/golden/nets_tiny.npz) run on the GPU through this
engine in replay mode and are compared step by step with the reference's own `controlled_sample` run
(tests/golden/g6_traj_mc_c1.npz: states, logits, scores of every step; reference diffusion_gosai.py:1021-1061).

argmax is discontinuous, so a last-bit difference between MIOpen / our kernels and the CPU's MKL-DNN convolutions can flip
a token at a near-tie, after which that ROW's trajectory differs (rows are independent). The report therefore gives, per
run: the first step at which any state differs, the error of logits and scores up to that step (north-star tolerance
1e-4), the fraction of rows / tokens of x_0 that still agree, and for the first diverging row how close the deciding
quantities were in the reference run. Used by tests/test_e2e_gpu.py and tools/e2e_parity_report.py.
Test infrastructure: lives under tests/, not in the product package."""
import numpy as np
import torch

from svdd_amd.backbone import CNNModel
from svdd_amd.config import Config, ModelConfig, SamplingConfig
from svdd_amd.diffusion import Diffusion
from svdd_amd.value_nets import ConvGRUTrunk, ConvHead


def _sd(g, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(prefix + ".")}


def tiny_engine(nets, L, S, device):
    """The reference's tiny nets (hidden 16 x 1 stack backbone; 8-channel ConvGRU value net) inside the engine."""
    cfg = Config(model=ModelConfig(hidden_dim=16, num_cnn_stacks=1, length=L), sampling=SamplingConfig(steps=S))
    bb = CNNModel(cfg.model, alphabet_size=5)
    bb.load_state_dict(_sd(nets, "backbone"), strict=True)
    model = Diffusion(cfg, backbone=bb)
    emb = ConvGRUTrunk(stem_in_channels=4, stem_channels=8, stem_kernel_size=15, n_conv=3, channel_init=8, kernel_size=5,
                       dropout=0.1)
    emb.load_state_dict(_sd(nets, "embedding"), strict=True)
    head = ConvHead(1, 8)
    head.load_state_dict(_sd(nets, "head"), strict=True)
    for m in (model, emb, head):
        m.to(device).eval()
    return model, emb, head


def compare_with_reference_run(g, nets, device="cuda:0", fuse_nets=True, value_batching="batched"):
    """The reference's TINY nets (hidden 16, GRU 8: below every gate of the hand-written net kernels, so they run as
    PyTorch-ROCm modules whatever fuse_nets says — what this pins is the sampler kernels + host loop on a whole real
    trajectory) -> report dict (see module docstring)."""
    S, L = int(g["S"]), int(g["L"])
    model, emb, head = tiny_engine(nets, L, S, device)
    return compare_engine_with_reference_run(g, model, emb, head, fuse_nets, value_batching)


def uses_hand_written_kernels(model, emb, head, L):
    """True when this engine evaluates (backbone, value net) through the one-launch backbone kernel and the
    tower / GRU / tail kernels of svdd_amd/csrc — i.e. the kernels that are 99 % of a decode's time."""
    from svdd_amd.fused import FusedValueNet
    fb = model._fused_backbone_or_none(L)
    fn = model.value_callable(emb, head)
    return fb is not None and isinstance(fn, FusedValueNet) and fn.kernels_ok(L)


def compare_engine_with_reference_run(g, model, emb, head, fuse_nets=True, value_batching="batched", precision="f32"):
    """Free-running decode of `model` in replay mode against a recorded reference controlled_sample run `g`."""
    S, B, L, M = int(g["S"]), int(g["B"]), int(g["L"]), int(g["M"])
    model.fuse_nets, model.value_batching, model.rng_mode, model.precision = fuse_nets, value_batching, "replay", precision
    model.trace, model.state_trace = [], []
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        x0 = model.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M)
    torch.cuda.synchronize()
    xs = np.stack([x.cpu().numpy() for x in model.state_trace])                # [S + 1, B, L] states fed to the backbone
    logits = [t[0].cpu().numpy() for t in model.trace]
    scores = [t[1].cpu().numpy() for t in model.trace[:-1]]
    same = (xs == g["xs"]).all(axis=2)                                            # [S + 1, B]
    first = next((i for i in range(S + 1) if not same[i].all()), None)
    upto = S + 1 if first is None else first                                      # steps whose INPUT state is still identical
    dl = max(float(np.abs(logits[i] - g["logits"][i]).max()) for i in range(upto)) if upto else 0.0
    ds = max([float(np.abs(scores[i] - g["scores"][i]).max()) for i in range(min(upto, S))] or [0.0])
    # row-wise: the error on rows that have not diverged yet, over the whole run
    dl_rows = max(float(np.abs(logits[i][same[i]] - g["logits"][i][same[i]]).max()) for i in range(S + 1) if same[i].any())
    ds_rows = max(float(np.abs(scores[i][same[i]] - g["scores"][i][same[i]]).max()) for i in range(S) if same[i].any())
    x0n = x0.cpu().numpy()
    rep = {"S": S, "B": B, "L": L, "M": M, "fuse_nets": fuse_nets, "value_batching": value_batching, "precision": precision,
           "hand_written_net_kernels": bool(fuse_nets and uses_hand_written_kernels(model, emb, head, L)),
           "first_divergence_step": first, "steps_compared": upto,
           "max_abs_logit_err_before_divergence": dl, "max_abs_score_err_before_divergence": ds,
           "max_abs_logit_err_on_undiverged_rows": dl_rows, "max_abs_score_err_on_undiverged_rows": ds_rows,
           "x0_exact": bool(np.array_equal(x0n, g["x0"])),
           "x0_rows_identical": float((x0n == g["x0"]).all(axis=1).mean()),
           "x0_tokens_identical": float((x0n == g["x0"]).mean())}
    if first is not None:
        i = first - 1                                                              # the step that produced the first differing state
        row = int(np.nonzero(~same[first])[0][0])
        sc = np.sort(g["scores"][i][row])[::-1]
        rep["first_divergence"] = {"row": row, "produced_by_step": i,
                                   "reference_score_gap_top2": float(sc[0] - sc[1]) if M > 1 else None,
                                   "gpu_minus_ref_scores": (scores[i][row] - g["scores"][i][row]).tolist()}
    model.trace = model.state_trace = None
    model.precision = "f32"
    return rep


def teacher_forced_report(g, model, emb, head, precision="f32"):
    """Every state of a recorded reference run `g` (full-size nets: tests/golden/g13_*.npz, reference
    diffusion_gosai.py:1021-1061, 1174-1228) is fed to the hand-written net kernels: x_t -> one-launch backbone -> raw
    logits; the reference's own M candidates -> value net three ways (whole-sequence tower; parent-sharing row windows;
    the compacted live-candidate path of the work-skipping decode) -> scores. Each step is compared with what the
    reference computed on the CPU at that step, so an error cannot hide behind an earlier divergence. Also: the select
    kernel applied to the GPU scores against the reference's next state (selection agreement), and the gap between the
    two best reference scores where a selection differs."""
    from svdd_amd import ops
    from svdd_amd.fused import FusedBackbone, FusedValueNet
    S, B, L, M = int(g["S"]), int(g["B"]), int(g["L"]), int(g["M"])
    dev = model.device
    model.fuse_nets, model.precision = True, precision
    fb, fn = model._fused_backbone_or_none(L), model.value_callable(emb, head)
    if not (isinstance(fb, FusedBackbone) and isinstance(fn, FusedValueNet) and fn.kernels_ok(L)):
        raise ops.SvddError("teacher_forced_report needs nets the hand-written kernels take (full-size CNN + ConvGRU)")
    dl, ds = np.zeros(S + 1), np.zeros((S, 3))
    agree = np.zeros((S, B), dtype=bool)
    gaps = []
    xs = torch.from_numpy(g["xs"]).to(dev)
    cands = torch.from_numpy(g["cand"]).to(dev)
    with torch.no_grad():
        for i in range(S + 1):
            x = xs[i].contiguous()
            lg = model._backbone_logits(x)
            dl[i] = float((lg.cpu() - torch.from_numpy(g["logits"][i])).abs().max())
            if i == S:
                break
            cand = cands[i].contiguous()
            onehot = ops.transform_samples(cand.view(B * M, L))
            ref = torch.from_numpy(g["scores"][i])
            whole = fn(onehot).reshape(B, M).float()
            windows = fn.forward_candidates(onehot, cand, x).reshape(B, M).float() if fn.candidates_ok(L, M) else whole
            ws = model._SkipWorkspace(B, M, dev)
            ws.parent_score.copy_(fn.forward_tokens(x).reshape(B))
            sc = fn.candidate_scores_compact(onehot, cand, x, ws).reshape(-1) if fn.candidates_ok(L, M) else None
            compact = model._dense_scores(sc, ws, B, M) if sc is not None else whole
            for k, v in enumerate((whole, windows, compact)):
                ds[i, k] = float((v.cpu() - ref).abs().max())
            x_next, _, _ = ops.select(compact.contiguous(), cand, mode=ops.SELECT_ARGMAX, want_soft=False)
            agree[i] = (x_next.cpu().numpy() == g["xs"][i + 1]).all(axis=1)
            for b in np.nonzero(~agree[i])[0]:
                top = np.sort(g["scores"][i][b])[::-1]
                gaps.append(float(top[0] - top[1]))
    model.precision = "f32"
    return {"S": S, "B": B, "L": L, "M": M, "precision": precision, "steps_compared": S + 1,
            "max_abs_logit_err": float(dl.max()), "max_abs_score_err_whole_tower": float(ds[:, 0].max()),
            "max_abs_score_err_windows": float(ds[:, 1].max()), "max_abs_score_err_compact": float(ds[:, 2].max()),
            "selection_agreement": float(agree.mean()), "row_steps": int(agree.size),
            "disagreeing_row_steps": int((~agree).sum()),
            "max_reference_top2_gap_where_selection_differs": max(gaps) if gaps else None}


def teacher_forced_pm_report(g, model, reward_model, precision="f32"):
    """The SVDD-PM twin of teacher_forced_report, on a recorded reference controlled_sample_tweedie(options="True") run `g`
    with full-size nets (tests/golden/g18_traj_pm_full_rna.npz; reference diffusion_gosai.py:1105-1145, 1373-1460): every
    state x_t and every candidate set goes through the one-launch backbone kernel (L = 50: several sequences per tile), the
    x0-hat one-hots (:1415-1419) are rebuilt from the GPU's candidate logits and compared with what the reward model saw in
    the reference run, and the reward model (hand-written tower / GRU / tail kernels) scores the REFERENCE's one-hots."""
    from svdd_amd import ops
    S, B, L, M = int(g["S"]), int(g["B"]), int(g["L"]), int(g["M"])
    dev = model.device
    model.fuse_nets, model.precision = True, precision
    fn = model.reward_callable(reward_model)
    dl, dcl, ds = np.zeros(S + 1), np.zeros(S), np.zeros(S)
    oh_same, oh_total = 0, 0
    agree = np.zeros((S, B), dtype=bool)
    gaps = []
    with torch.no_grad():
        for i in range(S + 1):
            x = torch.from_numpy(g["xs"][i]).to(dev).contiguous()
            dl[i] = float((model._backbone_logits(x).cpu() - torch.from_numpy(g["logits"][i])).abs().max())
            if i == S:
                break
            cand = torch.from_numpy(g["cand"][i]).to(dev).contiguous()                 # [B, M, L]
            flat = cand.view(B * M, L)
            cl = model._backbone_logits(flat)
            dcl[i] = float((cl.cpu().view(B, M, L, 5) - torch.from_numpy(g["cand_logits"][i])).abs().max())
            oh, _ = ops.x0hat(cl, flat)
            ref_oh = torch.from_numpy(g["x0hat_onehot_t"][i].astype(np.float32)).view(B * M, 4, L)
            same_rows = (oh.cpu() == ref_oh).all(dim=2).all(dim=1)
            oh_same += int(same_rows.sum())
            oh_total += B * M
            sc = fn(ref_oh.to(dev))[:, 0].reshape(B, M).float()
            ref = torch.from_numpy(g["scores"][i])
            ds[i] = float((sc.cpu() - ref).abs().max())
            x_next, _, _ = ops.select(sc.contiguous(), cand, mode=ops.SELECT_ARGMAX, want_soft=False)
            agree[i] = (x_next.cpu().numpy() == g["xs"][i + 1]).all(axis=1)
            for b in np.nonzero(~agree[i])[0]:
                top = np.sort(g["scores"][i][b])[::-1]
                gaps.append(float(top[0] - top[1]))
    model.precision = "f32"
    return {"S": S, "B": B, "L": L, "M": M, "precision": precision, "steps_compared": S + 1,
            "hand_written_net_kernels": bool(isinstance(fn, torch.nn.Module) and fn is not reward_model),
            "max_abs_logit_err": float(dl.max()), "max_abs_candidate_logit_err": float(dcl.max()),
            "max_abs_score_err": float(ds.max()), "x0hat_rows_identical": oh_same / max(oh_total, 1),
            "selection_agreement": float(agree.mean()), "disagreeing_row_steps": int((~agree).sum()),
            "max_reference_top2_gap_where_selection_differs": max(gaps) if gaps else None}


def free_running_pm_report(g, model, reward_model, precision="f32"):
    """Free-running controlled_sample_tweedie in replay mode with the hand-written net kernels against the reference's run."""
    S, B, M = int(g["S"]), int(g["B"]), int(g["M"])
    model.fuse_nets, model.rng_mode, model.precision = True, "replay", precision
    model.state_trace = []
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        x0 = model.controlled_sample_tweedie(reward_model, num_steps=S, eval_sp_size=B, sample_M=M, options="True")
    torch.cuda.synchronize()
    xs = np.stack([x.cpu().numpy() for x in model.state_trace])
    model.state_trace = None
    model.precision = "f32"
    n = min(len(xs), S + 1)
    same = (xs[:n] == g["xs"][:n]).all(axis=2)
    first = next((i for i in range(n) if not same[i].all()), None)
    x0n = x0.cpu().numpy()
    return {"precision": precision, "states_recorded": int(len(xs)), "first_divergence_step": first,
            "x0_exact": bool(np.array_equal(x0n, g["x0"])), "x0_rows_identical": float((x0n == g["x0"]).all(axis=1).mean())}


def tds_reference_run_report(g, model, reward_model, precision="f32"):
    """A recorded reference controlled_sample_TDS run `g` with full-size nets (tests/golden/g19_traj_tds_full.npz; reference
    diffusion_gosai.py:938-978, 1230-1284). Teacher-forced: every x_t and every proposal through the one-launch backbone, the
    numerator / denominator rewards (:1263-1277) through the hand-written reward kernels on the x0-hat of the GPU's logits.
    Free-running: the engine's decode in replay mode (torch + numpy streams seeded as in the reference run; exact reuse of
    forward(sample) and of the numerator reward across steps) against the reference's states and x_0."""
    from svdd_amd import ops
    S, B, L = int(g["S"]), int(g["B"]), int(g["L"])
    dev = model.device
    model.fuse_nets, model.precision = True, precision
    fn = model.reward_callable(reward_model)
    dl, dsl, dnum, dden = np.zeros(S + 1), np.zeros(S), np.zeros(S), np.zeros(S)
    oh_same, oh_total = 0, 0
    with torch.no_grad():
        for i in range(S + 1):
            x = torch.from_numpy(g["xs"][i]).to(dev).contiguous()
            lg = model._backbone_logits(x)
            dl[i] = float((lg.cpu() - torch.from_numpy(g["logits"][i])).abs().max())
            if i == S:
                break
            smp = torch.from_numpy(g["samples"][i]).to(dev).contiguous()
            ls = model._backbone_logits(smp)
            dsl[i] = float((ls.cpu() - torch.from_numpy(g["sample_logits"][i])).abs().max())
            # the reward model is scored on the x0-hat of the REFERENCE's logits (what it saw in the reference run); how often the
            # GPU logits give the same x0-hat rows is reported separately (an argmax over 4 near-uniform logits can flip at a tie)
            oh_num, _ = ops.x0hat(torch.from_numpy(g["sample_logits"][i]).to(dev).contiguous(), smp)
            oh_den, _ = ops.x0hat(torch.from_numpy(g["den_logits"][i]).to(dev).contiguous(), x)
            for mine, ref_oh in ((ops.x0hat(ls, smp)[0], oh_num), (ops.x0hat(lg, x)[0], oh_den)):
                oh_same += int((mine == ref_oh).all(dim=2).all(dim=1).sum())
                oh_total += B
            dnum[i] = float((fn(oh_num)[:, 0][:, 0].float().cpu() - torch.from_numpy(g["num"][i])).abs().max())
            dden[i] = float((fn(oh_den)[:, 0][:, 0].float().cpu() - torch.from_numpy(g["den"][i])).abs().max())
    model.rng_mode, model.state_trace = "replay", []
    torch.manual_seed(int(g["seed"]))
    np.random.seed(int(g["np_seed"]))
    with torch.no_grad():
        x0 = model.controlled_sample_TDS(reward_model, float(g["alpha"]), num_steps=S, eval_sp_size=B)
    torch.cuda.synchronize()
    xs = [x.cpu().numpy() for x in model.state_trace]
    model.state_trace = None
    model.precision = "f32"
    n = min(len(xs), S + 1)
    first = next((i for i in range(n) if not np.array_equal(xs[i], g["xs"][i])), None)
    x0n = x0.cpu().numpy()
    return {"precision": precision, "hand_written_net_kernels": bool(isinstance(fn, torch.nn.Module) and fn is not reward_model),
            "max_abs_logit_err": float(dl.max()), "max_abs_proposal_logit_err": float(dsl.max()),
            "max_abs_reward_num_err": float(dnum.max()), "max_abs_reward_den_err": float(dden.max()),
            "x0hat_rows_identical": oh_same / max(oh_total, 1), "states_recorded": len(xs), "first_divergence_step": first, "x0_exact": bool(np.array_equal(x0n, g["x0"])),
            "x0_rows_identical": float((x0n == g["x0"]).all(axis=1).mean())}


# ------------------------------------------------------------------------------------------------------------------
# g21: the reference's own runs AT the headline configs (BASELINE.json configs[1] = C2: controlled_sample, B = 256,
# L = 200, M = 10, 128 steps; configs[2] = C3: controlled_sample_tweedie(options="True"), B = 256, L = 50, M = 10).
# The fixtures are lean (states, delta-coded candidates, scores, selections; raw logits of a few calls only), so the
# candidates are RE-PROPOSED on the GPU from the replayed uniforms and must equal the reference's: at full occupancy that
# checks the fp32 backbone kernel on all 256 rows of every step through K1's argmaxes.
def cand_of(g, i):
    """Step i's candidates u8 [B, M, L] (stored in full, or as 255 = 'the parent's token' deltas)."""
    if "cand" in g:
        return g["cand"][i]
    d = g["cand_delta"][i]
    return np.where(d == 255, g["xs"][i][:, None, :], d)


def race_margin(logit_row, dm, mcs, u5):
    """Relative lead of the winner of ONE categorical draw of a MASKed position (diffusion_gosai.py:30-34 on q_xs of
    :1194-1196) over the runner-up, evaluated in float64 from the raw backbone output of that position."""
    l4 = np.asarray(logit_row[:4], dtype=np.float64)
    logp = l4 - (l4.max() + np.log(np.exp(l4 - l4.max()).sum()))
    q = np.concatenate([np.exp(logp) * float(dm), [float(mcs)]])
    r = np.sort(q / (1e-10 - np.log(np.asarray(u5, dtype=np.float64) + 1e-10)))[::-1]
    return float((r[0] - r[1]) / r[0])


def _explain_candidate_diffs(mine, ref, lg, u, dm, mcs):
    """Every (b, m, l) where the re-proposed candidates differ from the reference's -> its race margin (GPU logits)."""
    out = []
    for b, m, l in zip(*np.nonzero(mine != ref)):
        out.append(race_margin(lg[b, l], dm, mcs, u[m, b, :, l]))
    return out


def _replay_uniforms(model, M, B, L, logits):
    rng = model._rng(0, M, B, L, logits)                       # replay mode: M x rand_like(q_xs) of torch's CPU stream
    return rng, rng.uniforms


def teacher_forced_lean_report(g, model, emb, head, precision="f32"):
    """C2, teacher-forced on the reference's states: per step x_t -> one-launch backbone -> K1 with the replayed uniforms ->
    candidates (vs the reference's) ; the reference's candidates -> value kernels (compacted, parent-sharing path the
    decode takes, and the whole-sequence path) -> scores (vs the reference's, 1e-4) -> K2 -> next state (vs the reference's)."""
    from svdd_amd import ops
    from svdd_amd.fused import FusedBackbone, FusedValueNet
    S, B, L, M = int(g["S"]), int(g["B"]), int(g["L"]), int(g["M"])
    dev = model.device
    model.fuse_nets, model.precision, model.rng_mode = True, precision, "replay"
    fb, fn = model._fused_backbone_or_none(L), model.value_callable(emb, head)
    if not (isinstance(fb, FusedBackbone) and isinstance(fn, FusedValueNet) and fn.kernels_ok(L)):
        raise ops.SvddError("needs nets the hand-written kernels take (full-size CNN + ConvGRU)")
    sched, _, _ = model._schedule(S, 1e-5)
    kept = {int(s): k for k, s in enumerate(g["logits_steps"])} if "logits_steps" in g else {}
    dl, ds = [], np.zeros((S, 2))
    cand_rows_same, cand_margins, gaps = 0, [], []
    agree = np.zeros((S, B), dtype=bool)
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        for i in range(S + 1):
            x = torch.from_numpy(g["xs"][i]).to(dev).contiguous()
            lg = model._backbone_logits(x)
            if i in kept:
                dl.append(float((lg.cpu() - torch.from_numpy(g["logits"][kept[i]])).abs().max()))
            if i == S:
                break
            ref_cand = cand_of(g, i)
            rng, u = _replay_uniforms(model, M, B, L, lg)
            mine, _, _ = ops.propose(lg, x, sched[i, 2], sched[i, 1], M, rng)
            mine = mine.cpu().numpy()
            same = (mine == ref_cand).all(axis=2)
            cand_rows_same += int(same.sum())
            if not same.all():
                cand_margins += _explain_candidate_diffs(mine, ref_cand, lg.cpu().numpy(), u.cpu().numpy(), sched[i, 2], sched[i, 1])
            cand = torch.from_numpy(ref_cand).to(dev).contiguous()
            onehot = ops.transform_samples(cand.view(B * M, L))
            ref = torch.from_numpy(g["scores"][i])
            whole = fn(onehot).reshape(B, M).float()
            if fn.candidates_ok(L, M):
                ws = model._SkipWorkspace(B, M, dev)
                ws.parent_score.copy_(fn.forward_tokens(x).reshape(B))
                compact = model._dense_scores(fn.candidate_scores_compact(onehot, cand, x, ws).reshape(-1), ws, B, M)
            else:
                compact = whole
            ds[i] = [float((whole.cpu() - ref).abs().max()), float((compact.cpu() - ref).abs().max())]
            x_next, _, _ = ops.select(compact.contiguous(), cand, mode=ops.SELECT_ARGMAX, want_soft=False)
            agree[i] = (x_next.cpu().numpy() == g["xs"][i + 1]).all(axis=1)
            for b in np.nonzero(~agree[i])[0]:
                top = np.sort(g["scores"][i][b])[::-1]
                gaps.append(float(top[0] - top[1]))
    model.precision = "f32"
    return {"S": S, "B": B, "L": L, "M": M, "precision": precision, "steps_compared": S + 1,
            "max_abs_logit_err_kept_calls": max(dl) if dl else None, "logit_calls_compared": len(dl),
            "candidate_rows": S * B * M, "candidate_rows_identical": cand_rows_same,
            "candidate_tokens_differing": len(cand_margins),
            "max_race_margin_where_candidates_differ": max(cand_margins) if cand_margins else None,
            "max_abs_score_err_whole_tower": float(ds[:, 0].max()), "max_abs_score_err_compact": float(ds[:, 1].max()),
            "selection_agreement": float(agree.mean()), "row_steps": int(agree.size),
            "disagreeing_row_steps": int((~agree).sum()),
            "max_reference_top2_gap_where_selection_differs": max(gaps) if gaps else None}


def teacher_forced_lean_pm_report(g, model, reward_model, precision="f32"):
    """C3, teacher-forced: x_t -> backbone -> K1 (replayed uniforms) -> candidates vs the reference's; the reference's
    candidates -> backbone -> x0-hat rows vs what the reference's reward model was handed (:1415-1419; a differing position
    must be a near-tie of the GPU's two best real-token logits); reward kernels on the REFERENCE's x0-hat -> scores (1e-4)
    -> K2 -> next state."""
    from svdd_amd import ops
    S, B, L, M = int(g["S"]), int(g["B"]), int(g["L"]), int(g["M"])
    dev = model.device
    model.fuse_nets, model.precision, model.rng_mode = True, precision, "replay"
    fn = model.reward_callable(reward_model)
    sched, _, _ = model._schedule(S, 1e-5)
    kept = {int(s): k for k, s in enumerate(g["logits_calls"])}
    dl, ds = [], np.zeros(S)
    cand_rows_same, cand_margins, gaps, xh_gaps = 0, [], [], []
    xh_same = 0
    agree = np.zeros((S, B), dtype=bool)
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        for i in range(S + 1):
            x = torch.from_numpy(g["xs"][i]).to(dev).contiguous()
            lg = model._backbone_logits(x)
            if i * (1 + M) in kept:
                dl.append(float((lg.cpu() - torch.from_numpy(g["logits"][kept[i * (1 + M)]])).abs().max()))
            if i == S:
                break
            ref_cand = cand_of(g, i)
            rng, u = _replay_uniforms(model, M, B, L, lg)
            mine, _, _ = ops.propose(lg, x, sched[i, 2], sched[i, 1], M, rng)
            mine = mine.cpu().numpy()
            same = (mine == ref_cand).all(axis=2)
            cand_rows_same += int(same.sum())
            if not same.all():
                cand_margins += _explain_candidate_diffs(mine, ref_cand, lg.cpu().numpy(), u.cpu().numpy(), sched[i, 2], sched[i, 1])
            cand = torch.from_numpy(ref_cand).to(dev).contiguous()
            flat = cand.view(B * M, L)
            cl = model._backbone_logits(flat)
            _, xh = ops.x0hat(cl, flat, want_tokens=True, want_onehot=False)
            d = g["x0hat_delta"][i]
            ref_xh = np.where(d == 255, ref_cand, d).reshape(B * M, L)
            xh_np = xh.cpu().numpy()
            xh_same += int((xh_np == ref_xh).all(axis=1).sum())
            if not np.array_equal(xh_np, ref_xh):
                cln = cl.cpu().numpy()
                for r, l in zip(*np.nonzero(xh_np != ref_xh)):
                    top = np.sort(cln[r, l, :4])[::-1]
                    xh_gaps.append(float(top[0] - top[1]))
            ref_oh = torch.nn.functional.one_hot(torch.from_numpy(ref_xh).long(), 4).permute(0, 2, 1).float().contiguous()
            sc = fn(ref_oh.to(dev))[:, 0].reshape(B, M).float()
            ref = torch.from_numpy(g["scores"][i])
            ds[i] = float((sc.cpu() - ref).abs().max())
            x_next, _, _ = ops.select(sc.contiguous(), cand, mode=ops.SELECT_ARGMAX, want_soft=False)
            agree[i] = (x_next.cpu().numpy() == g["xs"][i + 1]).all(axis=1)
            for b in np.nonzero(~agree[i])[0]:
                top = np.sort(g["scores"][i][b])[::-1]
                gaps.append(float(top[0] - top[1]))
    model.precision = "f32"
    return {"S": S, "B": B, "L": L, "M": M, "precision": precision, "steps_compared": S + 1,
            "hand_written_net_kernels": bool(isinstance(fn, torch.nn.Module) and fn is not reward_model),
            "max_abs_logit_err_kept_calls": max(dl) if dl else None, "logit_calls_compared": len(dl),
            "candidate_rows": S * B * M, "candidate_rows_identical": cand_rows_same,
            "candidate_tokens_differing": len(cand_margins),
            "max_race_margin_where_candidates_differ": max(cand_margins) if cand_margins else None,
            "x0hat_rows_identical": xh_same, "x0hat_tokens_differing": len(xh_gaps),
            "max_logit_top2_gap_where_x0hat_differs": max(xh_gaps) if xh_gaps else None,
            "max_abs_score_err": float(ds.max()), "selection_agreement": float(agree.mean()), "row_steps": int(agree.size),
            "disagreeing_row_steps": int((~agree).sum()),
            "max_reference_top2_gap_where_selection_differs": max(gaps) if gaps else None}


def free_running_lean_report(g, model, run, precision="f32"):
    """Free-running replay decode at a headline config against the reference's run. `run(model)` calls the sampler. Rows are
    independent, so every row is followed to ITS first divergence: there the state the engine chose must be one of the
    reference's own candidates of that step whose REFERENCE score is within 2e-4 of the reference's best (a selection
    near-tie), or — a candidate the reference did not have — a proposal flip, which the teacher-forced report explains;
    scores are compared (1e-4) on every row-step whose input state and candidates are still the reference's. (SVDD-PM: a
    candidate's score is the reward of its x0-hat, an argmax over four near-uniform logits per position — where that argmax
    sits on a last-bit tie the GPU scores a DIFFERENT one-hot and the score differs by ~1e-3 legitimately; the teacher-forced
    report bounds the reward error on the reference's own x0-hat rows and explains every differing x0-hat token, so for PM
    the free run asserts the fraction of scores within tolerance, not the maximum.)"""
    S, B, L, M = int(g["S"]), int(g["B"]), int(g["L"]), int(g["M"])
    model.fuse_nets, model.rng_mode, model.precision = True, "replay", precision
    model.trace, model.state_trace = [], []
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        x0 = run(model)
    torch.cuda.synchronize()
    xs = np.stack([x.cpu().numpy() for x in model.state_trace])[:S + 1]
    scores = [t[1].cpu().numpy() for t in model.trace[:S]]
    model.trace = model.state_trace = None
    model.precision = "f32"
    same = (xs == g["xs"][:len(xs)]).all(axis=2)                                       # [S + 1, B]
    first_row = np.array([next((i for i in range(len(xs)) if not same[i, b]), -1) for b in range(B)])
    comparable = same[:S].copy()                                                      # row-steps whose candidates are the reference's
    near_tie, proposal, rescored, unexplained, gaps = 0, 0, 0, [], []
    for b in np.nonzero(first_row >= 0)[0]:
        i = int(first_row[b]) - 1                                                      # the step that produced the differing state
        refc = cand_of(g, i)[b]                                                        # [M, L]
        hit = np.nonzero((refc == xs[i + 1, b]).all(axis=1))[0]
        if len(hit):
            gap = float(g["scores"][i][b].max() - g["scores"][i][b][hit].max())
            if float(np.abs(scores[i][b] - g["scores"][i][b]).max()) > 1e-4:
                # SVDD-PM only: same x_t, same candidates, yet a score off by more than any net error — the reward was taken on a
                # DIFFERENT x0-hat one-hot (an argmax over four near-uniform logits flipped at a last-bit tie, :1415-1417);
                # the teacher-forced report counts and explains those flips
                rescored += 1
                comparable[i, b] = False
                continue
            gaps.append(gap)
            if gap <= 2e-4:
                near_tie += 1
            else:
                unexplained.append({"row": int(b), "step": i, "reference_score_gap": gap})
        else:
            proposal += 1
            comparable[i, b] = False                                                   # same x_t, but a candidate the reference did not have
    errs = np.concatenate([np.abs(scores[i][comparable[i]] - g["scores"][i][comparable[i]]).ravel() for i in range(S) if comparable[i].any()])
    ds = float(errs.max())
    x0n = x0.cpu().numpy()
    diverged = first_row[first_row >= 0]
    return {"S": S, "B": B, "L": L, "M": M, "precision": precision,
            "first_divergence_step": int(diverged.min()) if len(diverged) else None,
            "rows_diverged": int(len(diverged)), "rows_following_the_reference_to_the_end": int((first_row < 0).sum()),
            "divergences_at_selection_near_ties": near_tie, "divergences_by_proposal_flip": proposal,
            "divergences_by_x0hat_flip": rescored,
            "divergences_unexplained": unexplained, "max_reference_score_gap_at_a_selection_divergence": max(gaps) if gaps else None,
            "max_abs_score_err_on_undiverged_rows": ds, "scores_compared": int(errs.size),
            "frac_scores_within_1e-4": float((errs <= 1e-4).mean()),
            "x0_exact": bool(np.array_equal(x0n, g["x0"])),
            "x0_rows_identical": float((x0n == g["x0"]).all(axis=1).mean()),
            "x0_tokens_identical": float((x0n == g["x0"]).mean())}


def teacher_forced_lean_tds_report(g, model, reward_model, precision="f32"):
    """g23: the reference's controlled_sample_TDS at BASELINE configs[4]'s per-GPU shard size (256 particles, L = 200, 128 steps;
    reference diffusion_gosai.py:938-978, 1230-1284), teacher-forced: x_t -> backbone -> K1 (replayed uniforms, one proposal per
    particle) -> proposals vs the reference's; the reference's proposals -> backbone -> x0-hat rows vs what its reward model was
    handed; the reward kernels on the REFERENCE's x0-hat rows -> numerator rewards (1e-4); K4 on the reference's own reward vectors
    and numpy's replayed uniforms -> ancestor indices and next state, exact. Then the free-running decode (torch + numpy streams
    replayed; forward(sample) and the numerator reward reused across steps) against the reference's states."""
    from svdd_amd import ops
    S, B, L = int(g["S"]), int(g["B"]), int(g["L"])
    dev = model.device
    alpha = float(g["alpha"])
    model.fuse_nets, model.precision, model.rng_mode = True, precision, "replay"
    fn = model.reward_callable(reward_model)
    sched, _, _ = model._schedule(S, 1e-5)
    kept = {int(s): k for k, s in enumerate(g["logits_calls"])}
    dl, dnum = [], np.zeros(S)
    prop_same, prop_margins, xh_same, xh_gaps, idx_same, next_same, den_ok, den_n = 0, [], 0, [], 0, 0, 0, 0
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        for i in range(S + 1):
            x_np = g["xs"][i]
            x = torch.from_numpy(x_np).to(dev).contiguous()
            lg = model._backbone_logits(x)
            if 3 * i in kept:
                dl.append(float((lg.cpu() - torch.from_numpy(g["logits"][kept[3 * i]])).abs().max()))
            if i == S:
                break
            sd = g["sample_delta"][i]
            ref_sample = np.where(sd == 255, x_np, sd)
            rng, u = _replay_uniforms(model, 1, B, L, lg)
            mine, _, _ = ops.propose(lg, x, sched[i, 2], sched[i, 1], 1, rng)
            mine = mine.cpu().numpy()[:, 0]
            same = (mine == ref_sample).all(axis=1)
            prop_same += int(same.sum())
            if not same.all():
                prop_margins += _explain_candidate_diffs(mine[:, None], ref_sample[:, None], lg.cpu().numpy(), u.cpu().numpy(), sched[i, 2], sched[i, 1])
            smp = torch.from_numpy(ref_sample).to(dev).contiguous()
            ls = model._backbone_logits(smp)
            _, xh = ops.x0hat(ls, smp, want_tokens=True, want_onehot=False)
            xd = g["x0hat_num_delta"][i]
            ref_xh = np.where(xd == 255, ref_sample, xd)
            xh_np = xh.cpu().numpy()
            xh_same += int((xh_np == ref_xh).all(axis=1).sum())
            if not np.array_equal(xh_np, ref_xh):
                lsn = ls.cpu().numpy()
                for r, l in zip(*np.nonzero(xh_np != ref_xh)):
                    top = np.sort(lsn[r, l, :4])[::-1]
                    xh_gaps.append(float(top[0] - top[1]))
            ref_oh = torch.nn.functional.one_hot(torch.from_numpy(ref_xh).long(), 4).permute(0, 2, 1).float().contiguous().to(dev)
            num = fn(ref_oh)[:, 0][:, 0].float()
            dnum[i] = float((num.cpu() - torch.from_numpy(g["num"][i])).abs().max())
            oh_den, _ = ops.x0hat(lg, x)
            den = fn(oh_den)[:, 0][:, 0].float().cpu().numpy()
            den_ok += int((np.abs(den - g["den"][i]) <= 1e-4).sum())
            den_n += B
            # K4 on the reference's own reward vectors: ancestors and next state exact
            x_next, idx = ops.tds_resample(torch.from_numpy(g["num"][i]).to(dev), torch.from_numpy(g["den"][i]).to(dev), alpha, smp,
                                           torch.from_numpy(g["choice_u"][i]).to(dev))
            idx_same += int((idx.cpu().numpy() == g["idx"][i]).sum())
            next_same += int((x_next.cpu().numpy() == g["xs"][i + 1]).all(axis=1).sum())
    # free-running
    model.state_trace = []
    torch.manual_seed(int(g["seed"]))
    np.random.seed(int(g["np_seed"]))
    with torch.no_grad():
        x0 = model.controlled_sample_TDS(reward_model, alpha, num_steps=S, eval_sp_size=B)
    torch.cuda.synchronize()
    xs = [t.cpu().numpy() for t in model.state_trace]
    model.state_trace = None
    model.precision = "f32"
    n = min(len(xs), S + 1)
    first = next((i for i in range(n) if not np.array_equal(xs[i], g["xs"][i])), None)
    x0n = x0.cpu().numpy()
    return {"S": S, "B": B, "L": L, "precision": precision,
            "hand_written_net_kernels": bool(isinstance(fn, torch.nn.Module) and fn is not reward_model),
            "max_abs_logit_err_kept_calls": max(dl) if dl else None,
            "proposals": S * B, "proposals_identical": prop_same, "proposal_tokens_differing": len(prop_margins),
            "max_race_margin_where_proposals_differ": max(prop_margins) if prop_margins else None,
            "x0hat_rows_identical": xh_same, "x0hat_tokens_differing": len(xh_gaps),
            "max_logit_top2_gap_where_x0hat_differs": max(xh_gaps) if xh_gaps else None,
            "max_abs_reward_num_err": float(dnum.max()), "reward_den_within_1e-4": den_ok / max(den_n, 1),
            "resample_indices_identical": idx_same, "resample_next_rows_identical": next_same,
            "free_running": {"states_recorded": len(xs), "first_divergence_step": first,
                             "x0_exact": bool(np.array_equal(x0n, g["x0"])),
                             "x0_rows_identical": float((x0n == g["x0"]).all(axis=1).mean()),
                             "x0_tokens_identical": float((x0n == g["x0"]).mean())}}


def unguided_lean_report(g, model, precision="f32"):
    """g24: the reference's un-guided `decode_sample` (diffusion_gosai.py:888-936, 1147-1172) at B = 256, L = 200, 128 steps.
    Teacher-forced: x_t -> backbone -> K1 (one draw per position from the replayed stream) -> the reference's next state; the
    noise-removal argmax (:1049-1060) on the last state -> x_0. A differing token must be a near-tie of the categorical draw /
    of the two best real-token logits. Free-running: the engine's decode_sample in replay mode against the reference's x_0."""
    from svdd_amd import ops
    S, B, L = int(g["S"]), int(g["B"]), int(g["L"])
    dev = model.device
    model.fuse_nets, model.precision, model.rng_mode = True, precision, "replay"
    sched, _, _ = model._schedule(S, 1e-5)
    kept = {int(s): k for k, s in enumerate(g["logits_steps"])}
    dl, margins, fgaps = [], [], []
    rows_same = 0
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        for i in range(S + 1):
            x = torch.from_numpy(g["xs"][i]).to(dev).contiguous()
            lg = model._backbone_logits(x)
            if i in kept:
                dl.append(float((lg.cpu() - torch.from_numpy(g["logits"][kept[i]])).abs().max()))
            if i == S:
                x0 = ops.finalize(lg, x).cpu().numpy()
                fin_same = int((x0 == g["x0"]).all(axis=1).sum())
                if fin_same != B:
                    lgn = lg.cpu().numpy()
                    for b, l in zip(*np.nonzero(x0 != g["x0"])):
                        top = np.sort(lgn[b, l, :4])[::-1]
                        fgaps.append(float(top[0] - top[1]))
                break
            rng, u = _replay_uniforms(model, 1, B, L, lg)
            mine, _, _ = ops.propose(lg, x, sched[i, 2], sched[i, 1], 1, rng)
            mine = mine.cpu().numpy()[:, 0]
            ref = g["xs"][i + 1]
            same = (mine == ref).all(axis=1)
            rows_same += int(same.sum())
            if not same.all():
                margins += _explain_candidate_diffs(mine[:, None], ref[:, None], lg.cpu().numpy(), u.cpu().numpy(), sched[i, 2], sched[i, 1])
    torch.manual_seed(int(g["seed"]))
    with torch.no_grad():
        x0f = model.decode_sample(num_steps=S, eval_sp_size=B).cpu().numpy()
    model.precision = "f32"
    return {"S": S, "B": B, "L": L, "precision": precision, "max_abs_logit_err_kept_calls": max(dl) if dl else None,
            "row_steps": S * B, "next_states_identical": rows_same, "tokens_differing": len(margins),
            "max_race_margin_where_tokens_differ": max(margins) if margins else None,
            "noise_removal_rows_identical": fin_same, "max_logit_top2_gap_where_x0_differs": max(fgaps) if fgaps else None,
            "free_running": {"x0_exact": bool(np.array_equal(x0f, g["x0"])),
                             "x0_rows_identical": float((x0f == g["x0"]).all(axis=1).mean()),
                             "x0_tokens_identical": float((x0f == g["x0"]).mean())}}

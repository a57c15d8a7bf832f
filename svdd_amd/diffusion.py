"""`Diffusion` — host-side mirror of the reference sampler `diffusion_gosai.Diffusion` for the
decode hot path (reference diffusion_gosai.py:74-175, 286-377, 751-753, 820-1496).

Same method names, keyword arguments, return types and error behaviour as the reference for the
methods on the SVDD decode path, so `BaseModel`/`decode.py`-style callers work unchanged:

    forward, _process_sigma, _sample_prior, _sample, decode_sample, controlled_sample,
    controlled_sample_tweedie, controlled_sample_TDS, controlled_sample_DPS,
    _ddpm_update_finetune[_controlled[_twedie|_TDS|_DPS]], transform_samples

What differs is where the work runs: everything between "backbone logits" and "next x_t" is one
or two launches of the hand-written HIP kernels (svdd_amd/csrc, C ABI include/svdd_hip.h) instead
of ~16*M+20 tiny tensor ops and B host syncs per step; the nets stay PyTorch-ROCm modules.
There is no CPU fallback: the sampler methods need a gfx950 GPU and raise otherwise.

Engine knobs (attributes; defaults reproduce the reference's observable behaviour):
  rng_mode         "replay": the categorical uniforms are drawn from torch's global CPU mt19937
                   generator in exactly the order the reference's CPU path consumes them
                   (M x rand_like(q_xs), in q_xs' memory order) — token-exact parity. replay_rng = "device" (default):
                   the generator's 624-word state is uploaded once per sampler call, the stream is produced on the GPU by
                   svdd_mt19937_uniform_f32 a step ahead on a side stream, and the advanced state is written back into
                   torch's generator when the call returns; "host": torch.rand on the host + a 10 MB upload per step.
                   "philox": generated in-kernel from (seed, step, global row, m, l) — no host
                   traffic, identical results for any sharding of the batch over GPUs.
  value_batching   "batched": one value-net forward over all B*M candidates (default);
                   "reference": M forwards of batch B like diffusion_gosai.py:1207-1209.
  select_mode      "argmax" (reference, :1225) or "multinomial" (the commented-out :1223; philox only).
  fuse_nets        True: nets of known architecture (CNNModel backbone; ConvGRUTrunk+ConvHead value /
                   reward nets) are run through their MI355X formulations in svdd_amd/fused.py (same
                   weights, exact fp32: one-launch backbone kernel, conv-tower / GRU / value-tail kernels; in
                   SVDD-MC the conv tower is evaluated once per parent x_t and per candidate only around the
                   positions it changed — bit-identical, `FusedValueNet.share_parent_tower`).
                   False: call the modules as given.
  precision        "f32" (default): the fused nets compute in exact fp32 — the parity path and what bench.py's
                   headline measures. "f16x3" / "bf16x3": matrix products on the 16-bit matrix cores with operands
                   split hi + lo (3 MFMAs per product, fp32 accumulate) at ~2.5x the speed. ONLY f16x3 is fp32-class (a 22-bit
                   operand: logits 3e-6 from an fp64 forward, the fp32 kernels 4.5e-6); bf16x3 carries a 16-bit operand —
                   logits 4.5e-5 from fp64, ten times fp32's error, 1e-5-class. "f16" / "bf16": one 16-bit pass.
                   Tokens can differ from the fp32 decode at near-ties; tools/precision_agreement.py reports it.
  skip_unchanged   True (default): exact work-skipping. A candidate that unmasked nothing is a copy of
                   its parent x_t (:1203) and, time_conditioning being off (:334-335), every net output for it is the
                   parent's: SVDD-MC evaluates the value net only on the live candidates and takes a copy's score from
                   the parent (= the score of the candidate selected one step earlier); SVDD-PM runs the candidate
                   backbone forward, x0-hat and reward on the live candidates only and never re-runs the parent forward
                   (the selected candidate's logits ARE the next parent's). With logits_cache the backbone also skips
                   rows whose x_t did not change. Compaction is done on the device; no host round trip in the loop.
                   Decodes are bit-identical to skip_unchanged = False (tests/test_skip_gpu.py). Needs the fused nets.
  logits_cache     "auto" (default): SVDD-MC keeps a per-row logits cache when several sequences share a backbone tile
                   (L <= 104; at L = 200 one workgroup owns one sequence and skipping rows frees CUs but saves no
                   time); "on" / "off". (SVDD-PM always carries the selected candidate's logits forward.)
  dedup_prior      True (default): every row of the prior x_T is the same all-MASK row (_sample_prior, :751-753), and with the hand-written
                   kernels a row's net output does not depend on the batch around it — so the FIRST backbone forward of a decode, the
                   parents' first value / reward score and their first tower pass run on one row and are broadcast. Bit-identical to
                   forwarding all B rows (tests/test_skip_gpu.py); one backbone launch of 129 and a value-net pass less per decode.
  dps_one_launch   True (default): the differentiable backbone pass of a DPS step (forward2 on one_hot(x_t) + its input gradient) is ONE
                   launch each way (fp32, CNN backbone, 104 < L <= 208): the forward is the inference kernel bit for bit, so its logits
                   also give q_xs and the reference's second, identical forward (:1306 vs :1324) is not run. False: the layer-wise
                   autograd path of round 3 (20 + 20 convolution launches and a pass per layer each way).
  dps_single_forward  (only where the one-launch pair does not apply) False (default): a DPS step runs the no-grad forward for q_xs AND the differentiable pass for the gradient,
                   like the reference (:1306 and :1324 evaluate one function twice). True (opt-in): the differentiable pass
                   (forward2) also supplies the log-probs of q_xs — equal to round-off (~1e-6), so zero guidance is no longer
                   bit-for-bit the un-guided decode at near-ties.
  skip_stats       None, or a dict the samplers fill with device-side hit counters (live candidates, changed rows).
  skip_generic     False (default). True: SVDD-MC also skips the copies of the parent for an OPAQUE value function (any
                   nn.Module, e.g. the Enformer-shaped trunk): the live candidates are gathered into a smaller batch whose
                   size is read back once per step (one host sync, negligible next to a net that takes milliseconds).
                   Mathematically identical; bit-identical only if the module's kernels do not depend on the batch size
                   (vendor libraries pick — and sometimes build — kernels per size; with a large MIOpen-backed trunk the
                   first decode pays for every new live-batch size, rounded to 256 rows here), hence opt-in. The one-pass modes "bf16" / "f16" run an
                   opaque value / reward net under torch.autocast; the x3 modes leave it in fp32.
"""
import warnings
import weakref

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import noise_schedule, ops
from .backbone import CNNModel


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def weight_fingerprint(*modules, content=True):
    """Identity of the weights of `modules`: (storage address, in-place version counter, shape) of every parameter and
    buffer — changes when a tensor is replaced (load_state_dict with assign, .to()) or modified in place through autograd-
    visible ops (optimizer step, load_state_dict copy_) — plus, with `content`, a CONTENT checksum (1- and 2-norm of every
    floating tensor, two multi-tensor launches and one read-back): writes through `.data` do not bump `_version`
    (`p.data.copy_(...)` is how the reference swaps EMA weights around sampling, models/ema.py:62,87 with
    diffusion_gosai.py:1564-1574), and a cached re-packing of the weights must not survive them. -> (meta, content)."""
    meta, tensors = [], []
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            meta.append((t.data_ptr(), t._version, tuple(t.shape)))
            if t.is_floating_point() and t.numel():
                tensors.append(t.detach())
    chk = None
    if content and tensors and not _capturing():
        by_dev = {}
        for t in tensors:
            by_dev.setdefault((t.device, t.dtype), []).append(t)
        parts = []
        for ts in by_dev.values():
            parts.append(torch.stack(torch._foreach_norm(ts, 1) + torch._foreach_norm(ts, 2)).double().cpu())
        chk = tuple(torch.cat(parts).tolist())
    return tuple(meta), chk


def _same_weights(a, b):
    """Fingerprints equal; a side without a content checksum (taken during graph capture) compares by meta only."""
    return a[0] == b[0] and (a[1] is None or b[1] is None or a[1] == b[1])


def _decode_scope(fn):
    """Marks one call of a public sampler method: the fused-net caches are validated against the modules' weights once
    per outermost call (a decode), not at every diffusion step."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        if self._scope_depth == 0:
            self._scope_id += 1
        self._scope_depth += 1
        ok = False
        try:
            out = fn(self, *a, **k)
            ok = True
        finally:
            self._scope_depth -= 1
            if self._scope_depth == 0 and self._replay_stream is not None:
                st, self._replay_stream = self._replay_stream, None
                st.close()                 # the advanced mt19937 state goes back into torch's global CPU generator
        if ok and self._scope_depth == 0 and not _capturing():
            from .fused import check_backbone_split
            check_backbone_split()         # a small-batch backbone launch whose workgroups could not all be resident: raise, never return its tokens
        return out
    return wrapped


class Diffusion(nn.Module):
    replays_global_stream = True     # distributed.sharded_sample: replay mode shards by replaying the whole batch's stream per rank

    def __init__(self, config, backbone=None):
        super().__init__()
        self.config = config
        self.vocab_size = 4                                   # diffusion_gosai.py:85
        self.sampler = config.sampling.predictor
        self.mask_index = self.vocab_size                     # :94
        self.vocab_size += 1                                  # :95
        self.parameterization = config.parameterization
        if backbone is not None:
            self.backbone = backbone
        elif config.backbone == "cnn":
            self.backbone = CNNModel(config.model, alphabet_size=self.vocab_size, num_cls=3)   # :100-101
        elif config.backbone == "dit":
            # dead in the reference snapshot (models/__init__.py:1, needs CUDA-only flash_attn); here via ROCm SDPA
            from .dit import DIT
            self.backbone = DIT(config.model, vocab_size=self.vocab_size)                      # :102-104
        else:
            raise ValueError(f"Unknown backbone: {config.backbone}")
        if self.parameterization != "subs":
            raise ValueError("only the `subs` parameterization is on the reference's decode path "
                             "(configs_gosai/config_gosai.yaml:13)")
        self.T = config.T
        self.subs_masking = config.subs_masking
        self.noise = noise_schedule.get_noise(config)
        self.time_conditioning = config.time_conditioning
        self.neg_infinity = -1000000.0
        self.sampling_eps = config.training.sampling_eps
        # engine knobs
        self.rng_mode = "replay"
        self.value_batching = "batched"
        self.select_mode = "argmax"
        self.philox_seed = 0
        self.row_offset = 0
        self._shard = None               # (lo, hi, total, world) while distributed.sharded_sample runs this model
        self.fuse_nets = True
        self.precision = "f32"
        self.skip_unchanged = True
        self.late_steps_from = "auto"    # when FusedValueNet may run the live candidates as two parts (split_gru_rounds): "auto" = when the LAST
                                         # step's live count (read back asynchronously, never waited for) came within 3 % of one GRU round;
                                         # a number = from that fraction of the steps on (rounds 4-5: 0.8, tuned to the random-init benchmark)
        self.dedup_prior = True          # the prior's rows are identical (all MASK): its net evaluations run on ONE row (exact; see _prior_logits)
        self.pm_two_part = False         # SVDD-PM skipping loop: the live candidates as whole backbone rounds + remainder, the first part's reward net under the remainder (_pm_split_rows)
        self.dps_fused = True            # DPS: the whole step on hand-written kernels, no autograd (_dps_fused_nets); False: round 5's autograd path between the same big kernels
        self.dps_one_launch = True       # DPS: the differentiable backbone pass as one launch each way (svdd_backbone_cnn_save_f32 / _grad_f32) where it applies
        self._dps_hard_onehot, self._dps_raw_logits = False, None
        self.dps_single_forward = False  # DPS opt-in: q_xs from the differentiable pass's log-probs, not from a second backbone forward per step
        self.logits_cache = "auto"
        self.skip_stats = None
        self.skip_generic = False
        self.trace = None          # set to a list to record (logits, scores) of every step (tests / smoke)
        self.state_trace = None    # set to a list to record x_t (uint8 clone) at the start of every step + the final x
        self._sched_cache = {}
        self._fused = {}
        self._scope_depth, self._scope_id = 0, 0
        self.fuse_trunk_f32 = True       # precision "f32" + Enformer-shaped value trunk: the hand-written fp32 trunk kernels (False: the PyTorch modules)
        self.replay_rng = "device"       # rng_mode "replay": "device" = torch's CPU mt19937 stream continued by K8 on the GPU for
        self._replay_stream = None       # the span of a sampler call; "host" = torch.rand on the host + upload (round 1-3)
        self._replay_checked = None      # scope id of the sharded replay decode whose generator state was compared across ranks

    # ------------------------------------------------------------------ plumbing ----
    @property
    def device(self):
        return next(self.parameters()).device

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise ops.SvddError("the SVDD sampler runs on the GPU only (move the model with .cuda()); "
                                "there is no CPU fallback for the hot path")

    def _schedule(self, num_steps, eps):
        key = (num_steps, eps)
        if key not in self._sched_cache:
            tab, timesteps, dt = noise_schedule.move_chance_table(self.noise, num_steps, eps)
            self._sched_cache[key] = (tab.numpy().copy(), timesteps, dt)
        return self._sched_cache[key]

    def _tokens_u8(self, x):
        return x if x.dtype == torch.uint8 else x.to(torch.uint8)

    def clear_fused(self):
        """Drop the cached fused formulations and the backbone's cached zero-sigma time biases. Every cache entry carries
        a fingerprint of the weights it was built from (tensor identity, in-place version AND a content checksum, so
        `.data` / EMA swaps are caught too) and is rebuilt at the next decode when that no longer matches; call this to
        force it, or after changing weights in the middle of a per-step loop inside one sampler call."""
        self._fused = {}
        self._cpk_fp = None
        if isinstance(self.backbone, CNNModel):
            self.backbone.clear_time_bias_cache()
            self.backbone._cpk_key = None

    def _validate_conv_packs(self):
        """The packed dilated-conv weights of CNNModel._trunk_cl (DPS gradient path) are keyed by tensor identity + version
        inside the backbone; a `.data` / EMA swap changes neither. Validate them like the fused nets: content fingerprint,
        once per decode scope (every call outside one)."""
        bb = self.backbone
        if self._checked_now(getattr(self, "_cpk_stamp", None)):
            return
        fp = weight_fingerprint(*bb.convs)
        old = getattr(self, "_cpk_fp", None)
        if old is None or not _same_weights(old, fp):
            bb._cpk_key = None
        self._cpk_fp, self._cpk_stamp = fp, self._scope_id

    def _checked_now(self, stamp):
        return self._scope_depth > 0 and stamp == self._scope_id

    def _fused_backbone(self):
        ent = self._fused.get("backbone")
        if ent is None or not self._checked_now(ent[2]):
            fp = weight_fingerprint(self.backbone)
            if ent is None or not _same_weights(ent[0], fp):
                from .fused import FusedBackbone
                self.backbone.clear_time_bias_cache()
                ent = (fp, FusedBackbone(self.backbone).to(self.device).eval(), self._scope_id)
            else:
                ent = (ent[0], ent[1], self._scope_id)
            self._fused["backbone"] = ent
        ent[1].precision = self.precision
        return ent[1]

    def value_callable(self, embedding, head):
        """The callable the engine uses for `head(embedding(onehot))`: onehot fp32 [n,L,4] -> [n,1,1]."""
        from .value_nets import ConvGRUTrunk, ConvHead
        if (self.fuse_nets and isinstance(embedding, ConvGRUTrunk) and isinstance(head, ConvHead)
                and embedding.gru_tower.gru.hidden_size == 64 and embedding.gru_tower.gru.input_size == 64
                and embedding.gru_tower.gru.num_layers == 1 and next(embedding.parameters()).is_cuda):
            # The fused net holds re-packed COPIES of the weights, so an entry is valid only for these very module
            # objects (weak references: id() alone can be recycled after garbage collection) with these very
            # weights (fingerprint incl. a content checksum). Checked once per decode (_decode_scope), not per step.
            key = ("value", id(embedding), id(head))
            ent = self._fused.get(key)
            alive = ent is not None and ent[0]() is embedding and ent[1]() is head
            if not (alive and self._checked_now(ent[4])):
                fp = weight_fingerprint(embedding, head)
                if not (alive and _same_weights(ent[2], fp)):
                    from .fused import FusedValueNet
                    for k in [k for k, v in self._fused.items() if k != "backbone" and (v[0]() is None or v[1]() is None)]:
                        del self._fused[k]                          # entries of collected modules
                    ent = (weakref.ref(embedding), weakref.ref(head), fp,
                           FusedValueNet(embedding, head).to(self.device).eval(), self._scope_id)
                else:
                    ent = ent[:4] + (self._scope_id,)
                self._fused[key] = ent
            ent[3].precision = self.precision
            return ent[3]
        from .enformer_value import EnformerTrunk
        if (self.fuse_nets and (self.precision != "f32" or self.fuse_trunk_f32) and isinstance(embedding, EnformerTrunk)
                and isinstance(head, ConvHead) and next(embedding.parameters()).is_cuda):
            # BASELINE configs[3]'s Enformer-shaped trunk on the hand-written kernels (svdd_trunk.hip): "f32" = one fp32 operand
            # plane, fp32 MFMAs (the reference's precision; round 4); the x3 modes map to bf16x3 (bf16 keeps fp32's exponent
            # range: no operand scaling needed), the one-pass modes to bf16
            tp = "f32" if self.precision == "f32" else "bf16x3" if self.precision.endswith("x3") else "bf16"
            key = ("trunk", id(embedding), id(head), tp)
            ent = self._fused.get(key)
            alive = ent is not None and ent[0]() is embedding and ent[1]() is head
            if not (alive and self._checked_now(ent[4])):
                fp = weight_fingerprint(embedding, head)
                if not (alive and _same_weights(ent[2], fp)):
                    from .fused_trunk import FusedEnformerValueNet
                    for k in [k for k, v in self._fused.items() if k != "backbone" and (v[0]() is None or v[1]() is None)]:
                        del self._fused[k]
                    # a trunk whose GEMM shapes the kernels do not take (output channels must come in 128s, input channels in
                    # 32s: a 384-channel toy trunk has a 192-channel stem) stays on the PyTorch modules — said once, and ONLY
                    # for that documented reason: an assertion while packing a supported trunk is a bug and propagates
                    ok, why = FusedEnformerValueNet.supports(embedding, head)
                    if ok:
                        fused_net = FusedEnformerValueNet(embedding, head, tp)
                    else:
                        fused_net = None
                        warnings.warn(f"Enformer-shaped value trunk stays on the PyTorch modules: {why}", stacklevel=2)
                    ent = (weakref.ref(embedding), weakref.ref(head), fp, fused_net, self._scope_id)
                else:
                    ent = ent[:4] + (self._scope_id,)
                self._fused[key] = ent
            if ent[3] is not None:
                return ent[3]
        if self.precision in ("bf16", "f16"):                       # opaque nets: PyTorch-ROCm's own 16-bit kernels
            dt = torch.bfloat16 if self.precision == "bf16" else torch.float16

            def autocast_value(onehot):
                with torch.autocast("cuda", dtype=dt):
                    return head(embedding(onehot)).float()
            return autocast_value
        return lambda onehot: head(embedding(onehot))

    def reward_callable(self, reward_model):
        """The callable the engine uses for `reward_model(onehot_t [n,4,L])` -> [n,n_tasks,1]."""
        from .value_nets import RewardModel
        if self.fuse_nets and isinstance(reward_model, RewardModel):
            fused = self.value_callable(reward_model.embedding, reward_model.head)
            if isinstance(fused, nn.Module):
                return fused
        return reward_model

    def _backbone_logits(self, x_u8):
        """Raw backbone output for tokens x (sigma is zeroed when time_conditioning is False, :334-335)."""
        if isinstance(self.backbone, CNNModel) and not self.time_conditioning:
            if self.fuse_nets and x_u8.is_cuda and self.backbone.args.hidden_dim in (64, 128, 256):
                return self._fused_backbone()(x_u8)
            return self.backbone(x_u8, None, zero_sigma=True)
        sigma = torch.zeros(x_u8.shape[0], device=x_u8.device)
        return self.backbone(x_u8.long(), sigma).float()

    def _prior_dedup_ok(self, x_u8):
        """True when the net evaluations of the prior state x_u8 (every row all-MASK by construction) may run on one row: more than
        one row, and the one-launch backbone kernel (a row's logits are then the same bits wherever and with whomever it is evaluated)."""
        return (self.dedup_prior and x_u8.shape[0] > 1 and x_u8.is_cuda and not _capturing() and
                self._fused_backbone_or_none(x_u8.shape[1]) is not None)

    def _prior_logits(self, x_u8):
        """Backbone logits of the PRIOR state (the first forward of every sampler loop): one row forwarded, broadcast to all B."""
        if self._prior_dedup_ok(x_u8):
            B, L = x_u8.shape
            return self._backbone_logits(x_u8[:1].contiguous()).expand(B, L, self.vocab_size).contiguous()
        return self._backbone_logits(x_u8)

    def _step_scalars(self, t, dt):
        """(mct, mcs, mct - mcs) for an explicit (t, dt) as the per-step API receives them
        (:1176-1187), evaluated on the host with the reference's fp32 ops."""
        t0 = t.reshape(-1)[:1].detach().float().cpu().view(1, 1)
        sigma_t, _ = self.noise(t0)
        sigma_s, _ = self.noise(t0 - dt)
        mct = 1 - torch.exp(-sigma_t.squeeze(-1))
        mcs = 1 - torch.exp(-sigma_s.squeeze(-1))
        return float(mct), float(mcs), float(mct - mcs)

    def _step_index(self, t, dt):
        """Index i of the diffusion step a per-step call belongs to: the reference loops call the per-step methods with
        t_i = 1 - i * dt (diffusion_gosai.py:1036-1043). It keys the Philox counter, so that a caller who drives the
        per-step API draws fresh uniforms at every step (with a constant key a position that stays MASK would see the
        same Gumbel noise again and again)."""
        t0 = float(t.reshape(-1)[0])
        return max(0, int(round((1.0 - t0) / float(dt))))

    def _replay_layout(self, logits):
        """Memory order in which the REFERENCE consumes its uniforms: rand_like(q_xs) fills in the
        memory order of the reference backbone's output — [b][v][l] for its CNN, whose output is a
        permuted view (models/dnaconv.py:201) — whatever layout this engine's backbone emits."""
        if self.config.backbone == "cnn":
            return ops.LAYOUT_BVL
        return ops.layout_of(logits)[1]

    def _rng(self, step, M, B, L, logits):
        if self.rng_mode == "replay":
            ul = self._replay_layout(logits)
            # a rank of a batch-sharded decode replays the WHOLE batch's stream and reads its rows (svdd_rng.uniforms_rows)
            shard = self._shard if (self._shard is not None and self._shard[3] > 1) else None
            rows = shard[2] if shard else B
            shape = (M, rows, L, 5) if ul == ops.LAYOUT_BLV else (M, rows, 5, L)
            extra = dict(row_offset=shard[0], uniforms_rows=rows) if shard else {}
            if shard and not (self._scope_depth > 0 and self._replay_checked == self._scope_id):
                # every rank replays the whole batch's stream and slices its rows: only the reference's run if all ranks start from
                # the same generator state (a rank seeded per rank, or one that drew anything extra, would silently decode other tokens)
                import hashlib
                from . import distributed
                digest = hashlib.sha1(torch.get_rng_state().numpy().tobytes()).digest()
                distributed.assert_same_on_all_ranks(int.from_bytes(digest[:8], "little", signed=True),
                                                     "sharded replay decode: torch's CPU generator state (seed every rank identically)")
                self._replay_checked = self._scope_id
            if self.replay_rng == "device" and self._scope_depth > 0 and logits.is_cuda and not _capturing():
                # the same stream, generated on the device (svdd_mt19937_uniform_f32): the generator's state is uploaded at the
                # first draw of a sampler call and written back into torch's global generator when the call returns
                if self._replay_stream is None:
                    self._replay_stream = ops.DeviceReplayStream(logits.device)
                u = self._replay_stream.uniforms(M * rows * L * 5).view(shape)
            else:
                u = torch.rand(shape).to(logits.device, non_blocking=True)   # torch's global CPU generator
            return ops.Rng(uniforms=u, uniforms_layout=ul, **extra)
        if self.rng_mode == "philox":
            return ops.Rng(seed=self.philox_seed, row_offset=self.row_offset, step=step)
        raise ValueError(f"rng_mode {self.rng_mode!r}")

    def _select(self, scores, cand, step):
        mode = {"argmax": ops.SELECT_ARGMAX, "multinomial": ops.SELECT_MULTINOMIAL}[self.select_mode]
        rng = None
        if mode == ops.SELECT_MULTINOMIAL:
            if self.rng_mode != "philox":
                raise ValueError("select_mode='multinomial' needs rng_mode='philox'")
            rng = ops.Rng(seed=self.philox_seed, row_offset=self.row_offset, step=step)
        x_next, _, _ = ops.select(scores, cand, mode=mode, rng=rng, want_soft=False)
        return x_next

    def _value_scores(self, embedding, head, onehot, B, M, cand=None, x_u8=None):
        """scores[b, m] = head(embedding(onehot of candidate m of sample b)) (:1207-1209,1219)."""
        fn = self.value_callable(embedding, head)
        if self.value_batching == "batched":
            if cand is not None and hasattr(fn, "candidates_ok") and fn.candidates_ok(onehot.shape[1], M):
                return fn.forward_candidates(onehot, cand, x_u8).reshape(B, M).float()
            return fn(onehot).reshape(B, M).float()
        oh = onehot.view(B, M, onehot.shape[1], 4)
        return torch.stack([fn(oh[:, m].contiguous()).reshape(B) for m in range(M)], dim=1).float()

    def _batch_size(self, eval_sp_size):
        return self.config.loader.eval_batch_size if eval_sp_size is None else eval_sp_size

    def _num_steps(self, num_steps):
        return self.config.sampling.steps if num_steps is None else num_steps

    def _noise_removal(self, x_u8, logits=None):
        """:1049-1060 — x = forward(x, sigma(t_last))[:, :, :-1].argmax(-1) ; returns int64. `logits`: the backbone
        output for x_u8 when the caller already holds it (work-skipping SVDD-PM)."""
        if self.config.sampling.noise_removal:
            if self.sampler == "analytic":
                raise NotImplementedError("analytic sampler is not on the reference's decode path")
            if logits is None:
                logits = self._backbone_logits(x_u8)
            self._record(logits, None, x_u8)
            return ops.finalize(logits, x_u8)
        return x_u8.long()

    def _record(self, logits, scores, x=None):
        if self.trace is not None:
            self.trace.append((logits.detach().clone(), None if scores is None else scores.detach().clone()))
        if self.state_trace is not None and x is not None:
            self.state_trace.append(x.detach().clone())

    # ---------------------------------------------------------- reference API: basics ----
    def _process_sigma(self, sigma):
        if sigma.ndim > 1:
            sigma = sigma.squeeze(-1)
        if not self.time_conditioning:
            sigma = torch.zeros_like(sigma)
        assert sigma.ndim == 1, sigma.shape
        return sigma

    @_decode_scope
    def forward(self, x, sigma):
        """Returns log score: backbone logits under the SUBS parameterization (:339-357)."""
        self._require_gpu()
        sigma = self._process_sigma(sigma)
        x_u8 = self._tokens_u8(x)
        return ops.subs_logp(self._backbone_logits(x_u8), x_u8)

    def forward2(self, x_onehot, x, sigma):
        """Differentiable log score on a one-hot input (:359-377), used by the DPS baseline. Autograd
        must see every op, so the SUBS step is expressed in torch here."""
        sigma = self._process_sigma(sigma)
        fb = self._dps_one_launch(x_onehot) if self._dps_hard_onehot else None
        if fb is not None:
            # DPS: x_onehot IS one_hot(x) (:1308) — the whole backbone as ONE launch each way (forward = the inference kernel's
            # bits + saved statistics, backward = svdd_backbone_cnn_grad_f32): no layer-wise launches, no saved activations rows
            logits = fb.forward_with_grad(x_onehot, self._tokens_u8(x))
            self._dps_raw_logits = logits.detach()
        elif isinstance(self.backbone, CNNModel):
            # dilated convs on the hand-written kernel, both directions — for this call only (the flag does not outlive it:
            # a later backbone.forward2 by the user gets plain autograd unless asked otherwise)
            was = self.backbone.hip_convs
            self.backbone.hip_convs = bool(self.fuse_nets)
            if self.backbone.hip_convs:
                self._validate_conv_packs()
            try:
                logits = self.backbone.forward2(x_onehot, sigma)
            finally:
                self.backbone.hip_convs = was
        else:
            logits = self.backbone.forward2(x_onehot, sigma)
        neg = torch.zeros(self.vocab_size, device=logits.device)
        neg[self.mask_index] = self.neg_infinity
        logits = logits + neg
        logits = logits - torch.logsumexp(logits, dim=-1, keepdim=True)
        unmasked = (x != self.mask_index)
        fixed = torch.full_like(logits, self.neg_infinity).scatter(-1, x.clamp(max=self.vocab_size - 1)[..., None], 0.0)
        return torch.where(unmasked[..., None], fixed, logits)

    def _dps_one_launch(self, x_onehot):
        """The fused backbone when the differentiable pass of a DPS step can run as one launch each way, else None."""
        if not (self.fuse_nets and self.dps_one_launch and x_onehot.is_cuda and not self.time_conditioning
                and isinstance(self.backbone, CNNModel) and not self.backbone.training):
            return None
        fb = self._fused_backbone_or_none(x_onehot.shape[1])
        return fb if fb is not None and fb.grad_ok(x_onehot.shape[1]) else None

    def _sample_prior(self, *batch_dims):
        return self.mask_index * torch.ones(*batch_dims, dtype=torch.int64)          # :751-753

    def transform_samples(self, samples, num_classes=4):
        """tokens -> one-hot(4), MASK rows zero; int64 like the reference's F.one_hot (:1462-1470)."""
        if samples.is_cuda and num_classes == 4:
            return ops.transform_samples(self._tokens_u8(samples)).long()
        mask = samples != 4
        return F.one_hot((samples * mask).long(), num_classes=num_classes) * mask.unsqueeze(-1)

    # --------------------------------------------------------------- per-step updates ----
    @_decode_scope
    @torch.no_grad()
    def _ddpm_update_finetune(self, x, t, dt):
        """Un-guided ancestral step (:1147-1172) -> (x_next, x, q_xs, copy_flag)."""
        self._require_gpu()
        mct, mcs, dm = self._step_scalars(t, dt)
        x_u8 = self._tokens_u8(x)
        logits = self._backbone_logits(x_u8)
        B, L = x_u8.shape
        cand, _, q = ops.propose(logits, x_u8, dm, mcs, 1, self._rng(self._step_index(t, dt), 1, B, L, logits), want_q=True)
        return cand[:, 0].long(), x, q, (x != self.mask_index).to(x.dtype)

    @_decode_scope
    @torch.no_grad()
    def _ddpm_update_finetune_controlled(self, x, t, dt, pre_scorer_embedding, pre_scorer_head, repeats=10):
        """One SVDD-MC step (:1174-1228) -> (final_samples, x, q_xs, copy_flag)."""
        self._require_gpu()
        mct, mcs, dm = self._step_scalars(t, dt)
        x_u8 = self._tokens_u8(x)
        logits = self._backbone_logits(x_u8)
        B, L = x_u8.shape
        step = self._step_index(t, dt)
        cand, onehot, q = ops.propose(logits, x_u8, dm, mcs, repeats, self._rng(step, repeats, B, L, logits), want_q=True)
        scores = self._value_scores(pre_scorer_embedding, pre_scorer_head, onehot, B, repeats, cand, x_u8)
        x_next = self._select(scores, cand, step)
        return x_next.long(), x, q, (x != self.mask_index).to(x.dtype)

    @_decode_scope
    @torch.no_grad()
    def _ddpm_update_finetune_controlled_twedie(self, x, t, dt, reward_model, repeats=10, options="True", task="dna"):
        """One SVDD-PM step (:1373-1460) -> (final_samples, x, q_xs, copy_flag)."""
        self._require_gpu()
        mct, mcs, dm = self._step_scalars(t, dt)
        x_u8 = self._tokens_u8(x)
        logits = self._backbone_logits(x_u8)
        B, L = x_u8.shape
        step = self._step_index(t, dt)
        cand, _, q = ops.propose(logits, x_u8, dm, mcs, repeats, self._rng(step, repeats, B, L, logits), want_q=True)
        scores = self._tweedie_scores(cand, reward_model, options, task)
        x_next = self._select(scores, cand, step)
        return x_next.long(), x, q, (x != self.mask_index).to(x.dtype)

    def _tweedie_scores(self, cand, reward_model, options, task):
        """scores[b,m] = reward_model(x0hat(candidate))[:, 0] (:1413-1436)."""
        if task == "rna_saluki":
            raise NotImplementedError("rna_saluki needs a data file the reference hard-codes by absolute path "
                                      "(diffusion_gosai.py:1478); unsupported offline")
        B, M, L = cand.shape
        flat = cand.reshape(B * M, L)
        if options == "True":
            oh, _ = ops.x0hat(self._backbone_logits(flat), flat)          # :1415-1419
        else:
            oh = ops.transform_samples(flat, transposed=True)             # heuristic branch :1420-1424
        return self.reward_callable(reward_model)(oh)[:, 0].reshape(B, M).float()   # :1430,1436

    @_decode_scope
    @torch.no_grad()
    def _ddpm_update_finetune_controlled_TDS(self, x, t, dt, reward_model, alpha=1.0):
        """One SMC/TDS step (:1230-1284) -> x_next. Consumes B doubles of numpy's global RandomState,
        like the reference's np.random.choice."""
        self._require_gpu()
        mct, mcs, dm = self._step_scalars(t, dt)
        x_u8 = self._tokens_u8(x)
        return self._tds_step(x_u8, dm, mcs, reward_model, alpha, self._step_index(t, dt)).long()

    def _tds_step(self, x_u8, dm, mcs, reward_model, alpha, step, carry=None):
        """One SMC/TDS step. `carry`: None, or a dict the decode loop threads through the steps for EXACT reuse: the
        resampled particles x_next = sample[idx] are copies of proposals whose backbone logits and x0-hat reward this step
        already computed, so the next step's forward(x_next) and its denominator reward are row gathers of this step's
        forward(sample) and numerator reward (2 of the 3 net evaluations of a step; bit-identical because the hand-written
        kernels' output for a row does not depend on where the row sits in the batch)."""
        B, L = x_u8.shape
        have = carry is not None and "logits" in carry
        prior = carry is not None and carry.pop("prior", False)           # first step of a decode: x is the all-MASK prior
        logits = carry["logits"] if have else (self._prior_logits(x_u8) if prior else self._backbone_logits(x_u8))
        cand, _, _ = ops.propose(logits, x_u8, dm, mcs, 1, self._rng(step, 1, B, L, logits))
        sample = cand[:, 0].contiguous()
        logits_s = self._backbone_logits(sample)
        oh_num, _ = ops.x0hat(logits_s, sample)                           # :1263-1268
        reward_fn = self.reward_callable(reward_model)
        reward_num = reward_fn(oh_num)[:, 0][:, 0].float()                # :1269
        if have and "den" in carry:
            reward_den = carry["den"]
        else:
            # forward(x, sigma_s) == forward(x, sigma_t): sigma is zeroed (:334-335), so `logits` is reused (:1273)
            oh_den, _ = ops.x0hat(logits, x_u8)
            reward_den = reward_fn(oh_den)[:, 0][:, 0].float()            # :1277
        keep_logits = carry is not None and carry.get("keep_logits")
        keep_den = carry is not None and carry.get("keep_den")
        shard = getattr(self, "_shard", None)
        if shard is not None and shard[3] > 1:
            # batch sharded over GPUs: the resample draws ancestors from the WHOLE batch — one all-gather, then every rank
            # resamples the whole batch identically and keeps its rows (distributed.tds_exchange)
            from . import distributed
            u_all = torch.from_numpy(np.random.random_sample(shard[2]))
            extra = logits_s.reshape(B, -1) if keep_logits else None
            sample_all, num_all, den_all, u, extra_all = distributed.tds_exchange(shard, sample, reward_num, reward_den,
                                                                                  u_all, extra)
            x_all, idx = ops.tds_resample(num_all, den_all, alpha, sample_all, u)
            mine = idx[shard[0]:shard[1]].long()
            x_next, src_logits, src_num = x_all[shard[0]:shard[1]].contiguous(), extra_all, num_all
        else:
            u = torch.from_numpy(np.random.random_sample(B)).to(x_u8.device)   # what np.random.choice draws (:1282)
            x_next, idx = ops.tds_resample(reward_num, reward_den, alpha, sample, u)
            mine, src_logits, src_num = idx.long(), logits_s.reshape(B, -1), reward_num
        if carry is not None:
            carry.pop("logits", None), carry.pop("den", None)
            if keep_logits:
                carry["logits"] = src_logits.index_select(0, mine).view(B, *logits_s.shape[1:])
            if keep_den:
                carry["den"] = src_num.index_select(0, mine)
        return x_next

    def _tds_carry(self, reward_model, L):
        """What a TDS decode may reuse from step to step (see _tds_step): only results of kernels whose output for a row
        is independent of the batch around it — the one-launch backbone, the hand-written value-net kernels."""
        if not (self.skip_unchanged and self.fuse_nets):
            return None
        from .fused import FusedValueNet
        fn = self.reward_callable(reward_model)
        return {"keep_logits": self._fused_backbone_or_none(L) is not None,
                "keep_den": isinstance(fn, FusedValueNet) and fn.kernels_ok(L)}

    # ------------------------------------------------------------------ DPS baseline ----
    def compute_gradient_DPS(self, x_onehot, x, reward_model, sigma_s, copy_flag):
        """d mean(reward(softmax(E[x0|x_t]))) / d onehot(x_t)  (reference :1321-1330). Pure autograd: the
        differentiable backbone entry `forward2` and the reward net are run as plain torch modules."""
        x_onehot.requires_grad_(True)
        keep = copy_flag[:, :, None]
        logp = self.forward2(x_onehot, x, sigma_s)
        self._dps_logp = logp.detach()                            # log p(x0 | x_t): _dps_guided_q takes its q_xs from it
        expected_x0 = keep * x_onehot + (1 - keep) * logp
        probs = torch.softmax(expected_x0, dim=2)
        # MIOpen's fused RNN backward insists on train(). A GRU without inter-layer dropout computes the same function in
        # both modes, so only those modules are switched for the call (BatchNorm / Dropout stay in eval): 113 -> 54 ms per
        # gradient at B = 256 against the per-timestep native cells (400 cell launches forward + backward). Any other
        # recurrent module falls back to the native cells.
        # ... and where the GRU is the reward net's 64-unit bidirectional one, neither is used: the recurrence runs on the
        # hand-written forward + BPTT kernels (csrc/svdd_gru_train.hip; 35.9 -> ~1 ms of the gradient at B = 256).
        fn = self.reward_callable(reward_model) if (self.fuse_nets and self.dps_one_launch and x_onehot.is_cuda) else None
        from .fused import FusedValueNet
        if isinstance(fn, FusedValueNet) and fn.grad_ok(x_onehot.shape[1]):
            # the reward net without MIOpen (round 5): convolutions on svdd_conv1d_cl_f32 both ways, GRU on the BPTT kernels
            scores = fn.forward_grad(probs[:, :, 0:4].contiguous())[:, 0]
            scores.mean().backward()
            return x_onehot.grad.clone()
        hip = self._hip_gru_blocks(reward_model) if (self.fuse_nets and x_onehot.is_cuda) else []
        taken = {id(b.gru) for b in hip}
        rnns = [m for m in reward_model.modules() if isinstance(m, torch.nn.RNNBase) and id(m) not in taken]
        flip = [m for m in rnns if isinstance(m, torch.nn.GRU) and m.dropout == 0 and not m.training]
        native = len(flip) != len([m for m in rnns if not m.training])
        for m in flip:
            m.train()
        try:
            with torch.backends.cudnn.flags(enabled=not native):
                scores = reward_model(probs.transpose(1, 2)[:, 0:4, :])[:, 0]
            scores.mean().backward()
        finally:
            for m in flip:
                m.eval()
            for b in hip:
                b._hip_gru = None
        return x_onehot.grad.clone()

    def _hip_gru_blocks(self, reward_model):
        """GRUBlocks of `reward_model` whose nn.GRU the kernels of csrc/svdd_gru_train.hip take (64 -> 64, one layer,
        bidirectional, eval mode or no dropout), switched to them; weights re-packed when their fingerprint changes."""
        from .fused import GruBidirFunction, pack_gru, pack_gru_bwd
        from .value_nets import GRUBlock
        blocks = []
        for b in reward_model.modules():
            g = getattr(b, "gru", None)
            if not (isinstance(b, GRUBlock) and isinstance(g, torch.nn.GRU) and g.hidden_size == 64 and g.input_size == 64 and
                    g.num_layers == 1 and g.bidirectional and g.bias and g.batch_first and not g.training):
                continue
            cache = self.__dict__.setdefault("_gru_train_packs", {})   # id(gru) -> (weak ref, fingerprint, packs)
            fp = weight_fingerprint(g)
            ent = cache.get(id(g))
            if ent is None or ent[0]() is not g or not _same_weights(ent[1], fp):
                wpack, bpack = pack_gru(g)
                dev = next(g.parameters()).device
                ent = (weakref.ref(g), fp, (wpack.to(dev), bpack.to(dev), pack_gru_bwd(g).to(dev)))
                for k in [k for k, v in cache.items() if v[0]() is None]:
                    del cache[k]
                cache[id(g)] = ent
            packs = ent[2]
            b._hip_gru = lambda xx, p=packs: GruBidirFunction.apply(xx, *p)
            blocks.append(b)
        return blocks

    def _dps_guided_q(self, x_u8, mcs, dm, reward_model, guidance_scale):
        """The guided transition weights q_xs of one DPS step (:1306-1314) -> fp32 [B, L, 5]."""
        B, L = x_u8.shape
        fused = self._dps_fused_nets(x_u8, reward_model)
        if fused is not None:
            # round 6: the whole step without autograd — the one-launch backbone pair, the per-position pieces (K9: svdd_dps_probs /
            # _probs_bwd / _guided_q) and the reward net's gradient pass on hand-written kernels (FusedValueNet.mean_score_input_grad):
            # 22 launches where rounds 4-5 issued ~200 (torch element-wise ops and their autograd twins between the kernels)
            from .fused import backbone_cnn_grad, backbone_cnn_save
            fb, fn = fused
            with torch.no_grad():
                pk = fb.ol_pack()
                logits, saved = backbone_cnn_save(x_u8.contiguous(), pk)              # the inference kernel's bits + saved statistics
                dprobs = fn.mean_score_input_grad(ops.dps_probs(logits, x_u8))        # :1325-1329
                dlogits, direct = ops.dps_probs_bwd(logits, x_u8, dprobs)
                dx_bb = backbone_cnn_grad(dlogits, pk, fb.grad_pack(), saved)
                return ops.dps_guided_q(logits, x_u8, dx_bb, direct, dm, mcs, guidance_scale)     # :1306-1314
        x = x_u8.long()
        copy_flag = (x != self.mask_index).to(x.dtype)
        x_onehot = F.one_hot(x, num_classes=self.vocab_size).float()                          # :1308
        # The reference evaluates the backbone twice per step on the same x_t (forward() for q_xs, :1306 ; forward2() inside the
        # gradient, :1324). With the one-launch pair the differentiable forward IS the inference kernel, bit for bit (same kernel
        # template, plus stores): its raw logits serve both, through the same SUBS kernel — zero guidance stays bit-for-bit the
        # un-guided decode (tests/test_configs_gpu.py), and the second, identical forward is not run.
        # (a split-precision mode keeps its own sampling forward for q_xs — the mode's bits — and takes only the gradient from the
        #  fp32 pair: the differentiable pass has always been fp32)
        pair = self._dps_one_launch(x_onehot) is not None            # the gradient comes from the one-launch fp32 pair ...
        reuse = pair and self.precision == "f32"                     # ... whose forward logits also give q_xs
        legacy_single = self.dps_single_forward and not pair         # round 4's opt-in, layer-wise path only
        if not reuse and not legacy_single:
            with torch.no_grad():
                q_xs = torch.exp(ops.subs_logp(self._backbone_logits(x_u8), x_u8)) * float(dm)   # :1306-1307
        sigma = torch.zeros(B, device=x.device)
        self._dps_hard_onehot = True
        try:
            with torch.enable_grad():
                x_grad = self.compute_gradient_DPS(x_onehot, x, reward_model, sigma, copy_flag)   # :1310
        finally:
            self._dps_hard_onehot = False
        with torch.no_grad():
            if reuse:
                q_xs = torch.exp(ops.subs_logp(self._dps_raw_logits, x_u8)) * float(dm)       # :1306-1307 on the very same logits
            elif legacy_single:
                # The reference evaluates the backbone twice per step on the same x_t with the same (zeroed) sigma: forward() for
                # q_xs (:1306) and forward2() inside the gradient (:1324). They are one function; the differentiable pass's log-probs
                # are taken for both (round-off apart: ~1e-6, far inside the 1e-3 .. 1e-2 that the gradient's own ReLU decisions
                # move it by). dps_single_forward = False (the default) runs the second forward like the reference.
                q_xs = torch.exp(self._dps_logp) * float(dm)
            self._dps_raw_logits = self._dps_logp = None
            guidance = guidance_scale * (x_grad - x_grad[:, :, self.mask_index][:, :, None])   # :1311
            q_xs[:, :, self.mask_index] = float(mcs)                                          # :1312
            return q_xs * guidance.exp()                                                      # :1314

    def _dps_fused_nets(self, x_u8, reward_model):
        """(fused backbone, fused reward net) when a DPS step can run without autograd, else None: the one-launch fp32 backbone pair at
        this length, the reference-shaped ConvGRU reward net on the hand-written kernels, exact-fp32 mode (a split-precision mode keeps
        its own sampling forward for q_xs: the autograd path), dps_fused on."""
        if not (self.dps_fused and self.dps_one_launch and self.fuse_nets and self.precision == "f32" and x_u8.is_cuda
                and not self.time_conditioning and isinstance(self.backbone, CNNModel) and not self.backbone.training):
            return None
        L = x_u8.shape[1]
        fb = self._fused_backbone_or_none(L)
        if fb is None or not fb.grad_ok(L):
            return None
        from .fused import FusedValueNet
        fn = self.reward_callable(reward_model)
        if not (isinstance(fn, FusedValueNet) and fn.grad_ok(L)):
            return None
        return fb, fn

    def _dps_step(self, x_u8, mct, mcs, dm, reward_model, guidance_scale, step):
        B, L = x_u8.shape
        q_xs = self._dps_guided_q(x_u8, mcs, dm, reward_model, guidance_scale)
        with torch.no_grad():
            cand, _ = ops.sample_categorical(q_xs, x_u8, 1, self._rng(step, 1, B, L, q_xs))   # :1316-1319
        return cand.view(B, L)

    @_decode_scope
    def _ddpm_update_finetune_controlled_DPS(self, x, t, dt, reward_model, guidance_scale):
        """One DPS (gradient-guidance) step (:1286-1319) -> x_next."""
        self._require_gpu()
        mct, mcs, dm = self._step_scalars(t, dt)
        return self._dps_step(self._tokens_u8(x), mct, mcs, dm, reward_model, guidance_scale, self._step_index(t, dt)).long()

    @_decode_scope
    def controlled_sample_DPS(self, reward_model, guidance_scale, num_steps=None, eps=1e-5, eval_sp_size=None,
                              sample_M=10):
        """DPS baseline decode (:980-1019). Not under no_grad in the reference either: it back-propagates."""
        self._require_gpu()
        B, L, S = self._batch_size(eval_sp_size), self.config.model.length, self._num_steps(num_steps)
        sched, _, _ = self._schedule(S, eps)
        x = torch.full((B, L), self.mask_index, dtype=torch.uint8, device=self.device)
        for i in range(S):
            x = self._dps_step(x, sched[i, 0], sched[i, 1], sched[i, 2], reward_model, guidance_scale, i)
        with torch.no_grad():
            return self._noise_removal(x)

    # ------------------------------------------------------------------ outer loops ----
    @_decode_scope
    @torch.no_grad()
    def decode_sample(self, num_steps=None, eps=1e-5, eval_sp_size=None, cdq=False):
        """Un-guided decode (:888-936) -> LongTensor[B,L]."""
        self._require_gpu()
        B, L, S = self._batch_size(eval_sp_size), self.config.model.length, self._num_steps(num_steps)
        sched, _, _ = self._schedule(S, eps)
        x = torch.full((B, L), self.mask_index, dtype=torch.uint8, device=self.device)
        for i in range(S):
            logits = self._prior_logits(x) if i == 0 else self._backbone_logits(x)
            cand, _, _ = ops.propose(logits, x, sched[i, 2], sched[i, 1], 1, self._rng(i, 1, B, L, logits))
            x = cand.view(B, L)
        return self._noise_removal(x)

    @_decode_scope
    @torch.no_grad()
    def _sample(self, num_steps=None, eps=1e-5, eval_sp_size=None, cdq=False):
        """Un-guided decode that also returns the S-1 intermediate states (:820-886)."""
        self._require_gpu()
        if cdq:
            raise NotImplementedError("cdq=True is a value-function *training* data path (Enformer.py:163-267)")
        B, L, S = self._batch_size(eval_sp_size), self.config.model.length, self._num_steps(num_steps)
        sched, _, _ = self._schedule(S, eps)
        x = torch.full((B, L), self.mask_index, dtype=torch.uint8, device=self.device)
        mid_x = []
        for i in range(S):
            logits = self._prior_logits(x) if i == 0 else self._backbone_logits(x)
            cand, _, _ = ops.propose(logits, x, sched[i, 2], sched[i, 1], 1, self._rng(i, 1, B, L, logits))
            x = cand.view(B, L)
            if i != S - 1:
                mid_x.append(x.long())
        return self._noise_removal(x), mid_x

    @_decode_scope
    @torch.no_grad()
    def controlled_sample(self, pre_scorer_embedding, pre_scorer_head, num_steps=None, eps=1e-5,
                          eval_sp_size=None, sample_M=10):
        """SVDD-MC decode (:1021-1061): S x [backbone -> propose -> value net -> select], then noise removal."""
        self._require_gpu()
        B, L, S, M = self._batch_size(eval_sp_size), self.config.model.length, self._num_steps(num_steps), sample_M
        sched, _, _ = self._schedule(S, eps)
        x = torch.full((B, L), self.mask_index, dtype=torch.uint8, device=self.device)   # _sample_prior
        cand = torch.empty((B, M, L), dtype=torch.uint8, device=self.device)
        onehot = torch.empty((B * M, L, 4), dtype=torch.float32, device=self.device)
        fn = self.value_callable(pre_scorer_embedding, pre_scorer_head)
        if self._can_skip(fn, L, M):
            return self._controlled_sample_skipping(fn, x, cand, onehot, sched, B, L, S, M)
        if (self.skip_unchanged and self.skip_generic and M > 1 and self.value_batching == "batched" and
                self.select_mode in ("argmax", "multinomial")):
            return self._controlled_sample_generic_skipping(fn, x, cand, onehot, sched, B, L, S, M)
        for i in range(S):
            logits = self._prior_logits(x) if i == 0 else self._backbone_logits(x)
            ops.propose(logits, x, sched[i, 2], sched[i, 1], M, self._rng(i, M, B, L, logits), cand=cand, onehot=onehot)
            scores = self._value_scores(pre_scorer_embedding, pre_scorer_head, onehot, B, M, cand, x)
            self._record(logits, scores, x)
            x = self._select(scores, cand, i)
        return self._noise_removal(x)

    # ------------------------------------------------------------- exact work-skipping ----
    def _can_skip(self, fn, L, M):
        """The skipping paths hand the nets compacted batches whose size only the device knows: they need the
        hand-written kernels for the value / reward net and (PM, logits cache) the one-launch backbone kernel."""
        from .fused import FusedValueNet
        from .fused_trunk import FusedEnformerValueNet
        ok = ((isinstance(fn, FusedValueNet) and fn.kernels_ok(L) and fn.w_eff.shape[1] == 1) or
              (isinstance(fn, FusedEnformerValueNet) and fn.head_w.shape[1] == 1))
        return (self.skip_unchanged and self.fuse_nets and self.value_batching == "batched" and M > 1 and ok and
                self.select_mode in ("argmax", "multinomial"))

    def _fused_backbone_or_none(self, L):
        if isinstance(self.backbone, CNNModel) and not self.time_conditioning and self.fuse_nets:
            fb = self._fused_backbone()
            if fb.kernel_ok(L):
                return fb
        return None

    class _SkipWorkspace:
        def __init__(self, B, M, dev):
            i32 = dict(dtype=torch.int32, device=dev)
            self.flags, self.live_idx, self.slot = (torch.empty(B * M, **i32) for _ in range(3))
            self.count3 = torch.zeros(3, **i32)           # [live candidates, of them in the list's first part, in its second] (svdd_compact_by_key split)
            self.count = self.count3[0:1]
            self.row_idx, self.row_slot = torch.empty(B, **i32), torch.empty(B, **i32)
            self.row_count = torch.zeros(1, **i32)
            self.parent_score, self.sel_score = torch.empty(B, device=dev), torch.empty(B, device=dev)
            self.changed, self.idx = torch.empty(B, **i32), torch.empty(B, **i32)
            self.parent_out, self.parent_out_lp, self.seq = None, None, None   # the parents' tower output (FusedValueNet)
            self.M = M
            self.n_live = torch.zeros(1, dtype=torch.int64, device=dev)
            self.n_changed = torch.zeros(1, dtype=torch.int64, device=dev)
            self.late = False                   # set by the sampler per step: the live candidates may exceed one GRU round (FusedValueNet.split_gru_rounds)
            self.split_bufs = None              # persistent buffers + side stream of that path
            self.prior_rows_identical = False   # set by the sampler when x is the prior: the parents' first tower pass runs on one row
            self.n_win_rows = None          # with skip_stats: rows the value net's tower computed (the candidates' row windows)

    def _select_compact(self, sc, ws, cand, step):
        mode = {"argmax": ops.SELECT_ARGMAX, "multinomial": ops.SELECT_MULTINOMIAL}[self.select_mode]
        rng = None
        if mode == ops.SELECT_MULTINOMIAL:
            if self.rng_mode != "philox":
                raise ValueError("select_mode='multinomial' needs rng_mode='philox'")
            rng = ops.Rng(seed=self.philox_seed, row_offset=self.row_offset, step=step)
        x_next, _, _, _ = ops.select_compact(sc, ws.slot, ws.parent_score, cand, mode=mode, rng=rng, sel_score=ws.sel_score,
                                             changed=ws.changed, idx=ws.idx)
        ws.parent_score, ws.sel_score = ws.sel_score, ws.parent_score       # the selected candidate is the next parent
        if ws.parent_out is not None and ws.seq is not None:                # ... and so is its tower output
            ops.advance_rows(ws.seq, ws.slot, ws.idx, ws.parent_out, ws.M)
        if self.skip_stats is not None:
            ws.n_live += ws.count
            ws.n_changed += ws.changed.sum()
        return x_next

    def _dense_scores(self, sc, ws, B, M):
        """[B, M] scores from the compacted ones (only for the trace: the loop itself never materialises them)."""
        live = ws.slot >= 0
        return torch.where(live, sc[ws.slot.clamp(min=0).long()], ws.parent_score.repeat_interleave(M)).view(B, M)

    def _finish_stats(self, ws, B, M, S, kind):
        if self.skip_stats is not None:
            self.skip_stats.update(kind=kind, steps=S, candidates=B * M * S, live_candidates=int(ws.n_live),
                                   row_steps=B * S, changed_row_steps=int(ws.n_changed),
                                   tower_window_rows=None if ws.n_win_rows is None else int(ws.n_win_rows))

    def _use_logits_cache(self, L):
        """Per-row logits cache of the SVDD-MC skipping loop. "auto": where several sequences share a backbone tile
        (L <= 104) — at L = 200 one workgroup owns one sequence and skipping rows frees CUs but saves no time."""
        return self.logits_cache == "on" or (self.logits_cache == "auto" and 208 // L >= 2)

    def _controlled_sample_skipping(self, fn, x, cand, onehot, sched, B, L, S, M):
        """SVDD-MC with exact work-skipping (same tokens as the plain loop, bit for bit). 104 < L <= 208 (one sequence per
        tile): the value net's tower also shares the parent's rows (candidate_scores_compact); shorter sequences: the
        live candidates' token rows are gathered into a compact batch and scored whole (forward_tokens)."""
        from .fused import candidate_windows
        ws = self._SkipWorkspace(B, M, self.device)
        if self.skip_stats is not None:
            ws.n_win_rows = torch.zeros(1, dtype=torch.int64, device=self.device)
        from .fused import FusedValueNet
        dedup = self._prior_dedup_ok(x) and isinstance(fn, FusedValueNet)   # the parents are B copies of the all-MASK row
        ws.parent_score.copy_(fn.forward_tokens(x[:1].contiguous()).reshape(1).expand(B) if dedup
                              else fn.forward_tokens(x).reshape(B))          # scores of the all-MASK parents
        ws.prior_rows_identical = dedup                                      # -> the parents' first tower pass too (FusedValueNet)
        fb = self._fused_backbone_or_none(L) if self._use_logits_cache(L) else None
        share = hasattr(fn, "candidates_ok") and fn.candidates_ok(L, M)
        toks_c = None if share else torch.empty((B * M, L), dtype=torch.uint8, device=self.device)
        logits = None
        # the two-part late steps adapt to the decode in hand: the live count of a step is copied to pinned host memory without
        # waiting; a later step looks at the newest count that has arrived (a trained value net, another M or L, another chip change
        # when the live candidates outgrow one GRU round — the fixed "last 20 % of the steps" of rounds 4-5 fitted random-init nets)
        auto_late = share and self.late_steps_from == "auto" and not _capturing() and hasattr(fn, "gru_round_rows")
        if auto_late:
            live_host = torch.empty(1, dtype=torch.int32).pin_memory()
            live_ev, live_pending, live_last = torch.cuda.Event(), False, 0
            late_thr = 0.97 * fn.gru_round_rows()
        for i in range(S):
            if fb is None or logits is None:
                logits = self._prior_logits(x) if i == 0 else self._backbone_logits(x)
            else:                                                             # only the rows the last select changed
                ops.compact_flags(ws.changed, ws.row_idx, ws.row_slot, ws.row_count)
                fb.forward_rows(x, count=ws.row_count, out=logits, row_idx=ws.row_idx, scatter=True)
            ops.propose(logits, x, sched[i, 2], sched[i, 1], M, self._rng(i, M, B, L, logits), cand=cand, onehot=onehot)
            if share:
                if auto_late:
                    if live_pending and live_ev.query():
                        live_last, live_pending = int(live_host[0]), False
                    ws.late = live_last > late_thr
                else:
                    frac = 0.8 if self.late_steps_from == "auto" else float(self.late_steps_from)
                    ws.late = i >= int(frac * S)                             # late steps: (almost) every candidate is live
                sc = fn.candidate_scores_compact(onehot, cand, x, ws).reshape(-1)
                if auto_late and not live_pending:
                    live_host.copy_(ws.count, non_blocking=True)
                    live_ev.record()
                    live_pending = True
            else:
                candidate_windows(cand, x, margin=0, flags=ws.flags)
                ops.compact_flags(ws.flags, ws.live_idx, ws.slot, ws.count)
                ops.gather_rows(cand.view(B * M, L), ws.live_idx, ws.count, toks_c)
                if getattr(fn, "share_level0", False):                       # the Enformer-shaped trunk: first level on the changed windows only
                    sc = fn.forward_tokens(toks_c, count=ws.count, shared=(x, ws.live_idx, M)).reshape(-1)
                else:
                    sc = fn.forward_tokens(toks_c, count=ws.count).reshape(-1)
            if self.trace is not None or self.state_trace is not None:
                self._record(logits, self._dense_scores(sc, ws, B, M), x)
            x = self._select_compact(sc, ws, cand, i)
        self._finish_stats(ws, B, M, S, "mc")
        return self._noise_removal(x)

    def _controlled_sample_generic_skipping(self, fn, x, cand, onehot, sched, B, L, S, M):
        """SVDD-MC work-skipping for an opaque value function: live candidates gathered into a smaller batch (its size is
        read back once per step), copies take the parent's score (= the score of the candidate selected a step earlier)."""
        from .fused import candidate_windows
        ws = self._SkipWorkspace(B, M, self.device)
        first = fn(ops.transform_samples(x))
        if first.numel() != B:
            raise ops.SvddError("skip_generic needs a single-task value function (one score per candidate); "
                                f"got {tuple(first.shape)} for a batch of {B}")
        ws.parent_score.copy_(first.reshape(B).float())
        sc = torch.zeros(B * M, device=self.device)
        for i in range(S):
            logits = self._prior_logits(x) if i == 0 else self._backbone_logits(x)
            ops.propose(logits, x, sched[i, 2], sched[i, 1], M, self._rng(i, M, B, L, logits), cand=cand, onehot=onehot)
            candidate_windows(cand, x, margin=0, flags=ws.flags)
            ops.compact_flags(ws.flags, ws.live_idx, ws.slot, ws.count)
            k = int(ws.count)                                                 # the one host round trip of the step
            if k:
                # batch sizes in steps of 256 rows (padded with repeats of the first live row): vendor libraries pick and
                # sometimes compile kernels per problem size, and k takes a new value at almost every step
                kp = min(B * M, -(-k // 256) * 256)
                idx = ws.live_idx[:kp].long()
                if kp > k:
                    idx = torch.cat([idx[:k], idx[:1].expand(kp - k)])
                sc[:k] = fn(onehot.index_select(0, idx)).reshape(kp)[:k].float()
            if self.trace is not None or self.state_trace is not None:
                self._record(logits, self._dense_scores(sc, ws, B, M), x)
            x = self._select_compact(sc, ws, cand, i)
        self._finish_stats(ws, B, M, S, "mc-generic")
        return self._noise_removal(x)

    def _pm_split_rows(self, fb, rf, n, L):
        """Entries of part A of a two-part SVDD-PM step (0: one part): one full round of full backbone tiles — CUs x (208 // L)
        sequences — when several sequences share a tile, the candidates can exceed it and the reward net takes an output buffer."""
        from .fused import FusedValueNet
        if not (self.pm_two_part and isinstance(rf, FusedValueNet) and rf.w_eff.shape[1] == 1 and 208 // L >= 2) or _capturing():
            return 0
        from . import _lib
        n_a = _lib.device_info()[1] * (208 // L)
        return n_a if n_a < n else 0

    def _tweedie_sample_skipping(self, rf, x, sched, B, L, S, M, fb):
        """SVDD-PM with exact work-skipping: per step ONE backbone forward, on the live candidates only."""
        dev = self.device
        ws = self._SkipWorkspace(B, M, dev)
        n = B * M
        cand = torch.empty((B, M, L), dtype=torch.uint8, device=dev)
        toks_c = torch.empty((n, L), dtype=torch.uint8, device=dev)
        lg_c = torch.empty((n, L, 5), dtype=torch.float32, device=dev)
        from .fused import FusedValueNet
        dedup = self._prior_dedup_ok(x) and isinstance(rf, FusedValueNet)     # the parents are B copies of the all-MASK row
        logits = (fb.forward_rows(x[:1].contiguous()).expand(B, L, self.vocab_size).contiguous() if dedup
                  else fb.forward_rows(x))                                    # parents' logits; advanced, never recomputed
        _, xh = ops.x0hat(logits, x, want_tokens=True, want_onehot=False)
        ws.parent_score.copy_(rf.forward_tokens(xh[:1].contiguous()).reshape(1).expand(B) if dedup
                              else rf.forward_tokens(xh).reshape(B))          # reward of the parents' x0-hat
        from .fused import candidate_windows
        # Two-part steps (round 6): the live candidates of a step are rarely a whole number of rounds of backbone tiles (1400 of L = 50
        # are one full round of four-sequence tiles + a remainder round with a third of the chip idle). The compacted list runs as
        # A = its first `n_a` entries (whole rounds) and B = the rest: backbone(A), then backbone(B) with A's x0-hat + reward net on a
        # side stream filling the CUs B leaves idle. Same kernels on the same rows (a row's result does not depend on its batch or
        # tile): same bits (tests/test_skip_gpu.py).
        n_a = self._pm_split_rows(fb, rf, n, L)
        if n_a:
            side, ev_a, ev_done = ops.side_stream(dev, 0), torch.cuda.Event(), torch.cuda.Event()
            sc_buf = torch.empty((n, rf.w_eff.shape[1]), dtype=torch.float32, device=dev)
            c_a, c_b = ws.count3[1:2], ws.count3[2:3]
            cand_rows = cand.view(n, L)
        for i in range(S):
            ops.propose(logits, x, sched[i, 2], sched[i, 1], M, self._rng(i, M, B, L, logits), cand=cand)
            candidate_windows(cand, x, margin=0, flags=ws.flags)
            if n_a:
                main = torch.cuda.current_stream()
                ops.compact_by_key(ws.flags, ws.live_idx, ws.slot, ws.count3, split=n_a)     # keys are 0 / 1: the order of compact_flags
                ops.gather_rows(cand_rows, ws.live_idx, ws.count, toks_c)
                fb.forward_rows(cand_rows, count=c_a, out=lg_c[:n_a], row_idx=ws.live_idx, scatter=False)
                ev_a.record(main)
                fb.forward_rows(cand_rows, count=c_b, out=lg_c[n_a:], row_idx=ws.live_idx[n_a:], scatter=False)
                with torch.cuda.stream(side):
                    side.wait_event(ev_a)
                    _, xh_a = ops.x0hat(lg_c[:n_a], toks_c[:n_a], want_tokens=True, want_onehot=False)
                    rf.forward_tokens(xh_a, count=c_a, out=sc_buf[:n_a])
                    ev_done.record(side)
                _, xh_b = ops.x0hat(lg_c[n_a:], toks_c[n_a:], want_tokens=True, want_onehot=False)
                rf.forward_tokens(xh_b, count=c_b, out=sc_buf[n_a:])
                main.wait_event(ev_done)
                sc = sc_buf.reshape(-1)
            else:
                ops.compact_flags(ws.flags, ws.live_idx, ws.slot, ws.count)
                fb.forward_rows(cand.view(n, L), count=ws.count, out=lg_c, row_idx=ws.live_idx, scatter=False)
                ops.gather_rows(cand.view(n, L), ws.live_idx, ws.count, toks_c)
                _, xh = ops.x0hat(lg_c, toks_c, want_tokens=True, want_onehot=False)   # :1415-1419 on the compacted rows
                sc = rf.forward_tokens(xh, count=ws.count).reshape(-1)                 # :1430
            if self.trace is not None or self.state_trace is not None:
                self._record(logits, self._dense_scores(sc, ws, B, M), x)
            x_next = self._select_compact(sc, ws, cand, i)
            ops.advance_rows(lg_c, ws.slot, ws.idx, logits, M)                # the selected candidate's logits are the next parent's
            x = x_next
        self._finish_stats(ws, B, M, S, "pm")
        return self._noise_removal(x, logits)

    @_decode_scope
    @torch.no_grad()
    def controlled_sample_tweedie(self, reward_model, num_steps=None, eps=1e-5, eval_sp_size=None, sample_M=10,
                                  options=True, task="dna"):
        """SVDD-PM decode (:1105-1145). NB the reference compares `options == "True"` (a string, :1414):
        the default `options=True` therefore takes the heuristic branch there too."""
        self._require_gpu()
        B, L, S, M = self._batch_size(eval_sp_size), self.config.model.length, self._num_steps(num_steps), sample_M
        sched, _, _ = self._schedule(S, eps)
        x = torch.full((B, L), self.mask_index, dtype=torch.uint8, device=self.device)
        rf = self.reward_callable(reward_model)
        fb = self._fused_backbone_or_none(L)
        if options == "True" and task != "rna_saluki" and fb is not None and self._can_skip(rf, L, M):
            return self._tweedie_sample_skipping(rf, x, sched, B, L, S, M, fb)
        for i in range(S):
            logits = self._prior_logits(x) if i == 0 else self._backbone_logits(x)
            cand, _, _ = ops.propose(logits, x, sched[i, 2], sched[i, 1], M, self._rng(i, M, B, L, logits))
            scores = self._tweedie_scores(cand, reward_model, options, task)
            self._record(logits, scores, x)
            x = self._select(scores, cand, i)
        return self._noise_removal(x)

    @_decode_scope
    @torch.no_grad()
    def controlled_sample_TDS(self, reward_model, alpha, num_steps=None, eps=1e-5, eval_sp_size=None, sample_M=10):
        """SMC/TDS baseline decode (:938-978)."""
        self._require_gpu()
        B, L, S = self._batch_size(eval_sp_size), self.config.model.length, self._num_steps(num_steps)
        sched, _, _ = self._schedule(S, eps)
        x = torch.full((B, L), self.mask_index, dtype=torch.uint8, device=self.device)
        carry = self._tds_carry(reward_model, L)
        if carry is not None:
            carry["prior"] = True
        for i in range(S):
            if self.state_trace is not None:
                self.state_trace.append(x.detach().clone())
            x = self._tds_step(x, sched[i, 2], sched[i, 1], reward_model, alpha, i, carry)
        return self._noise_removal(x, carry.get("logits") if carry else None)

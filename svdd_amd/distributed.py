"""Batch-sharded decode over the GPUs of one node: one process per GPU, RCCL only for the single
all-gather of the final decoded tokens. Rows of the batch are independent in
SVDD-MC / SVDD-PM / un-guided decode, so nothing is exchanged per step; Philox draws are keyed by
the GLOBAL row index, so the decoded batch is identical for any number of GPUs. rng_mode="replay" (parity mode) shards too:
every rank — seeded identically, like the reference's single process — generates the WHOLE batch's uniforms of a step from
torch's mt19937 stream (on the device) and K1 reads the rows of its shard, so the gathered batch is token for token the
unsharded replay decode, i.e. the reference's run at the total batch size.

The SMC/TDS baseline is the one sampler whose step couples rows (the resample draws ancestors from the WHOLE batch,
reference diffusion_gosai.py:1279-1284): `tds_exchange` all-gathers each rank's proposals, both reward vectors and its
slice of the uniforms in ONE small collective per step ((L + 16) bytes per row), every rank runs the same K4 resample on
the whole batch and keeps its rows — token-exact against the unsharded decode.

Cost of the sharded replay mode, stated: every rank generates M * total_rows * L * 5 uniforms per step on the one-workgroup
mt19937 kernel (the stream is serial by definition), so the replay cost does NOT shrink with more ranks — it is the parity mode,
not the throughput mode. It is only correct when every rank's torch CPU generator is in the same state when the sampler starts
(seed every rank identically, NOT per rank): Diffusion verifies that with one 8-byte all-gather at the first draw of a sharded
replay decode and raises on a mismatch."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, world_size, local_rank). Initialises torch.distributed when launched by torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            # "nccl" is RCCL on ROCm. SVDD_DIST_BACKEND=gloo: dry runs of the N > 1 code path on a box with fewer GPUs
            # than ranks (several ranks share a device; RCCL refuses that) — never for measurements.
            # The ranks decide for themselves when there are fewer GPUs than local ranks (each rank is a fresh process: counting
            # devices here is free of the launcher's "never touch the GPU before spawning" constraint).
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
            shared = torch.cuda.is_available() and torch.cuda.device_count() < local_world
            backend = os.environ.get("SVDD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() and not shared else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world > 1 and torch.cuda.is_available() and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", str(world))):
        # several ranks on one device (whoever initialised the process group): the small-batch backbone (2 / 4 workgroups per
        # sequence that WAIT for each other, svdd_backbone_cnn_f32) may only be launched when all members of a group are resident at
        # once — which another process's kernels on the same CUs can prevent. One workgroup per sequence then (same bits, no
        # inter-workgroup wait).
        from . import _lib
        _lib.set_option(7, 1)
    return rank, world, local


def assert_same_on_all_ranks(value, what):
    """Collective: every rank contributes one int64; raises on every rank when they are not all equal."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    vals = [int(p.item()) for p in parts]
    if len(set(vals)) > 1:
        raise RuntimeError(f"{what} differs between ranks: {vals}")


def shard_rows(total_rows, rank, world):
    """Contiguous row range [lo, hi) of `rank` (the first total_rows % world ranks get one extra row)."""
    base, rem = divmod(total_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_tokens(local_tokens, total_rows=None):
    """All-gather of the final decoded tokens [B_local, L] -> [B_total, L] on every rank.
    Tokens travel as uint8 (B_total*L bytes: 0.4 MB at B=2048, L=200 — latency-bound over xGMI, so it is
    issued exactly once, after the last step); ragged shards are padded to the largest."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_tokens
    world = dist.get_world_size()
    out_dtype = local_tokens.dtype
    t = local_tokens.to(torch.uint8).contiguous()
    if total_rows is None:
        total_rows = t.shape[0] * world
    sizes = [shard_rows(total_rows, r, world) for r in range(world)]
    maxrows = max(hi - lo for lo, hi in sizes)
    if t.shape[0] < maxrows:
        t = torch.cat([t, t.new_zeros(maxrows - t.shape[0], t.shape[1])], dim=0)
    out = t.new_empty((world * maxrows, t.shape[1]))
    if dist.get_backend() == "gloo" and t.is_cuda:                     # dry-run path: gloo gathers through the host
        parts = [torch.empty_like(t, device="cpu") for _ in range(world)]
        dist.all_gather(parts, t.cpu())
        out = torch.cat(parts).to(t.device)
    else:
        dist.all_gather_into_tensor(out, t)
    parts = [out[r * maxrows: r * maxrows + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0).to(out_dtype)


def tds_exchange(shard, sample, reward_num, reward_den, u_all, extra=None):
    """The cross-rank half of a TDS step. shard = (lo, hi, total, world) ; sample [b, L] u8, reward_* [b] f32 of this
    rank's rows ; u_all [total] f64 this rank's draw of the step's uniforms (only its own slice travels, so all ranks
    agree on the assembled vector even if their numpy streams differ; with equal seeds it is the unsharded vector).
    extra: optional fp32 [b, K] per-row payload that rides along (the proposals' backbone logits, which the next step
    reuses for the resampled particles).
    -> (sample, reward_num, reward_den, u, extra | None) of the whole batch, identical on every rank."""
    lo, hi, total, world = shard
    if not dist.is_initialized() or dist.get_world_size() != world:
        raise RuntimeError("the TDS resample couples every row of the batch: a sharded TDS decode needs an initialised "
                           "process group of the shard's world size (launch one process per GPU)")
    b, L = sample.shape
    K = 0 if extra is None else 4 * extra.shape[1]
    pack = torch.empty((b, L + 16 + K), dtype=torch.uint8, device=sample.device)
    pack[:, :L] = sample
    pack[:, L:L + 4] = reward_num.float().contiguous().view(torch.uint8).view(b, 4)
    pack[:, L + 4:L + 8] = reward_den.float().contiguous().view(torch.uint8).view(b, 4)
    pack[:, L + 8:L + 16] = u_all[lo:hi].to(sample.device).contiguous().view(torch.uint8).view(b, 8)
    if K:
        pack[:, L + 16:] = extra.float().contiguous().view(torch.uint8).view(b, K)
    g = gather_tokens(pack, total)
    f32 = lambda a, c: g[:, a:a + c].contiguous().view(torch.float32 if c == 4 else torch.float64).view(total)  # noqa: E731
    extra_all = g[:, L + 16:].contiguous().view(torch.float32).view(total, K // 4) if K else None
    return g[:, :L].contiguous(), f32(L, 4), f32(L + 4, 4), f32(L + 8, 8), extra_all


def sharded_sample(model, total_rows, sampler, rank=None, world=None):
    """Runs `sampler(eval_sp_size=<rows of this rank>)` with the model's Philox row offset set to this
    rank's first global row, then gathers. `sampler` is e.g.
    `lambda **kw: model.controlled_sample(emb, head, sample_M=M, **kw)`."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world > 1 and getattr(model, "rng_mode", "philox") != "philox" and not getattr(model, "replays_global_stream", False):
        # replay mode draws from each process's own CPU mt19937 stream: with the usual identical manual_seed every rank
        # would decode the SAME rows and the gathered batch would be `world` copies of one shard. svdd_amd.Diffusion does it
        # properly (replays_global_stream): every rank generates the WHOLE batch's uniforms from its (identically seeded)
        # generator and K1 reads the rank's row slice (svdd_rng.uniforms_rows / row_offset) — SURVEY.md section 8e's parity mode
        raise ValueError("sharded_sample needs rng_mode='philox' when world > 1 (Philox is keyed by the global row; the "
                         "replay stream is per process)")
    lo, hi = shard_rows(total_rows, rank, world)
    prev = model.row_offset
    model.row_offset = lo
    model._shard = (lo, hi, total_rows, world)            # read by the one sampler that exchanges per step (TDS)
    try:
        local = sampler(eval_sp_size=hi - lo)
    finally:
        model.row_offset = prev
        model._shard = None
    return gather_tokens(local, total_rows)

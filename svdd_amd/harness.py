"""Decode harness — mirror of the reference's `Enformer.BaseModel.controlled_decode*`
(reference Enformer.py:399-477, 479-557, 719-813) for the SVDD decode path.

The reference's `BaseModel.__init__` hard-wires checkpoint paths, Hydra and `.cuda()`
(Enformer.py:75-131); there are no checkpoints offline, so here the three nets are passed in:

    BaseModel(embedding, head, ref_model, reward_model, batch_size, task="dna", n_tasks=1)

`controlled_decode*` keep the reference's call order (guided batches first, then
`gen_batch_num * sample_M` un-guided baseline batches, then the top-k) and return the same
5-tuple: (samples list, value_func_preds [N], reward_model_preds [N], top_k_values, baseline_preds [N]).
"""
import torch
from torch import nn

_M64 = (1 << 64) - 1


def batch_seed(base, k):
    """64-bit Philox key of batch k of a harness call whose model carries philox_seed = base: splitmix64 of
    (base, k). `base + k` would make batch k + 1 of seed s the same decode as batch k of seed s + 1, so runs at seeds
    0, 1, 2, ... would share all but one of their batches and understate the run-to-run variance."""
    z = (int(base) + 0x9E3779B97F4A7C15 * (int(k) + 1)) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


class BaseModel(nn.Module):
    def __init__(self, embedding, head, ref_model, reward_model, batch_size, task="dna", n_tasks=1,
                 val_batch_num=0):
        super().__init__()
        self.task = task
        self.n_tasks = n_tasks
        self.embedding = embedding
        self.head = head
        self.ref_model = ref_model.eval()
        self.reward_model = reward_model.eval()
        for m in (self.ref_model, self.reward_model):
            for p in m.parameters():
                p.requires_grad_(False)
        self.NUM_SAMPLES_PER_BATCH = batch_size
        self.val_data_num = val_batch_num * batch_size
        if task == "rna_saluki":
            raise NotImplementedError("rna_saluki needs a data file the reference hard-codes by absolute path")
        # The reference pre-samples val_batch_num batches here to build value-fn eval data
        # (Enformer.py:135-160), which advances the global RNG before any decode; kept for stream parity.
        self.eval_time_step_batches, self.eval_time_step_targets = [], []
        if val_batch_num > 0:
            self._presample(val_batch_num)

    @torch.no_grad()
    def _presample(self, val_batch_num):
        steps = self.ref_model.config.sampling.steps
        per_t_samples = [[] for _ in range(steps)]
        per_t_targets = [[] for _ in range(steps)]
        for _ in range(val_batch_num):
            samples, mid = self.ref_model._sample(eval_sp_size=self.NUM_SAMPLES_PER_BATCH)
            target = self._reward(samples)
            for j, s in enumerate(mid + [samples]):
                per_t_samples[j].append(self.transform_samples(s))
                per_t_targets[j].append(target)
        self.eval_time_step_batches = [torch.cat(s, dim=0) for s in per_t_samples]
        self.eval_time_step_targets = [torch.cat(t, dim=0) for t in per_t_targets]

    def transform_samples(self, samples, num_classes=4):
        return self.ref_model.transform_samples(samples, num_classes)        # Enformer.py:269-277

    def _reward(self, tokens):
        """reward_model(onehot.float().transpose(1, 2)) with the n_tasks convention of Enformer.py:446-449."""
        onehot = self.transform_samples(tokens).float().transpose(1, 2)
        pred = self.reward_model(onehot).detach()
        return pred[:, 0] if self.n_tasks == 1 else pred

    def _value(self, tokens):
        onehot = self.transform_samples(tokens).float()
        return self.head(self.embedding(onehot)).squeeze(2).detach()          # Enformer.py:443

    def _next_batch_seed(self):
        """Philox mode: the sampler is a pure function of (philox_seed, row, step), so consecutive batches would be
        copies of each other; the reference's batches differ because they share torch's global generator
        (Enformer.py:439-467). Batch k of a harness call therefore gets the key batch_seed(base, k) (deterministic given
        the model's philox_seed at entry, and never shared between different entry seeds); replay mode keeps drawing
        from the global generator like the reference."""
        m = self.ref_model
        if getattr(m, "rng_mode", "replay") == "philox":
            m.philox_seed = batch_seed(self._seed_base, self._batch_index)
        self._batch_index += 1

    def _decode(self, gen_batch_num, sample_M, guided, tweedie_quirks=False):
        self._seed_base, self._batch_index = int(getattr(self.ref_model, "philox_seed", 0)), 0
        try:
            return self._decode_inner(gen_batch_num, sample_M, guided, tweedie_quirks)
        finally:
            if hasattr(self.ref_model, "philox_seed"):
                self.ref_model.philox_seed = self._seed_base

    def _decode_inner(self, gen_batch_num, sample_M, guided, tweedie_quirks=False):
        """`tweedie_quirks`: what `controlled_decode_tweedie` does differently from its three siblings (pinned by
        tests/golden/g22_harness.npz, recorded from the reference's own method): `samples.extend(batch)` — a flat list of
        [L] rows instead of a list of [B, L] batches (Enformer.py:766 vs :441) — and `top_k_values = cat(baseline_preds)`,
        the top-k being commented out there (:802-811)."""
        samples, value_func_preds, reward_model_preds = [], [], []
        for _ in range(gen_batch_num):
            self._next_batch_seed()
            batch = guided()
            if tweedie_quirks:
                samples.extend(batch)
            else:
                samples.append(batch)
            value_func_preds.extend(self._value(batch))
            reward_model_preds.extend(self._reward(batch))
        print("Value-weighted sampling done.")
        baseline_preds, all_preds = [], []
        for i in range(gen_batch_num * sample_M):                             # Enformer.py:456-467
            self._next_batch_seed()
            batch = self.ref_model.decode_sample(eval_sp_size=self.NUM_SAMPLES_PER_BATCH)
            pred = self._reward(batch)
            if i < gen_batch_num:
                baseline_preds.extend(pred)
            all_preds.extend(pred)
        if tweedie_quirks:
            top_k_values = torch.cat(baseline_preds)                           # Enformer.py:802
        else:
            print("Baseline sampling done.")
            all_values = torch.cat(all_preds)
            k = int(len(all_values) / sample_M)                                # Enformer.py:471-475
            top_k_values, _ = torch.topk(all_values, k)
        return (samples, torch.cat(value_func_preds), torch.cat(reward_model_preds), top_k_values,
                torch.cat(baseline_preds))

    @torch.no_grad()
    def controlled_decode(self, gen_batch_num, sample_M):
        """SVDD-MC (reference Enformer.py:399-477)."""
        return self._decode(gen_batch_num, sample_M, lambda: self.ref_model.controlled_sample(
            self.embedding, self.head, eval_sp_size=self.NUM_SAMPLES_PER_BATCH, sample_M=sample_M))

    @torch.no_grad()
    def controlled_decode_tweedie(self, gen_batch_num, sample_M, options):
        """SVDD-PM (reference Enformer.py:719-813)."""
        return self._decode(gen_batch_num, sample_M, lambda: self.ref_model.controlled_sample_tweedie(
            self.reward_model, eval_sp_size=self.NUM_SAMPLES_PER_BATCH, sample_M=sample_M, options=options,
            task=self.task), tweedie_quirks=True)

    @torch.no_grad()
    def controlled_decode_TDS(self, gen_batch_num, sample_M, alpha):
        """SMC/TDS baseline (reference Enformer.py:479-557)."""
        return self._decode(gen_batch_num, sample_M, lambda: self.ref_model.controlled_sample_TDS(
            self.reward_model, alpha, eval_sp_size=self.NUM_SAMPLES_PER_BATCH, sample_M=sample_M))

    def controlled_decode_DPS(self, gen_batch_num, sample_M, guidance_scale):
        """DPS baseline (reference Enformer.py:560-637). The guided decodes back-propagate through the backbone
        and the reward net, so they run with autograd on; the evaluation / baseline part runs under no_grad.
        (Scoring consumes no RNG, so decoding all guided batches first keeps the reference's RNG order.)"""
        m = self.ref_model
        base, out = int(getattr(m, "philox_seed", 0)), []
        try:
            for k in range(gen_batch_num):
                if getattr(m, "rng_mode", "replay") == "philox":
                    m.philox_seed = batch_seed(base, k)           # the key _decode would give guided batch k
                out.append(m.controlled_sample_DPS(self.reward_model, guidance_scale,
                                                   eval_sp_size=self.NUM_SAMPLES_PER_BATCH, sample_M=sample_M))
        finally:
            if hasattr(m, "philox_seed"):
                m.philox_seed = base
        batches = iter(out)
        with torch.no_grad():
            return self._decode(gen_batch_num, sample_M, lambda: next(batches))

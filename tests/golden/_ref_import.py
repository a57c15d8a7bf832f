"""Import shim for the *reference* (masa-ue/SVDD, mounted read-only at /root/reference).

Used ONLY by `tests/golden/make_golden.py`, in the build container, to run the
reference's own Python on CPU and capture golden input/output vectors. Nothing
under `tests/` that runs on the GPU box imports this file, and no reference
source is copied: the reference modules are imported from where they lie.

The reference needs ~10 third-party packages that are not installed here
(lightning, hydra, torchmetrics, timm, wandb, grelu, enformer_pytorch, ...).
None of them is on the decode hot path; they are replaced by inert stubs.
"""
import importlib.machinery
import sys
import types

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []  # behave like a package so `import a.b` works
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch
    from torch import nn

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        @property
        def dtype(self):
            return torch.float32

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    lightning = _mod("lightning", LightningModule=LightningModule)
    pl = _mod("lightning.pytorch")
    utilities = _mod("lightning.pytorch.utilities", rank_zero_only=lambda f: f)
    pl.utilities = utilities
    lightning.pytorch = pl

    hydra = _mod("hydra", initialize=None, compose=None)
    hydra.utils = _mod("hydra.utils")
    core = _mod("hydra.core")
    gh = _mod("hydra.core.global_hydra", GlobalHydra=type("GlobalHydra", (), {}))
    core.global_hydra = gh
    hydra.core = core

    class _Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def set_dtype(self, *a, **k):
            return self

        def clone(self, *a, **k):
            return self

    tm = _mod("torchmetrics", MetricCollection=_Metric)
    agg = _mod("torchmetrics.aggregation", MeanMetric=_Metric)
    tm.aggregation = agg

    timm = _mod("timm")
    sched = _mod("timm.scheduler", CosineLRScheduler=type("CosineLRScheduler", (), {}))
    timm.scheduler = sched

    _mod("wandb")
    _mod("oracle")  # the reference's oracle.py (gReLU reward loader), not this repo's oracle/
    _mod("dataloader_gosai")
    grelu = _mod("grelu")
    grelu.lightning = _mod("grelu.lightning", LightningModel=type("LightningModel", (), {}))
    ep = _mod("enformer_pytorch")
    ep.modeling_enformer = _mod(
        "enformer_pytorch.modeling_enformer",
        GELU=nn.GELU, AttentionPool=None, relative_shift=None, Attention=None,
        exponential_linspace_int=None)


class Cfg(dict):
    """Attribute-style dict standing in for the reference's omegaconf config."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return Cfg(v) if isinstance(v, dict) else v


def make_cfg(length=200, hidden_dim=128, num_cnn_stacks=4, steps=128):
    """The keys `Diffusion.__init__` / the samplers read (configs_gosai/config_gosai.yaml)."""
    return Cfg(
        sampling=dict(predictor="ddpm", steps=steps, noise_removal=True),
        eval=dict(gen_ppl_eval_model_name_or_path="gpt2-large"),
        training=dict(antithetic_sampling=True, importance_sampling=False,
                      change_of_variables=False, ema=0.0, sampling_eps=1e-3),
        parameterization="subs", backbone="cnn", T=0, subs_masking=False,
        time_conditioning=False,
        model=dict(hidden_dim=hidden_dim, num_cnn_stacks=num_cnn_stacks, dropout=0.0,
                   clean_data=False, cls_free_guidance=False, length=length),
        noise=dict(type="loglinear"), optim=dict(lr=3e-4),
        loader=dict(eval_batch_size=4),
    )


def import_reference():
    """Returns (diffusion_gosai, Enformer) reference modules."""
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import diffusion_gosai  # noqa
    import Enformer  # noqa
    return diffusion_gosai, Enformer

"""Sampler configuration: the keys `diffusion_gosai.Diffusion.__init__` and its samplers read
from the reference's Hydra YAML (configs_gosai/config_gosai.yaml, configs_gosai/model/dnaconv.yaml;
SURVEY.md §5), as plain dataclasses. DNA vs RNA differ only in `model.length` (200 vs 50)."""
from dataclasses import dataclass, field


@dataclass
class ModelConfig:
    hidden_dim: int = 128          # configs_gosai/model/dnaconv.yaml:11
    num_cnn_stacks: int = 4        # :12
    dropout: float = 0.0           # :13
    clean_data: bool = False       # :14
    cls_free_guidance: bool = False  # :15
    length: int = 200              # :5 (50 for configs_gosai_rna)


@dataclass
class SamplingConfig:
    predictor: str = "ddpm"        # config_gosai.yaml:36
    steps: int = 128               # :37
    noise_removal: bool = True     # :38


@dataclass
class TrainingConfig:
    ema: float = 0.0               # decode never applies EMA weights
    antithetic_sampling: bool = True
    importance_sampling: bool = False
    change_of_variables: bool = False
    sampling_eps: float = 1e-3


@dataclass
class NoiseConfig:
    type: str = "loglinear"        # configs_gosai/noise/loglinear.yaml


@dataclass
class LoaderConfig:
    eval_batch_size: int = 256     # used only when eval_sp_size is None (diffusion_gosai.py:1025-1028)


@dataclass
class Config:
    model: ModelConfig = field(default_factory=ModelConfig)
    sampling: SamplingConfig = field(default_factory=SamplingConfig)
    training: TrainingConfig = field(default_factory=TrainingConfig)
    noise: NoiseConfig = field(default_factory=NoiseConfig)
    loader: LoaderConfig = field(default_factory=LoaderConfig)
    backbone: str = "cnn"          # config_gosai.yaml:12
    parameterization: str = "subs"  # :13
    time_conditioning: bool = False  # :14
    T: int = 0                     # :15
    subs_masking: bool = False     # :16


def dna_config(**model_overrides):
    return Config(model=ModelConfig(length=200, **model_overrides))


def rna_config(**model_overrides):
    return Config(model=ModelConfig(length=50, **model_overrides))


def dit_config(length=200, **dit_overrides):
    """backbone: dit (configs_gosai/model/small.yaml shape by default)."""
    from .dit import DiTModelConfig
    cfg = Config(backbone="dit")
    cfg.model = DiTModelConfig(length=length, **dit_overrides)
    return cfg

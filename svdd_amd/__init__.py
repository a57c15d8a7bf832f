"""svdd_amd — MI355X-native SVDD decoding engine (hot path: per-step propose / score / select).

The hot-path operators live in a hand-written HIP library (csrc/svdd_kernels.hip, C ABI in
include/svdd_hip.h); this package is the host-side mirror of the reference's
`diffusion_gosai.Diffusion` sampler API around it.
"""
__version__ = "0.1.0"

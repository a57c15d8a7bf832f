"""-m gpu: end-to-end parity with REAL nets.

 (1) The reference's tiny nets (tests/golden/nets_tiny.npz) run on the GPU through the engine in replay mode against the
     reference's own recorded `controlled_sample` trajectory (g6, BASELINE.json configs[0]: B=4, L=200, M=2, 128 steps):
     logits and scores within the north-star tolerance 1e-4 on every row for as long as that row's state is still the
     reference's; where the whole run stays identical the decoded x_0 must be exact (svdd_amd/e2e_parity.py; the numbers
     of the last run are kept in profiles/r02_e2e_parity.json).
 (2) Full-size nets: the reference's classes, random-initialised at torch.manual_seed(44) in the order synthetic.build
     uses, gave tests/golden/g12_fullsize_probe.npz; the HIP kernels (exact fp32 and the x3 split modes) must reproduce
     its logits / log-probabilities / value scores within 1e-4 on the GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4            # BASELINE.json north_star: "reward/soft-value tensors within 1e-4 fp32"


@pytest.mark.parametrize("fixture", ["g6_traj_mc_c1.npz", "g6_traj_mc_s16.npz"])
@pytest.mark.parametrize("fuse,batching", [(True, "batched"), (False, "batched"), (False, "reference")])
def test_real_tiny_nets_follow_the_reference_trajectory(golden, fixture, fuse, batching):
    from svdd_amd import e2e_parity
    rep = e2e_parity.compare_with_reference_run(golden(fixture), golden("nets_tiny.npz"), DEV, fuse, batching)
    assert rep["steps_compared"] >= 1
    assert rep["max_abs_logit_err_on_undiverged_rows"] <= TOL, rep
    assert rep["max_abs_score_err_on_undiverged_rows"] <= TOL, rep
    if rep["first_divergence_step"] is None:
        assert rep["x0_exact"], rep
    else:
        # a divergence must be a near-tie of the deciding scores in the reference run, never a wrong score
        fd = rep["first_divergence"]
        assert max(abs(v) for v in fd["gpu_minus_ref_scores"]) <= TOL, rep
        assert fd["reference_score_gap_top2"] is None or fd["reference_score_gap_top2"] <= 2 * TOL, rep


@pytest.fixture(scope="module")
def full_nets():
    from svdd_amd import synthetic
    return synthetic.build("dna", DEV)


@pytest.mark.parametrize("precision", ["f32", "f16x3", "bf16x3"])
def test_fullsize_probe_on_the_hip_kernels(golden, full_nets, precision):
    g = golden("g12_fullsize_probe.npz")
    model, emb, head, _ = full_nets
    model.precision = precision
    x = torch.from_numpy(g["x"]).to(DEV)
    try:
        with torch.no_grad():
            logits = model._backbone_logits(x)
            logp = model.forward(x.long(), torch.zeros(4, device=DEV))
            fn = model.value_callable(emb, head)
            assert isinstance(fn, torch.nn.Module)                       # the fused HIP formulation, not the plain modules
            onehot = model.transform_samples(x.long()).float()
            value = fn(onehot).reshape(-1)
    finally:
        model.precision = "f32"
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() <= TOL
    keep = g["logp"] > -1e5                                              # the -1e6 "impossible" entries are exact constants
    assert np.array_equal(logp.cpu().numpy() > -1e5, keep)
    assert np.abs(logp.cpu().numpy()[keep] - g["logp"][keep]).max() <= TOL
    assert np.abs(value.cpu().numpy() - g["value"]).max() <= TOL

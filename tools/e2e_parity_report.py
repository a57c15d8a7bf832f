"""Prints the end-to-end GPU-vs-reference-trajectory report (tests/e2e_parity.py) as JSON for profiles/.
Usage: python tools/e2e_parity_report.py > profiles/rNN_e2e_parity.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svdd_amd import synthetic
from tests import e2e_parity

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
nets = dict(np.load(os.path.join(G, "nets_tiny.npz")))
out = {"tiny_nets_free_running": {}, "fullsize_teacher_forced": {}, "fullsize_free_running": {}}
for name in ("g6_traj_mc_c1.npz", "g6_traj_mc_s16.npz"):
    g = dict(np.load(os.path.join(G, name)))
    out["tiny_nets_free_running"][name] = [e2e_parity.compare_with_reference_run(g, nets, fuse_nets=True, value_batching=vb)
                                           for vb in ("batched", "reference")]
model, emb, head, reward = synthetic.build("dna", "cuda:0")
for name in ("g13_traj_mc_full_c1.npz", "g13_traj_mc_full_m10.npz"):
    g = dict(np.load(os.path.join(G, name)))
    out["fullsize_teacher_forced"][name] = [e2e_parity.teacher_forced_report(g, model, emb, head, p)
                                            for p in ("f32", "f16x3", "bf16x3", "f16", "bf16")]
    out["fullsize_free_running"][name] = [e2e_parity.compare_engine_with_reference_run(g, model, emb, head, True, "batched", p)
                                          for p in ("f32", "f16x3", "bf16x3")]
# g19: the SMC / TDS baseline on the reference's full-size run (teacher-forced errors + the engine's free-running decode)
g = dict(np.load(os.path.join(G, "g19_traj_tds_full.npz")))
out["fullsize_tds"] = {"g19_traj_tds_full.npz": [e2e_parity.tds_reference_run_report(g, model, reward, p) for p in ("f32", "f16x3", "bf16x3")]}
# g18: full-size nets at L = 50 (several sequences per kernel tile): SVDD-MC and SVDD-PM (Tweedie)
rna, emb_r, head_r, reward_r = synthetic.build("rna", "cuda:0")
g = dict(np.load(os.path.join(G, "g18_traj_mc_full_rna.npz")))
out["fullsize_teacher_forced"]["g18_traj_mc_full_rna.npz"] = [e2e_parity.teacher_forced_report(g, rna, emb_r, head_r, p)
                                                              for p in ("f32", "f16x3", "bf16x3", "f16", "bf16")]
out["fullsize_free_running"]["g18_traj_mc_full_rna.npz"] = [
    e2e_parity.compare_engine_with_reference_run(g, rna, emb_r, head_r, True, "batched", p) for p in ("f32", "f16x3", "bf16x3")]
g = dict(np.load(os.path.join(G, "g18_traj_pm_full_rna.npz")))
out["fullsize_pm"] = {"g18_traj_pm_full_rna.npz": [
    {"teacher_forced": e2e_parity.teacher_forced_pm_report(g, rna, reward_r, p), "free_running": e2e_parity.free_running_pm_report(g, rna, reward_r, p)}
    for p in ("f32", "f16x3", "bf16x3")]}
# g21: the reference's runs AT the headline configs (C2: B = 256, L = 200, M = 10, 128 steps; C3: tweedie, B = 256, L = 50, M = 10)
g = dict(np.load(os.path.join(G, "g21_traj_mc_c2.npz")))
B, M, S = int(g["B"]), int(g["M"]), int(g["S"])
out["headline_c2"] = {"fixture": "g21_traj_mc_c2.npz", "runs": [
    {"teacher_forced": e2e_parity.teacher_forced_lean_report(g, model, emb, head, p),
     "free_running": e2e_parity.free_running_lean_report(
         g, model, lambda m: m.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M), p)}
    for p in ("f32", "f16x3", "bf16x3")]}
g = dict(np.load(os.path.join(G, "g21_traj_pm_c3.npz")))
B, M, S = int(g["B"]), int(g["M"]), int(g["S"])
out["headline_c3"] = {"fixture": "g21_traj_pm_c3.npz", "runs": [
    {"teacher_forced": e2e_parity.teacher_forced_lean_pm_report(g, rna, reward_r, p),
     "free_running": e2e_parity.free_running_lean_report(
         g, rna, lambda m: m.controlled_sample_tweedie(reward_r, num_steps=S, eval_sp_size=B, sample_M=M, options="True"), p)}
    for p in ("f32", "f16x3", "bf16x3")]}
# g23: the reference's SMC / TDS run at the configs[4] per-GPU shard size (256 particles, 128 steps)
g = dict(np.load(os.path.join(G, "g23_traj_tds_c5.npz")))
out["tds_c5_shard"] = {"fixture": "g23_traj_tds_c5.npz", "runs": [e2e_parity.teacher_forced_lean_tds_report(g, model, reward, p) for p in ("f32", "f16x3", "bf16x3")]}
# g24 / g25: the un-guided decode at B = 256 and SVDD-MC with M = 20 at the shard batch
g = dict(np.load(os.path.join(G, "g24_decode_sample_c2.npz")))
out["unguided_c2_batch"] = {"fixture": "g24_decode_sample_c2.npz", "runs": [e2e_parity.unguided_lean_report(g, model, p) for p in ("f32", "f16x3", "bf16x3")]}
g = dict(np.load(os.path.join(G, "g25_traj_mc_m20.npz")))
B, M, S = int(g["B"]), int(g["M"]), int(g["S"])
out["mc_m20_shard_batch"] = {"fixture": "g25_traj_mc_m20.npz", "runs": [
    {"teacher_forced": e2e_parity.teacher_forced_lean_report(g, model, emb, head, p),
     "free_running": e2e_parity.free_running_lean_report(
         g, model, lambda m: m.controlled_sample(emb, head, num_steps=S, eval_sp_size=B, sample_M=M), p)}
    for p in ("f32", "f16x3", "bf16x3")]}
print(json.dumps(out, indent=1))

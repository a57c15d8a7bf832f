"""Prints the end-to-end GPU-vs-reference-trajectory report (svdd_amd/e2e_parity.py) as JSON for profiles/.
Usage: python tools/e2e_parity_report.py > profiles/rNN_e2e_parity.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from svdd_amd import e2e_parity

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
nets = dict(np.load(os.path.join(G, "nets_tiny.npz")))
out = {}
for name in ("g6_traj_mc_c1.npz", "g6_traj_mc_s16.npz"):
    g = dict(np.load(os.path.join(G, name)))
    out[name] = [e2e_parity.compare_with_reference_run(g, nets, fuse_nets=f, value_batching=vb)
                 for f, vb in ((True, "batched"), (False, "batched"), (False, "reference"))]
print(json.dumps(out, indent=1))

import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import backbone, config
DEV="cuda:0"
for B, L in ((3, 50), (5, 200)):
    torch.manual_seed(L)
    cnn = backbone.CNNModel((config.dna_config() if L == 200 else config.rna_config()).model, alphabet_size=5).to(DEV).eval()
    with torch.no_grad():
        for nm in cnn.norms:
            nm.weight.uniform_(0.5, 1.5); nm.bias.uniform_(-0.3, 0.3)
    for p in cnn.parameters(): p.requires_grad_(False)
    x = torch.softmax(torch.randn(B, L, 5, device=DEV), dim=-1)
    g = torch.randn(B, L, 5, device=DEV)
    t = torch.linspace(0.0, 1.0, B, device=DEV)
    res = {}
    for key, (hip, fl, dt) in {"torch32": (False, False, torch.float32), "hip": (True, False, torch.float32), "fused": (True, True, torch.float32)}.items():
        cnn.hip_convs, cnn.fused_layers = hip, fl
        xi = x.clone().requires_grad_(True)
        y = cnn.forward2(xi, t); (y * g).sum().backward()
        res[key] = xi.grad.clone()
    c64 = cnn.double(); c64.hip_convs = False
    xi = x.double().clone().requires_grad_(True)
    y = c64.forward2(xi, t.double()); (y * g.double()).sum().backward()
    ref = xi.grad
    for k, v in res.items():
        d = (v.double() - ref).abs()
        print(B, L, k, "max err vs fp64 %.3e" % float(d.max()), "elements > 1e-4:", int((d > 1e-4).sum()), "positions:", sorted(set((d > 1e-4).nonzero()[:, 1].tolist()))[:12])

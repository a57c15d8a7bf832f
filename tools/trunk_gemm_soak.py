"""Race screen for trunk_gemm256_kernel (LDS-DMA stages, counted vmcnt, wave groups half a phase apart): many random GEMM shapes,
each run repeatedly under load and compared BIT FOR BIT with the register-staged 128 x 128 kernel (which has no asynchronous
LDS traffic). A hazard in the DMA / barrier protocol shows as a rare wrong tile that comes and goes with shape and timing.
Usage: python tools/trunk_gemm_soak.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import _lib
from svdd_amd.fused_trunk import GUARD, TAIL, pack_gemm_weight

DEV = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lib = _lib.lib()
g = torch.Generator().manual_seed(1)
bad = total = 0
for rnd in range(rounds):
    M = int(torch.randint(300, 120000, (1,), generator=g))
    N = 128 * int(torch.randint(1, 13, (1,), generator=g))
    Cin = 32 * int(torch.randint(1, 25, (1,), generator=g))
    T = [1, 1, 3, 5][int(torch.randint(0, 4, (1,), generator=g))]
    parts = 2 if rnd % 4 else 1
    a = torch.randn(M, Cin, generator=g)
    w = torch.randn(N, Cin, T, generator=g) * (Cin * T) ** -0.5
    hi = a.to(torch.bfloat16)
    planes = []
    for pl in ((hi, (a - hi.float()).to(torch.bfloat16)) if parts == 2 else (hi,)):
        buf = torch.zeros((GUARD + M + TAIL) * Cin, dtype=torch.bfloat16, device=DEV)
        buf[GUARD * Cin:(GUARD + M) * Cin] = pl.reshape(-1).to(DEV)
        planes.append(buf[GUARD * Cin:])
    wp = pack_gemm_weight(w, parts).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    outs = []
    for ver, reps in ((1, 1), (3, 6)):
        _lib.set_option(4, ver)
        for _ in range(reps):
            out = torch.empty((M, N), device=DEV)
            rc = lib.svdd_trunk_gemm(planes[0].data_ptr(), planes[1].data_ptr() if parts == 2 else None, wp.data_ptr(), bias.data_ptr(),
                                     None, out.data_ptr(), M, N, Cin, T, Cin, N, 1, None, 1, None, None, None, None, 0, 0, None)
            _lib.check(rc, "svdd_trunk_gemm")
            outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        total += 1
        if not torch.equal(o, outs[0]):
            bad += 1
            print(f"MISMATCH M={M} N={N} Cin={Cin} T={T} parts={parts}: {int((o != outs[0]).sum())} elements")
_lib.set_option(4, 2)
print(f"{total} runs of the 256 x 256 kernel over {rounds} random shapes compared bit for bit with the 128 x 128 kernel: {bad} mismatches")

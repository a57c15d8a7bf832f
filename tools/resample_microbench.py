"""K2 (select + index-gather compaction) and K4 (TDS resample) at sizes where launch latency is gone.
Usage (GPU box): python tools/resample_microbench.py  -> one line per case: us per launch, algorithmic GB/s, fraction of 8 TB/s.
Algorithmic bytes: K2 B*(4M + 2L + 4) (scores + winning row in, x_next + idx out); K4 B*(2L + 8 + 8 + 4 + 4)."""
import os
import sys

os.environ.setdefault("SVDD_EXPERIMENTS", "1")   # SVDD_OPT_CAND_ROW_STRIDE is honoured only in an opted-in process

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svdd_amd import ops

DEV = "cuda:0"
PEAK = 8000.0


def timed(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def k2(B, M, L=200, compact=False, ld=0, scale=1e-3):
    """ld: candidate rows at a stride of `ld` bytes (SVDD_OPT_CAND_ROW_STRIDE; the round-4 layout experiment: 256 = whole lines).
    The padded case is checked against the dense one (same scores, same candidates -> same x_next)."""
    from svdd_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(0)
    scores = torch.randn(B, M, device=DEV, generator=g) * scale
    cand = torch.randint(0, 5, (B, M, L), device=DEV, generator=g, dtype=torch.uint8)
    x_next = torch.empty(B, L, dtype=torch.uint8, device=DEV)
    if ld:
        ref, _, _ = ops.select(scores, cand, want_soft=False)
        wide = torch.zeros(B, M, ld, dtype=torch.uint8, device=DEV)
        wide[:, :, :L] = cand
        cand_dense, cand = cand, wide
        _lib.set_option(5, ld)

    if compact:
        live = torch.rand(B * M, device=DEV, generator=g) < 0.77
        slot = torch.where(live, torch.cumsum(live.int(), 0) - 1, -1).int()
        sc = scores.reshape(-1)[live].contiguous()
        parent = torch.randn(B, device=DEV, generator=g) * 1e-3
        sel, ch, idx = torch.empty(B, device=DEV), torch.empty(B, dtype=torch.int32, device=DEV), torch.empty(B, dtype=torch.int32, device=DEV)
        fn = lambda: ops.select_compact(sc, slot, parent, cand, x_next=x_next, sel_score=sel, changed=ch, idx=idx)   # noqa: E731
        nbytes = B * (4 * M + 4 * M + 2 * L + 4 + 12)
    else:
        idx = torch.empty(B, dtype=torch.int32, device=DEV)
        import ctypes

        def fn():      # the C ABI directly: L stays 200 while the rows of `cand` sit at the option's stride
            rc = _lib.lib().svdd_select(scores.data_ptr(), cand.data_ptr(), B, L, M, ops.SELECT_ARGMAX, None, x_next.data_ptr(), None,
                                        idx.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        nbytes = B * (4 * M + 2 * L + 4)
    us = timed(fn)
    if ld:
        torch.cuda.synchronize()
        if not compact:
            assert torch.equal(x_next, ref), "padded-row select differs from the dense one"
        _lib.set_option(5, 0)
    gbs = nbytes / us / 1e3
    print(f"K2 select{'_compact' if compact else ''} B={B} M={M} L={L} ld={ld or L} scores~{scale:g}: {us:9.1f} us  {gbs:8.1f} GB/s  frac {gbs / PEAK:.3f}  ({nbytes / 1e6:.1f} MB)")


def k4(B, L=200):
    g = torch.Generator(device=DEV).manual_seed(0)
    num, den = torch.randn(B, device=DEV, generator=g) * 0.1, torch.randn(B, device=DEV, generator=g) * 0.1
    sample = torch.randint(0, 5, (B, L), device=DEV, generator=g, dtype=torch.uint8)
    u = torch.rand(B, device=DEV, generator=g, dtype=torch.float64)
    us = timed(lambda: ops.tds_resample(num, den, 0.5, sample, u), iters=10, warm=2)
    nbytes = B * (2 * L + 24)
    gbs = nbytes / us / 1e3
    print(f"K4 tds_resample B={B} L={L}: {us:9.1f} us  {gbs:8.1f} GB/s  frac {gbs / PEAK:.4f}  ({nbytes / 1e6:.2f} MB)")


def k2_gather_split(B=1 << 18, M=10, L=200, scale=1e-2):
    """VERDICT r05 #8, one experiment: K2 WITHOUT the row gather (x_next = NULL: idx only), beside the fused kernel and beside the gather as a
    launch of its own (svdd_gather_rows through the flat index b * M + idx[b]) — what a consumer that reads cand[b, idx[b]] itself would
    pay per pass. Bytes: decision B (4M + 4); gather B 2L (+ 4 for the index)."""
    import ctypes
    from svdd_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(0)
    scores = torch.randn(B, M, device=DEV, generator=g) * scale
    cand = torch.randint(0, 5, (B, M, L), device=DEV, generator=g, dtype=torch.uint8)
    x_next, x2 = torch.empty(B, L, dtype=torch.uint8, device=DEV), torch.empty(B, L, dtype=torch.uint8, device=DEV)
    idx = torch.empty(B, dtype=torch.int32, device=DEV)
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)   # noqa: E731

    def fused():
        assert _lib.lib().svdd_select(scores.data_ptr(), cand.data_ptr(), B, L, M, ops.SELECT_ARGMAX, None, x_next.data_ptr(), None, idx.data_ptr(), st()) == 0

    def decide():
        assert _lib.lib().svdd_select(scores.data_ptr(), cand.data_ptr(), B, L, M, ops.SELECT_ARGMAX, None, None, None, idx.data_ptr(), st()) == 0
    fused()
    flat = (torch.arange(B, device=DEV, dtype=torch.int32) * M + idx).contiguous()
    count = torch.tensor([B], dtype=torch.int32, device=DEV)
    gather = lambda: ops.gather_rows(cand.view(B * M, L), flat, count, x2)   # noqa: E731
    gather()
    torch.cuda.synchronize()
    assert torch.equal(x2, x_next)
    for nb in (1, 2, 4, 3, 5):                                   # SVDD_OPT_SELECT_BATCHES: batches of row groups per wave (1 = the default; 2 / 4: the round-6 experiment)
        _lib.set_option(8, nb)
        x_next.zero_()
        us = timed(fused)
        torch.cuda.synchronize()
        assert torch.equal(x_next, x2), f"batches {nb}: other tokens"
        nbytes = B * (4 * M + 2 * L + 4)
        what = {1: "1 batch of 4 row groups (shipped)", 2: "2 batches of 4 groups", 4: "4 batches of 4 groups", 3: "2 batches of 2 groups (same waves)", 5: "4 batches of 1 group (same waves)"}[nb]
        print(f"K2 split B={B} M={M} L={L} scores~{scale:g}  fused, {what:36s} {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s  frac {nbytes / us / 1e3 / PEAK:.3f}")
    _lib.set_option(8, 0)
    for name, fn, nbytes in (("fused select + gather (the shipped K2)", fused, B * (4 * M + 2 * L + 4)),
                             ("decision only (x_next = NULL)", decide, B * (4 * M + 4)),
                             ("gather alone (svdd_gather_rows via idx)", gather, B * (2 * L + 4))):
        us = timed(fn)
        print(f"K2 split B={B} M={M} L={L} scores~{scale:g}  {name:42s} {us:8.1f} us  {nbytes / us / 1e3:8.1f} GB/s  frac {nbytes / us / 1e3 / PEAK:.3f}  ({nbytes / 1e6:.1f} MB)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "split":
        for scale in (1e-7, 1e-2):
            k2_gather_split(scale=scale)
        k2_gather_split(M=20)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ld":        # the padded-row layout experiment: near-tied leg first
        for scale in (1e-7, 1e-2):
            for M in (10, 20):
                for ld in (0, 208, 256):
                    k2(1 << 18, M, ld=ld, scale=scale)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "one":       # one case, for a PMC pass: one <M> <ld>
        k2(1 << 18, int(sys.argv[2]), ld=int(sys.argv[3]), scale=1e-2)
        sys.exit(0)
    for M in (10, 20):
        k2(1 << 18, M)
    k2(1 << 18, 10, compact=True)
    k2(256, 10)
    k2(256, 10, compact=True)
    for B in (256, 2048, 65536):
        k4(B)

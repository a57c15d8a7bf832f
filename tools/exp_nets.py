"""Experiment: per-forward time of the nets in different formulations (GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from svdd_amd import synthetic

dev = "cuda:0"
model, emb, head, _ = synthetic.build("dna", dev)
bb = model.backbone
B, L, M = 256, 200, 10

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

x = torch.randint(0, 5, (B, L), device=dev).to(torch.uint8)
oh = torch.randn(B * M, L, 4, device=dev)
with torch.no_grad():
    print("backbone plain            %.3f ms" % timeit(lambda: bb(x, None, zero_sigma=True)))
    print("value net plain           %.3f ms" % timeit(lambda: head(emb(oh))))
    ct = emb.conv_tower
    xin = oh.transpose(1, 2).contiguous()
    print("  conv tower              %.3f ms" % timeit(lambda: ct(xin)))
    y = ct(xin)
    print("  gru (MIOpen)            %.3f ms" % timeit(lambda: emb.gru_tower.gru(y.transpose(1, 2))))
    g = emb.gru_tower.gru(y.transpose(1, 2))[0]
    print("  ffn                     %.3f ms" % timeit(lambda: emb.gru_tower.ffn(g[:, :, :64] + g[:, :, 64:])))

    # --- channels-last 4D formulation of the backbone
    H = 128
    tb = bb.zero_time_biases(B, dev)
    w_first = bb.linear.weight.unsqueeze(2).contiguous(memory_format=torch.channels_last)
    ws = [c.weight.unsqueeze(2).contiguous(memory_format=torch.channels_last) for c in bb.convs]
    dil = [c.dilation[0] for c in bb.convs]
    wf1 = bb.final_conv[0].weight.unsqueeze(2).contiguous(memory_format=torch.channels_last)
    wf2 = bb.final_conv[2].weight.unsqueeze(2).contiguous(memory_format=torch.channels_last)
    tbl = [t.reshape(B, 1, 1, H)[:1] for t in tb]   # [1,1,1,H] broadcast on the NHWC view
    def bb_cl():
        oh5 = bb._eye[x.long()]                       # [B,L,5]
        f = oh5.view(B, 1, L, 5).permute(0, 3, 1, 2)  # [B,5,1,L] channels_last view
        f = F.relu(F.conv2d(f, w_first, bb.linear.bias, padding=(0, 4)))
        for i in range(20):
            hN = f.permute(0, 2, 3, 1)                # [B,1,L,H] contiguous view
            hN = F.layer_norm(hN + tbl[i], (H,), bb.norms[i].weight, bb.norms[i].bias)
            h = F.relu(F.conv2d(hN.permute(0, 3, 1, 2), ws[i], bb.convs[i].bias, padding=(0, 4 * dil[i]), dilation=(1, dil[i])))
            f = h + f
        f = F.conv2d(F.relu(F.conv2d(f, wf1, bb.final_conv[0].bias)), wf2, bb.final_conv[2].bias)
        return f
    ref = bb(x, None, zero_sigma=True)
    out = bb_cl()
    print("cl output", out.shape, out.stride(), "max diff vs plain", (out.squeeze(2).permute(0, 2, 1) - ref).abs().max().item())
    print("backbone channels-last    %.3f ms" % timeit(bb_cl))

    # --- value conv tower channels-last with folded BN
    blocks = ct.blocks
    wst = blocks[0].conv.weight.unsqueeze(2).contiguous(memory_format=torch.channels_last)
    folded = []
    for blk in blocks[1:]:
        bn = blk.norm.layer
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = (blk.conv.weight * s[:, None, None]).unsqueeze(2).contiguous(memory_format=torch.channels_last)
        b_ = (blk.conv.bias - bn.running_mean) * s + bn.bias
        folded.append((w, b_))
    def ct_cl():
        f = oh.view(B * M, 1, L, 4).permute(0, 3, 1, 2)
        f = F.relu(F.conv2d(f, wst, blocks[0].conv.bias, padding=(0, 7)))
        for w, b_ in folded:
            f = F.relu(F.conv2d(f, w, b_, padding=(0, 2)) + f)
        return f
    o2 = ct_cl()
    print("ct cl max diff", (o2.squeeze(2) - y).abs().max().item(), o2.stride())
    print("  conv tower cl+foldedBN  %.3f ms" % timeit(ct_cl))

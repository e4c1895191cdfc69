"""Dev (GPU): NTM forward / BPTT alone per batch size (one workgroup per sequence): do the workgroups of an XCD, which stream the same
weights at the same time, slow each other down?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntmtrack import tracker
T = 20
S = T * 65
dev = torch.device("cuda:0")
for B in (1, 8, 16, 32, 64, 128, 256):
    trk = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=dev, seed=1)
    g = torch.Generator().manual_seed(0)
    fmap = torch.relu(torch.randn((B * T, 28, 28, 512), generator=g)).to(dev)
    gts0 = torch.rand((B, 64), generator=g).to(dev)
    offs = (torch.rand((B, T, 2), generator=g) - 0.5).to(dev)
    X = trk.serialize(fmap, gts0); st0 = trk.cell.zero_state(B)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    best = [1e9, 1e9]
    for _ in range(3):
        e[0].record()
        logits, _o, new, rec = trk.cell.run_sequence(X, st0, record=True, want_outputs=False)
        e[1].record()
        loss, pred, dlog = tracker.offset_loss(logits, offs, T)
        g0 = trk.cell.backward_sequence(X, st0, rec, dlog)
        e[2].record(); torch.cuda.synchronize()
        best = [min(best[0], e[0].elapsed_time(e[1])), min(best[1], e[1].elapsed_time(e[2]))]
    print("B %3d: forward %.2f us/step, BPTT (+ loss, weight-gradient GEMMs) %.2f us/step" % (B, best[0] * 1e3 / S, best[1] * 1e3 / S), flush=True)
    del trk, fmap, X, rec
    torch.cuda.empty_cache()

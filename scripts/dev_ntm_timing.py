"""Time the NTM sequence forward / BPTT kernels at a benchmark shape (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import tracker
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
trk = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=dev, seed=1)
g = torch.Generator().manual_seed(0)
fmap = torch.relu(torch.randn((B * T, 28, 28, 512), generator=g)).to(dev)
gts0 = torch.rand((B, 64), generator=g).to(dev)
offs = (torch.rand((B, T, 2), generator=g) - 0.5).to(dev)
def ev(): return torch.cuda.Event(enable_timing=True)
for it in range(3):
    e = [ev() for _ in range(5)]
    e[0].record()
    X = trk.serialize(fmap, gts0); st0 = trk.cell.zero_state(B)
    e[1].record()
    logits, _o, new, rec = trk.cell.run_sequence(X, st0, record=True, want_outputs=False)
    e[2].record()
    loss, pred, dlog = tracker.offset_loss(logits, offs, T)
    g0 = trk.cell.backward_sequence(X, st0, rec, dlog); trk.cell.init_state_backward(g0, B)
    e[3].record()
    trk.opt.step()
    e[4].record(); torch.cuda.synchronize()
    S = T * 65
    print("iter %d: serialize %.3f ms | fwd(xproj+seq) %.3f ms (%.2f us/step) | loss+bwd+wgrad %.3f ms (%.2f us/step) | opt %.3f ms | loss %.5f"
          % (it, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[1].elapsed_time(e[2]) * 1e3 / S,
             e[2].elapsed_time(e[3]), e[2].elapsed_time(e[3]) * 1e3 / S, e[3].elapsed_time(e[4]), float(loss.cpu())), flush=True)

"""Dev (GPU): time the NTM forward forms alone at B32 x S1300: the wave-specialised kernel per split variant, and round 2's kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntmtrack import tracker
B, T = 32, 20
dev = torch.device("cuda:0")
trk = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=dev, seed=1)
g = torch.Generator().manual_seed(0)
fmap = torch.relu(torch.randn((B * T, 28, 28, 512), generator=g)).to(dev)
gts0 = torch.rand((B, 64), generator=g).to(dev)
X = trk.serialize(fmap, gts0); st0 = trk.cell.zero_state(B)
S = T * 65
def run(record):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(4):
        e0.record(); out = trk.cell.run_sequence(X, st0, record=record, want_outputs=False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts[1:]), out[0]
ref = None
variants = [("res", None)] + [("ws", str(i)) for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10)]
for form, sp in variants:
    os.environ["NTK_NTM_FWD_FORM"] = "res" if form == "res" else "ws"
    if sp is not None: os.environ["NTK_NTM_WS_SPLIT"] = sp
    for record in (False, True):
        ms, logits = run(record)
        if ref is None: ref = logits.clone()
        err = float((logits - ref).abs().max())
        print("%s split %s record=%d: %.3f ms  %.2f us/step  (max |logit - res| %.2e)" % (form, sp, record, ms, ms * 1e3 / S, err), flush=True)

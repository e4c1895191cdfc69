"""Dev: can the deep-controller cell's step-wise sequence (StackedNTMCell.run_sequence: a Python loop of ~25 launches per step) be
captured into one HIP graph and replayed?  Eager against replay: same bits, and the time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntmtrack.ntm import StackedNTMCell
dev = torch.device("cuda:0")
B, S, D, L = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 65, 516, 2
cell = StackedNTMCell(2, mem_size=128, mem_dim=20, controller_hidden_size=200, controller_num_layers=L, write_head_size=1,
                      read_head_size=4, input_dim=D, device=dev, seed=1)
g = torch.Generator().manual_seed(0)
X = torch.randn((B, S, D), generator=g).to(dev)
st0 = cell.zero_state(B)
def eager():
    return cell.run_sequence(X, {k: v.clone() for k, v in st0.items()}, record=False)
for _ in range(2):
    lg, out, st, _ = eager()
torch.cuda.synchronize()
t0 = time.time(); lg, out, st, _ = eager(); torch.cuda.synchronize(); te = time.time() - t0
print("eager: %.1f ms for %d steps = %.1f us per step" % (te * 1e3, S, te * 1e6 / S), flush=True)
# capture
Xs = X.clone()
sts = {k: v.clone() for k, v in st0.items()}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        cell.run_sequence(Xs, {k: v.clone() for k, v in sts.items()}, record=False)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    lg2, out2, st2, _ = cell.run_sequence(Xs, {k: v.clone() for k, v in sts.items()}, record=False)
torch.cuda.synchronize()
gr.replay(); torch.cuda.synchronize()
t0 = time.time(); gr.replay(); torch.cuda.synchronize(); tg = time.time() - t0
print("graph replay: %.1f ms = %.1f us per step; same bits: logits %s, final M %s" % (tg * 1e3, tg * 1e6 / S, torch.equal(lg, lg2), torch.equal(st["M"], st2["M"])), flush=True)

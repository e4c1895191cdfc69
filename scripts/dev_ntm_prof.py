"""Dev: per-phase cycle shares of the NTM sequence forward kernel (diagnostic library: make -C ntm-tracker_amd/csrc prof;
run with NTK_LIB_PATH=ntm-tracker_amd/libntmtrack_hip_prof.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import tracker, _lib
B, T = 32, 20
dev = torch.device("cuda:0")
trk = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=dev, seed=1)
g = torch.Generator().manual_seed(0)
fmap = torch.relu(torch.randn((B * T, 28, 28, 512), generator=g)).to(dev)
gts0 = torch.rand((B, 64), generator=g).to(dev)
record = len(sys.argv) > 1 and sys.argv[1] == "rec"
X = trk.serialize(fmap, gts0); st0 = trk.cell.zero_state(B)
for _ in range(2):
    trk.cell.run_sequence(X, st0, record=record, want_outputs=False)
torch.cuda.synchronize()
fn = _lib.lib().ntk_ntm_fwd_prof
fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 16)()
assert fn(buf) == 0
names = ["P1 gate partials (Wr stream)", "P2 LSTM || column norms", "P3 unpack partials (Wa stream)", "P4 activations", "P5-P7 addressing (wave per head)",
         "P8a read partials", "P8b memory update + reads"]
S = T * 65
tot = float(sum(buf[:7]))
print("ntm fwd B=%d S=%d record=%s: %.0f cycles/step (workgroup 0, stamped build)" % (B, S, record, tot / S))
for i, nm in enumerate(names):
    print("  %-36s %8.0f cyc/step  %5.1f %%" % (nm, buf[i] / S, 100.0 * buf[i] / tot))

# ---- BPTT kernel: shares between consecutive workgroup barriers of a step
_loss_offs = (torch.rand((B, T, 2), generator=g) - 0.5).to(dev)
logits, _o, new, rec = trk.cell.run_sequence(X, st0, record=True, want_outputs=False)
loss, pred, dlog = tracker.offset_loss(logits, _loss_offs, T)
for _ in range(2):
    trk.cell.backward_sequence(X, st0, rec, dlog)
torch.cuda.synchronize()
fnb = _lib.lib().ntk_ntm_bwd_prof
fnb.restype = ctypes.c_int
bb = (ctypes.c_ulonglong * 16)()
assert fnb(bb) == 0
nb = ["X1 memory-shaped elementwise, column norms", "X2 d(w_t) per head, R1 sums", "R2 sharpen bwd", "R3 shift + gate bwd", "R4 content softmax bwd",
      "B7 dMhat (to its first barrier)", "B7/B8 second part", "B8 third part", "B9 dh = dU . Wa^T (Wa^T stream)", "B10 LSTM cell bwd",
      "B11 d[read;h] = dgates . Wr^T (Wr^T stream)", "B11 reduce + carry", "(B7a) per-head scalar controls", "(B7b) dMhat loop"]
totb = float(sum(bb[:14]))
print("ntm bwd: %.0f cycles/step (workgroup 0, stamped build)" % (totb / S))
for i, nm in enumerate(nb):
    print("  %-48s %8.0f cyc/step  %5.1f %%" % (nm, bb[i] / S, 100.0 * bb[i] / totb))

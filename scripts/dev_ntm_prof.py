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

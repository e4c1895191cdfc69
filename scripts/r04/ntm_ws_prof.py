"""Dev (GPU): where a step of the wave-specialised NTM forward kernel goes (diagnostic library: make -C ntm-tracker_amd/csrc prof;
run with NTK_LIB_PATH=ntm-tracker_amd/libntmtrack_hip_prof.so).  Compute wave 0 and stream wave 0 of workgroup 0: cycles of
work before each of the step's seven barriers and cycles waiting at it."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntmtrack import tracker, _lib
B, T = 32, 20
dev = torch.device("cuda:0")
trk = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=dev, seed=1)
g = torch.Generator().manual_seed(0)
fmap = torch.relu(torch.randn((B * T, 28, 28, 512), generator=g)).to(dev)
gts0 = torch.rand((B, 64), generator=g).to(dev)
X = trk.serialize(fmap, gts0); st0 = trk.cell.zero_state(B)
S = T * 65
fn = _lib.lib().ntk_ntm_ws_prof
fn.restype = ctypes.c_int
names = ["P1 read rows (resident)", "P2 LSTM || norms", "P3 unpack (Wa stream)", "P4 activations", "P5-7 addressing", "P8a read", "P8b write"]
for sp in (sys.argv[1:] or ["0"]):
    os.environ["NTK_NTM_WS_SPLIT"] = sp
    for _ in range(2):
        trk.cell.run_sequence(X, st0, record=True, want_outputs=False)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    assert fn(buf) == 0
    for role, nm in ((0, "compute wave 0"), (1, "stream wave 0")):
        v = [buf[role * 16 + i] / S for i in range(14)]
        print("split %s, %s: %.0f cycles/step" % (sp, nm, sum(v)))
        for i in range(7):
            print("   before B%d (%-24s) work %7.0f  wait %7.0f" % (i + 1, names[i], v[2 * i], v[2 * i + 1]))

import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ntmtrack import tracker, vgg
dev = torch.device("cuda")
B, T = 32, 20
g = torch.Generator().manual_seed(42)
ws = {}
for name, cin, cout, _ in vgg.VGG_LAYERS:
    ws[name] = ((torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5).numpy(), np.zeros(cout, np.float32))
trk = tracker.NTMOffsetTracker(B, T, vgg_weights=ws, device=dev, seed=1)
frames = (torch.rand((B * T, 224, 224, 3), generator=g) * 255 - 117).to(dev)
gts0 = torch.rand((B, 64), generator=g).to(dev)
offs = (torch.rand((B, T, 2), generator=g) - 0.5).to(dev)
L = ctypes.CDLL(os.path.join(ROOT, "ntm-tracker_amd", "libntmtrack_hip.so"))
buf = (ctypes.c_longlong * 40)()
def read():
    torch.cuda.synchronize(); L.ntk_ntm_debug_read(buf); return np.array(list(buf), dtype=np.float64)
fmap = trk.features(frames)
for _ in range(2):
    trk.loss_and_grads(fmap, gts0, offs)
alone = read()
trk.submit_features(frames)
for i in range(3):
    trk.train_on_submitted(gts0, offs)
    trk.submit_features(frames)
trk.train_on_submitted(gts0, offs)
trk.submit_features(frames)
trk.join()
cont = read()
S = T * 65
names = ["X1 mem elementwise", "X2 d(w_t)+R1", "R2 sharpen", "R3 shift+gate", "R4 softmax", "B7a slot reductions", "B7b", "B7c keys+dM",
         "B9 dU.WaT", "B10 LSTM", "B11 dgates.WrT", "final reduce+commit"]
print("phase                      alone us/step   contended   ratio")
for i in range(12):
    print("%-26s %8.2f %12.2f   %.2f" % (names[i], alone[i] / S / 100, cont[i] / S / 100, cont[i] / max(alone[i], 1)))
print("%-26s %8.2f %12.2f   %.2f" % ("sum", alone.sum() / S / 100, cont.sum() / S / 100, cont.sum() / alone.sum()))

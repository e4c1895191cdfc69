"""Dev: per-section cycles of the eight-wave F(4x4,3x3) kernel's K loop (diagnostic library: make -C ntm-tracker_amd/csrc prof;
run with NTK_LIB_PATH=ntm-tracker_amd/libntmtrack_hip_prof.so).  One workgroup, all eight waves."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import vgg, _lib
dev = torch.device("cuda")
F, H, cin, cout = [int(v) for v in (sys.argv[1:5] + ["640", "56", "256", "256"][len(sys.argv) - 1:])][:4]
x = torch.randn((F, H, H, cin), device=dev)
w = torch.randn((3, 3, cin, cout), device=dev) * 0.02
b = torch.zeros(cout, device=dev)
up = vgg.pack_weights_wino43(w)
out = torch.empty((F, H, H, cout), device=dev)
for _ in range(2):
    vgg.conv3x3_relu_wino43(x, up, b, cin, cout, out=out)
torch.cuda.synchronize()
fn = _lib.lib().ntk_vgg_wino43d_prof
fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 96)()
assert fn(buf) == 0
n8 = cin // 8
for wv in range(8):
    v = buf[12 * wv:12 * wv + 12]
    if wv < 4:
        names = ["-> barrier arrive", "barrier wait", "previous step's group 1 (8 / 16 MFMAs), stage request, transform", "group 0 (12 MFMAs), stage store, operand requests"]
        idx = [0, 1, 2, 8]
        role = "T " + ("rows 1,2", "rows 3,4", "row 0", "row 5")[wv]
    else:
        names = ["-> barrier arrive", "barrier wait", "48 MFMAs issued (A, B operand waits)"]
        idx = [0, 1, 2]
        role = "S"
    tot = float(sum(v[i] for i in idx))
    print("wave %d (%s): %.0f cycles per K step (x %d); prologue %.0f, loop exit %.0f, epilogue %.0f" % (wv, role, tot / n8, n8, v[6], v[4], v[5]))
    for nm, i in zip(names, idx):
        print("   %-74s %7.0f  %5.1f %%" % (nm, v[i] / n8, 100.0 * v[i] / tot))

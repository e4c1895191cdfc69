"""Dev: per-section cycle shares of the F(4x4,3x3) K loop (diagnostic library: make -C ntm-tracker_amd/csrc prof; run with
NTK_LIB_PATH=ntm-tracker_amd/libntmtrack_hip_prof.so).  One workgroup, all four waves (= transform tasks)."""
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
fn = _lib.lib().ntk_vgg_wino43_prof
fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 96)()
assert fn(buf) == 0
names = {15: "loop tail -> barrier arrive", 0: "barrier wait", 1: "issue A/window reads, stage+U loads (x3)"}
for g in range(3):
    names[2 + 4 * g] = "MFMA g%d first half (incl. operand waits)" % g
    names[3 + 4 * g] = "gap g%d a: rows(2g), reads(2g+1)" % g
    names[4 + 4 * g] = "MFMA g%d second half" % g
    names[5 + 4 * g] = "gap g%d b: A reads, stage store, rows(2g+1), reads/finish" % g
n8 = cin // 8
for wv, role in enumerate(("rows 1,2", "rows 3,4", "row 0", "row 5")):
    v = buf[24 * wv:24 * wv + 24]
    tot = float(sum(v[:16]))
    print("wave %d (%s): %.0f cycles per K step (x %d K steps); whole workgroup %.0f cycles" % (wv, role, tot / n8, n8, float(sum(v))))
    for i in [15, 0, 1] + list(range(2, 14)):
        print("   %-62s %7.0f  %5.1f %%" % (names[i], v[i] / n8, 100.0 * v[i] / tot))
    for i, nm in ((16, "setup (slot table)"), (17, "prologue: two patches staged, barrier"), (18, "prologue: transform of K step 0, U request"),
                  (19, "epilogue: barrier before Z (x2)"), (20, "epilogue: accumulators -> LDS (x2)"), (21, "epilogue: barrier (x2)"),
                  (22, "epilogue: A^T M A, bias, ReLU, stores (x2)")):
        print("   %-62s %7.0f cycles" % (nm, v[i]))

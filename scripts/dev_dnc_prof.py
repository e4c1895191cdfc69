"""Dev: per-phase cycle shares of the DNC cluster forward kernel (diagnostic library: make -C ntm-tracker_amd/csrc prof;
run with NTK_LIB_PATH=ntm-tracker_amd/libntmtrack_hip_prof.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import dnc as G, _lib
dev = torch.device("cuda:0")
N, W, B, T, k = [int(v) for v in (sys.argv[1:6] + ["256", "64", "32", "20", "8"][len(sys.argv) - 1:])][:5]
record = len(sys.argv) > 6 and sys.argv[6] == "rec"
S = T * 65
x = (torch.randn((S, B, 514), generator=torch.Generator().manual_seed(0)) * 0.5).to(dev)
core = G.DNC({"memory_size": N, "word_size": W, "num_reads": 4, "num_writes": 1}, {"hidden_size": 200}, 2, 20.0, input_dim=514, device=dev, seed=1)
core.cluster_k = k
for _ in range(2):
    core.run_sequence(x, record=record)
torch.cuda.synchronize(); core.check_cluster()
L = _lib.lib()
fn = L.ntk_dnc_cluster_prof
fn.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 16)()
assert fn(buf) == 0
names = ["P1 gates+LSTM", "P2 ifc partial", "E0 publish", "E0 wait", "E0 consume+act", "recs/readmode/usage", "P4 write sims", "softmax+P5 alloc",
         "P6 M update+read sims", "P7a link update", "P7b MFMA reads", "fwd reduce+softmax+E1 publish", "E1 wait", "P8 consume+rw", "reads+output", "loop top"]
order = [15, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14]
tot = float(sum(buf))
print("cluster fwd N=%d W=%d B=%d S=%d k=%d record=%s: %.0f cycles/step (workgroup 0, stamped build)" % (N, W, B, S, core.last_cluster_k, record, tot / S))
for i in order:
    print("  %-32s %8.0f cyc/step  %5.1f %%" % (names[i], buf[i] / S, 100.0 * buf[i] / tot))

# ---- backward
if record:
    dout = torch.randn((B, S, 2), device=dev)
    for _ in range(2):
        core.run_sequence(x, record=True)
        core.backward_sequence(core.last_X, dout)
    torch.cuda.synchronize(); core.check_cluster()
    fnb = L.ntk_dnc_cluster_bwd_prof
    fnb.restype = ctypes.c_int
    bb = (ctypes.c_ulonglong * 20)()
    assert fnb(bb) == 0
    nb = ["top (prev consume tail)", "load records+B1", "S1 Wy^T dy, norms, rank", "rank sum + B2", "B3 head reductions", "B4 dM pass + col fold",
          "reload M + B5ab link elementwise", "B5c MFMA + publish", "X0 wait", "X0 consume", "B6 precedence", "B7 write bwd pass", "B8/B9 alloc bwd",
          "B10b/B11/colsums/dxi", "B14/B15 dh + LSTM bwd", "B16 dz partial + publish", "X1 wait", "X1 consume"]
    totb = float(sum(bb))
    print("cluster bwd: %.0f cycles/step (workgroup 0, stamped build; the stamps still cost ~30 spilled registers in this 246-VGPR kernel:"
          " the product build is ~10 %% faster and the sections after a record prefetch read too high)" % (totb / S))
    for i, nm in enumerate(nb):
        print("  %-36s %8.0f cyc/step  %5.1f %%" % (nm, bb[i] / S, 100.0 * bb[i] / totb))

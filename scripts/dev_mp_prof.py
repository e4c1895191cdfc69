"""Dev: per-phase cycle shares of the memory-partitioned DNC cluster kernels (diagnostic library: make -C ntm-tracker_amd/csrc prof;
run with NTK_LIB_PATH=ntm-tracker_amd/libntmtrack_hip_prof.so).  s_memtime ticks at the shader clock here (~2.4 GHz): the us figures assume 2400 ticks per us; read SHARES."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import dnc as G, _lib
dev = torch.device("cuda:0")
N, W, B, S, k = [int(v) for v in (sys.argv[1:6] + ["512", "128", "64", "300", "4"][len(sys.argv) - 1:])][:5]
x = (torch.randn((S, B, 514), generator=torch.Generator().manual_seed(0)) * 0.5).to(dev)
core = G.DNC({"memory_size": N, "word_size": W, "num_reads": 4, "num_writes": 1}, {"hidden_size": 200}, 2, 20.0, input_dim=514, device=dev, seed=1)
core.cluster_k, core.cluster_form = k, "mp"
L = _lib.lib()
FN = ["loop top", "P1 gates+LSTM", "P2 ifc partial", "publish 0 (+ deferred output)", "wait 0", "consume 0 + records", "key norms + P3 usage",
      "P4 write scores + P5a rank partial", "publish A + wait A", "P5b-d allocation, ww", "P6 M write + read scores", "P7 link rows",
      "P7 column reduction", "publish B + wait B", "P8 read weights", "reads partial + publish C", "wait C"]
BN = ["loop top", "records->LDS, B1, norms, rank partial", "B2", "publish 1 (+ gate record requests)", "wait 1", "consume 1", "B3", "B4 + col sums",
      "B7", "B5 link rows", "B5 column reductions", "-", "publish 2 (+ M_{t-1} requests) + wait 2", "consume 2", "B6 B8 B9", "B10b B11",
      "publish 3 + free gates + sync", "B14 part 1 + wait 3 + consume + dxi", "B14 part 2 + B15", "B16 + publish 4", "wait 4 + consume", "B10b rows", "B10b fold", "B10b park (barrier)"]
for record in (False, True):
    for _ in range(2):
        core.run_sequence(x, record=record)
    torch.cuda.synchronize(); core.check_cluster()
    fn = L.ntk_dnc_mp_fwd_prof; fn.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 24)()
    assert fn(buf) == 0
    tot = float(sum(buf))
    print("mp fwd N=%d W=%d B=%d S=%d k=%d record=%s: %.2f us/step (workgroup 0, stamped build)" % (N, W, B, S, core.last_cluster_k, record, tot / S / 2400.0))
    for i, nm in enumerate(FN):
        print("  %-40s %7.2f us  %5.1f %%" % (nm, buf[i] / S / 2400.0, 100.0 * buf[i] / tot))
dout = torch.randn((B, S, 2), device=dev)
for _ in range(2):
    core.run_sequence(x, record=True)
    core.backward_sequence(core.last_X, dout)
torch.cuda.synchronize(); core.check_cluster()
fnb = L.ntk_dnc_mp_bwd_prof; fnb.restype = ctypes.c_int
bb = (ctypes.c_ulonglong * 32)()
assert fnb(bb) == 0
totb = float(sum(bb))
print("mp bwd: %.2f us/step (workgroup 0, stamped build)" % (totb / S / 2400.0))
for i, nm in enumerate(BN):
    print("  %-40s %7.2f us  %5.1f %%" % (nm, bb[i] / S / 2400.0, 100.0 * bb[i] / totb))

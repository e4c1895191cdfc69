"""Dev: DNC sequence kernels, forward (inference / recording) and BPTT, one-workgroup vs cluster form."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import dnc as G
dev = torch.device("cuda:0")
N, W, B, T = [int(v) for v in (sys.argv[1:5] + ["256", "64", "32", "20"][len(sys.argv) - 1:])][:4]
ks = [int(v) for v in sys.argv[5:]] or [0, 2, 4, 8]
S = T * 65
g = torch.Generator().manual_seed(0)
x = (torch.randn((S, B, 514), generator=g) * 0.5).to(dev)
for k in ks:
    core = G.DNC({"memory_size": N, "word_size": W, "num_reads": 4, "num_writes": 1}, {"hidden_size": 200}, 2, 20.0, input_dim=514,
                 device=dev, seed=1)
    core.cluster_k = k
    for record in (False, True):
        try:
            core.run_sequence(x, record=record); torch.cuda.synchronize()
        except Exception as e:
            print("k=%d record=%s: %s" % (k, record, e)); continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(3):
            e0.record(); core.run_sequence(x, record=record); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        core.check_cluster()
        ms = sorted(ts)[1]
        print("N=%d W=%d B=%d S=%d k=%d (used %d) record=%s: %.2f ms = %.2f us/step" % (N, W, B, S, k, core.last_cluster_k, record, ms, ms * 1e3 / S), flush=True)
        if record and hasattr(core, "backward_sequence"):
            dout = torch.randn((B, S, 2), device=dev)
            X = core.last_X
            try:
                core.backward_sequence(X, dout); torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    core.run_sequence(x, record=True)
                    e0.record(); core.backward_sequence(X, dout); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
                core.check_cluster()
                ms = sorted(ts)[1]
                print("   BPTT (incl. weight-gradient GEMMs): %.2f ms = %.2f us/step" % (ms, ms * 1e3 / S), flush=True)
            except Exception as e:
                print("   BPTT: %s" % e)

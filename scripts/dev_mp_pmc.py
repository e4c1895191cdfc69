"""Dev: one inference forward, one recorded forward and one BPTT of the DNC core at BASELINE configs[4]'s shape (512 x 128, B 64) on the
memory-partitioned cluster kernels, for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) and --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import dnc as G
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N, W, B, K = [int(v) for v in (sys.argv[2:6] + ["512", "128", "64", "4"][len(sys.argv[2:6]):])]
x = (torch.randn((S, B, 514), generator=torch.Generator().manual_seed(0)) * 0.5).to(dev)
core = G.DNC({"memory_size": N, "word_size": W, "num_reads": 4, "num_writes": 1}, {"hidden_size": 200}, 2, 20.0, input_dim=514, device=dev, seed=1)
core.cluster_form, core.cluster_k = "mp", K
dout = torch.randn((B, S, 2), device=dev)
for _ in range(2):
    core.run_sequence(x, record=False)
    core.run_sequence(x, record=True)
    core.backward_sequence(core.last_X, dout)
torch.cuda.synchronize(); core.check_cluster()
assert core.last_cluster_form == "mp" and core.last_cluster_bwd_form == "mp"
print("done: S=%d B=%d k=%d" % (S, B, core.last_cluster_k))

"""Dev: one recorded forward + one BPTT of the DNC core at config 3 (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import dnc as G
dev = torch.device("cuda:0")
N, W, B, T = 256, 64, 32, 20
S = T * 65
x = (torch.randn((S, B, 514), generator=torch.Generator().manual_seed(0)) * 0.5).to(dev)
core = G.DNC({"memory_size": N, "word_size": W, "num_reads": 4, "num_writes": 1}, {"hidden_size": 200}, 2, 20.0, input_dim=514, device=dev, seed=1)
dout = torch.randn((B, S, 2), device=dev)
for _ in range(2):
    core.run_sequence(x, record=True)
    core.backward_sequence(core.last_X, dout)
    core.run_sequence(x, record=False)
torch.cuda.synchronize(); core.check_cluster()
print("done")

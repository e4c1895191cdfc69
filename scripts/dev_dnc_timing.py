"""Time the DNC forward sequence kernel at benchmark shapes (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack.dnc import DNC
dev = torch.device("cuda:0")
for (N, W, B, S) in ((256, 64, 32, 1300), (512, 128, 64, 650)):
    core = DNC({"memory_size": N, "word_size": W, "num_reads": 4, "num_writes": 1}, {"hidden_size": 200}, 2, 20, input_dim=514, device=dev)
    x = torch.relu(torch.randn((S, B, 514), generator=torch.Generator().manual_seed(0))).to(dev)
    for it in range(2):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); out, st = core.run_sequence(x); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        byt = (2 * N * W + 2 * N * N + 2 * 5 * N + 2 * N + 2 * N) * 4      # SURVEY 8(d) algorithmic bytes / sequence-step
        print("DNC N=%d W=%d B=%d S=%d: %.2f ms  %.2f us/step  algorithmic state traffic %.1f GB/s" % (N, W, B, S, ms, ms * 1e3 / S, byt * B * S / ms / 1e6), flush=True)

"""Dev / profiling target: N passes of the default fp32 VGG trunk (conv1_1 row kernel + nine Winograd layers) over 640
synthetic frames.  Used under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for the HBM-side traffic of one trunk pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ntmtrack import vgg

F = int(sys.argv[1]) if len(sys.argv) > 1 else 640
algo = sys.argv[2] if len(sys.argv) > 2 else "split3"        # split3 | winograd | winograd2 | direct | bf16 (config 5's bf16 MFMA trunk)
dev = torch.device("cuda")
g = torch.Generator().manual_seed(42)
ws = {}
for name, cin, cout, _ in vgg.VGG_LAYERS:
    ws[name] = (torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5, torch.zeros(cout))
net = vgg.VGG16Conv43(ws, device=dev, dtype="bf16") if algo == "bf16" else vgg.VGG16Conv43(ws, device=dev, algo=algo)
frames = (torch.rand((F, 224, 224, 3), generator=g) * 255 - 117.0).to(dev)
out = torch.empty((F, 28, 28, 512), device=dev)
for _ in range(3):
    net(frames, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    net(frames, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("trunk %s F=%d: %.3f ms/pass, %.1f TFLOP/s (algorithmic)" % (algo, F, ms, vgg.conv_flops_per_frame() * F / ms / 1e9))

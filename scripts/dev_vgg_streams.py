"""Trunk time when the batch is split over several HIP streams (tails of one stream's layer overlap the other's)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ntmtrack import vgg
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ws = {n: ((rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32), np.zeros(co, np.float32)) for n, ci, co, _ in vgg.VGG_LAYERS}
net = vgg.VGG16Conv43(ws, device=dev)
F = 640
frames = (torch.rand((F, 224, 224, 3)) * 255 - 117).to(dev)
out = torch.empty((F, 28, 28, 512), device=dev)
for ns in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    bounds = [F * i // ns for i in range(ns + 1)]
    def run():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                net.forward_chunk(frames[bounds[i]:bounds[i + 1]], out=out[bounds[i]:bounds[i + 1]])
        for s in streams: cur.wait_stream(s)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = min(ts)
    print("streams=%d: %.2f ms  %.1f TF  (%.1f%% of 157.3)" % (ns, ms, vgg.conv_flops_per_frame() * F / ms / 1e9, vgg.conv_flops_per_frame() * F / ms / 1e9 / 1.573), flush=True)

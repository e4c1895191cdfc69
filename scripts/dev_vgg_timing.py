"""Per-layer timing of the VGG conv stack (dev tool; run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ntmtrack import vgg

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
H = W = 224
x = (torch.rand((F, H, W, 3), generator=g) * 255 - 117).to(dev)
tot_t = 0.0; tot_f = 0.0
for name, cin, cout, pool in vgg.VGG_LAYERS:
    w = (torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev)
    wp = vgg.pack_weights(w)
    for _ in range(2):
        y = vgg.conv3x3_relu(x, wp, b, cin, cout, fuse_pool=pool)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        y = vgg.conv3x3_relu(x, wp, b, cin, cout, fuse_pool=pool)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * F * H * W * 9 * cin * cout
    print("%-8s F=%d %dx%d %4d->%4d pool=%d  %8.3f ms  %7.2f TFLOP/s" % (name, F, H, W, cin, cout, pool, ms, fl / ms / 1e9), flush=True)
    tot_t += ms; tot_f += fl
    x = y
    if pool: H //= 2; W //= 2
print("TOTAL %.3f ms for %d frames -> %.1f frames/s, %.2f TFLOP/s (%.1f%% of 157.3)" % (tot_t, F, F / tot_t * 1e3, tot_f / tot_t / 1e9, tot_f / tot_t / 1e9 / 157.3 * 100))

"""Per-layer timing of the bf16 VGG trunk (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import vgg
F = int(sys.argv[1]) if len(sys.argv) > 1 else 640
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
H = W = 224
tot_t = 0.0; tot_f = 0.0
x = None
for name, cin, cout, pool in vgg.VGG_LAYERS:
    w = (torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev)
    last = name == "conv4_3"
    if name == "conv1_1":
        xin = (torch.rand((F, H, W, 3), generator=g) * 255 - 117).to(dev)
        wp = vgg.pack_weights(w)
        run = lambda: (lambda y: (_ := vgg._lib.check(vgg._lib.lib().ntk_vgg_conv3x3_relu_f32_to_bf16(vgg._lib.ptr(xin), vgg._lib.ptr(wp), vgg._lib.ptr(b), vgg._lib.ptr(y), F, H, W, 3, 64, vgg._lib.stream()), "c11"), y)[1])(torch.empty((F, H, W, 64), device=dev, dtype=torch.bfloat16))
    else:
        wp = vgg.pack_weights_bf16(w); xi = x
        run = lambda: vgg.conv3x3_relu_bf16(xi, wp, b, cin, cout, fuse_pool=pool, out_f32=last)
    for _ in range(2): y = run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): y = run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * F * H * W * 9 * cin * cout
    print("%-8s F=%d %dx%d %4d->%4d pool=%d  %8.3f ms  %8.2f TFLOP/s" % (name, F, H, W, cin, cout, pool, ms, fl / ms / 1e9), flush=True)
    tot_t += ms; tot_f += fl; x = y
    if pool: H //= 2; W //= 2
print("TOTAL %.3f ms for %d frames -> %.1f frames/s, %.2f TFLOP/s" % (tot_t, F, F / tot_t * 1e3, tot_f / tot_t / 1e9))

"""Dev: the eight-wave F(4x4,3x3) kernel against the four-wave one: bitwise equality on test shapes, then
per-layer timing of both at the trunk's shapes.  usage: python scripts/dev_wino43d.py [frames]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntmtrack import vgg

dev = torch.device("cuda")
rng = np.random.default_rng(0)


def run(variant, *args, **kw):
    return vgg.conv3x3_relu_wino43(*args, waves=8 if variant else 4, **kw)


bad = 0
for (F, H, W, cin, cout, pool) in [(1, 16, 32, 32, 64, False), (2, 16, 32, 32, 64, True), (1, 16, 16, 32, 128, False), (3, 32, 16, 64, 64, True),
                                   (2, 8, 8, 32, 64, False), (5, 8, 24, 64, 128, True), (3, 4, 4, 32, 64, False), (7, 28, 28, 32, 64, True),
                                   (2, 12, 20, 32, 512, False), (1, 112, 112, 64, 128, False), (3, 56, 56, 128, 256, True), (2, 28, 28, 256, 512, False),
                                   (3, 224, 224, 64, 64, True)]:
    x = torch.from_numpy(np.maximum(rng.standard_normal((F, H, W, cin)), 0).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1).to(dev)
    u = vgg.pack_weights_wino43(w)
    r0 = run(0, x, u, b, cin, cout, fuse_pool=pool)
    r1 = run(1, x, u, b, cin, cout, fuse_pool=pool)
    torch.cuda.synchronize()
    same = bool(torch.equal(r0, r1))
    print("F%d %dx%d %d->%d pool=%d  bitwise %s  max diff %.2e" % (F, H, W, cin, cout, pool, same, float((r0 - r1).abs().max())), flush=True)
    bad += not same
print("BAD" if bad else "OK", flush=True)
if bad:
    sys.exit(1)

F = int(sys.argv[1]) if len(sys.argv) > 1 else 640
H = 224
tot = [0.0, 0.0]
for name, cin, cout, pool in vgg.VGG_LAYERS:
    if cin != 3:
        x = torch.randn((F, H, H, cin), device=dev)
        w = torch.randn((3, 3, cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
        b = torch.zeros(cout, device=dev)
        u = vgg.pack_weights_wino43(w)
        oh = H // 2 if pool else H
        out = torch.empty((F, oh, oh, cout), device=dev)
        res = []
        for v in (0, 1):
            run(v, x, u, b, cin, cout, fuse_pool=pool, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(v, x, u, b, cin, cout, fuse_pool=pool, out=out)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 3)
            tot[v] += res[-1]
        print("%-8s H%3d %3d->%3d  four waves %7.3f ms | eight waves %7.3f ms  x%.3f" % (name, H, cin, cout, res[0], res[1], res[0] / res[1]), flush=True)
        del x, out
    if pool:
        H //= 2
print("sum (9 layers) four waves %.2f ms, eight waves %.2f ms" % tuple(tot))

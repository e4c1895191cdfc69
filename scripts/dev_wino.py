"""Dev: Winograd conv kernel -- correctness vs the direct MFMA kernel and the oracle, then per-layer timing."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from ntmtrack import vgg
from oracle import ntm_oracle as O

import os
from ntmtrack import _lib
dev = torch.device("cuda")
rng = np.random.default_rng(0)



def check(F, H, W, cin, cout, pool):
    x = rng.standard_normal((F, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    tx, tw, tb = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
    ref = vgg.conv3x3_relu(tx, vgg.pack_weights(tw), tb, cin, cout, fuse_pool=pool).cpu().numpy()
    got = vgg.conv3x3_relu_wino(tx, vgg.pack_weights_wino(tw), tb, cin, cout, fuse_pool=pool).cpu().numpy()
    err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)
    orc = O.conv3x3_same_relu(x[:1].astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    if pool:
        orc = O.maxpool2x2(orc)
    err_o = np.abs(got[:1] - orc).max() / (np.abs(orc).max() + 1e-30)
    err_d = np.abs(ref[:1] - orc).max() / (np.abs(orc).max() + 1e-30)
    print("F%d %dx%d %d->%d pool=%d  wino-vs-direct %.2e  wino-vs-f64 %.2e  direct-vs-f64 %.2e" % (F, H, W, cin, cout, pool, err, err_o, err_d), flush=True)
    return err


if len(sys.argv) > 1 and sys.argv[1] == "check":
    bad = 0
    for cfg in [(2, 8, 28, 32, 64, False), (2, 8, 28, 64, 64, True), (1, 28, 28, 128, 128, False), (3, 12, 56, 32, 256, True),
                (1, 28, 28, 64, 512, False), (2, 8, 16, 32, 64, False), (1, 16, 48, 64, 128, True), (1, 112, 112, 64, 128, False), (1, 8, 24, 32, 64, False), (3, 56, 56, 128, 256, True), (1, 8, 8, 32, 64, True), (1, 4, 4, 32, 64, False), (2, 12, 20, 32, 128, True)]:
        bad += check(*cfg) > 1e-4
    print("BAD" if bad else "OK")
    sys.exit(1 if bad else 0)

F = int(sys.argv[1]) if len(sys.argv) > 1 else 640
H = 224
tot_d = tot_w = 0.0
x = None
for name, cin, cout, pool in vgg.VGG_LAYERS:
    if cin == 3:
        H2 = H
    else:
        x = torch.randn((F, H, H, cin), device=dev)
        w = torch.randn((3, 3, cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
        b = torch.zeros(cout, device=dev)
        wp, up = vgg.pack_weights(w), vgg.pack_weights_wino(w)
        oh = H // 2 if pool else H
        out = torch.empty((F, oh, oh, cout), device=dev)
        res = []
        for fn, pk in ((vgg.conv3x3_relu, wp), (vgg.conv3x3_relu_wino, up)):
            fn(x, pk, b, cin, cout, fuse_pool=pool, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn(x, pk, b, cin, cout, fuse_pool=pool, out=out)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 3)
        fl = 2.0 * F * H * H * 9 * cin * cout
        tot_d += res[0]; tot_w += res[1]
        print("%-8s H%3d %3d->%3d  direct %7.3f ms %6.1f TF | wino %7.3f ms %6.1f TF(eff)  x%.2f" %
              (name, H, cin, cout, res[0], fl / res[0] / 1e9, res[1], fl / res[1] / 1e9, res[0] / res[1]), flush=True)
        del x, out
    if pool:
        H //= 2
print("sum (9 layers) direct %.2f ms, wino %.2f ms" % (tot_d, tot_w))

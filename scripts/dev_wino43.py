"""Dev: Winograd F(4x4,3x3) kernel -- correctness vs the direct MFMA kernel and a float64 convolution, then per-layer timing
against the direct and F(2x2,3x3) kernels."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntmtrack import vgg
from oracle import ntm_oracle as O

dev = torch.device("cuda")
rng = np.random.default_rng(0)


def check(F, H, W, cin, cout, pool):
    x = np.maximum(rng.standard_normal((F, H, W, cin)), 0).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    tx, tw, tb = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev)
    ref = vgg.conv3x3_relu(tx, vgg.pack_weights(tw), tb, cin, cout, fuse_pool=pool).cpu().numpy()
    got = vgg.conv3x3_relu_wino43(tx, vgg.pack_weights_wino43(tw), tb, cin, cout, fuse_pool=pool).cpu().numpy()
    torch.cuda.synchronize()
    err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)
    orc = O.conv3x3_same_relu(x[:1].astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    if pool:
        orc = O.maxpool2x2(orc)
    err_o = np.abs(got[:1] - orc).max() / (np.abs(orc).max() + 1e-30)
    print("F%d %dx%d %d->%d pool=%d  wino43-vs-direct %.2e  wino43-vs-f64 %.2e" % (F, H, W, cin, cout, pool, err, err_o), flush=True)
    return max(err, err_o)


if len(sys.argv) > 1 and sys.argv[1] == "check":
    bad = 0
    for cfg in [(1, 16, 32, 32, 64, False), (2, 16, 32, 32, 64, True), (1, 16, 16, 32, 128, False), (3, 32, 16, 64, 64, True),
                (2, 8, 8, 32, 64, False), (5, 8, 24, 64, 128, True), (3, 4, 4, 32, 64, False), (7, 28, 28, 32, 64, True),
                (2, 12, 20, 32, 512, False), (1, 112, 112, 64, 128, False), (3, 56, 56, 128, 256, True), (2, 28, 28, 256, 512, False)]:
        bad += check(*cfg) > 1e-4
    print("BAD" if bad else "OK")
    sys.exit(1 if bad else 0)

F = int(sys.argv[1]) if len(sys.argv) > 1 else 640
H = 224
tot = [0.0, 0.0, 0.0]
for name, cin, cout, pool in vgg.VGG_LAYERS:
    if cin != 3:
        x = torch.randn((F, H, H, cin), device=dev)
        w = torch.randn((3, 3, cin, cout), device=dev) * (2.0 / (9 * cin)) ** 0.5
        b = torch.zeros(cout, device=dev)
        packs = (vgg.pack_weights(w), vgg.pack_weights_wino(w), vgg.pack_weights_wino43(w))
        oh = H // 2 if pool else H
        out = torch.empty((F, oh, oh, cout), device=dev)
        res = []
        for fn, pk in zip((vgg.conv3x3_relu, vgg.conv3x3_relu_wino, vgg.conv3x3_relu_wino43), packs):
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
        for i in range(3):
            tot[i] += res[i]
        print("%-8s H%3d %3d->%3d  direct %7.3f ms | F(2x2) %7.3f ms %6.1f TF(eff) | F(4x4) %7.3f ms %6.1f TF(eff)  x%.2f vs F(2x2)" %
              (name, H, cin, cout, res[0], res[1], fl / res[1] / 1e9, res[2], fl / res[2] / 1e9, res[1] / res[2]), flush=True)
        del x, out
    if pool:
        H //= 2
print("sum (9 layers) direct %.2f ms, F(2x2) %.2f ms, F(4x4) %.2f ms" % tuple(tot))

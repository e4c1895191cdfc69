"""A/B of conv kernel variants in ONE process, interleaved rounds (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import vgg, _lib
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 640
g = torch.Generator().manual_seed(0)
layers = [("conv1_2", 224, 64, 64, True), ("conv2_2", 112, 128, 128, True), ("conv3_2", 56, 256, 256, False), ("conv4_2", 28, 512, 512, False)]
L = _lib.lib()
for name, H, cin, cout, pool in layers:
    x = torch.randn((F, H, H, cin), generator=g).to(dev)
    w = (torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
    b = torch.zeros(cout, device=dev); wp = vgg.pack_weights(w)
    ref = None; res = {}
    for rnd in range(3):
        for v in (4, 2):
            L.ntk_vgg_set_conv_variant(v)
            y = vgg.conv3x3_relu(x, wp, b, cin, cout, fuse_pool=pool); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); y = vgg.conv3x3_relu(x, wp, b, cin, cout, fuse_pool=pool); e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1))
            if ref is None: ref = y.clone()
            else: assert torch.equal(ref, y), "variant %d differs" % v
    fl = 2.0 * F * H * H * 9 * cin * cout
    print(name, " ".join("V%d: %.3f ms (%.1f TF)" % (v, min(t), fl / min(t) / 1e9) for v, t in res.items()), flush=True)
L.ntk_vgg_set_conv_variant(4)

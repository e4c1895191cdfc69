"""Dev: the fp32 trunk with its early layers in the split form (algo="split3") against the all-Winograd trunk: difference of the
conv4_3 maps, both against torch's fp64 trunk on a few frames, and the time of a 640-frame pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ntmtrack import vgg

F = int(sys.argv[1]) if len(sys.argv) > 1 else 640
dev = torch.device("cuda")
g = torch.Generator().manual_seed(42)
ws = {}
for name, cin, cout, _ in vgg.VGG_LAYERS:
    ws[name] = (torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5, torch.randn((cout,), generator=g) * 0.05)
frames = (torch.rand((F, 224, 224, 3), generator=g) * 255 - 117.0).to(dev)


def ref64(x):
    x = x.double().permute(0, 3, 1, 2)
    for name, cin, cout, pool in vgg.VGG_LAYERS:
        w, b = ws[name]
        x = torch.nn.functional.conv2d(x, w.double().permute(3, 2, 0, 1).to(dev), b.double().to(dev), padding=1).clamp_min(0)
        if pool and name != "conv4_3":
            x = torch.nn.functional.max_pool2d(x, 2)
    return x.permute(0, 2, 3, 1)


def timeit(net, parts):
    net.split_streams = parts
    out = torch.empty((F, 28, 28, 512), device=dev)
    for _ in range(2):
        net(frames, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        net(frames, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4


r = ref64(frames[:3])
s = r.abs().max().item()
nets = [("winograd", vgg.VGG16Conv43(ws, device=dev, algo="winograd"))]
for upto in ("conv3_3", "conv2_2", "conv4_3"):
    n = vgg.VGG16Conv43(ws, device=dev, algo="split3")
    n.split3_upto = upto
    nets.append(("split3 up to " + upto, n))
for label, net in nets:
    y = net(frames[:3].contiguous())
    err = (y.double() - r).abs().max().item() / s
    print("%-24s max err / max|y| vs fp64 trunk %.2e   640-frame pass: %.3f ms (1 part)  %.3f ms (2 parts)  %.3f ms (3)  %.3f ms (4)"
          % (label, err, timeit(net, 1), timeit(net, 2), timeit(net, 3), timeit(net, 4)), flush=True)

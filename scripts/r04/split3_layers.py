"""Dev: the SPLIT form (fp32 convolution as three bf16 MFMA products, csrc/conv_bf16p.hip X3) layer by layer: error against an fp64
convolution next to the F(4x4) Winograd kernel's, and time per layer on `frames` frames (default 640)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
vgg = importlib.import_module("ntm-tracker_amd.vgg")
dev = torch.device("cuda:0")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 640
only3 = len(sys.argv) > 2 and sys.argv[2] == "only3"        # ablation builds: time the split form only
g = torch.Generator(device="cpu").manual_seed(0)


def ref64(x, w, b, pool):
    y = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), b.double(), padding=1).clamp_min(0)
    if pool:
        y = torch.nn.functional.max_pool2d(y, 2)
    return y.permute(0, 2, 3, 1)


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot3 = totw = 0.0
for name, cin, cout, pool, H in (("conv1_2", 64, 64, True, 224), ("conv2_1", 64, 128, False, 112), ("conv2_2", 128, 128, True, 112),
                                  ("conv3_1", 128, 256, False, 56), ("conv3_2", 256, 256, False, 56), ("conv3_3", 256, 256, True, 56),
                                  ("conv4_1", 256, 512, False, 28), ("conv4_2", 512, 512, False, 28), ("conv4_3", 512, 512, False, 28)):
    w = (torch.randn((3, 3, cin, cout), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
    b = (torch.randn((cout,), generator=g) * 0.1).to(dev)
    # parity on two frames
    x = (torch.randn((2, H, H, cin), generator=g).clamp_min(0) * 2).to(dev)
    wp = vgg.pack_weights_split3(w, H, H)
    last = name == "conv4_3"
    y = vgg.conv3x3_relu_split3(vgg.to_split(x), wp, b, cin, cout, fuse_pool=pool, out_f32=last)
    y = y if last else vgg.from_split(y)
    r = ref64(x, w, b, pool)
    s = r.abs().max().item()
    e3 = (y.double() - r).abs().max().item() / s
    ew = tw = float("nan")
    if not only3:
        u = vgg.pack_weights_wino43(w)
        yw = vgg.conv3x3_relu_wino43(x, u, b, cin, cout, fuse_pool=pool)
        ew = (yw.double() - r).abs().max().item() / s
    # time
    xs = vgg.to_split(torch.randn((frames, H, H, cin), generator=g).clamp_min(0).to(dev))
    t3 = timeit(lambda: vgg.conv3x3_relu_split3(xs, wp, b, cin, cout, fuse_pool=pool, out_f32=last))
    if not only3:
        xb = torch.empty((frames, H, cin // 8, H, 8), device=dev) if name != "conv1_2" else torch.empty((frames, H, H, cin), device=dev)
        xb.normal_().clamp_(min=0)                           # post-ReLU-like, as the split form's input (zeros change the clock the chip holds)
        tw = timeit(lambda: vgg.conv3x3_relu_wino43_blocked(xb, u, b, cin, cout, fuse_pool=pool, out_blocked=not last))
        del xb
    fl = 2.0 * 9 * cin * cout * H * H * frames
    tot3 += t3; totw += tw
    print("%s: split3 %.3f ms (%.0f TF fp32-equivalent, MFMA pipe %.1f %% of 2.5 PF)  winograd43 %.3f ms   max err / max|y|: split3 %.2e  winograd43 %.2e"
          % (name, t3, fl / t3 * 1e-9, 3 * fl / t3 * 1e-9 / 2500 * 100, tw, e3, ew), flush=True)
    del xs
print("nine layers: split3 %.2f ms, winograd43 (blocked) %.2f ms" % (tot3, totw))

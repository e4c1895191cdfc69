"""Dev (GPU): the fp32 F(4x4) trunk on 640 frames, blocked layout against NHWC: per layer alone (events) and the whole pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ntmtrack import vgg
F = int(sys.argv[1]) if len(sys.argv) > 1 else 640
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ws = {n: ((rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32), np.zeros(co, np.float32)) for n, ci, co, _ in vgg.VGG_LAYERS}
net = vgg.VGG16Conv43(ws, device=dev)
frames = (torch.rand((F, 224, 224, 3), device=dev) * 255.0 - 120.0)
def t_ms(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
out = torch.empty((F, 28, 28, 512), device=dev)
for layout in ("nhwc", "blocked"):
    net.layout = layout
    for split in (2, 1):
        net.split_streams = split
        print("%s trunk pass, %d stream part(s): %.3f ms" % (layout, split, t_ms(lambda: net(frames, out=out))), flush=True)
# per layer alone
h = w = 224
x_n = frames
wp, b = net.packed["conv1_1"]
y_n = vgg.conv3x3_relu(x_n, wp, b, 3, 64)
y_b = y_n
print("conv1_1 (NHWC either way): %.3f ms" % t_ms(lambda: vgg.conv3x3_relu(x_n, wp, b, 3, 64, out=y_n)), flush=True)
tot_n = tot_b = 0.0
for name, cin, cout, pool in vgg.VGG_LAYERS[1:]:  # y_b starts NHWC (conv1_1), conv1_2 writes the first blocked map
    u, bb = net.packed_wino43[name], net.packed[name][1]
    o_n = vgg.conv3x3_relu_wino43(y_n, u, bb, cin, cout, fuse_pool=pool)
    o_b = vgg.conv3x3_relu_wino43_blocked(y_b, u, bb, cin, cout, fuse_pool=pool)
    tn = t_ms(lambda: vgg.conv3x3_relu_wino43(y_n, u, bb, cin, cout, fuse_pool=pool, out=o_n))
    tb = t_ms(lambda: vgg.conv3x3_relu_wino43_blocked(y_b, u, bb, cin, cout, fuse_pool=pool, out=o_b))
    same = torch.equal(vgg.blocked_to_nhwc(o_b), o_n)
    print("%s: nhwc %.3f ms, blocked %.3f ms (%+.1f %%) same bits: %s" % (name, tn, tb, 100.0 * (tb - tn) / tn, same), flush=True)
    tot_n += tn; tot_b += tb
    y_n, y_b = o_n, o_b
print("nine layers: nhwc %.2f ms, blocked %.2f ms" % (tot_n, tot_b))

"""Dev (GPU): config 5's bf16 trunk on 640 frames: the patch-form kernel (csrc/conv_bf16p.hip) against round 2's tile kernel,
per layer alone and the whole pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ntmtrack import vgg
F = int(sys.argv[1]) if len(sys.argv) > 1 else 640
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ws = {n: ((rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32), np.zeros(co, np.float32)) for n, ci, co, _ in vgg.VGG_LAYERS}
net = vgg.VGG16Conv43(ws, device=dev, dtype="bf16")
frames = (torch.rand((F, 224, 224, 3), device=dev) * 255.0 - 120.0)
def t_ms(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
out = torch.empty((F, 28, 28, 512), device=dev)
flops = vgg.conv_flops_per_frame() * F
for form, split in (("tile", 1), ("patch", 1), ("patch", 2)):
    net.bf16_form = form
    net.split_streams = split
    ms = t_ms(lambda: net(frames, out=out))
    print("%s trunk pass, %d stream part(s): %.3f ms = %.0f TFLOP/s (%.1f %% of 2.5 PF)" % (form, split, ms, flops / ms / 1e9, 100 * flops / ms / 1e9 / 2500), flush=True)
net.split_streams = 1
from ntmtrack import _lib
wp11, b11 = net.packed["conv1_1"]
x11 = torch.empty((F, 224, 224, 64), device=dev, dtype=torch.bfloat16)
t11 = t_ms(lambda: _lib.check(_lib.lib().ntk_vgg_conv3x3_relu_f32_to_bf16(_lib.ptr(frames), _lib.ptr(wp11), _lib.ptr(b11), _lib.ptr(x11), F, 224, 224, 3, 64, _lib.stream()), "c11"))
print("conv1_1 (fp32 frames -> bf16): %.3f ms = %.2f TB/s of stores" % (t11, F * 224 * 224 * 64 * 2 / t11 / 1e9), flush=True)
x = torch.relu(torch.randn((F, 224, 224, 64), device=dev)).to(torch.bfloat16)
h = w = 224
tot_t = tot_p = 0.0
for name, cin, cout, pool in vgg.VGG_LAYERS[1:]:
    last = name == "conv4_3"
    wt, b = net.packed[name]
    wp = vgg.pack_weights_bf16p(net._w_hwio[name], h, w)
    o_t = vgg.conv3x3_relu_bf16(x, wt, b, cin, cout, fuse_pool=pool, out_f32=last)
    o_p = vgg.conv3x3_relu_bf16p(x, wp, b, cin, cout, fuse_pool=pool, out_f32=last)
    tt = t_ms(lambda: vgg.conv3x3_relu_bf16(x, wt, b, cin, cout, fuse_pool=pool, out_f32=last, out=o_t))
    tp = t_ms(lambda: vgg.conv3x3_relu_bf16p(x, wp, b, cin, cout, fuse_pool=pool, out_f32=last, out=o_p))
    fl = 2.0 * h * w * 9 * cin * cout * F
    d = float((o_t.float() - o_p.float()).abs().max()) / float(o_t.float().abs().max())
    print("%s: tile %.3f ms (%.0f TF), patch %.3f ms (%.0f TF = %.1f %% of peak)  max rel diff %.1e" % (name, tt, fl / tt / 1e9, tp, fl / tp / 1e9, 100 * fl / tp / 1e9 / 2500, d), flush=True)
    tot_t += tt; tot_p += tp
    x = o_p if not last else x
    if pool: h //= 2; w //= 2
print("nine layers: tile %.2f ms, patch %.2f ms" % (tot_t, tot_p))

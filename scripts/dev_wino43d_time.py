"""Dev: time two layer shapes with the eight-wave F(4x4,3x3) kernel of the library in NTK_LIB_PATH (ablation builds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import vgg
dev = torch.device("cuda")
WAVES = 8 if (sys.argv[1] if len(sys.argv) > 1 else "1") == "1" else 4
res = []
for (H, cin, cout) in ((56, 256, 256), (224, 64, 64), (28, 512, 512)):
    x = torch.randn((640, H, H, cin), device=dev)
    w = torch.randn((3, 3, cin, cout), device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    up = vgg.pack_weights_wino43(w)
    out = torch.empty((640, H, H, cout), device=dev)
    vgg.conv3x3_relu_wino43(x, up, b, cin, cout, out=out, waves=WAVES)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        vgg.conv3x3_relu_wino43(x, up, b, cin, cout, out=out, waves=WAVES)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 3)
    del x, out
print("%-40s conv3_2 %.3f ms  conv1_2 %.3f ms  conv4_2 %.3f ms" % (os.environ.get("NTK_LIB_PATH", "product"), res[0], res[1], res[2]), flush=True)

import sys, os, torch
sys.path.insert(0, os.getcwd())
from ntmtrack import vgg
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
ws = {n: (torch.randn((3, 3, ci, co), generator=g) * (2.0 / (9 * ci)) ** 0.5, torch.zeros(co)) for n, ci, co, _ in vgg.VGG_LAYERS}
for F in (1, 4, 8, 16):
    x = (torch.rand((F, 224, 224, 3), generator=g) * 255 - 117).to(dev)
    for algo in ("split3", "winograd"):
        net = vgg.VGG16Conv43(ws, device=dev, algo=algo)
        for _ in range(3): net(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): net(x)
        e1.record(); torch.cuda.synchronize()
        print("F=%d %s: %.3f ms per pass" % (F, algo, e0.elapsed_time(e1) / 10))

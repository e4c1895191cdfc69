"""Dev: time conv1_1 (3 -> 64 channels, 224 x 224, 640 frames) -- the row kernel of csrc/mfma_f32.hip."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ntmtrack import vgg
dev = torch.device("cuda")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 640
x = torch.randn((F, 224, 224, 3), device=dev)
w = torch.randn((3, 3, 3, 64), device=dev) * 0.2
b = torch.zeros(64, device=dev)
wp = vgg.pack_weights(w)
out = torch.empty((F, 224, 224, 64), device=dev)
vgg.conv3x3_relu(x, wp, b, 3, 64, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    vgg.conv3x3_relu(x, wp, b, 3, 64, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("%s conv1_1 %d frames: %.3f ms  (%.2f TB/s written)" % (os.environ.get("NTK_LIB_PATH", "product"), F, ms, F * 224 * 224 * 64 * 4 / ms / 1e9), flush=True)

"""Dev: a few launches of one Winograd layer shape (for rocprofv3 --pmc)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntmtrack import vgg
dev = torch.device("cuda")
F, H, cin, cout = int(sys.argv[1]) if len(sys.argv) > 1 else 640, 56, 256, 256
if len(sys.argv) > 3:                      # H,cin,cout of another layer shape
    H, cin, cout = [int(v) for v in sys.argv[3].split(",")]
algo = sys.argv[2] if len(sys.argv) > 2 else "wino"
x = torch.randn((F, H, H, cin), device=dev)
w = torch.randn((3, 3, cin, cout), device=dev) * 0.02
b = torch.zeros(cout, device=dev)
pack, conv = (vgg.pack_weights_wino43, vgg.conv3x3_relu_wino43) if algo == "wino43" else (vgg.pack_weights_wino, vgg.conv3x3_relu_wino)
up = pack(w)
out = torch.empty((F, H, H, cout), device=dev)
for _ in range(3):
    conv(x, up, b, cin, cout, out=out)
torch.cuda.synchronize()

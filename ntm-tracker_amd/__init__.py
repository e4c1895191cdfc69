"""ntmtrack -- MI355X-native tracking hot path (VGG-16 conv features -> NTM/DNC
memory cell -> bbox offsets) behind the reference's NTMCell / LoopNTMTracker /
DNC operator API.  Host code is Python on PyTorch-ROCm (device memory, streams,
torch.distributed); all arithmetic on the path runs in hand-written HIP kernels
for gfx950 reached through the C ABI in include/ntmtrack.h."""
__version__ = "0.1.0"

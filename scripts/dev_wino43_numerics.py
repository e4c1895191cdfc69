"""Dev: fp32 error of Winograd F(4x4,3x3) vs F(2x2,3x3) vs a float64 direct convolution on VGG-like layer data
(numpy emulation of the kernel's arithmetic order: fp32 transforms, fp32 products accumulated over channels in fp32)."""
import numpy as np
rng = np.random.default_rng(0)

BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def wino(x, w, BT, G, AT, m):
    """x [H,W,C] fp32 (H, W multiples of m), w [3,3,C,O] -> y [H,W,O] fp32; SAME padding."""
    H, W, C = x.shape
    O = w.shape[3]
    a = m + 2
    f32 = np.float32
    U = np.einsum("ik,klco,jl->ijco", G, w.astype(np.float64), G).astype(f32)      # weights transformed in float64, stored fp32
    xp = np.zeros((H + 2, W + 2, C), f32)
    xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, O), f32)
    BTf, ATf = BT.astype(f32), AT.astype(f32)
    for ty in range(H // m):
        for tx in range(W // m):
            d = xp[ty * m:ty * m + a, tx * m:tx * m + a]                              # [a,a,C]
            t = np.einsum("ik,klc->ilc", BTf, d).astype(f32)
            V = np.einsum("ilc,jl->ijc", t, BTf).astype(f32)
            M = np.zeros((a, a, O), f32)
            for c0 in range(0, C, 2):                                                   # MFMA k = 2 accumulation order
                M = (M + np.einsum("ijc,ijco->ijo", V[:, :, c0:c0 + 2], U[:, :, c0:c0 + 2])).astype(f32)
            z = np.einsum("ik,klo->ilo", ATf, M).astype(f32)
            y[ty * m:(ty + 1) * m, tx * m:(tx + 1) * m] = np.einsum("ilo,jl->ijo", z, ATf).astype(f32)
    return y


def direct64(x, w):
    H, W, C = x.shape
    xp = np.zeros((H + 2, W + 2, C)); xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, w.shape[3]))
    for ky in range(3):
        for kx in range(3):
            y += xp[ky:ky + H, kx:kx + W] @ w[ky, kx].astype(np.float64)
    return y


for C, O, H in ((64, 64, 16), (128, 128, 16), (256, 256, 12), (512, 512, 12)):
    x = np.maximum(rng.standard_normal((H, H, C)), 0).astype(np.float32) * 2       # post-ReLU-like activations
    w = (rng.standard_normal((3, 3, C, O)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    ref = direct64(x, w)
    s = np.abs(ref).max()
    e2 = np.abs(wino(x, w, BT2, G2, AT2, 2) - ref).max() / s
    e4 = np.abs(wino(x, w, BT4, G4, AT4, 4) - ref).max() / s
    print("C%4d O%4d: F(2x2) %.2e   F(4x4) %.2e   (max abs error / max |y|)" % (C, O, e2, e4), flush=True)

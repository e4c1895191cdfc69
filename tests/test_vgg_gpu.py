"""GPU parity: HIP MFMA conv / VGG trunk vs the numpy oracle (oracle/ntm_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import ntm_oracle as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


@pytest.mark.parametrize("F,H,W,cin,cout,pool", [
    (2, 8, 12, 3, 64, False),      # conv1_1 path (tiny Cin, scalar gather loader)
    (1, 16, 16, 3, 64, True),
    (2, 8, 32, 3, 64, False),      # conv1_1 dedicated persistent row kernel (W a multiple of 32, no pool)
    (1, 4, 96, 3, 64, False),
    (3, 8, 8, 64, 64, True),       # BN=64 tile, fused pool
    (2, 12, 8, 64, 128, False),    # BN=128 tile
    (1, 28, 28, 256, 512, False),  # conv4_1 shape, one frame (ragged last row-tile: 784 = 6*128+16)
    (3, 4, 4, 32, 64, False),      # smallest legal image
    (2, 8, 8, 128, 256, True),
])
def test_conv3x3_relu_matches_oracle(cuda, F, H, W, cin, cout, pool):
    from ntmtrack import vgg
    rng = np.random.default_rng(7)
    x = rng.standard_normal((F, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    ref = O.conv3x3_same_relu(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    if pool:
        ref = O.maxpool2x2(ref)
    wp = vgg.pack_weights(torch.from_numpy(w).to(cuda))
    out = vgg.conv3x3_relu(torch.from_numpy(x).to(cuda), wp, torch.from_numpy(b).to(cuda), cin, cout, fuse_pool=pool)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    # fp32 MFMA = k-ordered fmaf chain: error ~1e-7 * sum|a*b|; tolerance 1e-5 relative to max
    assert _rel(got, ref) < 1e-5


def test_conv_rejects_bad_shapes(cuda):
    from ntmtrack import vgg, _lib
    x = torch.zeros((1, 6, 8, 32), device=cuda)
    wp = torch.zeros((64, 288), device=cuda)
    b = torch.zeros(64, device=cuda)
    with pytest.raises(_lib.NtkError):
        vgg.conv3x3_relu(x, wp, b, 32, 64)       # H not a multiple of 4
    x = torch.zeros((1, 8, 8, 48), device=cuda)
    with pytest.raises(_lib.NtkError):
        vgg.conv3x3_relu(x, wp, b, 48, 64)       # Cin neither 3 nor multiple of 32


def test_vgg_trunk_matches_oracle_small(cuda):
    """Full conv1_1..conv4_3 stack on 2 frames of 32x32 (oracle finishes in seconds)."""
    from ntmtrack import vgg
    rng = np.random.default_rng(42)
    ws = O.init_vgg_weights(rng)
    frames = (rng.uniform(0, 255, size=(2, 32, 32, 3)).astype(np.float32) - O.VGG_MEAN)
    ref = O.vgg16_conv43(frames.astype(np.float64), {k: (w.astype(np.float64), b.astype(np.float64)) for k, (w, b) in ws.items()})
    net = vgg.VGG16Conv43(ws, device=cuda)
    out = net(torch.from_numpy(frames).to(cuda))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got.shape == (2, 4, 4, 512)
    assert _rel(got, ref) < 1e-4   # north_star tolerance (1e-4 fp32) through 10 layers


# ---- bf16 trunk (BASELINE config 5)
@pytest.mark.parametrize("F,H,W,cin,cout,pool,out_f32", [
    (2, 8, 12, 64, 64, True, False),
    (1, 16, 16, 64, 128, False, False),
    (1, 28, 28, 256, 512, False, True),
    (3, 4, 4, 128, 256, True, False),
])
def test_conv3x3_relu_bf16_matches_bf16_oracle(cuda, F, H, W, cin, cout, pool, out_f32):
    """Operands rounded to bf16 in the oracle exactly as the kernel sees them; fp32 accumulation order is the
    only difference, plus one bf16 rounding of the stored result (half an ulp = 2^-9 relative)."""
    from ntmtrack import vgg
    rng = np.random.default_rng(9)
    x = O.bf16_round(rng.standard_normal((F, H, W, cin)).astype(np.float32))
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    ref = O.conv3x3_same_relu(x.astype(np.float64), O.bf16_round(w).astype(np.float64), b.astype(np.float64))
    if pool:
        ref = O.maxpool2x2(ref)
    wp = vgg.pack_weights_bf16(torch.from_numpy(w).to(cuda))
    out = vgg.conv3x3_relu_bf16(torch.from_numpy(x).to(cuda).to(torch.bfloat16), wp, torch.from_numpy(b).to(cuda), cin, cout,
                                fuse_pool=pool, out_f32=out_f32)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert got.shape == ref.shape
    scale = np.max(np.abs(ref))
    if out_f32:
        assert np.max(np.abs(got - ref)) / scale < 1e-5
    else:
        assert np.max(np.abs(got - ref) / (np.abs(ref) + 1e-3 * scale)) < 2.0 ** -8      # one bf16 rounding of the output


@pytest.mark.parametrize("F,H,W,cin,cout,pool,out_f32", [
    (2, 16, 32, 64, 64, True, False),        # 32x16 sub-blocks, 64 columns (conv1_2's form), fused pool
    (1, 32, 64, 64, 128, False, False),      # 32x16, 128 columns, several blocks per frame
    (3, 16, 16, 128, 128, True, False),      # 16x16 sub-blocks x 2 (conv2_2's form): a block spans two frames
    (5, 8, 8, 128, 256, False, False),       # 8x8 sub-blocks x 8: a ragged last block (5 sub-blocks of 8)
    (2, 24, 8, 256, 256, True, False),       # 8x8 with the pool (conv3_3's form)
    (3, 28, 28, 256, 512, False, True),      # 4x4 sub-blocks x 32, 16-channel chunks, fp32 output (conv4_3's form), 49 tiles per frame
    (1, 4, 4, 32, 64, False, False),         # one sub-block
])
def test_conv3x3_relu_bf16_patch_form_matches_bf16_oracle(cuda, F, H, W, cin, cout, pool, out_f32):
    """csrc/conv_bf16p.hip (the patch-resident kernel config 5's trunk runs): every sub-block shape, both column-block widths,
    the fused pool and the fp32 output, against the bf16-emulating oracle -- same bounds as the tile kernel's test above."""
    from ntmtrack import vgg, _lib
    assert _lib.lib().ntk_vgg_bf16p_supported(H, W, cin, cout, 1 if pool else 0) == 1
    rng = np.random.default_rng(19)
    x = O.bf16_round(rng.standard_normal((F, H, W, cin)).astype(np.float32))
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    ref = O.conv3x3_same_relu(x.astype(np.float64), O.bf16_round(w).astype(np.float64), b.astype(np.float64))
    if pool:
        ref = O.maxpool2x2(ref)
    wp = vgg.pack_weights_bf16p(torch.from_numpy(w).to(cuda), H, W)
    out = vgg.conv3x3_relu_bf16p(torch.from_numpy(x).to(cuda).to(torch.bfloat16), wp, torch.from_numpy(b).to(cuda), cin, cout,
                                 fuse_pool=pool, out_f32=out_f32)
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    assert got.shape == ref.shape
    scale = np.max(np.abs(ref))
    if out_f32:
        assert np.max(np.abs(got - ref)) / scale < 1e-5
    else:
        assert np.max(np.abs(got - ref) / (np.abs(ref) + 1e-3 * scale)) < 2.0 ** -8      # one bf16 rounding of the output
    assert _lib.lib().ntk_vgg_bf16p_supported(12, 12, cin, cout, 1) == 0                # 4x4 sub-blocks have no pooled form: tile kernel


def test_vgg_trunk_bf16_matches_bf16_oracle(cuda):
    from ntmtrack import vgg
    rng = np.random.default_rng(43)
    ws = O.init_vgg_weights(rng)
    frames = (rng.uniform(0, 255, size=(2, 32, 32, 3)).astype(np.float32) - O.VGG_MEAN)
    ref = O.vgg16_conv43_bf16(frames, ws)
    ref32 = O.vgg16_conv43(frames.astype(np.float64), {k: (w.astype(np.float64), b.astype(np.float64)) for k, (w, b) in ws.items()})
    net = vgg.VGG16Conv43(ws, device=cuda, dtype="bf16")
    got = net(torch.from_numpy(frames).to(cuda)).cpu().numpy()
    torch.cuda.synchronize()
    scale = np.max(np.abs(ref))
    # vs the bf16-emulating oracle: activation roundings can land on the other side of a tie after a different
    # accumulation order -> a few bf16 ulps through 9 bf16 layers
    assert np.max(np.abs(got - ref)) / scale < 2e-2
    # and the bf16 trunk stays within ~1 % of the fp32 trunk (the stated tolerance of config 5's conv)
    assert np.max(np.abs(got - ref32)) / np.max(np.abs(ref32)) < 3e-2


# ---------------------------------------------------------------------------------------------------------
# the SPLIT form of the fp32 operator (csrc/conv_bf16p.hip, X3): every fp32 value as fp16 hi + lo, x w = xh wh + xh wl + xl wh
# with fp32 accumulators.  Tolerance: 4e-6 of the activation scale per layer (measured 0.7e-6 .. 1.7e-6; the F(4x4) Winograd
# kernel's bound below is 3e-5), 1e-4 through the trunk (north_star).
@pytest.mark.parametrize("F,H,W,cin,cout,pool,in_f32,out_f32", [
    (2, 16, 32, 64, 64, True, True, False),       # conv1_2's form: four waves, 32x8 sub-blocks, fp32 NHWC input split by the staging, pool
    (2, 16, 32, 64, 64, False, False, False),     # ... the same shape from a split map, un-pooled
    (3, 16, 16, 64, 64, True, True, True),        # four waves, 16x16 sub-blocks, fp32 in AND out
    (5, 8, 8, 32, 64, False, True, False),        # four waves, 8x8 sub-blocks x 4, ragged last block, two chunks
    (1, 32, 64, 64, 128, False, False, False),    # eight waves, 32x16 sub-blocks, 128 columns (conv2_1's form)
    (3, 16, 16, 128, 128, True, False, False),    # 16x16 sub-blocks x 2: a block spans two frames (conv2_2's form)
    (5, 8, 8, 128, 256, False, False, False),     # 8x8 sub-blocks x 8: ragged last block (conv3_1's form)
    (2, 24, 8, 256, 256, True, False, True),      # 8x8 with the pool, fp32 output (conv3_3's form when the trunk leaves the split form there)
    (3, 28, 28, 256, 512, False, False, False),   # 28-wide maps: runs of 512 consecutive pixels (18.3 rows), five workgroups, two of them across a frame boundary
    (2, 28, 28, 512, 512, False, False, True),    # conv4_3's form: fp32 output
    (5, 20, 28, 16, 64, False, False, False),     # 28 wide, 20 rows (the shortest frame the form takes), one chunk, 64 columns on eight waves
])
def test_conv3x3_relu_split_form_matches_float64_oracle(cuda, F, H, W, cin, cout, pool, in_f32, out_f32):
    from ntmtrack import vgg
    assert vgg.split3_supported(H, W, cin, cout, pool)
    rng = np.random.default_rng(23)
    x = (np.maximum(rng.standard_normal((F, H, W, cin)), 0) * 3).astype(np.float32)       # post-ReLU-like
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    ref = O.conv3x3_same_relu(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    if pool:
        ref = O.maxpool2x2(ref)
    wp = vgg.pack_weights_split3(torch.from_numpy(w).to(cuda), H, W)
    xt = torch.from_numpy(x).to(cuda)
    out = vgg.conv3x3_relu_split3(xt if in_f32 else vgg.to_split(xt), wp, torch.from_numpy(b).to(cuda), cin, cout, fuse_pool=pool, out_f32=out_f32)
    torch.cuda.synchronize()
    got = (out if out_f32 else vgg.from_split(out)).cpu().numpy()
    assert got.shape == ref.shape
    err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    assert err < 4e-6, err
    if not out_f32:
        # the split map a kernel writes is EXACTLY to_split() of the fp32 map the same launch writes with out_f32 (the format is pinned
        # bit for bit: hi rounded toward zero, lo to nearest), and a second launch gives the same bits
        y32 = vgg.conv3x3_relu_split3(xt if in_f32 else vgg.to_split(xt), wp, torch.from_numpy(b).to(cuda), cin, cout, fuse_pool=pool, out_f32=True)
        assert torch.equal(out, vgg.to_split(y32))
        again = vgg.conv3x3_relu_split3(xt if in_f32 else vgg.to_split(xt), wp, torch.from_numpy(b).to(cuda), cin, cout, fuse_pool=pool)
        assert torch.equal(out, again)
    if in_f32:                                    # the staging's split is to_split(): both routes give the same bits
        via_map = vgg.conv3x3_relu_split3(vgg.to_split(xt), wp, torch.from_numpy(b).to(cuda), cin, cout, fuse_pool=pool, out_f32=out_f32)
        assert torch.equal(out, via_map)


def test_split_form_range_and_refusals(cuda):
    """fp16's narrow exponent on both sides: weights of any magnitude (scaled by a power of two when packed: 1e-6 and 1e+6 give the
    relative error of ordinary weights), activations beyond 65504 saturate into the low part instead of overflowing (exact up to
    131008 at fp16 spacing), tiny activations keep fp16's absolute 6e-8; shapes the form does not take are refused, not mangled."""
    from ntmtrack import vgg, _lib
    rng = np.random.default_rng(29)
    F, H, W, cin, cout = 1, 16, 16, 64, 64
    b = np.zeros(cout, np.float32)
    for wmag, xmag, bound in ((1e-6, 1.0, 4e-6), (1e6, 1.0, 4e-6), (1.0, 1e-2, 4e-6), (1.0, 1.0e4, 4e-6),
                              (1.0, 2.0e4, 3e-4),               # activations up to ~9e4: those beyond 65504 carried at fp16 spacing
                              (1.0, 1e-4, 5e-4)):               # a map that is ALL below 5e-4: the low parts are under fp16's 6e-8 floor
        x = (np.abs(rng.standard_normal((F, H, W, cin))) * xmag).astype(np.float32)
        w = (rng.standard_normal((3, 3, cin, cout)) * wmag / np.sqrt(9 * cin)).astype(np.float32)
        ref = O.conv3x3_same_relu(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
        wp = vgg.pack_weights_split3(torch.from_numpy(w).to(cuda), H, W)
        got = vgg.conv3x3_relu_split3(torch.from_numpy(x).to(cuda), wp, torch.from_numpy(b).to(cuda), cin, cout, out_f32=True).cpu().numpy()
        err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
        assert np.isfinite(got).all() and err < bound, (wmag, xmag, err)
    xt = torch.tensor([0.0, 1.0, -1.0, 65504.0, 70000.0, -100000.0, 131008.0, 1e9, 3e-5, -7e-8] + [0.5] * 6, device=cuda).view(1, 1, 1, 16)
    back = vgg.from_split(vgg.to_split(xt)).view(-1)[:10].cpu().numpy()
    assert np.allclose(back[:7], [0.0, 1.0, -1.0, 65504.0, 70000.0, -100000.0, 131008.0], rtol=2.0 ** -20, atol=0)
    assert back[7] == 131008.0 and abs(back[8] - 3e-5) < 6e-8 and abs(back[9] + 7e-8) < 6e-8
    assert not vgg.split3_supported(12, 12, 64, 64, False)            # neither multiples of 8 nor 28 wide
    assert not vgg.split3_supported(12, 28, 64, 64, False)            # 28 wide but fewer than 20 rows (a run could cross two frame boundaries)
    assert not vgg.split3_supported(28, 28, 256, 256, True)           # 28-wide maps have no pooled form
    assert not vgg.split3_supported(16, 16, 24, 64, False) and not vgg.split3_supported(16, 16, 64, 96, False)
    wp = vgg.pack_weights_split3(torch.zeros((3, 3, 128, 128), device=cuda), 16, 16)
    with pytest.raises(_lib.NtkError):                                # an fp32 map is read by the four-wave form only (cin <= 64, cout = 64)
        vgg.conv3x3_relu_split3(torch.zeros((1, 16, 16, 128), device=cuda), wp, torch.zeros(128, device=cuda), 128, 128)
    with pytest.raises(_lib.NtkError):
        vgg.conv3x3_relu_split3(vgg.to_split(torch.zeros((1, 16, 16, 64), device=cuda)), wp, torch.zeros(128, device=cuda), 128, 128)


# ---------------------------------------------------------------------------------------------------------
# fused Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip): same operator, tolerance-level parity with the oracle
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("F,H,W,cin,cout,pool", [
    (2, 8, 28, 16, 64, False),      # smallest legal block, one K iteration
    (2, 8, 28, 64, 64, True),       # conv1_2-like: one column block, fused pool
    (1, 28, 28, 128, 128, False),   # two column blocks (4 XCDs each)
    (3, 12, 28, 32, 256, True),     # four column blocks, ragged XCD groups (14x2 tile blocks)
    (1, 28, 28, 64, 512, False),    # eight column blocks = one per XCD
    (1, 4, 28, 32, 1024, False),    # sixteen column blocks (two per XCD)
    (2, 8, 16, 32, 64, False),      # 8x4 tile blocks (all 32 MFMA rows): smallest
    (1, 16, 48, 64, 128, True),     # 8x4 tile blocks, several blocks per row, fused pool
    (1, 8, 24, 32, 64, False),      # 4x4x2 sub-block pairs (tile grid a multiple of 4x4 only), odd sub-block count
    (3, 56, 56, 128, 256, True),    # conv3-like: sub-block pairs straddling rows and frames, fused pool
    (1, 4, 4, 32, 64, False),       # 2x2x8 sub-blocks: a single tile block with 7 empty sub-blocks
    (2, 12, 20, 32, 128, True),     # 2x2x8 sub-blocks, ragged tail, fused pool
    (1, 28, 28, 32, 512, False),    # conv4-like tile grid (14 x 14) through the 2x2x8 path
])
def test_conv3x3_relu_winograd_matches_oracle(cuda, F, H, W, cin, cout, pool):
    from ntmtrack import vgg
    rng = np.random.default_rng(9)
    x = rng.standard_normal((F, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    ref = O.conv3x3_same_relu(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    if pool:
        ref = O.maxpool2x2(ref)
    assert vgg.wino_supported(cin, cout, H, W)
    up = vgg.pack_weights_wino(torch.from_numpy(w).to(cuda))
    got = vgg.conv3x3_relu_wino(torch.from_numpy(x).to(cuda), up, torch.from_numpy(b).to(cuda), cin, cout, fuse_pool=pool).cpu().numpy()
    assert got.shape == ref.shape
    assert _rel(got, ref) < 1e-5


def test_winograd_rejects_bad_shapes(cuda):
    from ntmtrack import vgg, _lib
    b = torch.zeros(64, device=cuda)
    u = torch.zeros(16 * 32 * 64, device=cuda)
    with pytest.raises(_lib.NtkError):
        vgg.conv3x3_relu_wino(torch.zeros((1, 8, 30, 32), device=cuda), u, b, 32, 64)      # W not a multiple of 4
    with pytest.raises(_lib.NtkError):
        vgg.conv3x3_relu_wino(torch.zeros((1, 6, 28, 32), device=cuda), u, b, 32, 64)      # H not a multiple of 4
    assert not vgg.wino_supported(3, 64, 224, 224) and not vgg.wino_supported(64, 192, 224, 224)


# ---------------------------------------------------------------------------------------------------------
# fused Winograd F(4x4,3x3) kernel (csrc/conv_wino43.hip): same operator; rounding error ~16x that of F(2x2,3x3)
# (4e-6 .. 1.1e-5 of the activation scale per layer), bound 3e-5 per layer here, 1e-4 through the trunk (north_star)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("F,H,W,cin,cout,pool", [
    (1, 16, 32, 32, 64, False),     # 8x4x1 tile block: one workgroup per column block, two K steps
    (2, 16, 32, 32, 64, True),      # ... fused pool
    (1, 16, 16, 32, 128, False),    # 4x4x2 sub-block pairs (tile grid 4 x 4), two column blocks
    (3, 32, 16, 64, 64, True),      # 4x4x2, sub-block pairs straddling frames, fused pool
    (2, 8, 8, 32, 64, False),       # 2x2x8 sub-blocks: one workgroup with 6 empty sub-blocks
    (5, 8, 24, 64, 128, True),      # 2x2x8, ragged tail, fused pool
    (3, 4, 4, 32, 64, False),       # 1x1x32: single tiles, 29 empty sub-blocks
    (7, 28, 28, 32, 64, True),      # 1x1x32 on the conv4 tile grid (7 x 7): sub-blocks straddling frames, fused pool
    (2, 12, 20, 32, 512, False),    # eight column blocks = one per XCD
    (1, 4, 28, 32, 1024, False),    # sixteen column blocks (two per XCD)
    (1, 112, 112, 64, 128, False),  # conv2-like
    (3, 56, 56, 128, 256, True),    # conv3-like, fused pool
    (2, 28, 28, 256, 512, False),   # conv4-like: 32 K steps
    (1, 180, 180, 64, 64, False),   # 45 x 45 tiles: 1x1x32 blocks spanning > 16 MB of input (the eight-wave form hands over to the four-wave kernel)
])
@pytest.mark.parametrize("waves", [8, 4])
def test_conv3x3_relu_winograd43_matches_oracle(cuda, F, H, W, cin, cout, pool, waves):
    """Both forms of the kernel (csrc/conv_wino43.hip: eight waves per workgroup = the default, four = round 2's) against the
    float64 oracle, and against each other bit for bit."""
    from ntmtrack import vgg
    rng = np.random.default_rng(9)
    x = rng.standard_normal((F, H, W, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    ref = O.conv3x3_same_relu(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    if pool:
        ref = O.maxpool2x2(ref)
    assert vgg.wino43_supported(cin, cout, H, W)
    up = vgg.pack_weights_wino43(torch.from_numpy(w).to(cuda))
    tx, tb = torch.from_numpy(x).to(cuda), torch.from_numpy(b).to(cuda)
    got_t = vgg.conv3x3_relu_wino43(tx, up, tb, cin, cout, fuse_pool=pool, waves=waves)
    got = got_t.cpu().numpy()
    assert got.shape == ref.shape
    assert _rel(got, ref) < 3e-5
    if waves == 8:
        assert torch.equal(got_t, vgg.conv3x3_relu_wino43(tx, up, tb, cin, cout, fuse_pool=pool))           # the default IS this form
        assert torch.equal(got_t, vgg.conv3x3_relu_wino43(tx, up, tb, cin, cout, fuse_pool=pool, waves=4))  # same bits


def test_winograd43_rejects_bad_shapes(cuda):
    from ntmtrack import vgg, _lib
    b = torch.zeros(64, device=cuda)
    u = torch.zeros(36 * 32 * 64, device=cuda)
    with pytest.raises(_lib.NtkError):
        vgg.conv3x3_relu_wino43(torch.zeros((1, 8, 30, 32), device=cuda), u, b, 32, 64)      # W not a multiple of 4
    with pytest.raises(_lib.NtkError):
        vgg.conv3x3_relu_wino43(torch.zeros((1, 8, 8, 24), device=cuda), torch.zeros(36 * 24 * 64, device=cuda), b, 24, 64)   # cin % 16
    assert not vgg.wino43_supported(3, 64, 224, 224) and not vgg.wino43_supported(64, 192, 224, 224)


def test_conv3x3_relu_winograd43_is_deterministic_and_frame_invariant(cuda):
    """A frame computed inside a batch equals the same frame computed alone, bit for bit (tile blocks that straddle
    frames on the 7 x 7 tile grid included), and two launches give identical bits."""
    from ntmtrack import vgg
    g = torch.Generator().manual_seed(3)
    x = torch.randn((5, 28, 28, 64), generator=g).to(cuda)
    w = (torch.randn((3, 3, 64, 128), generator=g) * 0.04).to(cuda)
    b = torch.randn(128, generator=g).to(cuda)
    up = vgg.pack_weights_wino43(w)
    full = vgg.conv3x3_relu_wino43(x, up, b, 64, 128)
    again = vgg.conv3x3_relu_wino43(x, up, b, 64, 128)
    assert torch.equal(full, again)
    for i in (0, 2, 4):
        alone = vgg.conv3x3_relu_wino43(x[i:i + 1].contiguous(), up, b, 64, 128)
        assert torch.equal(alone[0], full[i])


def test_vgg_trunk_winograd_equals_direct_fullsize(cuda):
    """The fp32 trunks against the all-direct trunk on 224x224 frames: all are fp32 with different summation orders.
    F(2x2,3x3) ("winograd2") agrees to 1e-5 of the activation scale through ten layers; the default F(4x4,3x3)
    ("winograd") to 2e-5 (measured 4.5e-6; north_star allows 1e-4)."""
    from ntmtrack import vgg
    rng = np.random.default_rng(5)
    ws = O.init_vgg_weights(rng)
    frames = torch.from_numpy((rng.uniform(0, 255, size=(2, 224, 224, 3)).astype(np.float32) - O.VGG_MEAN)).to(cuda)
    a4 = vgg.VGG16Conv43(ws, device=cuda, algo="winograd")
    a2 = vgg.VGG16Conv43(ws, device=cuda, algo="winograd2")
    d = vgg.VGG16Conv43(ws, device=cuda, algo="direct")
    assert len(a4.packed_wino43) == 9 and len(a2.packed_wino) == 9 and not a2.packed_wino43 and not d.packed_wino
    y4, y2, yd = a4(frames).cpu().numpy(), a2(frames).cpu().numpy(), d(frames).cpu().numpy()
    print("trunk vs direct trunk at 224x224: F(2x2) %.3e, F(4x4) %.3e" % (_rel(y2, yd), _rel(y4, yd)))
    assert _rel(y2, yd) < 1e-5
    assert _rel(y4, yd) < 2e-5


def test_winograd43_window_equals_whole_frame_inside_and_touches_nothing_outside(cuda):
    """ntk_vgg_conv3x3_relu_wino43_window_f32: the tiles of the window are bit-identical to the whole-frame call, every
    position outside keeps the buffer's previous content; windows that select each tile-block shape; bad windows refused."""
    from ntmtrack import vgg, _lib
    rng = np.random.default_rng(12)
    for (F, H, W, cin, cout, win) in [(3, 28, 28, 32, 64, (4, 4, 24, 24)),      # conv4_3's case: 5 x 5 of 7 x 7 tiles, 1x1x32 blocks
                                      (2, 32, 32, 16, 64, (8, 16, 24, 32)),      # 4 x 4 tiles at even offsets: 2x2x8 blocks
                                      (2, 64, 64, 16, 128, (16, 0, 48, 32)),     # 8 wide, 8 high at multiples of (8, 4) tiles: 8x4x1 blocks
                                      (1, 20, 12, 16, 64, (0, 4, 20, 8))]:
        x = torch.from_numpy(rng.standard_normal((F, H, W, cin)).astype(np.float32)).to(cuda)
        w = torch.from_numpy((rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)).to(cuda)
        b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(cuda)
        up = vgg.pack_weights_wino43(w)
        full = vgg.conv3x3_relu_wino43(x, up, b, cin, cout)
        y0, x0, y1, x1 = win
        mask = torch.zeros((H, W), dtype=torch.bool, device=cuda)
        mask[y0:y1, x0:x1] = True
        for waves in (None, 4, 8):
            out = torch.full_like(full, -7.0)
            vgg.conv3x3_relu_wino43(x, up, b, cin, cout, out=out, window=win, waves=waves)
            assert torch.equal(out[:, mask], full[:, mask]), (win, waves)
            assert bool((out[:, ~mask] == -7.0).all()), (win, waves)
    with pytest.raises(_lib.NtkError):
        vgg.conv3x3_relu_wino43(x, up, b, cin, cout, out=out, window=(0, 2, 20, 8))      # not a multiple of 4
    with pytest.raises(_lib.NtkError):
        vgg.conv3x3_relu_wino43(x, up, b, cin, cout, out=out, window=(0, 4, 24, 8))      # outside the frame


def test_latency_calls_of_a_few_frames_run_the_winograd_form(cuda):
    """`net(frames, latency=True)` runs a call of fewer than split3_latency_frames (12) frames in the F(4x4) Winograd form (shorter
    critical path: the online tracker's one frame per call) and larger calls, every chunk of them, in the split form; without the
    flag every call runs the split form (a frame's features then do not depend on the batch it is in)."""
    from ntmtrack import vgg
    rng = np.random.default_rng(31)
    ws = O.init_vgg_weights(rng)
    net = vgg.VGG16Conv43(ws, device=cuda, chunk_frames=16)
    wino = vgg.VGG16Conv43(ws, device=cuda, algo="winograd", chunk_frames=16)
    x = torch.from_numpy((rng.uniform(0, 255, size=(20, 224, 224, 3)).astype(np.float32) - O.VGG_MEAN)).to(cuda)
    small = x[:4].contiguous()
    assert torch.equal(net(small, latency=True), wino(small))                          # small latency call: the Winograd form's bits
    assert not torch.equal(net(small), wino(small)) and torch.equal(net(small), net(x)[:4])      # default: the split form, batch-invariant
    assert torch.equal(net(x, latency=True), net(x))                                   # 16 + 4 frames: both chunks in the split form
    assert _rel(net(small).cpu().numpy(), wino(small).cpu().numpy()) < 1e-5            # the two forms agree to fp32 rounding


def test_trackers_pick_the_trunk_form_per_pass(cuda):
    """The NTM tracker runs the split-form trunk everywhere; the DNC tracker runs it for trunk passes that are alone or beside an
    inference pass and the F(4x4) Winograd form for the passes submit_features() puts beside a TRAINING pass (its cluster kernels
    are the bound there and run faster beside the Winograd trunk: tracker.py); an explicit conv_algo is obeyed everywhere."""
    from ntmtrack import tracker
    rng = np.random.default_rng(17)
    ws = O.init_vgg_weights(rng)
    B, T = 2, 2
    frames = torch.from_numpy((rng.uniform(0, 255, size=(B * T, 224, 224, 3)).astype(np.float32) - O.VGG_MEAN)).to(cuda)
    gts0 = torch.from_numpy(rng.uniform(0, 1, size=(B, 64)).astype(np.float32)).to(cuda)

    def forms_seen(trk, fn):
        seen = []
        orig = trk.vgg.forward_chunk
        def spy(fr, upto="conv4_3", out=None):
            seen.append(bool(trk.vgg.split3) and trk.vgg._call_split3 and trk.vgg.split3_trunk_supported(fr.shape))
            return orig(fr, upto=upto, out=out)
        trk.vgg.forward_chunk = spy
        try:
            fn()
            trk.join() if hasattr(trk, "join") else None
            torch.cuda.synchronize()
        finally:
            trk.vgg.forward_chunk = orig
        return seen

    def drain(trk):
        trk._pending.clear()

    ntm = tracker.NTMOffsetTracker(B, T, vgg_weights=ws, device=cuda, seed=3)
    assert forms_seen(ntm, lambda: ntm.submit_features(frames)) == [True]
    drain(ntm)
    dnc = tracker.DNCOffsetTracker(B, T, vgg_weights=ws, device=cuda, seed=3, mem_size=64, mem_dim=16)
    assert forms_seen(dnc, lambda: dnc.submit_features(frames)) == [False]                     # beside a training pass: Winograd
    drain(dnc)
    assert forms_seen(dnc, lambda: dnc.submit_features(frames, beside="infer")) == [True]
    drain(dnc)
    assert forms_seen(dnc, lambda: dnc.infer(frames, gts0)) == [True]                          # alone: the split form
    assert dnc.vgg.split3                                                                      # ... and the choice is per pass, not sticky
    asked = tracker.DNCOffsetTracker(B, T, vgg_weights=ws, device=cuda, seed=3, mem_size=64, mem_dim=16, conv_algo="split3")
    assert forms_seen(asked, lambda: asked.submit_features(frames)) == [True]
    drain(asked)
    wino = tracker.DNCOffsetTracker(B, T, vgg_weights=ws, device=cuda, seed=3, mem_size=64, mem_dim=16, conv_algo="winograd")
    assert forms_seen(wino, lambda: wino.infer(frames, gts0)) == [False]


def test_tracker_features_roi_is_invisible_to_the_recurrent_core(cuda):
    """features_roi=True computes conv4_3 only where extract_features reads it (a window of the F(4x4) Winograd kernel: the trunk then
    runs the Winograd form): the serialised NTM input, the loss and the gradients are bit-identical to the whole-map Winograd tracker's."""
    from ntmtrack import tracker
    rng = np.random.default_rng(13)
    ws = O.init_vgg_weights(rng)
    B, T = 2, 2
    frames = torch.from_numpy((rng.uniform(0, 255, size=(B * T, 224, 224, 3)).astype(np.float32) - O.VGG_MEAN)).to(cuda)
    gts0 = torch.from_numpy(rng.uniform(0, 1, size=(B, 64)).astype(np.float32)).to(cuda)
    offs = torch.from_numpy(rng.uniform(-0.5, 0.5, size=(B, T, 2)).astype(np.float32)).to(cuda)
    res = []
    for roi in (False, True):
        trk = tracker.NTMOffsetTracker(B, T, vgg_weights=ws, device=cuda, seed=3, features_roi=roi, conv_algo="winograd")
        assert trk.features_roi == roi and (trk.vgg.features_window == (4, 4, 24, 24)) == roi
        fmap = trk.vgg(frames)
        X = trk.serialize(fmap, gts0)
        loss, _ = trk.loss_and_grads(fmap, gts0, offs)
        res.append((X.clone(), float(loss.cpu()), trk._flat_grad().clone(), fmap.clone()))
    assert torch.equal(res[0][0], res[1][0]) and res[0][1] == res[1][1] and torch.equal(res[0][2], res[1][2])
    assert not torch.equal(res[0][3], res[1][3])          # the maps themselves differ outside the window (zeros there)
    # ... and "zeros there" holds for a map the trunk allocates itself (VGG16Conv43.__call__ without `out`): inside the
    # window the whole-map values, outside exactly zero -- never uninitialised memory
    full, roi = res[0][3], res[1][3]
    assert torch.equal(roi[:, 4:24, 4:24], full[:, 4:24, 4:24])
    outside = roi.clone()
    outside[:, 4:24, 4:24] = 0
    assert float(outside.abs().max()) == 0.0


def test_vgg_trunk_stream_parts_do_not_change_the_result(cuda):
    """The default trunk runs a pass as two half batches on two streams (they fill each other's launch tails): the
    features are bit-identical to a one-stream pass and to a three-part pass, on the caller's current stream and on a side
    stream, and a consumer that only orders itself after the caller's stream sees the finished buffer."""
    from ntmtrack import vgg
    rng = np.random.default_rng(11)
    ws = O.init_vgg_weights(rng)
    frames = torch.from_numpy((rng.uniform(0, 255, size=(96, 64, 64, 3)).astype(np.float32) - O.VGG_MEAN)).to(cuda)
    net = vgg.VGG16Conv43(ws, device=cuda, algo="winograd")
    assert net.split_streams == 2
    y2 = net(frames).clone()
    net.split_streams = 1
    y1 = net(frames).clone()
    net.split_streams = 3
    y3 = net(frames).clone()
    assert torch.equal(y1, y2) and torch.equal(y1, y3)
    net.split_streams = 2
    side = torch.cuda.Stream(device=cuda)
    out = torch.zeros_like(y1)
    side.wait_stream(torch.cuda.current_stream(cuda))
    with torch.cuda.stream(side):
        net(frames, out=out)
        total = out.sum()                     # ordered after the pass on the caller's (side) stream only
    torch.cuda.current_stream(cuda).wait_stream(side)
    assert torch.equal(out, y1) and float(total) == float(y1.sum())


def test_vgg_trunk_fullsize_matches_float64_oracle(cuda):
    """One 224x224 frame (plus a second, so frame strides are exercised) through the DEFAULT trunk (conv1_1 direct + nine layers in
    the split form: four-wave 32x8 sub-blocks reading conv1_1's fp32 map at W = 224, 16x16 at 112, 8x8 at 56, linear 28-wide tiles at 28),
    the F(4x4,3x3) Winograd trunk (the 8x4x1 tile path at W = 224, 4x4x2 at 112, 2x2x8 at 56, 1x1x32 at 28),
    the F(2x2,3x3) trunk and the all-direct trunk, against a float64 convolution oracle (torch-CPU conv2d in double: the second restatement of
    direct_offset_output.py:417-422 / vgg.py:155-161).  north_star tolerance: 1e-4 of the activation scale."""
    from ntmtrack import vgg
    from oracle import ntm_oracle_torch as OT
    rng = np.random.default_rng(11)
    ws = O.init_vgg_weights(rng)
    for k in ws:                                       # non-zero biases: the epilogue's bias add is on the path
        ws[k] = (ws[k][0], (rng.standard_normal(ws[k][1].shape) * 0.05).astype(np.float32))
    frames = (rng.uniform(0, 255, size=(2, 224, 224, 3)).astype(np.float32) - O.VGG_MEAN)
    ref = OT.vgg16_conv43(frames.astype(np.float64), {k: (w.astype(np.float64), b.astype(np.float64)) for k, (w, b) in ws.items()})
    assert ref.dtype == np.float64 and ref.shape == (2, 28, 28, 512)
    x = torch.from_numpy(frames).to(cuda)
    assert vgg.VGG16Conv43(ws, device=cuda).split3                        # the default IS the split form ...
    for algo, bound in (("split3", 1e-5), ("winograd", 2e-5), ("winograd2", 1e-5), ("direct", 1e-5)):
        net = vgg.VGG16Conv43(ws, device=cuda, algo=algo)
        assert algo != "split3" or (net.split3_trunk_supported(x.shape) and len(net.packed_wino43) == 9)    # ... and takes this shape
        got = net(x).cpu().numpy()
        err = _rel(got, ref)
        print("fp32 %s trunk at 224x224 vs float64: max error / max |ref| = %.3e" % (algo, err))
        assert err < bound, (algo, err)


def test_vgg_trunk_bf16_fullsize_matches_bf16_oracle(cuda):
    """Config 5's bf16 trunk on a 224x224 frame vs the bf16-emulating oracle (same roundings, float64 sums)."""
    from ntmtrack import vgg
    from oracle import ntm_oracle_torch as OT
    rng = np.random.default_rng(12)
    ws = O.init_vgg_weights(rng)
    frames = (rng.uniform(0, 255, size=(1, 224, 224, 3)).astype(np.float32) - O.VGG_MEAN)
    ref = OT.vgg16_conv43_bf16(frames, ws)
    ref32 = OT.vgg16_conv43(frames.astype(np.float64), {k: (w.astype(np.float64), b.astype(np.float64)) for k, (w, b) in ws.items()})
    got = vgg.VGG16Conv43(ws, device=cuda, dtype="bf16")(torch.from_numpy(frames).to(cuda)).cpu().numpy()
    scale = np.max(np.abs(ref))
    e_max, e_mean = np.max(np.abs(got - ref)) / scale, np.mean(np.abs(got - ref)) / scale
    print("bf16 trunk at 224x224 vs bf16 oracle: max %.3e mean %.3e; vs fp64 trunk: %.3e" % (e_max, e_mean, _rel(got, ref32)))
    assert e_max < 2e-2 and e_mean < 1e-3
    assert _rel(got, ref32) < 3e-2


@pytest.mark.parametrize("dtype,algo", [("f32", "split3"), ("f32", "winograd"), ("bf16", None)])
def test_vgg_trunk_is_frame_invariant_across_chunk_boundaries(cuda, dtype, algo):
    """BASELINE configs[3] / [4] push 1920 / 3200 frames per step through `VGG16Conv43.__call__`, i.e. through its
    `F > chunk_frames` loop (two or more chunks, each split over the stream parts).  Here: 160 frames of 224x224 with
    chunk_frames = 64 -- chunks of 64, 64 and 32 frames, the first two in two stream parts, the last in one -- against
    ONE-FRAME launches of the same network: bit-identical, frame by frame, for the fp32 split-form trunk, the F(4x4) trunk and the bf16 trunk
    (a frame's result may not depend on its position in a chunk, on the chunk's size or on the stream it ran on)."""
    from ntmtrack import vgg
    rng = np.random.default_rng(21)
    ws = O.init_vgg_weights(rng)
    for k in ws:
        ws[k] = (ws[k][0], (rng.standard_normal(ws[k][1].shape) * 0.05).astype(np.float32))
    F = 160
    base = (rng.uniform(0, 255, size=(8, 224, 224, 3)).astype(np.float32) - O.VGG_MEAN)
    # 160 different frames from 8 random ones: frame f = base[f % 8] rolled by f // 8 pixels (cheap on the host, no two alike)
    frames = torch.from_numpy(base).to(cuda)
    frames = torch.stack([torch.roll(frames[f % 8], shifts=f // 8, dims=1) for f in range(F)]).contiguous()
    net = vgg.VGG16Conv43(ws, device=cuda, dtype=dtype, chunk_frames=64, algo=algo)
    got = net(frames)
    torch.cuda.synchronize()
    assert got.shape == (F, 28, 28, 512) and got.dtype == torch.float32
    one = vgg.VGG16Conv43(ws, device=cuda, dtype=dtype, algo=algo)
    for f in range(F):
        ref = one(frames[f:f + 1])
        assert torch.equal(got[f:f + 1], ref), "frame %d (chunk %d, position %d) differs from its one-frame launch" % (f, f // 64, f % 64)
    # and the whole batch in one chunk (what the 640-frame bench pass does)
    assert torch.equal(vgg.VGG16Conv43(ws, device=cuda, dtype=dtype, chunk_frames=1024, algo=algo)(frames), got)


def test_blocked_trunk_layers_equal_nhwc_layers_bit_for_bit(cuda):
    """The channel-blocked activation layout [H][C/8][W][8] between the layers of the F(4x4) trunk changes addresses only: every
    tile-block shape of the eight-wave kernel (8x4x1, 4x4x2, 2x2x8, 1x1x32; with and without the fused pool; blocked or NHWC on
    either side) gives exactly the bits of the NHWC call, and the whole 224x224 trunk (which is blocked by default) equals the
    NHWC trunk bit for bit -- so every parity bound measured on one holds for the other."""
    from ntmtrack import vgg
    rng = np.random.default_rng(5)
    # the F(4x4) kernel, one case per tile-block shape
    for F, H, W, cin, cout, pool in ((2, 16, 32, 64, 64, True), (3, 16, 16, 64, 128, False), (5, 8, 8, 128, 256, True),
                                     (3, 28, 28, 256, 512, False), (40, 4, 4, 32, 64, False)):
        w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        x = torch.from_numpy(np.maximum(rng.standard_normal((F, H, W, cin)), 0).astype(np.float32)).to(cuda)
        u, bt = vgg.pack_weights_wino43(torch.from_numpy(w).to(cuda)), torch.from_numpy(b).to(cuda)
        ref = vgg.conv3x3_relu_wino43(x, u, bt, cin, cout, fuse_pool=pool)
        xb = vgg.nhwc_to_blocked(x)
        yb = vgg.conv3x3_relu_wino43_blocked(xb, u, bt, cin, cout, fuse_pool=pool, out_blocked=True)
        yn = vgg.conv3x3_relu_wino43_blocked(xb, u, bt, cin, cout, fuse_pool=pool, out_blocked=False)
        ynb = vgg.conv3x3_relu_wino43_blocked(x, u, bt, cin, cout, fuse_pool=pool, out_blocked=True)       # NHWC in, blocked out
        torch.cuda.synchronize()
        assert yb.shape == (F, ref.shape[1], cout // 8, ref.shape[2], 8)
        assert torch.equal(yn, ref), (F, H, W, cin, cout, pool)
        assert torch.equal(vgg.blocked_to_nhwc(yb), ref) and torch.equal(ynb, yb), (F, H, W, cin, cout, pool)
        assert torch.equal(vgg.nhwc_to_blocked(ref), yb)
    # whole trunk
    ws = O.init_vgg_weights(rng)
    for k in ws:
        ws[k] = (ws[k][0], (rng.standard_normal(ws[k][1].shape) * 0.05).astype(np.float32))
    frames = torch.from_numpy(rng.uniform(0, 255, size=(3, 224, 224, 3)).astype(np.float32) - O.VGG_MEAN).to(cuda)
    net = vgg.VGG16Conv43(ws, device=cuda)
    assert net.layout == "blocked" and vgg.blocked_trunk_supported(3, 224, 224)
    yb = net(frames).clone()
    net.layout = "nhwc"
    assert torch.equal(net(frames), yb)
    assert not vgg.blocked_trunk_supported(1, 36, 36)          # falls back to NHWC where a layer is outside the kernel's shapes

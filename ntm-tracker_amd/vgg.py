"""VGG-16 conv1_1 .. conv4_3 feature extractor on the HIP MFMA conv kernel.

Replaces the frozen-GraphDef import of the reference
(direct_offset_output.py:417-422; layer spec vgg.py:155-161): ten
[conv3x3 SAME + bias + ReLU] layers with 2x2/2 max-pools after conv1_2,
conv2_2 and conv3_3 (fused into the producing conv's epilogue), NHWC fp32,
TF HWIO weights.  The extractor is frozen (constants in the reference), so it
is inference-only.
"""
import os
import torch

from . import _lib

# (name, Cin, Cout, pool_after) -- vgg.py:155-160
VGG_LAYERS = [
    ("conv1_1", 3, 64, False), ("conv1_2", 64, 64, True),
    ("conv2_1", 64, 128, False), ("conv2_2", 128, 128, True),
    ("conv3_1", 128, 256, False), ("conv3_2", 256, 256, False), ("conv3_3", 256, 256, True),
    ("conv4_1", 256, 512, False), ("conv4_2", 512, 512, False), ("conv4_3", 512, 512, False),
]

# algorithmic MACs per 224x224 frame: sum H*W*9*Cin*Cout
def conv_flops_per_frame(h=224, w=224):
    total = 0
    for _name, cin, cout, pool in VGG_LAYERS:
        total += 2 * h * w * 9 * cin * cout
        if pool:
            h //= 2
            w //= 2
    return total


def conv3x3_relu(x, w_packed, bias, cin, cout, fuse_pool=False, out=None):
    """x [F,H,W,Cin] fp32 NHWC device tensor -> relu(conv3x3_same(x)+b), optionally 2x2 max-pooled."""
    F, H, W, C = x.shape
    if C != cin:
        raise _lib.NtkError("conv3x3_relu: input has %d channels, layer expects %d" % (C, cin))
    oh, ow = (H // 2, W // 2) if fuse_pool else (H, W)
    if out is None:
        out = torch.empty((F, oh, ow, cout), device=x.device, dtype=torch.float32)
    L = _lib.lib()
    _lib.check(L.ntk_vgg_conv3x3_relu_f32(_lib.ptr(x), _lib.ptr(w_packed), _lib.ptr(bias), _lib.ptr(out),
                                          F, H, W, cin, cout, 1 if fuse_pool else 0, _lib.stream()),
               "ntk_vgg_conv3x3_relu_f32")
    return out


def pack_weights(w_hwio):
    """[3,3,Cin,Cout] (TF HWIO) device tensor -> kernel layout [Cout][Kp]."""
    kh, kw, cin, cout = w_hwio.shape
    assert kh == 3 and kw == 3
    L = _lib.lib()
    kp = L.ntk_vgg_packed_k(cin)
    wp = torch.empty((cout, kp), device=w_hwio.device, dtype=torch.float32)
    _lib.check(L.ntk_vgg_pack_weights(_lib.ptr(w_hwio.contiguous()), _lib.ptr(wp), cin, cout, _lib.stream()),
               "ntk_vgg_pack_weights")
    return wp


def wino_supported(cin, cout, H, W, frames=1):
    """Shapes the fused Winograd kernel takes (otherwise the direct kernel runs): channel multiples, H and W
    multiples of 4, and an input small enough for its 32-bit element offsets."""
    nCB = cout // 64
    return (cin % 16 == 0 and cout % 64 == 0 and nCB >= 1 and (8 % nCB == 0 if nCB <= 8 else nCB % 8 == 0)
            and H % 4 == 0 and W % 4 == 0 and frames * H * W * cin < 0xffffffff)


def pack_weights_wino(w_hwio):
    """[3,3,Cin,Cout] (TF HWIO) -> Winograd-domain weights U = G g G^T packed for the MFMA B operand."""
    kh, kw, cin, cout = w_hwio.shape
    assert kh == 3 and kw == 3
    L = _lib.lib()
    u = torch.empty(L.ntk_vgg_wino_packed_floats(cin, cout), device=w_hwio.device, dtype=torch.float32)
    w = w_hwio.contiguous()
    _lib.check(L.ntk_vgg_pack_weights_wino(_lib.ptr(w), _lib.ptr(u), cin, cout, _lib.stream()), "ntk_vgg_pack_weights_wino")
    return u


def conv3x3_relu_wino(x, u_packed, bias, cin, cout, fuse_pool=False, out=None):
    """Same operator as conv3x3_relu by fused Winograd F(2x2,3x3) (csrc/conv_wino.hip)."""
    F, H, W, C = x.shape
    if C != cin:
        raise _lib.NtkError("conv3x3_relu_wino: input has %d channels, layer expects %d" % (C, cin))
    oh, ow = (H // 2, W // 2) if fuse_pool else (H, W)
    if out is None:
        out = torch.empty((F, oh, ow, cout), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().ntk_vgg_conv3x3_relu_wino_f32(_lib.ptr(x), _lib.ptr(u_packed), _lib.ptr(bias), _lib.ptr(out),
                                                        F, H, W, cin, cout, 1 if fuse_pool else 0, _lib.stream()),
               "ntk_vgg_conv3x3_relu_wino_f32")
    return out


def wino43_supported(cin, cout, H, W, frames=1):
    """Shapes the fused Winograd F(4x4,3x3) kernel takes."""
    nCB = cout // 64
    return (cin % 16 == 0 and cout % 64 == 0 and nCB >= 1 and (8 % nCB == 0 if nCB <= 8 else nCB % 8 == 0)
            and H % 4 == 0 and W % 4 == 0 and 2 * H * W * cin * 4 <= 0x40000000)


def pack_weights_wino43(w_hwio):
    """[3,3,Cin,Cout] (TF HWIO) -> the 36 Winograd-domain planes U = G g G^T of F(4x4,3x3), packed for the MFMA B operand."""
    kh, kw, cin, cout = w_hwio.shape
    assert kh == 3 and kw == 3
    L = _lib.lib()
    u = torch.empty(L.ntk_vgg_wino43_packed_floats(cin, cout), device=w_hwio.device, dtype=torch.float32)
    w = w_hwio.contiguous()
    _lib.check(L.ntk_vgg_pack_weights_wino43(_lib.ptr(w), _lib.ptr(u), cin, cout, _lib.stream()), "ntk_vgg_pack_weights_wino43")
    return u


def conv3x3_relu_wino43(x, u_packed, bias, cin, cout, fuse_pool=False, out=None, window=None, waves=None):
    """Same operator as conv3x3_relu by fused Winograd F(4x4,3x3) (csrc/conv_wino43.hip).
    window = (y0, x0, y1, x1), multiples of 4: compute only that part of the un-pooled output (the rest of `out` is not touched).
    waves = 4 / 8: the one- / two-waves-per-SIMD form of the kernel (default: 8; same bits)."""
    F, H, W, C = x.shape
    if C != cin:
        raise _lib.NtkError("conv3x3_relu_wino43: input has %d channels, layer expects %d" % (C, cin))
    oh, ow = (H // 2, W // 2) if fuse_pool else (H, W)
    if out is None:
        out = torch.empty((F, oh, ow, cout), device=x.device, dtype=torch.float32)
    if waves is not None:
        y0, x0, y1, x1 = [int(v) for v in window] if window is not None else (0, 0, H, W)
        _lib.check(_lib.lib().ntk_vgg_conv3x3_relu_wino43_form_f32(_lib.ptr(x), _lib.ptr(u_packed), _lib.ptr(bias), _lib.ptr(out),
                                                                   F, H, W, cin, cout, 1 if fuse_pool else 0, y0, x0, y1, x1,
                                                                   int(waves), _lib.stream()), "ntk_vgg_conv3x3_relu_wino43_form_f32")
        return out
    if window is not None:
        y0, x0, y1, x1 = [int(v) for v in window]
        _lib.check(_lib.lib().ntk_vgg_conv3x3_relu_wino43_window_f32(_lib.ptr(x), _lib.ptr(u_packed), _lib.ptr(bias), _lib.ptr(out),
                                                                     F, H, W, cin, cout, 1 if fuse_pool else 0, y0, x0, y1, x1,
                                                                     _lib.stream()), "ntk_vgg_conv3x3_relu_wino43_window_f32")
        return out
    _lib.check(_lib.lib().ntk_vgg_conv3x3_relu_wino43_f32(_lib.ptr(x), _lib.ptr(u_packed), _lib.ptr(bias), _lib.ptr(out),
                                                          F, H, W, cin, cout, 1 if fuse_pool else 0, _lib.stream()),
               "ntk_vgg_conv3x3_relu_wino43_f32")
    return out


def _wino43_shape(H, W):
    """Tile-block shape the F(4x4) kernel picks for a frame (csrc/conv_wino43.hip, wino43_launch): 0 = 8x4x1, 1 = 4x4x2, 2 = 2x2x8, 3 = 1x1x32."""
    gw, gh = W // 4, H // 4
    for i, (tw, th) in enumerate(((8, 4), (4, 4), (2, 2))):
        if gw % tw == 0 and gh % th == 0:
            return i
    return 3


def blocked_trunk_supported(frames, H, W):
    """Can conv1_2 .. conv4_3 of a [frames,H,W,3] batch hand channel-blocked maps [H][C/8][W][8] to each other?  Every one of them
    must be a shape of the eight-wave F(4x4) kernel (a 1x1x32-block layer only while its blocks span less than 16 MB of input:
    there the slot words are packed)."""
    if H % 32 or W % 32 or H < 32 or W < 32:
        return False
    h, w = H, W
    for _name, cin, cout, pool in VGG_LAYERS[1:]:
        if not wino43_supported(cin, cout, h, w, frames):
            return False
        if _wino43_shape(h, w) == 3:
            tpf = (w // 4) * (h // 4)
            if ((32 + tpf - 1) // tpf + 1) * h * w * cin * 4 > 0xfffff0:
                return False
        if pool:
            h //= 2
            w //= 2
    return True


def conv3x3_relu_wino43_blocked(x, u_packed, bias, cin, cout, fuse_pool=False, out_blocked=True, out=None):
    """conv3x3_relu_wino43 with channel-blocked maps: the input is blocked [F,H,cin/8,W,8] (5-D) or NHWC [F,H,W,cin] (4-D); the
    output blocked [F,Ho,cout/8,Wo,8] or NHWC [F,Ho,Wo,cout].  Same arithmetic and bits as the NHWC call (csrc/conv_wino43.hip)."""
    in_blocked = x.dim() == 5
    if in_blocked:
        F, H, CB, W, E = x.shape
        if CB * 8 != cin or E != 8:
            raise _lib.NtkError("conv3x3_relu_wino43_blocked: input is %s, layer expects %d channels in blocks of 8" % (tuple(x.shape), cin))
    else:
        F, H, W, C = x.shape
        if C != cin:
            raise _lib.NtkError("conv3x3_relu_wino43_blocked: input has %d channels, layer expects %d" % (C, cin))
    oh, ow = (H // 2, W // 2) if fuse_pool else (H, W)
    if out is None:
        out = torch.empty((F, oh, cout // 8, ow, 8) if out_blocked else (F, oh, ow, cout), device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().ntk_vgg_conv3x3_relu_wino43_layout_f32(_lib.ptr(x), _lib.ptr(u_packed), _lib.ptr(bias), _lib.ptr(out),
                                                                 F, H, W, cin, cout, 1 if fuse_pool else 0, 1 if in_blocked else 0,
                                                                 1 if out_blocked else 0,
                                                                 _lib.stream()), "ntk_vgg_conv3x3_relu_wino43_layout_f32")
    return out


def nhwc_to_blocked(x):
    """[F,H,W,C] -> [F,H,C/8,W,8] (host-side helper for tests and step-wise callers)."""
    F, H, W, C = x.shape
    return x.view(F, H, W, C // 8, 8).permute(0, 1, 3, 2, 4).contiguous()


def blocked_to_nhwc(x):
    F, H, CB, W, E = x.shape
    return x.permute(0, 1, 3, 2, 4).reshape(F, H, W, CB * E).contiguous()


def conv3x3_relu_bf16(x, w_packed, bias, cin, cout, fuse_pool=False, out_f32=False, out=None):
    """bf16 NHWC activations [F,H,W,Cin] -> relu(conv3x3_same(x)+b) in bf16 (or fp32 when out_f32)."""
    F, H, W, C = x.shape
    if C != cin or x.dtype != torch.bfloat16:
        raise _lib.NtkError("conv3x3_relu_bf16: expected bf16 input with %d channels" % cin)
    oh, ow = (H // 2, W // 2) if fuse_pool else (H, W)
    if out is None:
        out = torch.empty((F, oh, ow, cout), device=x.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    _lib.check(_lib.lib().ntk_vgg_conv3x3_relu_bf16(_lib.ptr(x), _lib.ptr(w_packed), _lib.ptr(bias), _lib.ptr(out), F, H, W,
                                                    cin, cout, 1 if fuse_pool else 0, 1 if out_f32 else 0, _lib.stream()),
               "ntk_vgg_conv3x3_relu_bf16")
    return out


def conv3x3_relu_bf16p(x, w_packed, bias, cin, cout, fuse_pool=False, out_f32=False, out=None):
    """conv3x3_relu_bf16 in patch form (csrc/conv_bf16p.hip); w_packed from pack_weights_bf16p for THIS frame shape."""
    F, H, W, C = x.shape
    if C != cin or x.dtype != torch.bfloat16:
        raise _lib.NtkError("conv3x3_relu_bf16p: expected bf16 input with %d channels" % cin)
    oh, ow = (H // 2, W // 2) if fuse_pool else (H, W)
    if out is None:
        out = torch.empty((F, oh, ow, cout), device=x.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    _lib.check(_lib.lib().ntk_vgg_conv3x3_relu_bf16p(_lib.ptr(x), _lib.ptr(w_packed), _lib.ptr(bias), _lib.ptr(out), F, H, W,
                                                     cin, cout, 1 if fuse_pool else 0, 1 if out_f32 else 0, _lib.stream()),
               "ntk_vgg_conv3x3_relu_bf16p")
    return out


def pack_weights_bf16p(w_hwio, H, W):
    kh, kw, cin, cout = w_hwio.shape
    wp = torch.empty(_lib.lib().ntk_vgg_bf16p_packed_elems(cin, cout), device=w_hwio.device, dtype=torch.bfloat16)
    _lib.check(_lib.lib().ntk_vgg_pack_weights_bf16p(_lib.ptr(w_hwio.contiguous()), _lib.ptr(wp), cin, cout, H, W, _lib.stream()),
               "ntk_vgg_pack_weights_bf16p")
    return wp


# ---- the SPLIT form of the fp32 trunk (csrc/conv_bf16p.hip, template flag X3): every fp32 value travels as two fp16 numbers
def to_split(x):
    """fp32 NHWC [F,H,W,C] (C a multiple of 16) -> split map [F,H,W,C/16,2,16] fp16, as the kernels' epilogues write it
    (csrc/conv_bf16p.hip s3_split4): hi = fp16(x) rounded toward zero (saturating at 65504), lo = fp16(x - hi)."""
    F, H, W, C = x.shape
    t = x.clamp(-65504.0, 65504.0)
    hi = t.to(torch.float16)                                  # round to nearest ...
    over = hi.to(torch.float32).abs() > t.abs()               # ... stepped back where that rounded away from zero
    hi_i = hi.view(torch.int16)
    hi = torch.where(over, hi_i - 1, hi_i).view(torch.float16)    # (sign-magnitude: one ulp toward zero is bits - 1)
    lo = (x - hi.to(torch.float32)).clamp(-65504.0, 65504.0).to(torch.float16)
    return torch.stack((hi.view(F, H, W, C // 16, 16), lo.view(F, H, W, C // 16, 16)), dim=4).contiguous()


def from_split(s):
    """split map -> fp32 NHWC (hi + lo)."""
    F, H, W, G = s.shape[:4]
    return (s[:, :, :, :, 0].to(torch.float32) + s[:, :, :, :, 1].to(torch.float32)).reshape(F, H, W, G * 16)


def split3_supported(H, W, cin, cout, fuse_pool=False):
    return bool(_lib.lib().ntk_vgg_split3_supported(H, W, cin, cout, 1 if fuse_pool else 0))


def pack_weights_split3(w_hwio, H, W):
    kh, kw, cin, cout = w_hwio.shape
    wp = torch.empty(_lib.lib().ntk_vgg_split3_packed_elems(cin, cout), device=w_hwio.device, dtype=torch.float16)
    _lib.check(_lib.lib().ntk_vgg_pack_weights_split3(_lib.ptr(w_hwio.contiguous()), _lib.ptr(wp), cin, cout, H, W, _lib.stream()),
               "ntk_vgg_pack_weights_split3")
    return wp


def conv3x3_relu_split3(x, w_packed, bias, cin, cout, fuse_pool=False, out_f32=False, out=None):
    """split map [F,H,W,Cin/16,2,16] (or an fp32 NHWC map [F,H,W,Cin]: the kernel's staging splits it; cin <= 64 and cout == 64 only)
    -> relu(conv3x3_same(x) + b) as a split map (or fp32 NHWC when out_f32): the fp32 product x w accumulated as
    xh wh + xh wl + xl wh on the 16-bit matrix pipe (fp16 parts, fp32 accumulators)."""
    F, H, W = x.shape[:3]
    in_f32 = x.dtype == torch.float32
    if in_f32:
        if x.dim() != 4 or x.shape[3] != cin:
            raise _lib.NtkError("conv3x3_relu_split3: expected an fp32 NHWC map of %d channels" % cin)
    elif x.dim() != 6 or x.shape[3] * 16 != cin or x.dtype != torch.float16 or tuple(x.shape[4:]) != (2, 16):
        raise _lib.NtkError("conv3x3_relu_split3: expected a split map of %d channels" % cin)
    oh, ow = (H // 2, W // 2) if fuse_pool else (H, W)
    if out is None:
        out = (torch.empty((F, oh, ow, cout), device=x.device, dtype=torch.float32) if out_f32
               else torch.empty((F, oh, ow, cout // 16, 2, 16), device=x.device, dtype=torch.float16))
    _lib.check(_lib.lib().ntk_vgg_conv3x3_relu_split3(_lib.ptr(x), _lib.ptr(w_packed), _lib.ptr(bias), _lib.ptr(out), F, H, W,
                                                      cin, cout, 1 if fuse_pool else 0, 1 if in_f32 else 0, 1 if out_f32 else 0,
                                                      _lib.stream()), "ntk_vgg_conv3x3_relu_split3")
    return out


def pack_weights_bf16(w_hwio):
    kh, kw, cin, cout = w_hwio.shape
    wp = torch.empty((cout, 9 * cin), device=w_hwio.device, dtype=torch.bfloat16)
    _lib.check(_lib.lib().ntk_vgg_pack_weights_bf16(_lib.ptr(w_hwio.contiguous()), _lib.ptr(wp), cin, cout, _lib.stream()),
               "ntk_vgg_pack_weights_bf16")
    return wp


class VGG16Conv43(object):
    """Frozen VGG-16 trunk up to conv4_3/Relu.

    weights: {layer_name: (w_hwio [3,3,Cin,Cout], b [Cout])} as numpy arrays or tensors.
    dtype "f32" (BASELINE configs 2-4: fp32 values and accumulators; `algo` picks the form -- None / "split3": conv1_2 .. conv4_3
    as three fp16 MFMA products per fp32 product of hi / lo parts (DESIGN.md 4.0''; error at or below the Winograd form's),
    "winograd" / "winograd2": fused Winograd on the fp32 MFMA pipe, "direct": the implicit-GEMM kernel; NTK_TRUNK_ALGO overrides
    the default) or "bf16" (config 5: bf16 operands, fp32 accumulate; conv1_1 reads the fp32 frames, conv4_3 writes fp32 for the
    memory cell).
    """

    def __init__(self, weights, device="cuda", chunk_frames=1024, dtype="f32", algo=None):
        self.device = torch.device(device)
        self.chunk_frames = int(chunk_frames)
        if dtype not in ("f32", "bf16"):
            raise _lib.NtkError("VGG16Conv43: dtype must be 'f32' or 'bf16'")
        if algo is None:                                        # the fp32 trunk's default form (NTK_TRUNK_ALGO overrides)
            algo = os.environ.get("NTK_TRUNK_ALGO", "split3")
        if algo not in ("split3", "winograd", "winograd2", "direct"):
            raise _lib.NtkError("VGG16Conv43: algo must be 'split3', 'winograd', 'winograd2' or 'direct'")
        self.dtype = dtype
        # (y0, x0, y1, x1) in conv4_3 output pixels, multiples of 4, or None: compute conv4_3 only there (F(4x4) fp32 trunk).  The
        # tracker's extract_features reads 64 fixed points of the 28x28 map (rows / columns 6..20): window (4, 4, 24, 24) = 25
        # of its 49 tiles.  Positions outside the window keep whatever the buffer held (pass a zeroed `out`).  Off by default:
        # the benchmark computes the whole map, as the reference graph does.
        self.features_window = None
        # parts / streams of a trunk pass (see __call__): two for the default F(4x4) fp32 trunk; measured a loss for the direct
        # kernels (149 -> 173 ms per step) and for round 2's bf16 tile kernel (59.5 -> 59.9), no change for F(2x2); the bf16 patch-form
        # kernel (one workgroup per CU, like the F(4x4) kernel) gains 2 % (640 frames: 18.07 -> 17.67 ms)
        self.split_streams = int(os.environ.get("NTK_TRUNK_SPLIT", "2" if ((dtype == "f32" and algo in ("winograd", "split3")) or dtype == "bf16") else "1"))
        self._side = []
        # fp32 trunk: "winograd" = fused Winograd F(4x4,3x3) wherever the layer shape allows (conv1_2 .. conv4_3 on
        # 224x224 frames; 4x fewer multiplies than the direct form, error ~1e-5 of the activation scale per layer),
        # "winograd2" = fused Winograd F(2x2,3x3) (2.25x fewer multiplies, error ~3e-7 per layer); the direct
        # implicit-GEMM kernel runs where neither applies (conv1_1, odd frame sizes) and everywhere with "direct"
        # "split3" = the SPLIT form (csrc/conv_bf16p.hip X3: every fp32 product as three fp16 MFMA products of hi / lo parts, fp32
        # accumulators) on conv1_2 .. split3_upto,
        # the F(4x4) Winograd kernel on the layers after it (on the 28 x 28 maps of conv4_x the two are within 2 %).  conv1_1 keeps its
        # fp32 kernel (conv1_2's staging splits its map), the last split layer writes fp32 NHWC for the Winograd layers.
        self.split3 = (algo == "split3" and dtype == "f32")
        self.split3_upto = os.environ.get("NTK_SPLIT3_UPTO", "conv4_3")
        # __call__(..., latency=True) runs a call of fewer frames than this in the Winograd form: with a handful of frames a layer is
        # a few workgroups per CU at most and the pass is bound by one workgroup's critical path, which is shorter in the F(4x4)
        # kernel (224 x 224, one MI355X: 1 frame 0.78 against 0.96 ms, 8 frames 1.00 against 1.09, 12 frames 1.39 against 1.27,
        # 16 frames 1.57 against 1.44 -- scripts/r04/small_batch_trunk.py).  What the online tracker asks for (one frame per call).
        # Without the flag every call runs the same form, so a frame's features do not depend on the size of the batch it is in
        # (bit for bit: tests/test_fullsize_gpu.py).
        self.split3_latency_frames = 12
        self._call_split3 = True
        self._packed_split3 = {}
        if self.split3:
            algo = "winograd"                                # everything else (weights packed, layouts, fallbacks) as the Winograd trunk
        self.algo = algo
        # form of the F(4x4) kernel: None = the library's default (eight waves per workgroup); 4 = round 2's one-wave-per-SIMD kernel
        self.wino_waves = None
        # activation layout BETWEEN the layers of the F(4x4) fp32 trunk: "blocked" = [H][C/8][W][8] (a K step's eight channels
        # contiguous per pixel, blocks interleaved per image row: the patch staging reads whole cache lines), "nhwc" = round 3's.
        # Blocked is the default wherever the eight-wave kernel takes every layer (blocked_trunk_supported); frames, conv1_1's
        # output and conv4_3's output are NHWC either way; same bits.
        self.layout = os.environ.get("NTK_TRUNK_LAYOUT", "blocked")
        self._blocked_ws = {}
        self.packed = {}
        self._w_hwio = {}
        self._packed_bf16p = {}
        # bf16 trunk: "patch" = csrc/conv_bf16p.hip wherever it takes the layer shape (round 4), "tile" = round 2's kernel
        self.bf16_form = os.environ.get("NTK_BF16_FORM", "patch")
        self.packed_wino = {}
        self.packed_wino43 = {}
        for name, cin, cout, _pool in VGG_LAYERS:
            w, b = weights[name]
            w = torch.as_tensor(w, dtype=torch.float32).to(self.device)
            b = torch.as_tensor(b, dtype=torch.float32).to(self.device).contiguous()
            if tuple(w.shape) != (3, 3, cin, cout):
                raise _lib.NtkError("%s: weight shape %s != (3,3,%d,%d)" % (name, tuple(w.shape), cin, cout))
            if self.split3 and dtype == "f32" and cin % 16 == 0:
                self._w_hwio[name] = w                     # packed per frame shape on first use
            if dtype == "bf16" and cin % 64 == 0:
                self.packed[name] = (pack_weights_bf16(w), b)
                self._w_hwio[name] = w                     # the patch-form kernel packs per frame shape, on first use
            else:
                self.packed[name] = (pack_weights(w), b)
                if dtype == "f32" and algo in ("winograd", "winograd2") and cin % 16 == 0:
                    self.packed_wino[name] = pack_weights_wino(w)
                    if algo == "winograd":
                        self.packed_wino43[name] = pack_weights_wino43(w)

    def _forward_chunk_bf16(self, frames, out=None):
        F, H, W, _ = frames.shape
        L = _lib.lib()
        wp, b = self.packed["conv1_1"]
        x = torch.empty((F, H, W, 64), device=frames.device, dtype=torch.bfloat16)
        _lib.check(L.ntk_vgg_conv3x3_relu_f32_to_bf16(_lib.ptr(frames), _lib.ptr(wp), _lib.ptr(b), _lib.ptr(x), F, H, W, 3, 64,
                                                      _lib.stream()), "ntk_vgg_conv3x3_relu_f32_to_bf16")
        for name, cin, cout, pool in VGG_LAYERS[1:]:
            wp, b = self.packed[name]
            last = (name == "conv4_3")
            h, w = x.shape[1], x.shape[2]
            if self.bf16_form == "patch" and L.ntk_vgg_bf16p_supported(h, w, cin, cout, 1 if pool else 0):
                key = (name, h % 8 == 0 and w % 8 == 0)
                if key not in self._packed_bf16p:
                    self._packed_bf16p[key] = pack_weights_bf16p(self._w_hwio[name], h, w)
                    torch.cuda.current_stream(frames.device).synchronize()    # packed once, read from every stream a pass runs on
                x = conv3x3_relu_bf16p(x, self._packed_bf16p[key], b, cin, cout, fuse_pool=pool, out_f32=last, out=out if last else None)
            else:
                x = conv3x3_relu_bf16(x, wp, b, cin, cout, fuse_pool=pool, out_f32=last, out=out if last else None)
        return x

    def _trunk_ws(self, frames):
        """two ping-pong workspaces per (stream, chunk shape) for the maps between the layers, allocated once: a training loop then
        makes no allocator calls in its trunk passes (the largest map, conv1_1's, is 12.8 MB per frame)"""
        F, H, W, _ = frames.shape
        key = (torch.cuda.current_stream(frames.device).cuda_stream, F, H, W)
        ws = self._blocked_ws.get(key)
        if ws is None:
            if len(self._blocked_ws) >= 8:                        # shapes come and go (tests, online tracking): keep the table small
                self._blocked_ws.clear()
            ws = (torch.empty(F * H * W * 64, device=frames.device), torch.empty(F * (H // 2) * (W // 2) * 64, device=frames.device))
            self._blocked_ws[key] = ws
        return ws

    def split3_trunk_supported(self, frames_shape):
        F, H, W = frames_shape[:3]
        return (self.wino_waves in (None, 8) and self.features_window is None and blocked_trunk_supported(F, H, W)
                and split3_supported(H, W, 64, 64, True))

    def _forward_chunk_split3(self, frames, out):
        """conv1_1 (fp32 kernel) -> conv1_2 .. split3_upto in the split form -> the rest on the F(4x4) kernel with blocked maps."""
        F, H, W, _ = frames.shape
        ws = self._trunk_ws(frames)
        wp, b = self.packed["conv1_1"]
        x = conv3x3_relu(frames, wp, b, 3, 64, out=ws[0][:F * H * W * 64].view(F, H, W, 64))
        names = [l[0] for l in VGG_LAYERS]
        n_split = names.index(self.split3_upto) if self.split3_upto in names else 0
        mode = "nhwc"                                            # what x is: fp32 "nhwc", a "split" map, or a "blocked" fp32 map
        h, w = H, W
        for li, (name, cin, cout, pool) in enumerate(VGG_LAYERS[1:]):
            last = (name == "conv4_3")
            oh, ow = (h // 2, w // 2) if pool else (h, w)
            buf = ws[(li + 1) & 1][:F * oh * ow * cout]
            use_split = (li + 1 <= n_split and mode != "blocked" and split3_supported(h, w, cin, cout, pool)
                         and (mode == "split" or (cin <= 64 and cout == 64)))
            if use_split:
                nxt = VGG_LAYERS[li + 2] if not last else None
                nh, nw_ = oh, ow
                stay = (nxt is not None and li + 2 <= n_split and split3_supported(nh, nw_, nxt[1], nxt[2], nxt[3]))
                key = (name, h, w)
                if key not in self._packed_split3:
                    self._packed_split3[key] = pack_weights_split3(self._w_hwio[name], h, w)
                    torch.cuda.current_stream(frames.device).synchronize()    # packed once, read from every stream a pass runs on
                dst = out if last else (buf.view(torch.float16).view(F, oh, ow, cout // 16, 2, 16) if stay else buf.view(F, oh, ow, cout))
                x = conv3x3_relu_split3(x, self._packed_split3[key], self.packed[name][1], cin, cout, fuse_pool=pool,
                                        out_f32=(last or not stay), out=dst)
                mode = "split" if stay else "nhwc"
            else:
                if mode == "split":
                    raise _lib.NtkError("split3 trunk: %s cannot read a split map" % name)
                dst = out if last else buf.view(F, oh, cout // 8, ow, 8)
                x = conv3x3_relu_wino43_blocked(x, self.packed_wino43[name], self.packed[name][1], cin, cout, fuse_pool=pool,
                                                out_blocked=not last, out=dst)
                mode = "blocked"
            h, w = oh, ow
        return x

    def forward_chunk(self, frames, upto="conv4_3", out=None):
        if self.dtype == "bf16":
            if upto != "conv4_3":
                raise _lib.NtkError("bf16 trunk runs to conv4_3 only")
            return self._forward_chunk_bf16(frames.contiguous(), out=out)
        x = frames
        if self.split3 and self._call_split3 and upto == "conv4_3" and self.split3_trunk_supported(frames.shape):
            if out is None:
                out = torch.empty((frames.shape[0], frames.shape[1] // 8, frames.shape[2] // 8, 512), device=frames.device)
            return self._forward_chunk_split3(frames.contiguous(), out)
        if (self.layout == "blocked" and self.algo == "winograd" and self.wino_waves in (None, 8) and self.features_window is None
                and upto == "conv4_3" and blocked_trunk_supported(*frames.shape[:3])):
            # the maps between the layers live in two ping-pong workspaces per (stream, chunk shape), allocated once: a training
            # loop then makes no allocator calls in its trunk passes (the largest map, conv1_1's, is 12.8 MB per frame)
            F, H, W, _ = frames.shape
            ws = self._trunk_ws(frames)
            wp, b = self.packed["conv1_1"]
            x = conv3x3_relu(frames, wp, b, 3, 64, out=ws[0][:F * H * W * 64].view(F, H, W, 64))   # NHWC: conv1_2 reads it as it is
            h, w = H, W
            for li, (name, cin, cout, pool) in enumerate(VGG_LAYERS[1:]):
                last = (name == "conv4_3")
                oh, ow = (h // 2, w // 2) if pool else (h, w)
                dst = out if last else ws[(li + 1) & 1][:F * oh * ow * cout].view(F, oh, cout // 8, ow, 8)
                x = conv3x3_relu_wino43_blocked(x, self.packed_wino43[name], self.packed[name][1], cin, cout, fuse_pool=pool,
                                                out_blocked=not last, out=dst)
                h, w = oh, ow
            return x
        for name, cin, cout, pool in VGG_LAYERS:
            wp, b = self.packed[name]
            last = (name == upto)
            if name in self.packed_wino43 and wino43_supported(cin, cout, x.shape[1], x.shape[2], x.shape[0]):
                win = self.features_window if (last and name == "conv4_3") else None
                if win is not None and out is None:
                    out = torch.zeros((x.shape[0], x.shape[1], x.shape[2], cout), device=x.device, dtype=torch.float32)
                x = conv3x3_relu_wino43(x, self.packed_wino43[name], b, cin, cout, fuse_pool=(pool and not last),
                                        out=out if last else None, window=win, waves=self.wino_waves)
            elif name in self.packed_wino and wino_supported(cin, cout, x.shape[1], x.shape[2], x.shape[0]):
                x = conv3x3_relu_wino(x, self.packed_wino[name], b, cin, cout, fuse_pool=(pool and not last),
                                      out=out if last else None)
            else:
                x = conv3x3_relu(x, wp, b, cin, cout, fuse_pool=(pool and not last), out=out if last else None)
            if last:
                break
        return x

    def __call__(self, frames, out=None, latency=False):
        """frames [F,224,224,3] mean-subtracted fp32 NHWC -> [F,28,28,512].  latency=True: a call of a few frames may run the form with
        the shorter critical path (split3_latency_frames) -- same operator, results equal to fp32 rounding, not bit for bit."""
        if frames.dim() != 4 or frames.shape[3] != 3:
            raise _lib.NtkError("frames must be [F,H,W,3] NHWC")
        F, H, W, _ = frames.shape
        if out is None:
            # with a features_window the window kernel leaves everything outside the window untouched: the map a caller
            # gets back must be zero there, not uninitialised memory
            alloc = torch.zeros if getattr(self, "features_window", None) is not None else torch.empty
            out = alloc((F, H // 8, W // 8, 512), device=frames.device, dtype=torch.float32)
        self._call_split3 = not (latency and F < self.split3_latency_frames)
        try:
            return self._run_chunks(frames, out)
        finally:
            self._call_split3 = True

    def _run_chunks(self, frames, out):
        F = frames.shape[0]
        for f0 in range(0, F, self.chunk_frames):
            f1 = min(F, f0 + self.chunk_frames)
            n = self.split_streams if (f1 - f0) >= 32 * self.split_streams else 1
            if n <= 1:
                self.forward_chunk(frames[f0:f1], out=out[f0:f1])
                continue
            # the chunk in n parts on n streams: the kernels of the parts fill each other's launch tails (a layer's last
            # workgroups leave CUs idle until the next launch; frames are independent, the layers of one frame are not).
            # Measured, 640 frames: one stream 55.4 ms, two 54.5, three 54.4; whole step 64.45 -> 63.65 ms with two.
            cur = torch.cuda.current_stream(frames.device)
            while len(self._side) < n - 1:
                self._side.append(torch.cuda.Stream(device=frames.device, priority=cur.priority))
            cuts = [f0 + (f1 - f0) * i // n for i in range(n + 1)]
            for i in range(1, n):
                self._side[i - 1].wait_stream(cur)
                with torch.cuda.stream(self._side[i - 1]):
                    self.forward_chunk(frames[cuts[i]:cuts[i + 1]], out=out[cuts[i]:cuts[i + 1]])
            self.forward_chunk(frames[cuts[0]:cuts[1]], out=out[cuts[0]:cuts[1]])
            for i in range(1, n):
                cur.wait_stream(self._side[i - 1])
        return out

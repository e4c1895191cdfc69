"""CPU: the C-ABI library loads and exports every symbol include/ntmtrack.h declares
(no compute call is made without a GPU); host-side validation refuses bad shapes."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ntmtrack.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ntk_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("ntk_vgg_conv3x3_relu_f32", "ntk_ntm_seq_fwd", "ntk_ntm_seq_bwd", "ntk_gather_serialize",
                 "ntk_offset_loss", "ntk_rmsprop_clip_step", "ntk_version", "ntk_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from ntmtrack import _lib
    L = _lib.lib()                              # raises if the .so is missing: no silent fallback
    for s in declared_symbols():
        assert hasattr(L, s), "libntmtrack_hip.so does not export %s" % s
    assert L.ntk_version() >= 100
    # the binding's own table covers the same surface
    assert set(declared_symbols()) <= set(_lib.exported_symbols()) | {"ntk_version", "ntk_last_error"}


def test_host_side_validation_refuses_bad_arguments_without_a_gpu():
    from ntmtrack import _lib
    L = _lib.lib()
    # null pointers / bad shapes are rejected before any launch
    assert L.ntk_vgg_conv3x3_relu_f32(None, None, None, None, 1, 8, 8, 3, 64, 0, None) == -2
    assert L.ntk_gemm_nt_f32(None, 4, None, 4, None, None, 4, 1, 1, 4, None) == -2
    one = ctypes.c_void_p(16)                   # non-null, aligned, never dereferenced: shape check fires first
    assert L.ntk_vgg_conv3x3_relu_f32(one, one, one, one, 1, 6, 8, 32, 64, 0, None) == -1      # H % 4
    assert L.ntk_vgg_conv3x3_relu_f32(one, one, one, one, 1, 8, 8, 48, 64, 0, None) == -1      # Cin
    assert b"cin=48" in L.ntk_last_error()
    assert L.ntk_offset_loss(one, one, one, one, one, 2, 1, 64, 2, None) == -1                 # T < 2
    assert L.ntk_vgg_packed_k(3) == 32 and L.ntk_vgg_packed_k(64) == 576
    vals = [ctypes.c_int() for _ in range(5)]
    assert L.ntk_ntm_padded_dims(128, 20, 4, 1, 200, 1, 2, *[ctypes.byref(v) for v in vals]) == 0
    P, PP, K, ldz, ldh = [v.value for v in vals]
    assert (P, PP, K, ldz, ldh) == (170, 172, 280, 284, 204)     # P = 170 at R4/W1


def test_product_path_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ntmtrack import _lib, vgg
    with pytest.raises(_lib.NtkError):
        vgg.conv3x3_relu(torch.zeros((1, 8, 8, 32)), torch.zeros((64, 288)), torch.zeros(64), 32, 64)

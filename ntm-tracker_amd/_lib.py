"""ctypes binding to libntmtrack_hip.so (the C ABI declared in include/ntmtrack.h).

The product path has no CPU fallback: if the shared library is missing or a
call fails, an exception is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NTK_LIB_PATH: developer override (e.g. the diagnostic build `make -C csrc prof`); the product loads the in-tree library
LIB_PATH = os.environ.get("NTK_LIB_PATH") or os.path.join(_HERE, "libntmtrack_hip.so")

_lib = None


class NtkError(RuntimeError):
    pass


def _sig(lib):
    c_int, c_void_p, c_size_t = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
    P = c_void_p
    sigs = {
        "ntk_version": (c_int, []),
        "ntk_last_error": (ctypes.c_char_p, []),
        "ntk_vgg_packed_k": (c_int, [c_int]),
        "ntk_vgg_pack_weights": (c_int, [P, P, c_int, c_int, P]),
        "ntk_vgg_conv3x3_relu_f32": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
        "ntk_vgg_pack_weights_bf16": (c_int, [P, P, c_int, c_int, P]),
        "ntk_vgg_conv3x3_relu_bf16": (c_int, [P, P, P, P] + [c_int] * 7 + [P]),
        "ntk_vgg_conv3x3_relu_f32_to_bf16": (c_int, [P, P, P, P] + [c_int] * 5 + [P]),
        "ntk_vgg_bf16p_packed_elems": (c_size_t, [c_int, c_int]),
        "ntk_vgg_bf16p_supported": (c_int, [c_int] * 5),
        "ntk_vgg_pack_weights_bf16p": (c_int, [P, P] + [c_int] * 4 + [P]),
        "ntk_vgg_conv3x3_relu_bf16p": (c_int, [P, P, P, P] + [c_int] * 7 + [P]),
        "ntk_vgg_split3_packed_elems": (c_size_t, [c_int, c_int]),
        "ntk_vgg_split3_supported": (c_int, [c_int] * 5),
        "ntk_vgg_pack_weights_split3": (c_int, [P, P] + [c_int] * 4 + [P]),
        "ntk_vgg_conv3x3_relu_split3": (c_int, [P, P, P, P] + [c_int] * 8 + [P]),
        "ntk_gemm_nt_f32": (c_int, [P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, P]),
        "ntk_gemm_tn_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
        "ntk_gemm_tn_f32": (c_int, [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
        "ntk_ntm_padded_dims": (c_int, [c_int] * 7 + [ctypes.POINTER(c_int)] * 5),
        "ntk_ntm_seq_fwd": (c_int, [c_int] * 10 + [P] * 23 + [P]),
        "ntk_transpose_pad": (c_int, [P, c_int, P, c_int, c_int, c_int, P]),
        "ntk_ntm_seq_bwd": (c_int, [c_int] * 10 + [P, c_int, P, c_int] + [P] * 21 + [P]),
        "ntk_dnc_padded_dims": (c_int, [c_int] * 6 + [ctypes.POINTER(c_int)] * 8),
        "ntk_dnc_seq_fwd": (c_int, [c_int] * 8 + [ctypes.c_float] + [P] * 13 + [P] * 18 + [P]),
        "ntk_dnc_cluster_plan": (c_int, [c_int] * 8 + [ctypes.POINTER(c_int), ctypes.POINTER(c_size_t)]),
        "ntk_dnc_cluster_status": (c_int, [P, c_int, c_int, P]),
        "ntk_dnc_cluster_placement": (c_int, [P, c_int, c_int, P, P]),
        "ntk_dnc_cluster_guard": (c_int, [P, c_size_t, c_int, c_int, c_int, P, P, c_size_t, P]),
        "ntk_dnc_cluster_inject_abort": (c_int, [P, c_size_t, c_int, c_int, c_int, P]),
        "ntk_dnc_cluster_fwd": (c_int, [c_int] * 8 + [ctypes.c_float, c_int] + [P] * 13 + [P] * 18 + [P, P]),
        "ntk_dnc_cluster_bwd_plan": (c_int, [c_int] * 8 + [ctypes.POINTER(c_int), ctypes.POINTER(c_size_t)]),
        "ntk_dnc_cluster_bwd": (c_int, [c_int] * 8 + [ctypes.c_float, c_int] + [P, c_int, P, P] + [P] * 7 + [P] * 15 + [P] * 6 + [P, c_int, P, P]),
        "ntk_cu_count": (c_int, []),
        "ntk_dnc_mp_plan": (c_int, [c_int] * 8 + [ctypes.POINTER(c_int), ctypes.POINTER(c_size_t)]),
        "ntk_dnc_mp_compiled_shape": (c_int, [c_int] * 7),
        "ntk_dnc_mp_status": (c_int, [P, c_size_t, c_int, c_int, c_int, P]),
        "ntk_dnc_mp_placement": (c_int, [P, c_int, c_int, P, P]),
        "ntk_dnc_mp_fwd": (c_int, [c_int] * 8 + [ctypes.c_float, c_int] + [P] * 13 + [P] * 18 + [P, P]),
        "ntk_dnc_mp_bwd_plan": (c_int, [c_int] * 8 + [ctypes.POINTER(c_int), ctypes.POINTER(c_size_t)]),
        "ntk_dnc_mp_bwd": (c_int, [c_int] * 8 + [ctypes.c_float, c_int] + [P, c_int, P, P] + [P] * 7 + [P] * 15 + [P] * 6 + [P, c_int, P, P]),
        "ntk_dnc_cosine_weights": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
        "ntk_dnc_linkage": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
        "ntk_dnc_directional_read_weights": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
        "ntk_dnc_freeness": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
        "ntk_dnc_write_allocation_weights": (c_int, [P, P, P, c_int, c_int, c_int, P]),
        "ntk_dnc_interface_activations": (c_int, [P, c_int, P] + [c_int] * 5 + [P]),
        "ntk_dnc_write_weights": (c_int, [P] * 8 + [c_int] * 4 + [P]),
        "ntk_dnc_erase_and_write": (c_int, [P] * 5 + [c_int] * 4 + [P]),
        "ntk_dnc_read_weights": (c_int, [P] * 8 + [c_int] * 5 + [P]),
        "ntk_dnc_read_words": (c_int, [P] * 3 + [c_int] * 4 + [P]),
        "ntk_dnc_access_step_workspace_bytes": (ctypes.c_size_t, [c_int] * 5),
        "ntk_dnc_access_step_fwd": (c_int, [P, c_int] + [P] * 14 + [c_int] * 5 + [P]),
        "ntk_dnc_access_step_bwd_workspace_bytes": (ctypes.c_size_t, [c_int] * 5),
        "ntk_dnc_access_step_bwd": (c_int, [P, c_int] + [P] * 14 + [c_int] * 5 + [P]),
        "ntk_ntm_cosine_similarity": (c_int, [P] * 3 + [c_int] * 5 + [P]),
        "ntk_ntm_circular_convolution": (c_int, [P] * 3 + [c_int] * 4 + [P]),
        "ntk_ntm_step_debug": (c_int, [P] + [c_int] * 7 + [P] * 9 + [c_int] * 5 + [P]),
        "ntk_ntm_step_fwd": (c_int, [c_int] * 9 + [P] * 24),
        "ntk_ntm_step_bwd": (c_int, [c_int] * 9 + [P, c_int, P, c_int] + [P] * 22),
        "ntk_lstm_step_fwd": (c_int, [P, P, ctypes.c_float, P, P, P, c_int, c_int, P]),
        "ntk_lstm_step_bwd": (c_int, [P] * 7 + [c_int, c_int, P]),
        "ntk_vgg_wino_packed_floats": (ctypes.c_size_t, [c_int, c_int]),
        "ntk_vgg_pack_weights_wino": (c_int, [P, P, c_int, c_int, P]),
        "ntk_vgg_conv3x3_relu_wino_f32": (c_int, [P] * 4 + [c_int] * 6 + [P]),
        "ntk_vgg_wino43_packed_floats": (ctypes.c_size_t, [c_int, c_int]),
        "ntk_vgg_pack_weights_wino43": (c_int, [P, P, c_int, c_int, P]),
        "ntk_vgg_conv3x3_relu_wino43_f32": (c_int, [P] * 4 + [c_int] * 6 + [P]),
        "ntk_vgg_conv3x3_relu_wino43_window_f32": (c_int, [P] * 4 + [c_int] * 10 + [P]),
        "ntk_vgg_conv3x3_relu_wino43_form_f32": (c_int, [P] * 4 + [c_int] * 11 + [P]),
        "ntk_vgg_conv3x3_relu_wino43_layout_f32": (c_int, [P] * 4 + [c_int] * 8 + [P]),
        "ntk_maxpool2x2": (c_int, [P, P] + [c_int] * 4 + [P]),
        "ntk_offset_loss_fwd": (c_int, [P] * 4 + [c_int] * 4 + [P]),
        "ntk_offset_loss_bwd": (c_int, [P] * 3 + [c_int] * 4 + [P]),
        "ntk_dnc_seq_bwd": (c_int, [c_int] * 8 + [ctypes.c_float] + [P, c_int, P, c_int, P] + [P] * 7 + [P] * 15 + [P] * 6 + [P, c_int, P]),
        "ntk_gather_serialize": (c_int, [P, P, P] + [c_int] * 9 + [P]),
        "ntk_gather_serialize_online": (c_int, [P, P, P] + [c_int] * 9 + [P]),
        "ntk_crop_and_resize": (c_int, [P, c_int, c_int, c_int, P] + [ctypes.c_float] * 4 + [P, c_int, c_int, ctypes.c_float, P]),
        "ntk_resize_bilinear": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, P]),
        "ntk_offset_loss": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
        "ntk_log_loss": (c_int, [P, P, P, P, c_int, P]),
        "ntk_serialize_sequential": (c_int, [P, P, P] + [c_int] * 5 + [P]),
        "ntk_heatmap_ce_loss": (c_int, [P] * 5 + [c_int] * 3 + [P]),
        "ntk_serialize_two_step": (c_int, [P, P, P] + [c_int] * 5 + [P]),
        "ntk_two_step_ce_loss": (c_int, [P] * 5 + [c_int] * 3 + [P]),
        "ntk_ntm_init_state": (c_int, [P, P, c_int, c_int, c_int, P]),
        "ntk_ntm_init_state_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
        "ntk_global_norm_workspace_bytes": (c_size_t, [c_size_t]),
        "ntk_global_norm": (c_int, [P, c_size_t, P, P, P]),
        "ntk_rmsprop_clip_step": (c_int, [P, P, P, P, c_size_t] + [ctypes.c_float] * 5 + [P, P]),
        "ntk_rmsprop_clip_step_checked": (c_int, [P, P, P, P, c_size_t] + [ctypes.c_float] * 5 + [P, P, P, P]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return sigs


def lib():
    """Load (once) and return the C-ABI library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NtkError(
                "libntmtrack_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C ntm-tracker_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        l._ntk_sigs = _sig(l)
        _lib = l
    return _lib


def exported_symbols():
    return sorted(lib()._ntk_sigs.keys())


def check(rc, what):
    if rc != 0:
        msg = lib().ntk_last_error()
        raise NtkError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))


def ptr(t):
    """Device pointer of a contiguous fp32 CUDA(HIP) tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NtkError("expected a device tensor; the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise NtkError("expected a contiguous tensor")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

"""Module-level addressing ops of the reference's ops.py on HIP kernels (same names and argument meaning).

``batched_smooth_cosine_similarity(memory [B,N,M], keys [B,H,M]) -> [B,H,N]`` (ops.py:135-158) and
``batched_circular_convolution(tensor [B,H,N], kernel [B,H,S]) -> [B,H,N]`` (ops.py:180-214),
``circular_shift(tensor, shift)`` (ops.py:216-242).  The NTM sequence kernels fuse the same arithmetic; these
are the stand-alone forms the reference's ops_test.py exercises.

``similarity="as_coded"`` (default) is what the shipped code computes (what NTMCell uses);
``similarity="smooth_cosine"`` is the Torch7 nn.SmoothCosineSimilarity the reference's own test expects.
"""
import torch

from . import _lib

_P = _lib.ptr
_MODES = {"as_coded": 0, "smooth_cosine": 1}


def _dev(t, device):
    return torch.as_tensor(t, dtype=torch.float32).to(device).contiguous()


def batched_smooth_cosine_similarity(memory, keys, name=None, scope=None, similarity="as_coded", device="cuda"):
    if similarity not in _MODES:
        raise _lib.NtkError("similarity=%r (expected 'as_coded' or 'smooth_cosine')" % (similarity,))
    m, k = _dev(memory, device), _dev(keys, device)
    B, N, Md = m.shape
    H = k.shape[1]
    out = torch.empty((B, H, N), device=m.device)
    _lib.check(_lib.lib().ntk_ntm_cosine_similarity(_P(m), _P(k), _P(out), B, N, Md, H, _MODES[similarity], _lib.stream()),
               "ntk_ntm_cosine_similarity")
    return out


def batched_circular_convolution(tensor, kernel, name=None, scope=None, device="cuda"):
    w, s = _dev(tensor, device), _dev(kernel, device)
    B, H, N = w.shape
    out = torch.empty_like(w)
    _lib.check(_lib.lib().ntk_ntm_circular_convolution(_P(w), _P(s), _P(out), B, H, N, s.shape[2], _lib.stream()),
               "ntk_ntm_circular_convolution")
    return out


def circular_shift(tensor, shift, device="cuda"):
    """result[..., i] = tensor[..., (i + shift) mod N]: a one-tap circular convolution."""
    t = _dev(tensor, device)
    N = t.shape[-1]
    if not -N < shift < N:
        raise _lib.NtkError("shift=%d out of range for a length-%d axis" % (shift, N))
    flat = t.reshape(1, -1, N)
    # taps of a kernel of width SS start at floor(-SS/2): place the single 1 so that the tap offset equals `shift`
    SS = 2 * abs(shift) + 3 if shift <= 0 else 2 * shift + 2
    start = -((SS + 1) // 2)
    kern = torch.zeros((1, flat.shape[1], SS), device=t.device)
    kern[:, :, shift - start] = 1.0
    return batched_circular_convolution(flat, kern, device=device).reshape(t.shape)

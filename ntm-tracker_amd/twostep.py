"""Two-step presentation + (F+1)-way head (main.py's ntm_two_step, :862-977, over
ntm_tracker_new.NTMTracker(two_step=True), :112-195) on the HIP path.

A whole frame is ONE step of the cell: its flattened (in the reference: 1x1-conv compressed) feature map.  Step 0 shows
frame 0 together with the target heat-map; every later frame takes a presentation step [0, feat, 0] and a query step
[1, 0, 0] -- S = 2T - 1 steps.  The cell has F + 1 outputs (F positions + background); the labels are the background
row at step 0 and at every presentation step and [gt_t, 0] at the query steps, passed through a softmax before the
cross entropy (as coded, :943-947); loss = sum / ((2T - 1) B).  ``InputCompressor`` is the 1x1 convolution in front
(:882-887: conv2d(features, w [1,1,C,compress_dim], VALID) = one GEMM over the positions), trained with the cell.
"""
import torch

from . import _lib
from .ntm import NTMCell, _P, _np, gemm_nt, gemm_tn
from .tracker import RMSPropClip, _Checkpointing


def two_step_steps(T):
    return 2 * T - 1


def serialize_two_step(feat, target, ldx, out=None):
    """feat [B, T, D], target [B, F] (or None) -> X [B, 2T-1, ldx], rows [switch, feat, target, 0 pad]."""
    B, T, D = feat.shape
    F = target.shape[1] if target is not None else ldx - 1 - D
    if out is None:
        out = torch.empty((B, 2 * T - 1, ldx), device=feat.device, dtype=torch.float32)
    _lib.check(_lib.lib().ntk_serialize_two_step(_P(feat.contiguous()), _np(target), _P(out), B, T, D, F, ldx, _lib.stream()),
               "ntk_serialize_two_step")
    return out


def two_step_ce_loss(logits, gt, want_grad=True):
    """logits [B, 2T-1, F+1], gt [B, T, F] -> (loss [1], probs [B, 2T-1, F+1], dlogits or None)."""
    B, S, K = logits.shape
    T, F = gt.shape[1], gt.shape[2]
    if S != 2 * T - 1 or K != F + 1 or gt.shape[0] != B:
        raise _lib.NtkError("two_step_ce_loss: logits %s do not match gt %s" % (tuple(logits.shape), tuple(gt.shape)))
    probs = torch.empty_like(logits)
    loss = torch.empty(1, device=logits.device)
    dlogits = torch.empty_like(logits) if want_grad else None
    _lib.check(_lib.lib().ntk_two_step_ce_loss(_P(logits.contiguous()), _P(gt.contiguous()), _P(probs), _P(loss), _np(dlogits),
                                              B, T, F, _lib.stream()), "ntk_two_step_ce_loss")
    return loss, probs, dlogits


class InputCompressor(object):
    """main.py:882-887 (also :709-716, :799-803, :1005-1011): features [.., C] -> [.., compress_dim] by a bias-free 1x1
    convolution, xavier-uniform initialised.  Forward = one fp32 MFMA GEMM; backward fills ``grad`` (d loss / d w)."""

    def __init__(self, channels, compress_dim, device="cuda", seed=0):
        self.C, self.Cd, self.device = int(channels), int(compress_dim), torch.device(device)
        if self.C % 4:
            raise _lib.NtkError("InputCompressor: channels=%d must be a multiple of 4" % self.C)
        g = torch.Generator().manual_seed(int(seed))
        lim = (6.0 / (self.C + self.Cd)) ** 0.5
        self.wT = ((torch.rand((self.Cd, self.C), generator=g) * 2 - 1) * lim).to(self.device)      # [compress_dim][C] = w^T
        self.grad = torch.zeros_like(self.wT)

    def load_w(self, w_hwio):
        """w [1,1,C,compress_dim] (TF HWIO) or [C, compress_dim]."""
        w = torch.as_tensor(w_hwio, dtype=torch.float32).reshape(self.C, self.Cd)
        self.wT.copy_(w.t().to(self.device))

    def w(self):
        return self.wT.t().contiguous().view(1, 1, self.C, self.Cd)

    def __call__(self, features):
        x = features.contiguous().view(-1, self.C)
        self._x = x
        return gemm_nt(x, self.wT).view(tuple(features.shape[:-1]) + (self.Cd,))

    def backward(self, d_out):
        """d_out [.., compress_dim] -> fills self.grad [compress_dim][C] (= (d loss / d w)^T); the frozen features get none."""
        gemm_tn(d_out.contiguous().view(-1, self.Cd), self._x, self.grad)
        return self.grad


class NTMTwoStepTracker(_Checkpointing):
    """Per-frame feature vectors -> two-step serialisation -> NTMCell(output_dim F + 1) -> softmax-CE on soft labels."""

    def __init__(self, batch_size, sequence_length, num_features, feature_dim, mem_size=128, mem_dim=20, hidden_size=200,
                 read_head_size=4, write_head_size=1, write_first=False, init_scale=0.05, learning_rate=1e-4, decay=0.95,
                 momentum=0.9, max_gradient_norm=5.0, device="cuda", seed=42, compressor=None):
        """compressor: optional InputCompressor applied to feature maps [B, T, F, C] -> the cell sees F * compress_dim values
        per frame (feature_dim must then equal F * compress_dim).  loss_and_grads leaves its weight gradient in
        ``compressor.grad``; train_step updates the CELL only (the reference clips the joint global norm of all variables
        and applies RMSProp to all of them, main.py:953-961: the caller owns that policy for the extra variable)."""
        self.compressor = compressor
        self.B, self.T, self.F, self.D = int(batch_size), int(sequence_length), int(num_features), int(feature_dim)
        self.S = two_step_steps(self.T)
        self.device = torch.device(device)
        self.cell = NTMCell(self.F + 1, mem_size=mem_size, mem_dim=mem_dim, controller_hidden_size=hidden_size,
                            controller_num_layers=1, write_head_size=write_head_size, read_head_size=read_head_size,
                            write_first=write_first, input_dim=1 + self.D + self.F, device=self.device, init_scale=init_scale,
                            seed=seed)
        self.opt = RMSPropClip(self.cell.params, learning_rate, decay, momentum, 1e-10, max_gradient_norm)

    def _ckpt_params(self):
        return self.cell.params

    def _core(self):
        return self.cell

    def forward_features(self, feat, target, record=False):
        X = serialize_two_step(feat, target, self.cell.input_ldx)
        st0 = self.cell.zero_state(self.B)
        logits, _o, _new, rec = self.cell.run_sequence(X, st0, record=record, want_outputs=False)
        return X, st0, logits, rec

    def loss_and_grads(self, feat, gts):
        """feat [B, T, D] (or, with a compressor, feature maps [B, T, F, C]); gts [B, T, F]: frame 0 is the target shown to
        the tracker, frames 1.. are the labels.  With a compressor its weight gradient lands in ``compressor.grad``
        (un-clipped: joint clipping with the cell's gradients is the caller's, see train_step)."""
        if self.compressor is not None:
            feat = self.compressor(feat).view(self.B, self.T, self.D)
        self.cell.want_input_grad = self.compressor is not None
        X, st0, logits, rec = self.forward_features(feat, gts[:, 0].contiguous(), record=True)
        loss, probs, dlogits = two_step_ce_loss(logits, gts)
        g0 = self.cell.backward_sequence(X, st0, rec, dlogits)
        self.cell.init_state_backward(g0, self.B)
        if self.compressor is not None:
            dX = self.cell.last_dX                                           # [B, 2T-1, ldx]
            d_feat = torch.empty((self.B, self.T, self.D), device=self.device)
            d_feat[:, 0] = dX[:, 0, 1:1 + self.D]                            # presentation steps 0, 1, 3, 5, ... carry the frames
            if self.T > 1:
                d_feat[:, 1:] = dX[:, 1::2, 1:1 + self.D]
            self.compressor.backward(d_feat.view(self.B, self.T, self.F, -1))
        return loss, probs

    def train_step(self, feat, gts):
        loss, _ = self.loss_and_grads(feat, gts)
        self.opt.step()
        return loss

    def infer(self, feat, target):
        """-> probabilities [B, T-1, F+1] at the query steps (softmax over the F positions + background)."""
        _X, _st, logits, _ = self.forward_features(feat, target)
        dummy = torch.zeros((self.B, self.T, self.F), device=self.device)
        _l, probs, _ = two_step_ce_loss(logits, dummy, want_grad=False)
        return probs[:, 2::2]

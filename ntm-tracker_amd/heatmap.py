"""Sequential presentation + heat-map head (main.py's ntm_sevenbyseven, :1646-1969, and the same
serialisation in its sequential tracker, :979-1291) on the HIP path.

Every position of the feature map is one feature (F = Hf * Wf; 7 x 7 = 49 on the reference's pooled map).  Frame 0 is
shown as F rows carrying the target heat-map in the last column; every later frame as one frame-delimiter row followed
by, per feature, the feature row and a feature-delimiter row -- S = F + (T - 1)(2 F + 1) steps (:1701-1775).  The cell
has ONE output; its logits at the feature-delimiter steps of a frame are that frame's F-way scores, trained with
softmax cross entropy against the frame's heat-map, summed over sequences and frames and divided by T - 1
(:1880-1922).  Optimiser as the other trackers (:1928-1940: clip_by_global_norm + RMSProp).
"""
import torch

from . import _lib
from .ntm import NTMCell, _P, _np
from .tracker import RMSPropClip, _Checkpointing


def sequential_steps(T, F):
    return F + (T - 1) * (2 * F + 1)


def serialize_sequential(fmap, gts0, B, T, ldx, out=None):
    """fmap [B*T, Hf, Wf, C] (or [B*T, F, C]) features, gts0 [B, F] frame-0 heat-map (or None) -> X [B, S, ldx]."""
    C = fmap.shape[-1]
    F = fmap.numel() // (B * T * C)
    S = sequential_steps(T, F)
    if out is None:
        out = torch.empty((B, S, ldx), device=fmap.device, dtype=torch.float32)
    _lib.check(_lib.lib().ntk_serialize_sequential(_P(fmap.contiguous()), _np(gts0), _P(out), B, T, F, C, ldx, _lib.stream()),
               "ntk_serialize_sequential")
    return out


def heatmap_ce_loss(logits, gt, T, want_grad=True):
    """logits [B, S, 1], gt [B, T-1, F] -> (loss [1], probs [B, T-1, F], dlogits [B, S, 1] or None)."""
    B, S = logits.shape[0], logits.shape[1]
    F = gt.shape[2]
    if S != sequential_steps(T, F) or logits.numel() != B * S:
        raise _lib.NtkError("heatmap_ce_loss: logits %s do not match T=%d F=%d (one output per step)" % (tuple(logits.shape), T, F))
    probs = torch.empty((B, T - 1, F), device=logits.device)
    loss = torch.empty(1, device=logits.device)
    dlogits = torch.empty_like(logits) if want_grad else None
    _lib.check(_lib.lib().ntk_heatmap_ce_loss(_P(logits.contiguous()), _P(gt.contiguous()), _P(probs), _P(loss), _np(dlogits),
                                             B, T, F, _lib.stream()), "ntk_heatmap_ce_loss")
    return loss, probs, dlogits


class NTMHeatmapTracker(_Checkpointing):
    """Feature map -> sequential serialisation -> NTMCell(output_dim 1) -> per-frame F-way softmax-CE."""

    def __init__(self, batch_size, sequence_length, num_features, feature_channels, mem_size=128, mem_dim=20, hidden_size=200,
                 read_head_size=4, write_head_size=1, init_scale=0.05, learning_rate=1e-4, decay=0.95, momentum=0.9,
                 max_gradient_norm=5.0, device="cuda", seed=42):
        self.B, self.T, self.F, self.C = int(batch_size), int(sequence_length), int(num_features), int(feature_channels)
        self.S = sequential_steps(self.T, self.F)
        self.device = torch.device(device)
        self.cell = NTMCell(1, mem_size=mem_size, mem_dim=mem_dim, controller_hidden_size=hidden_size, controller_num_layers=1,
                            write_head_size=write_head_size, read_head_size=read_head_size, input_dim=self.C + 3,
                            device=self.device, init_scale=init_scale, seed=seed)
        self.opt = RMSPropClip(self.cell.params, learning_rate, decay, momentum, 1e-10, max_gradient_norm)

    def _ckpt_params(self):
        return self.cell.params

    def _core(self):
        return self.cell

    def forward_features(self, fmap, gts0, record=False):
        X = serialize_sequential(fmap, gts0, self.B, self.T, self.cell.input_ldx)
        st0 = self.cell.zero_state(self.B)
        logits, _o, _new, rec = self.cell.run_sequence(X, st0, record=record, want_outputs=False)
        return X, st0, logits, rec

    def loss_and_grads(self, fmap, gts):
        """gts [B, T, F]: frame 0 is the target shown to the tracker, frames 1.. are the labels."""
        X, st0, logits, rec = self.forward_features(fmap, gts[:, 0].contiguous(), record=True)
        loss, probs, dlogits = heatmap_ce_loss(logits, gts[:, 1:].contiguous(), self.T)
        g0 = self.cell.backward_sequence(X, st0, rec, dlogits)
        self.cell.init_state_backward(g0, self.B)
        return loss, probs

    def train_step(self, fmap, gts):
        loss, _ = self.loss_and_grads(fmap, gts)
        self.opt.step()
        return loss

    def infer(self, fmap, gts0):
        """-> predicted heat-maps [B, T-1, F] (softmax over the features of each frame)."""
        _X, _st, logits, _ = self.forward_features(fmap, gts0)
        dummy = torch.zeros((self.B, self.T - 1, self.F), device=self.device)
        _l, probs, _ = heatmap_ce_loss(logits, dummy, self.T, want_grad=False)
        return probs

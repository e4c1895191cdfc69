"""NTM copy task (main.py:1540-1644, BASELINE configs[0]) on the HIP path: a regression demo that exercises
the sequence kernel, BPTT and the optimiser end to end without VGG.

Inputs [B, 2L+1, width+1]: L steps of `width` random bits (+ a zero indicator column), one delimiter step
[0..0,1], L zero steps; labels are zero for the first L+1 steps and the input block for the last L
(main.py:1546-1559).  Loss = tf.losses.log_loss(labels, sigmoid(logits)) (:1603-1610); clip_by_global_norm +
RMSProp as in the tracker (:1613-1620).  The reference's own copy_paste() is broken at HEAD (it unpacks five
return values from LoopNTMTracker, which returns two) -- this follows its graph construction as a spec.
"""
import torch

from . import _lib
from .ntm import NTMCell, _P
from .tracker import RMSPropClip


def make_batch(bits):
    """bits [B, L, width] in {0,1} -> (inputs [B, 2L+1, width+1], labels [B, 2L+1, width+1])."""
    B, L, width = bits.shape
    S = 2 * L + 1
    x = torch.zeros((B, S, width + 1), dtype=torch.float32, device=bits.device)
    y = torch.zeros_like(x)
    x[:, :L, :width] = bits
    x[:, L, width] = 1.0
    y[:, L + 1:, :width] = bits
    return x, y


class CopyTask(object):
    def __init__(self, batch_size, length, width=3, mem_size=128, mem_dim=20, hidden_size=100, read_head_size=1,
                 write_head_size=1, init_scale=0.05, learning_rate=1e-4, decay=0.95, momentum=0.9,
                 max_gradient_norm=5.0, device="cuda", seed=0):
        self.B, self.L, self.width = batch_size, length, width
        self.S = 2 * length + 1
        self.device = torch.device(device)
        self.cell = NTMCell(width + 1, mem_size=mem_size, mem_dim=mem_dim, controller_hidden_size=hidden_size,
                            controller_num_layers=1, write_head_size=write_head_size, read_head_size=read_head_size,
                            input_dim=width + 1, device=self.device, init_scale=init_scale, seed=seed)
        self.opt = RMSPropClip(self.cell.params, learning_rate, decay, momentum, 1e-10, max_gradient_norm)

    def loss_and_grads(self, x, y):
        X = self.cell._pad_inputs(x)
        st0 = self.cell.zero_state(self.B)
        logits, _o, _n, rec = self.cell.run_sequence(X, st0, record=True, want_outputs=False)
        loss = torch.empty(1, device=self.device)
        dlogits = torch.empty_like(logits)
        _lib.check(_lib.lib().ntk_log_loss(_P(logits), _P(y.contiguous()), _P(loss), _P(dlogits), logits.numel(), _lib.stream()),
                   "ntk_log_loss")
        g0 = self.cell.backward_sequence(X, st0, rec, dlogits)
        self.cell.init_state_backward(g0, self.B)
        return loss, logits

    def train_step(self, bits):
        x, y = make_batch(bits)
        loss, _ = self.loss_and_grads(x, y)
        self.opt.step()
        return loss

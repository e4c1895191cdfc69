"""GPU: the sequential presentation (2 F + 1 steps per frame) and the F-way heat-map head with softmax cross entropy
of main.py's earlier trackers (ntm_sevenbyseven, main.py:1646-1969) against the oracle restatement
(oracle/ntm_oracle.py: serialize_sequential / heatmap_ce_loss, gradients from torch autograd)."""
import numpy as np
import pytest
import torch

from oracle import ntm_oracle as O
from oracle import ntm_oracle_torch as OT

pytestmark = pytest.mark.gpu


def test_sequential_serialiser_is_bit_exact(cuda):
    from ntmtrack import heatmap
    rng = np.random.default_rng(3)
    B, T, F, C = 2, 3, 9, 16
    feats = rng.standard_normal((B, T, F, C)).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, F)).astype(np.float32)
    ref = O.serialize_sequential(feats, gts)
    assert ref.shape == (B, F + (T - 1) * (2 * F + 1), C + 3)
    X = heatmap.serialize_sequential(torch.from_numpy(feats.reshape(B * T, 3, 3, C)).to(cuda), torch.from_numpy(gts[:, 0].copy()).to(cuda),
                                     B, T, 20)
    got = X.cpu().numpy()
    assert np.array_equal(got[:, :, :C + 3], ref) and not got[:, :, C + 3:].any()
    # layout contract: frame delimiter first, then (feature, feature delimiter) pairs
    s0 = F
    assert got[0, s0, C + 1] == 1 and not got[0, s0, :C].any() and got[0, s0 + 2, C] == 1 and np.array_equal(got[0, s0 + 1, :C], feats[0, 1, 0])


def test_heatmap_ce_loss_and_gradient_match_oracle(cuda):
    from ntmtrack import heatmap
    rng = np.random.default_rng(4)
    B, T, F = 3, 4, 49
    S = F + (T - 1) * (2 * F + 1)
    logits = rng.standard_normal((B, S, 1)).astype(np.float32) * 2
    gt = rng.uniform(0, 1, size=(B, T - 1, F)).astype(np.float32)
    gt /= gt.sum(2, keepdims=True)
    gt[0, 0] *= 0.7                                   # labels need not sum to one (softmax_cross_entropy_with_logits does not require it)
    loss_ref, probs_ref = O.heatmap_ce_loss(logits, gt, T)
    lt = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    z = lt.reshape(B, -1)[:, F:].reshape(B, T - 1, 2 * F + 1)[:, :, 1:].reshape(B, T - 1, F, 2)[:, :, :, 1]
    l2 = -(torch.tensor(gt, dtype=torch.float64) * torch.log_softmax(z, dim=2)).sum() / (T - 1)
    l2.backward()
    loss, probs, dlog = heatmap.heatmap_ce_loss(torch.from_numpy(logits).to(cuda), torch.from_numpy(gt).to(cuda), T)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), loss_ref, rtol=1e-5)
    np.testing.assert_allclose(float(l2.detach()), loss_ref, rtol=1e-12)
    np.testing.assert_allclose(probs.cpu().numpy(), probs_ref, atol=1e-6)
    np.testing.assert_allclose(dlog.cpu().numpy(), lt.grad.numpy(), atol=1e-6)


def test_heatmap_tracker_gradients_and_learning(cuda):
    """End to end: loss and every gradient of the NTM under the sequential presentation vs torch autograd, then a few
    optimiser steps on one batch lower the loss."""
    from ntmtrack import heatmap
    B, T, F, C = 2, 3, 9, 16
    rng = np.random.default_rng(6)
    trk = heatmap.NTMHeatmapTracker(B, T, F, C, mem_size=64, mem_dim=8, hidden_size=32, read_head_size=2, write_head_size=1,
                                    init_scale=0.2, learning_rate=3e-3, device=cuda, seed=5)
    sd = {k: v.numpy() for k, v in trk.cell.state_dict().items()}
    cfg = O.NTMConfig(C + 3, 1, mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=32, controller_num_layers=1,
                      write_head_size=1, read_head_size=2)
    feats = np.maximum(rng.standard_normal((B, T, F, C)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, F)).astype(np.float32)
    gts /= gts.sum(2, keepdims=True)
    x = O.serialize_sequential(feats, gts)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sd.items()}
    logits, _ = OT.loop(cfg, pt, torch.tensor(x, dtype=torch.float64))
    z = logits.reshape(B, -1)[:, F:].reshape(B, T - 1, 2 * F + 1)[:, :, 1:].reshape(B, T - 1, F, 2)[:, :, :, 1]
    loss_ref = -(torch.tensor(gts[:, 1:], dtype=torch.float64) * torch.log_softmax(z, dim=2)).sum() / (T - 1)
    loss_ref.backward()
    fmap = torch.from_numpy(feats.reshape(B * T, 3, 3, C)).to(cuda)
    g = torch.from_numpy(gts).to(cuda)
    loss, probs = trk.loss_and_grads(fmap, g)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), float(loss_ref.detach()), rtol=1e-4)
    got = trk.cell.params.to_tf(grad=True)
    gmax = max(float(np.abs(pt[k].grad.numpy()).max()) for k in sd)
    for k in sorted(sd):
        ref = pt[k].grad.numpy()
        # the labels of a frame sum to one, so d loss / d (output bias) = sum(p) - sum(y) is exactly zero: what the kernels
        # return there is float32 rounding (measured 3-4e-6 of the largest gradient, moving with the summation order of the
        # forward pass): compare against the tensor's own scale, floored at 2e-3 of the largest gradient
        err = np.max(np.abs(got[k].numpy() - ref)) / max(np.max(np.abs(ref)), 2e-3 * gmax)
        assert err < 3e-3, (k, err)
    assert probs.shape == (B, T - 1, F) and torch.allclose(probs.sum(2), torch.ones((B, T - 1), device=cuda), atol=1e-5)
    first = float(loss.cpu())
    for _ in range(30):
        last = trk.train_step(fmap, g)
    assert float(last.cpu()) < first

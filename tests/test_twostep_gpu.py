"""GPU: the two-step presentation (one step per frame + one query step, 2T - 1 steps) and its (F + 1)-way head with
softmax cross entropy on softmaxed labels (main.py's ntm_two_step, :862-977; ntm_tracker_new.py:112-195) against the
oracle restatement (oracle/ntm_oracle.py: two_step_inputs / two_step_labels / two_step_ce_loss, gradients from torch
autograd).  Parity unpinned: the reference holds no fixture for this variant."""
import numpy as np
import pytest
import torch

from oracle import ntm_oracle as O
from oracle import ntm_oracle_torch as OT

pytestmark = pytest.mark.gpu


def test_two_step_serialiser_is_bit_exact(cuda):
    from ntmtrack import twostep
    rng = np.random.default_rng(3)
    B, T, D, F = 3, 4, 37, 9
    feat = rng.standard_normal((B, T, D)).astype(np.float32)
    target = rng.uniform(0, 1, size=(B, F)).astype(np.float32)
    ref = O.two_step_inputs(feat, target)
    assert ref.shape == (B, 2 * T - 1, 1 + D + F)
    X = twostep.serialize_two_step(torch.from_numpy(feat).to(cuda), torch.from_numpy(target).to(cuda), 48).cpu().numpy()
    assert np.array_equal(X[:, :, :1 + D + F], ref) and not X[:, :, 1 + D + F:].any()
    # layout contract (ntm_tracker_new.py:150-181): presentation steps carry switch 0, query steps switch 1 and nothing else
    assert X[0, 2, 0] == 1 and not X[0, 2, 1:].any() and X[0, 1, 0] == 0 and np.array_equal(X[0, 1, 1:1 + D], feat[0, 1])
    assert np.array_equal(X[:, 0, 1 + D:1 + D + F], target) and not X[:, 1:, 1 + D:].any()


def test_two_step_ce_loss_and_gradient_match_oracle(cuda):
    from ntmtrack import twostep
    rng = np.random.default_rng(4)
    B, T, F = 3, 4, 49
    logits = (rng.standard_normal((B, 2 * T - 1, F + 1)) * 2).astype(np.float32)
    gt = (rng.uniform(0, 1, size=(B, T, F)) > 0.8).astype(np.float32)                 # 0/1 heat-maps, as the reference's
    gt[0, 1] = rng.uniform(0, 3, size=F).astype(np.float32)                           # ... and an arbitrary one
    loss_ref, probs_ref, dl_ref = O.two_step_ce_loss(logits.astype(np.float64), gt.astype(np.float64))
    lt = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    q = torch.softmax(torch.tensor(O.two_step_labels(gt.astype(np.float64))), dim=2)
    l2 = -(q * torch.log_softmax(lt, dim=2)).sum() / ((2 * T - 1) * B)
    l2.backward()
    np.testing.assert_allclose(float(l2.detach()), loss_ref, rtol=1e-12)
    np.testing.assert_allclose(lt.grad.numpy(), dl_ref, atol=1e-14)
    loss, probs, dlog = twostep.two_step_ce_loss(torch.from_numpy(logits).to(cuda), torch.from_numpy(gt).to(cuda))
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), loss_ref, rtol=1e-5)
    np.testing.assert_allclose(probs.cpu().numpy(), probs_ref, atol=1e-6)
    np.testing.assert_allclose(dlog.cpu().numpy(), dl_ref, atol=1e-7)


def test_two_step_tracker_gradients_and_learning(cuda):
    """End to end: loss and every gradient of the NTM under the two-step presentation vs torch autograd on the oracle,
    then a few optimiser steps on one batch lower the loss."""
    from ntmtrack import twostep
    B, T, F, D = 2, 3, 9, 36
    rng = np.random.default_rng(6)
    trk = twostep.NTMTwoStepTracker(B, T, F, D, mem_size=64, mem_dim=8, hidden_size=32, read_head_size=2, write_head_size=1,
                                    write_first=True, init_scale=0.2, learning_rate=3e-3, device=cuda, seed=5)
    sd = {k: v.numpy() for k, v in trk.cell.state_dict().items()}
    cfg = O.NTMConfig(1 + D + F, F + 1, mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=32, controller_num_layers=1,
                      write_head_size=1, read_head_size=2, write_first=True)
    feat = np.maximum(rng.standard_normal((B, T, D)), 0).astype(np.float32)
    gts = (rng.uniform(0, 1, size=(B, T, F)) > 0.7).astype(np.float32)
    x = O.two_step_inputs(feat, gts[:, 0])
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sd.items()}
    logits, _ = OT.loop(cfg, pt, torch.tensor(x, dtype=torch.float64))
    q = torch.softmax(torch.tensor(O.two_step_labels(gts.astype(np.float64))), dim=2)
    loss_ref = -(q * torch.log_softmax(logits, dim=2)).sum() / ((2 * T - 1) * B)
    loss_ref.backward()
    f, g = torch.from_numpy(feat).to(cuda), torch.from_numpy(gts).to(cuda)
    loss, probs = trk.loss_and_grads(f, g)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), float(loss_ref.detach()), rtol=1e-4)
    got = trk.cell.params.to_tf(grad=True)
    gmax = max(float(np.abs(pt[k].grad.numpy()).max()) for k in sd)
    for k in sorted(sd):
        ref = pt[k].grad.numpy()
        err = np.max(np.abs(got[k].numpy() - ref)) / max(np.max(np.abs(ref)), 1e-3 * gmax)
        assert err < 3e-3, (k, err)
    assert probs.shape == (B, 2 * T - 1, F + 1) and torch.allclose(probs.sum(2), torch.ones((B, 2 * T - 1), device=cuda), atol=1e-5)
    first = float(loss.cpu())
    for _ in range(30):
        last = trk.train_step(f, g)
    assert float(last.cpu()) < first
    assert trk.infer(f, g[:, 0].contiguous()).shape == (B, T - 1, F + 1)


@pytest.mark.parametrize("two_step", [False, True], ids=["one_step_per_frame", "two_step"])
def test_static_unroll_trackers_match_oracle(cuda, two_step):
    """ntm_tracker_new.NTMTracker (one step per frame with the target indicator; two_step) and PlainNTMTracker mirrors:
    logits of the single launch vs the oracle cell looped over the same rows."""
    from ntmtrack.ntm import NTMTracker, PlainNTMTracker
    rng = np.random.default_rng(8)
    B, T, D, F = 2, 4, 20, 9
    kw = dict(mem_size=64, mem_dim=8, controller_hidden_size=32, controller_num_layers=1, read_head_size=2, write_head_size=1)
    trk = NTMTracker(T, B, F + 1, two_step=two_step, device=cuda, seed=3, **kw)
    feat = rng.standard_normal((B, T, D)).astype(np.float32)
    target = rng.uniform(0, 1, size=(B, F)).astype(np.float32)
    outputs, logits, states, debugs = trk(torch.from_numpy(feat).to(cuda), torch.from_numpy(target).to(cuda))
    if two_step:
        x = O.two_step_inputs(feat, target)
    else:
        x = np.concatenate([feat, np.zeros((B, T, F), np.float32)], 2)
        x[:, 0, D:] = target
    cfg = O.NTMConfig(x.shape[2], F + 1, mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=32, controller_num_layers=1,
                      write_head_size=1, read_head_size=2)
    sd = {k: v.numpy() for k, v in trk.cell.state_dict().items()}
    out_ref, logits_ref, fin, states_ref = O.loop_ntm_tracker(cfg, sd, x, return_states=True)
    torch.cuda.synchronize()
    # ntm_tracker_new.py:95-100: the initial state and the state after EVERY step
    assert logits.shape == (B, x.shape[1], F + 1) and len(states) == x.shape[1] + 1
    np.testing.assert_allclose(logits.cpu().numpy(), logits_ref, atol=2e-5)
    np.testing.assert_allclose(outputs.cpu().numpy(), out_ref, atol=2e-5)
    for t, ref in enumerate(states_ref):
        for key in ("M", "w", "read", "controller_state"):
            np.testing.assert_allclose(states[t + 1][key].cpu().numpy(), ref[key], atol=2e-5, err_msg="state %d %s" % (t + 1, key))
    np.testing.assert_allclose(states[-1]["M"].cpu().numpy(), fin["M"], atol=2e-5)
    assert debugs["M"].shape[:2] == (B, x.shape[1])
    # PlainNTMTracker on the same rows gives the same thing
    plain = PlainNTMTracker(x.shape[1], F + 1, device=cuda, seed=3, **kw)
    o2, l2, s2, _d = plain(torch.from_numpy(x).to(cuda))
    assert torch.equal(l2, logits) and torch.equal(o2, outputs) and len(s2) == x.shape[1] + 1
    assert all(torch.equal(s2[t][key], states[t][key]) for t in range(1, len(s2)) for key in ("M", "w", "read", "controller_state"))


def test_two_step_tracker_with_input_compressor_gradients(cuda):
    """main.py:882-887: a 1x1 convolution compresses the feature map in front of the two-step tracker and trains with it:
    loss, the compressor's weight gradient and the cell's gradients vs torch autograd on the oracle."""
    from ntmtrack import twostep
    B, T, F, C, Cd = 2, 3, 9, 16, 4
    rng = np.random.default_rng(9)
    comp = twostep.InputCompressor(C, Cd, device=cuda, seed=2)
    trk = twostep.NTMTwoStepTracker(B, T, F, F * Cd, mem_size=64, mem_dim=8, hidden_size=32, read_head_size=2, write_head_size=1,
                                    init_scale=0.2, device=cuda, seed=5, compressor=comp)
    sd = {k: v.numpy() for k, v in trk.cell.state_dict().items()}
    cfg = O.NTMConfig(1 + F * Cd + F, F + 1, mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=32, controller_num_layers=1,
                      write_head_size=1, read_head_size=2)
    fmap = np.maximum(rng.standard_normal((B, T, F, C)), 0).astype(np.float32)
    gts = (rng.uniform(0, 1, size=(B, T, F)) > 0.7).astype(np.float32)
    w = comp.w().cpu().numpy().reshape(C, Cd)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sd.items()}
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    feat = (torch.tensor(fmap, dtype=torch.float64) @ wt).reshape(B, T, F * Cd)
    # the oracle's two_step_inputs on torch tensors
    x = torch.zeros((B, 2 * T - 1, 1 + F * Cd + F), dtype=torch.float64)
    x[:, 0, 1:1 + F * Cd] = feat[:, 0]
    x[:, 0, 1 + F * Cd:] = torch.tensor(gts[:, 0], dtype=torch.float64)
    for t in range(1, T):
        x[:, 2 * t - 1, 1:1 + F * Cd] = feat[:, t]
        x[:, 2 * t, 0] = 1
    np.testing.assert_allclose(x.detach().numpy(), O.two_step_inputs(feat.detach().numpy(), gts[:, 0].astype(np.float64)), atol=0)
    logits, _ = OT.loop(cfg, pt, x)
    q = torch.softmax(torch.tensor(O.two_step_labels(gts.astype(np.float64))), dim=2)
    loss_ref = -(q * torch.log_softmax(logits, dim=2)).sum() / ((2 * T - 1) * B)
    loss_ref.backward()
    loss, _p = trk.loss_and_grads(torch.from_numpy(fmap).to(cuda), torch.from_numpy(gts).to(cuda))
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), float(loss_ref.detach()), rtol=1e-4)
    gw = comp.grad.t().cpu().numpy()                                    # [C, Cd]
    ref = wt.grad.numpy()
    assert np.max(np.abs(gw - ref)) / np.max(np.abs(ref)) < 3e-3
    got = trk.cell.params.to_tf(grad=True)
    gmax = max(float(np.abs(pt[k].grad.numpy()).max()) for k in sd)
    for k in sorted(sd):
        r = pt[k].grad.numpy()
        assert np.max(np.abs(got[k].numpy() - r)) / max(np.max(np.abs(r)), 1e-3 * gmax) < 3e-3, k

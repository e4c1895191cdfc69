"""GPU parity of the training path: serialiser, loss, BPTT gradients, clip + RMSProp
against the numpy / torch-autograd oracles."""
import numpy as np
import pytest
import torch

from oracle import ntm_oracle as O
from oracle import ntm_oracle_torch as OT

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def test_gather_serialize_matches_oracle(cuda):
    from ntmtrack import tracker
    rng = np.random.default_rng(1)
    B, T, C = 2, 3, 512
    fmap = rng.standard_normal((B * T, 28, 28, C)).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
    ref = O.serialize_inputs(O.extract_features(fmap).reshape(B, T, 64, C), gts)
    X = tracker.gather_serialize(torch.from_numpy(fmap).to(cuda), torch.from_numpy(gts[:, 0].copy()).to(cuda), B, T, 516)
    torch.cuda.synchronize()
    got = X.cpu().numpy()
    assert got.shape == (B, T * 65, 516)
    assert np.array_equal(got[:, :, :514], ref)          # pure copies: bit exact
    assert np.all(got[:, :, 514:] == 0)


def test_offset_loss_matches_oracle(cuda):
    from ntmtrack import tracker
    rng = np.random.default_rng(2)
    B, T = 3, 5
    logits = rng.standard_normal((B, T * 65, 2)).astype(np.float32)
    offs = rng.uniform(-0.5, 0.5, size=(B, T, 2)).astype(np.float32)
    loss_ref, pred_ref = O.offset_loss(logits.astype(np.float64), offs.astype(np.float64))
    loss, pred, dlog = tracker.offset_loss(torch.from_numpy(logits).to(cuda), torch.from_numpy(offs).to(cuda), T)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), loss_ref, rtol=1e-5)
    np.testing.assert_allclose(pred.cpu().numpy(), pred_ref, atol=1e-6)
    lt = torch.tensor(logits, dtype=torch.float64, requires_grad=True)
    l2, _ = OT.offset_loss(lt, torch.tensor(offs, dtype=torch.float64))
    l2.backward()
    np.testing.assert_allclose(dlog.cpu().numpy(), lt.grad.numpy(), atol=1e-6)


GRAD_CASES = [
    ("c2_shape_T3", dict(mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=200, controller_num_layers=1,
                         write_head_size=1, read_head_size=4), 514, 3, 2, 0.05),
    ("r1w1_T2", dict(mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=100, controller_num_layers=1,
                     write_head_size=1, read_head_size=1), 514, 2, 2, 0.1),
    ("write_first_2w", dict(mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=64, controller_num_layers=1,
                            write_head_size=2, read_head_size=2, write_first=True), 514, 2, 3, 0.2),
    ("shift_range_3", dict(mem_size=64, mem_dim=8, shift_range=3, controller_hidden_size=48, controller_num_layers=1,
                           write_head_size=1, read_head_size=2), 514, 2, 2, 0.2),
    ("shift_range_4", dict(mem_size=128, mem_dim=8, shift_range=4, controller_hidden_size=48, controller_num_layers=1,
                           write_head_size=2, read_head_size=1, write_first=True), 514, 2, 2, 0.2),
]


@pytest.mark.parametrize("name,kw,D,T,B,scale", GRAD_CASES, ids=[c[0] for c in GRAD_CASES])
def test_bptt_gradients_match_autograd_oracle(cuda, name, kw, D, T, B, scale):
    from ntmtrack.ntm import NTMCell
    from ntmtrack import tracker
    cfg = O.NTMConfig(D, 2, **kw)
    rng = np.random.default_rng(21)
    params = O.init_params(cfg, rng, scale=scale)
    for k in params:
        if k.endswith("biases"):
            params[k] = rng.uniform(-scale, scale, size=params[k].shape).astype(np.float32)
    S = T * 65
    # serialised inputs of the tracking task: sparse relu-like features + delimiter/target columns
    feats = np.maximum(rng.standard_normal((B, T, 64, 512)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
    x = O.serialize_inputs(feats, gts)
    offs = rng.uniform(-0.5, 0.5, size=(B, T, 2)).astype(np.float32)
    loss_ref, grads_ref, logits_ref, _ = OT.loss_and_grads(cfg, params, x, offs)

    cell = NTMCell(2, mem_size=cfg.mem_size, mem_dim=cfg.mem_dim, shift_range=cfg.shift_range,
                   controller_hidden_size=cfg.hidden, controller_num_layers=1, write_head_size=cfg.write_heads,
                   read_head_size=cfg.read_heads, write_first=cfg.write_first, device=cuda)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, input_dim=D)
    X = cell._pad_inputs(torch.from_numpy(x).to(cuda))
    st0 = cell.zero_state(B)
    logits, _o, _new, rec = cell.run_sequence(X, st0, record=True)
    loss, pred, dlogits = tracker.offset_loss(logits, torch.from_numpy(offs).to(cuda), T)
    g0 = cell.backward_sequence(X, st0, rec, dlogits)
    cell.init_state_backward(g0, B)
    torch.cuda.synchronize()
    np.testing.assert_allclose(logits.cpu().numpy(), logits_ref, atol=2e-5)
    np.testing.assert_allclose(float(loss.cpu()), loss_ref, rtol=1e-4)
    got = cell.params.to_tf(grad=True)
    # bound per tensor, relative to its largest entry: 1e-4 -- or, where the float32 evaluation of the SAME restatement is
    # itself further than that from float64 (sums with heavy cancellation), no further from float64 than 3x what it is
    _l32, grads32, _lg32, _p32 = OT.loss_and_grads(cfg, params, x, offs, dtype=torch.float32)
    worst, bad = {}, {}
    for k in sorted(grads_ref):
        err, err32 = _relerr(got[k].numpy(), grads_ref[k]), _relerr(grads32[k].astype(np.float64), grads_ref[k])
        worst[k] = (err, err32)
        if err > max(1e-4, 3 * err32):
            bad[k] = (err, err32)
    print("%s relative gradient error (HIP, float32 oracle) vs float64: %s" % (name, {k: ("%.1e" % a, "%.1e" % b_) for k, (a, b_) in worst.items()}))
    assert not bad, bad


def test_train_step_matches_oracle_update(cuda):
    """One full optimiser step (no VGG): clip_by_global_norm(5) + TF RMSProp on the packed buffer
    equals the oracle update applied to the autograd gradients."""
    from ntmtrack import tracker
    name, kw, D, T, B, scale = GRAD_CASES[0]
    cfg = O.NTMConfig(D, 2, **kw)
    rng = np.random.default_rng(33)
    params = O.init_params(cfg, rng, scale=0.3)       # large weights -> gradient norm above the clip
    feats = np.maximum(rng.standard_normal((B, T, 64, 512)), 0).astype(np.float32) * 3
    gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
    offs = rng.uniform(-0.5, 0.5, size=(B, T, 2)).astype(np.float32)
    x = O.serialize_inputs(feats, gts)
    loss_ref, grads_ref, _, _ = OT.loss_and_grads(cfg, params, x, offs)
    names = sorted(grads_ref)
    clipped, gn = O.clip_by_global_norm([grads_ref[k] for k in names], 5.0)
    new_ref = {}
    for k, g in zip(names, clipped):
        pnew, _, _ = O.rmsprop_step(params[k].astype(np.float64), g, np.ones_like(g), np.zeros_like(g))
        new_ref[k] = pnew

    trk = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=cuda)
    flat_before = trk.cell.params.flat
    # the documented way to import reference weights; the optimiser built in __init__ must stay bound
    trk.cell.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, input_dim=D)
    assert trk.cell.params.flat is flat_before and trk.opt.p is trk.cell.params
    # feed the features through a fake conv4_3 map so the gather kernel is on the path too
    fmap = np.zeros((B * T, 28, 28, 512), np.float32)
    for i, (y, xx) in enumerate(O.CONV43_POINTS):
        fmap[:, y, xx, :] = feats.reshape(B * T, 64, 512)[:, i]
    loss, _ = trk.loss_and_grads(torch.from_numpy(fmap).to(cuda), torch.from_numpy(gts[:, 0].copy()).to(cuda),
                                 torch.from_numpy(offs).to(cuda))
    trk.opt.step()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(trk.opt.gnorm.cpu()), gn, rtol=2e-3)
    assert gn > 5.0
    got = trk.cell.state_dict()
    for k in names:
        delta_ref = new_ref[k] - params[k]
        err = np.max(np.abs(got[k].numpy().astype(np.float64) - new_ref[k]))
        # fp32 storage of the parameter (|p| <= 0.3 -> half an ulp = 1.5e-8) plus 0.5 % of the step
        assert err <= 3e-8 + 5e-3 * np.max(np.abs(delta_ref)), "%s: %.3e (step %.3e)" % (k, err, np.max(np.abs(delta_ref)))


def test_checkpoint_roundtrip_resumes_bit_identically(cuda, tmp_path):
    """save after step 1, keep training to step 3; a fresh tracker restored from the checkpoint reaches the same
    parameters bit for bit (parameters, RMSProp slots and global_step all restored)."""
    from ntmtrack import tracker
    B, T = 2, 2
    g = torch.Generator().manual_seed(1)
    fmap = torch.relu(torch.randn((B * T, 28, 28, 512), generator=g)).to(cuda)
    gts0 = torch.rand((B, 64), generator=g).to(cuda)
    offs = (torch.rand((B, T, 2), generator=g) - 0.5).to(cuda)
    a = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=cuda, seed=3, learning_rate=1e-2)
    a.loss_and_grads(fmap, gts0, offs); a.opt.step()
    path = a.save_checkpoint(str(tmp_path / "ck.pt"))
    for _ in range(2):
        a.loss_and_grads(fmap, gts0, offs); a.opt.step()
    b = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=cuda, seed=99, learning_rate=1e-2)
    b.load_checkpoint(path)
    assert b.opt.global_step == 1
    for _ in range(2):
        b.loss_and_grads(fmap, gts0, offs); b.opt.step()
    torch.cuda.synchronize()
    assert torch.equal(a.cell.params.flat, b.cell.params.flat) and torch.equal(a.opt.ms, b.opt.ms)
    d = tracker.DNCOffsetTracker(B, T, vgg_weights=None, mem_size=32, mem_dim=16, hidden_size=32, device=cuda)
    with pytest.raises(Exception):
        d.load_checkpoint(path)                      # wrong tracker kind is refused


def test_load_state_dict_keeps_the_optimiser_bound(cuda):
    """After tracker.load_state_dict / cell.load_state_dict the model must still train: the optimiser updates the
    buffer the kernels read (same layout -> same buffer; different layout -> the tracker re-binds)."""
    from ntmtrack import tracker
    B, T = 2, 2
    g = torch.Generator().manual_seed(5)
    fmap = torch.relu(torch.randn((B * T, 28, 28, 512), generator=g)).to(cuda)
    gts0 = torch.rand((B, 64), generator=g).to(cuda)
    offs = (torch.rand((B, T, 2), generator=g) - 0.5).to(cuda)
    src = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=cuda, seed=7)
    for trk in (tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=cuda, seed=1, learning_rate=1e-2),
                tracker.DNCOffsetTracker(B, T, vgg_weights=None, mem_size=32, mem_dim=16, hidden_size=32, device=cuda,
                                         learning_rate=1e-2)):
        sd = (src if isinstance(trk, tracker.NTMOffsetTracker) else trk).state_dict()
        sd = {k: v + 0.01 for k, v in sd.items()}
        trk.load_state_dict(sd)
        core = trk._core()
        assert trk.opt.p is core.params
        before = core.params.flat.clone()
        trk.loss_and_grads(fmap, gts0, offs)
        trk.opt.step()
        torch.cuda.synchronize()
        assert float((core.params.flat - before).abs().max()) > 0, "parameters did not move after load_state_dict"
        got = trk.state_dict()
        assert any(float((got[k] - sd[k]).abs().max()) > 0 for k in sd)


def test_full_length_bptt_gradients_match_autograd_oracle(cuda):
    """BASELINE config 2's full horizon: T = 20 frames -> S = 1300 strictly sequential steps of BPTT, B = 2, gradients
    of every tensor against the float64 torch-autograd restatement (direct_offset_output.py:611-621).  The bound (5e-5 per
    tensor, relative to the tensor's largest entry) is ~10x what fp32 storage + fp32 accumulation over 1300 dependent
    steps measures against float64; the per-tensor errors are printed."""
    from ntmtrack.ntm import NTMCell
    from ntmtrack import tracker
    name, kw, D, _T, _B, scale = GRAD_CASES[0]
    T, B = 20, 2
    cfg = O.NTMConfig(D, 2, **kw)
    rng = np.random.default_rng(77)
    params = O.init_params(cfg, rng, scale=scale)
    for k in params:
        if k.endswith("biases"):
            params[k] = rng.uniform(-scale, scale, size=params[k].shape).astype(np.float32)
    feats = np.maximum(rng.standard_normal((B, T, 64, 512)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
    x = O.serialize_inputs(feats, gts)
    offs = rng.uniform(-0.5, 0.5, size=(B, T, 2)).astype(np.float32)
    loss_ref, grads_ref, logits_ref, _ = OT.loss_and_grads(cfg, params, x, offs)

    cell = NTMCell(2, mem_size=cfg.mem_size, mem_dim=cfg.mem_dim, shift_range=cfg.shift_range,
                   controller_hidden_size=cfg.hidden, controller_num_layers=1, write_head_size=cfg.write_heads,
                   read_head_size=cfg.read_heads, write_first=cfg.write_first, device=cuda)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, input_dim=D)
    X = cell._pad_inputs(torch.from_numpy(x).to(cuda))
    st0 = cell.zero_state(B)
    logits, _o, _new, rec = cell.run_sequence(X, st0, record=True)
    assert logits.shape == (B, 1300, 2)
    loss, pred, dlogits = tracker.offset_loss(logits, torch.from_numpy(offs).to(cuda), T)
    g0 = cell.backward_sequence(X, st0, rec, dlogits)
    cell.init_state_backward(g0, B)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), loss_ref, rtol=1e-4)
    got = cell.params.to_tf(grad=True)
    worst = {k: _relerr(got[k].numpy(), grads_ref[k]) for k in sorted(grads_ref)}
    print("full-length (S=1300) NTM gradient error vs float64 autograd, max|d|/max|ref| per tensor:")
    for k, v in worst.items():
        print("  %-24s %.3e" % (k, v))
    assert max(worst.values()) < 5e-5, worst          # measured 2e-8 .. 4e-6 (GPUTEST r2)


@pytest.mark.parametrize("layers", [2, 3])
def test_deep_controller_bptt_matches_autograd_oracle(cuda, layers):
    """controller_num_layers > 1 (MultiRNNCell of BasicLSTMCells, ntm_cell.py:45-50; the constructor default is 10):
    loss and the gradient of EVERY variable, lower LSTM layers included, through the tracking head over T = 2 frames
    (130 steps) against the float64 torch-autograd restatement; then one clipped RMSProp step moves all of them."""
    from ntmtrack import tracker
    from ntmtrack.ntm import StackedNTMCell
    B, T = 2, 2
    kw = dict(mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=32, controller_num_layers=layers,
              write_head_size=1, read_head_size=2)
    cfg = O.NTMConfig(514, 2, **kw)
    rng = np.random.default_rng(50 + layers)
    params = O.init_params(cfg, rng, scale=0.15)
    for k in params:
        if k.endswith("biases"):
            params[k] = rng.uniform(-0.15, 0.15, size=params[k].shape).astype(np.float32)
    feats = np.maximum(rng.standard_normal((B, T, 64, 512)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
    x = O.serialize_inputs(feats, gts)
    offs = rng.uniform(-0.5, 0.5, size=(B, T, 2)).astype(np.float32)
    loss_ref, grads_ref, logits_ref, _ = OT.loss_and_grads(cfg, params, x, offs)
    assert any(k.startswith("lstm/cell_%d/" % (layers - 1)) for k in grads_ref)

    trk = tracker.NTMOffsetTracker(B, T, vgg_weights=None, mem_size=64, mem_dim=8, hidden_size=32, num_layers=layers,
                                   read_head_size=2, write_head_size=1, device=cuda, learning_rate=1e-2)
    assert isinstance(trk.cell, StackedNTMCell)
    trk.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    assert trk.opt.p is trk.cell.params
    fmap = np.zeros((B * T, 28, 28, 512), np.float32)
    for i, (y, xx) in enumerate(O.CONV43_POINTS):
        fmap[:, y, xx, :] = feats.reshape(B * T, 64, 512)[:, i]
    loss, _ = trk.loss_and_grads(torch.from_numpy(fmap).to(cuda), torch.from_numpy(gts[:, 0].copy()).to(cuda),
                                 torch.from_numpy(offs).to(cuda))
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), loss_ref, rtol=1e-4)
    got = trk.cell.state_dict(grad=True)
    assert sorted(got) == sorted(grads_ref)
    for k in sorted(grads_ref):
        err = _relerr(got[k].numpy(), grads_ref[k])
        assert err < 2e-3, "%s: relative error %.3e" % (k, err)
    before = {k: v.clone() for k, v in trk.cell.state_dict().items()}
    trk.opt.step()
    torch.cuda.synchronize()
    after = trk.cell.state_dict()
    gmax = max(float(np.abs(g).max()) for g in grads_ref.values())
    for k in before:
        if float(np.abs(grads_ref[k]).max()) > 1e-3 * gmax:          # a vanishing gradient may not change an fp32 parameter
            assert float((after[k] - before[k]).abs().max()) > 0, "%s did not move" % k
    assert all(float((after["lstm/cell_%d/weights" % l] - before["lstm/cell_%d/weights" % l]).abs().max()) > 0 for l in range(layers))

"""CPU: pin the NTM oracle against the reference's own vectors, cross-check the two independent
restatements, and freeze it against the committed golden vectors."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ntm_oracle as O
from oracle import ntm_oracle_torch as OT

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = json.load(open(os.path.join(GOLD, "reference_vectors.json")))


def test_smooth_cosine_matches_reference_test_vector():
    """ops_test.py:20-34 pins the opt-in smooth-cosine mode ..."""
    v = REF["ops_test.py:20-34 (Torch7 nn.SmoothCosineSimilarity values; contradicts shipped ops.py, see SURVEY 4)"]
    mem, keys = np.array(v["memory"], np.float32), np.array(v["keys"], np.float32)
    got = O.batched_smooth_cosine_similarity(mem, keys, mode="smooth_cosine")
    np.testing.assert_allclose(got, np.array(v["expected"]), atol=1e-5)
    # ... and is a NEGATIVE test for the shipped code (quirk Q1): as coded the values differ
    as_coded = O.batched_smooth_cosine_similarity(mem, keys)
    assert np.max(np.abs(as_coded - np.array(v["expected"]))) > 0.1
    hand = REF["ops.py:147-156 as coded, hand-evaluated on the same inputs (SURVEY 4)"]["expected"]
    np.testing.assert_allclose(as_coded, np.array(hand), atol=1e-4)


def test_as_coded_similarity_normalises_feature_columns_over_slots():
    rng = np.random.default_rng(0)
    M = rng.standard_normal((2, 16, 5))
    k = rng.standard_normal((2, 3, 5))
    sim = O.batched_smooth_cosine_similarity(M, k)
    Mh = M / np.sqrt((M ** 2).sum(axis=1, keepdims=True))      # each column (feature) over the N slots
    kh = k / np.sqrt((k ** 2).sum(axis=2, keepdims=True))
    np.testing.assert_allclose(sim, np.einsum("bhm,bnm->bhn", kh, Mh), atol=1e-12)


def test_shift_taps_follow_python2_floor_division():
    """ops.py:204-209 without `from __future__ import division`: -3/2 == -2 -> taps (-2,-1,0)."""
    assert O.shift_offsets(3) == [-2, -1, 0]
    assert O.shift_offsets(5) == [-3, -2, -1, 0, 1]
    w = np.arange(6, dtype=np.float64)[None, None, :]
    k = np.array([[[0.5, 0.3, 0.2]]])
    out = O.batched_circular_convolution(w, k)[0, 0]
    exp = np.array([0.5 * w[0, 0, (i - 2) % 6] + 0.3 * w[0, 0, (i - 1) % 6] + 0.2 * w[0, 0, i] for i in range(6)])
    np.testing.assert_allclose(out, exp)
    np.testing.assert_array_equal(O.circular_shift(np.arange(5), -2), [3, 4, 0, 1, 2])


def test_numpy_and_torch_restatements_agree():
    cfg = O.NTMConfig(10, 2, mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=32,
                      controller_num_layers=2, write_head_size=2, read_head_size=3)
    rng = np.random.default_rng(0)
    p = O.init_params(cfg, rng, scale=0.3, dtype=np.float64)
    x = rng.standard_normal((2, 9, 10))
    _, l, st = O.loop_ntm_tracker(cfg, p, x)
    lt, stt = OT.loop(cfg, {k: torch.tensor(v) for k, v in p.items()}, torch.tensor(x))
    np.testing.assert_allclose(l, lt.numpy(), atol=1e-12)
    np.testing.assert_allclose(st["M"], stt["M"].numpy(), atol=1e-12)
    # write_first variant
    cfg.write_first = True
    _, l2, _ = O.loop_ntm_tracker(cfg, p, x)
    lt2, _ = OT.loop(cfg, {k: torch.tensor(v) for k, v in p.items()}, torch.tensor(x))
    np.testing.assert_allclose(l2, lt2.numpy(), atol=1e-12)
    assert np.max(np.abs(l2 - l)) > 1e-6


def test_zero_state_is_trainable_unnormalised_and_tiled():
    cfg = O.NTMConfig(4, 2, mem_size=8, mem_dim=3, controller_hidden_size=5, controller_num_layers=1,
                      write_head_size=1, read_head_size=2)
    p = O.init_params(cfg, np.random.default_rng(1))
    st = O.zero_state(cfg, p, 3)
    np.testing.assert_allclose(st["M"][2], np.tanh(p["init_state/M"]))
    np.testing.assert_allclose(st["w"][1], 1 / (1 + np.exp(-p["init_state/w"])), rtol=1e-6)
    assert abs(st["w"][0].sum(axis=1)[0] - 1.0) > 0.5          # Q5: sigmoid, NOT a distribution
    assert st["controller_state"].shape == (3, 10) and not st["controller_state"].any()


def test_serializer_layout():
    """direct_offset_output.py:439-500: 64 feature rows [feat,0,tgt] then the delimiter row [0..0,1,0]."""
    rng = np.random.default_rng(2)
    B, T, C = 2, 3, 6
    feats = rng.standard_normal((B, T, 64, C)).astype(np.float32)
    gts = rng.uniform(size=(B, T, 64)).astype(np.float32)
    X = O.serialize_inputs(feats, gts)
    assert X.shape == (B, T * 65, C + 2)
    for t in range(T):
        np.testing.assert_array_equal(X[:, t * 65:t * 65 + 64, :C], feats[:, t])
        np.testing.assert_array_equal(X[:, t * 65 + 64, :C], 0)
        assert (X[:, t * 65 + 64, C] == 1).all() and (X[:, t * 65:t * 65 + 64, C] == 0).all()
    np.testing.assert_array_equal(X[:, :64, C + 1], gts[:, 0])
    assert not X[:, 64:, C + 1].any()                           # target only on frame 0 (:492-500)


def test_offset_loss_reads_delimiter_steps_of_frames_1_onwards():
    B, T = 2, 4
    logits = np.zeros((B, T * 65, 2))
    offs = np.zeros((B, T, 2))
    logits[:, 64, :] = 5.0                                      # frame 0's delimiter step is dropped (:581)
    loss, pred = O.offset_loss(logits, offs)
    assert loss == 0 and pred.shape == (B, T - 1, 2)
    logits[1, 2 * 65 + 64, 0] = 0.3
    loss, pred = O.offset_loss(logits, offs)
    np.testing.assert_allclose(loss, 0.5 * np.tanh(0.3) ** 2)
    assert pred[1, 1, 0] == np.tanh(0.3)


def test_extract_features_points():
    assert O.CONV43_POINTS[0] == (6, 6) and O.CONV43_POINTS[7] == (6, 20) and O.CONV43_POINTS[8] == (8, 6)
    assert O.CONV43_POINTS[-1] == (20, 20) and len(O.CONV43_POINTS) == 64
    fm = np.arange(2 * 28 * 28 * 3).reshape(2, 28, 28, 3)
    f = O.extract_features(fm)
    np.testing.assert_array_equal(f[1, 9], fm[1, 8, 8])


def test_rmsprop_and_clip_formulas():
    g = [np.array([3.0, 4.0]), np.array([12.0])]
    clipped, gn = O.clip_by_global_norm(g, 5.0)
    assert gn == 13.0
    np.testing.assert_allclose(clipped[0], np.array([3.0, 4.0]) * 5 / 13)
    p, ms, mom = O.rmsprop_step(np.array([1.0]), np.array([2.0]), np.ones(1), np.zeros(1),
                                lr=0.1, decay=0.9, momentum=0.5, eps=0.0)
    np.testing.assert_allclose(ms, 0.9 + 0.1 * 4)               # ms slot starts at ONE
    np.testing.assert_allclose(mom, 0.1 * 2 / np.sqrt(1.3))
    np.testing.assert_allclose(p, 1 - mom)


def test_autograd_oracle_gradients_by_finite_differences():
    cfg = O.NTMConfig(6, 2, mem_size=8, mem_dim=4, shift_range=1, controller_hidden_size=5,
                      controller_num_layers=1, write_head_size=1, read_head_size=2)
    rng = np.random.default_rng(5)
    p = O.init_params(cfg, rng, scale=0.5, dtype=np.float64)
    x = rng.standard_normal((2, 4, 6))

    def loss_of(pp):
        _, l, _ = O.loop_ntm_tracker(cfg, pp, x)
        return 0.5 * float((np.tanh(l) ** 2).sum())

    pt = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    lt, _ = OT.loop(cfg, pt, torch.tensor(x))
    (0.5 * (torch.tanh(lt) ** 2).sum()).backward()
    for name in ("init_state/M", "addressing/weights", "lstm/cell_0/biases", "init_state/w"):
        g = pt[name].grad.numpy()
        idxs = [tuple(rng.integers(0, s) for s in p[name].shape) for _ in range(4)]
        for idx in idxs:
            d = 1e-6
            pp = {k: v.copy() for k, v in p.items()}
            pp[name][idx] += d
            up = loss_of(pp)
            pp[name][idx] -= 2 * d
            dn = loss_of(pp)
            np.testing.assert_allclose(g[idx], (up - dn) / (2 * d), rtol=2e-4, atol=1e-9)


def test_vgg_numpy_vs_torch_conv_restatement():
    rng = np.random.default_rng(3)
    ws = O.init_vgg_weights(rng)
    frame = (rng.uniform(0, 255, size=(1, 16, 16, 3)).astype(np.float32) - O.VGG_MEAN)
    a = O.vgg16_conv43(frame, ws)
    b = OT.vgg16_conv43(frame, ws)
    assert a.shape == (1, 2, 2, 512)
    np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-3 * np.abs(a).max())


def test_vgg_bf16_numpy_vs_torch_restatement():
    """The two restatements of config 5's bf16 trunk (numpy, torch float64 conv2d) agree: the torch one is what the
    GPU test uses at 224x224, where the numpy one would take minutes."""
    rng = np.random.default_rng(4)
    ws = O.init_vgg_weights(rng)
    frame = (rng.uniform(0, 255, size=(1, 16, 16, 3)).astype(np.float32) - O.VGG_MEAN)
    a = O.vgg16_conv43_bf16(frame, ws)
    b = OT.vgg16_conv43_bf16(frame, ws)
    # identical bf16 roundings except where a float64 sum lands within an ulp of a rounding boundary
    assert np.max(np.abs(a - b)) <= 2e-2 * np.abs(a).max()
    assert np.mean(np.abs(a - b)) <= 1e-4 * np.abs(a).max()


# ---- golden vectors (regression freeze of the oracle; generated by tests/golden/make_golden.py)
def test_golden_ntm_seq_small():
    g = np.load(os.path.join(GOLD, "ntm_seq_small.npz"))
    cfg = O.NTMConfig(10, 2, mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=32,
                      controller_num_layers=1, write_head_size=1, read_head_size=2)
    p = {k[len("param:"):]: g[k].astype(np.float64) for k in g.files if k.startswith("param:")}
    outs, logits, fin = O.loop_ntm_tracker(cfg, p, g["x"].astype(np.float64))
    np.testing.assert_allclose(logits, g["logits"], atol=1e-12)
    np.testing.assert_allclose(fin["M"], g["M_final"], atol=1e-12)


def test_golden_ntm_step_c2_and_vgg():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(GOLD, "ntm_step_c2.npz"))
    cfg, p = mg.c2_params(int(g["seed"]))
    st = {"M": g["M"], "w": g["w"], "read": g["read"], "controller_state": g["cs"]}
    out, logit, new, dbg = O.ntm_step(cfg, p, g["x"], st)           # float32 run vs float64 golden
    np.testing.assert_allclose(logit, g["logit"], atol=2e-6)
    np.testing.assert_allclose(new["M"], g["M_new"], atol=2e-6)
    np.testing.assert_allclose(new["w"], g["w_new"], atol=2e-6)
    v = np.load(os.path.join(GOLD, "vgg_small.npz"))
    ws = O.init_vgg_weights(np.random.default_rng(int(v["seed"])))
    f = O.vgg16_conv43(v["frame"], ws)
    np.testing.assert_allclose(f, v["conv4_3"], rtol=1e-4, atol=1e-4 * np.abs(v["conv4_3"]).max())


def test_sequential_serialisation_layout():
    """main.py:1701-1775: S = F + (T-1)(2F+1); frame 0 = F feature rows carrying the target; later frames = frame delimiter,
    then (feature, feature delimiter) pairs; the heat-map gather (main.py:1880-1897) picks the feature-delimiter steps."""
    rng = np.random.default_rng(0)
    B, T, F, C = 2, 3, 4, 5
    feats = rng.standard_normal((B, T, F, C)).astype(np.float32)
    gts = rng.uniform(size=(B, T, F)).astype(np.float32)
    x = O.serialize_sequential(feats, gts)
    assert x.shape == (B, F + (T - 1) * (2 * F + 1), C + 3)
    assert np.array_equal(x[:, :F, :C], feats[:, 0]) and np.array_equal(x[:, :F, C + 2], gts[:, 0]) and not x[:, F:, C + 2].any()
    for t in range(1, T):
        base = F + (t - 1) * (2 * F + 1)
        assert np.all(x[:, base, C + 1] == 1) and not x[:, base, :C + 1].any()
        for i in range(F):
            assert np.array_equal(x[:, base + 1 + 2 * i, :C], feats[:, t, i]) and not x[:, base + 1 + 2 * i, C:].any()
            assert np.all(x[:, base + 2 + 2 * i, C] == 1) and not x[:, base + 2 + 2 * i, :C].any()
    logits = np.arange(B * x.shape[1], dtype=np.float64).reshape(B, -1, 1)
    g = O.heatmap_gather(logits, T, F)
    assert g.shape == (B, T - 1, F) and g[0, 0, 0] == F + 2 and g[0, 1, 3] == F + (2 * F + 1) + 2 + 6
    loss, p = O.heatmap_ce_loss(np.zeros((B, x.shape[1], 1)), np.full((B, T - 1, F), 1.0 / F), T)
    np.testing.assert_allclose(loss, B * np.log(F))              # uniform scores: (T-1) B log F / (T-1)
    np.testing.assert_allclose(p, 1.0 / F)


def test_two_step_layout_known_answers():
    """main.py:903-934 / ntm_tracker_new.py:150-181 on a hand-written case: T = 3 frames, F = 2 positions, D = 2."""
    feat = np.arange(12, dtype=np.float32).reshape(1, 3, 4)[:, :, :2] + 1          # frames [1,2], [5,6], [9,10]
    target = np.array([[0.25, 0.75]], np.float32)
    X = O.two_step_inputs(feat, target)
    assert X.shape == (1, 5, 5)
    np.testing.assert_array_equal(X[0], np.array([[0, 1, 2, 0.25, 0.75],        # frame 0 with the target
                                                   [0, 5, 6, 0, 0],              # frame 1 shown
                                                   [1, 0, 0, 0, 0],              # ... and asked for
                                                   [0, 9, 10, 0, 0],
                                                   [1, 0, 0, 0, 0]], np.float32))
    gt = np.array([[[9, 9], [1, 0], [0, 1]]], np.float32)                          # row 0 (the target frame) is not a label
    lab = O.two_step_labels(gt)
    np.testing.assert_array_equal(lab[0], np.array([[0, 0, 1], [0, 0, 1], [1, 0, 0], [0, 0, 1], [0, 1, 0]], np.float32))
    # the loss softmaxes the labels (as coded): a one-hot row becomes [e, 1, 1] / (e + 2)
    logits = np.zeros((1, 5, 3))
    loss, p, dl = O.two_step_ce_loss(logits, gt.astype(np.float64))
    assert abs(loss - np.log(3.0)) < 1e-12 and np.allclose(p, 1 / 3)             # uniform logits: CE = log 3 whatever the labels
    q = np.exp([0, 0, 1.0]) / (np.e + 2)
    np.testing.assert_allclose(dl[0, 0], (1 / 3 - q) / 5, atol=1e-15)


"""CPU: pin the DNC oracle with the reference's own module tests (dnc/*_test.py), restated on the
numpy oracle: fixed vectors, planted one-hot cases and the properties those tests assert."""
import json
import os

import numpy as np

from oracle import dnc_oracle as D

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF = json.load(open(os.path.join(GOLD, "reference_vectors.json")))


def test_batch_gather_fixed_vector():
    v = REF["dnc/util_test.py:51-53 batch_gather"]
    np.testing.assert_array_equal(D.batch_gather(np.array(v["values"]), np.array(v["indices"])), np.array(v["target"]))


def test_batch_invert_permutation():                      # dnc/util_test.py:27-45
    rng = np.random.default_rng(0)
    perms = np.stack([rng.permutation(7) for _ in range(5)])
    inv = D.batch_invert_permutation(perms)
    for i in range(5):
        for j in range(7):
            assert perms[i][inv[i][j]] == j


def test_weighted_softmax_identity():                      # addressing_test.py:31-53
    rng = np.random.default_rng(1)
    a = rng.standard_normal((5, 3, 7))
    np.testing.assert_allclose(D.weighted_softmax(a, np.ones((5, 3)), lambda x: x), D.softmax(a, 2))


def test_cosine_weights_values():                          # addressing_test.py:72-118 (explicit numpy loop)
    rng = np.random.default_rng(2)
    B, H, N, W = 5, 4, 10, 2
    mem = rng.standard_normal((B, N, W))
    mem[0, 0], mem[0, 1], mem[0, 2] = [1, 2], [3, 4], [5, 6]
    keys = rng.standard_normal((B, H, W))
    keys[0, 0], keys[0, 1], keys[0, 2], keys[0, 3] = [5, 6], [1, 2], [5, 6], [3, 4]
    strengths = rng.standard_normal((B, H))
    res = D.cosine_weights(mem, keys, strengths)
    sp = np.log(1 + np.exp(strengths))
    for b in range(B):
        for h in range(H):
            sim = np.array([np.dot(keys[b, h], mem[b, m]) / (np.linalg.norm(keys[b, h]) * np.linalg.norm(mem[b, m]))
                            for m in range(N)])
            sim = np.exp(sim * sp[b, h])
            sim /= sim.sum()
            np.testing.assert_allclose(res[b, h], sim, atol=1e-4, rtol=1e-4)


def test_cosine_weights_divide_by_zero():                  # addressing_test.py:120-145
    res = D.cosine_weights(np.zeros((5, 10, 2)), np.random.default_rng(3).standard_normal((5, 4, 2)),
                           np.random.default_rng(4).standard_normal((5, 4)))
    assert np.isfinite(res).all()


def test_temporal_linkage_planted_transitions():           # addressing_test.py:150-236
    rng = np.random.default_rng(5)
    B, N, R, Wn = 7, 4, 11, 5
    link = np.zeros((B, Wn, N, N))
    prec = np.zeros((B, Wn, N))
    steps = 5
    for i in range(steps):
        ww = rng.random((B, Wn, N))
        ww /= ww.sum(2, keepdims=True) + 1
        if i == steps - 2:
            ww[0, 0, :] = D.one_hot(N, 0)
            ww[0, 1, :] = D.one_hot(N, 3)
        elif i == steps - 1:
            ww[0, 0, :] = D.one_hot(N, 1)
            ww[0, 1, :] = D.one_hot(N, 2)
        link, prec = D.link_update(link, prec, ww), D.precedence_weights(prec, ww)
    assert link.min() >= 0 and link.max() <= 1
    np.testing.assert_array_equal(link[:, :, range(N), range(N)], np.zeros((B, Wn, N)))
    assert link.sum(2).max() <= 1 + 1e-12 and link.sum(3).max() <= 1 + 1e-12
    np.testing.assert_array_equal(link[0, 0, :, 0], D.one_hot(N, 1))     # transition 0 -> 1
    np.testing.assert_array_equal(link[0, 1, :, 3], D.one_hot(N, 2))     # transition 3 -> 2
    prw = rng.random((B, R, N))
    prw[0, 5, :] = D.one_hot(N, 0)
    prw[0, 6, :] = D.one_hot(N, 2)
    fwd = D.directional_read_weights(link, prw, True)
    bwd = D.directional_read_weights(link, prw, False)
    np.testing.assert_array_equal(fwd[0, 5, 0, :], D.one_hot(N, 1))
    np.testing.assert_array_equal(bwd[0, 6, 1, :], D.one_hot(N, 3))


def test_precedence_weights_cases():                       # addressing_test.py:238-272
    rng = np.random.default_rng(6)
    prev = rng.random((7, 5, 3))
    ww = rng.random((7, 5, 3))
    ww /= ww.sum(2, keepdims=True) + 1
    prev /= prev.sum(2, keepdims=True) + 1
    ww[0, 1, :] = 0
    ww[1, 2, :] /= ww[1, 2, :].sum()
    p = D.precedence_weights(prev, ww)
    assert p.min() >= 0 and p.max() <= 1
    np.testing.assert_allclose(p[0, 1], prev[0, 1])
    np.testing.assert_allclose(p[1, 2], ww[1, 2])


def test_freeness_full_write_and_full_free():              # addressing_test.py:277-314
    rng = np.random.default_rng(7)
    B, N, R, Wn = 5, 11, 3, 7
    fg = rng.random((B, R))
    prw = rng.random((B, R, N))
    prw[1, :, 3] = 0
    prw /= prw.sum(2, keepdims=True)
    pww = rng.random((B, Wn, N))
    pww /= pww.sum(2, keepdims=True)
    pu = rng.random((B, N))
    pww[1, 2, 3] = 1
    prw[2, 0, 4] = 1
    fg[2, 0] = 1
    u = D.freeness(pww, fg, prw, pu)
    assert u.min() >= 0 and u.max() <= 1
    assert u[1][3] == 1 and u[2][4] == 0


def test_write_allocation_weights_cases():                 # addressing_test.py:316-366
    rng = np.random.default_rng(8)
    B, N, Wn = 7, 23, 5
    usage = rng.random((B, N))
    wg = rng.random((B, Wn))
    wg[0, 1] = wg[0, 3] = 0
    wg[0, 0] = wg[0, 2] = 1
    usage[1] = usage[1] * 0.9 + 0.1
    usage[1][4] = 0
    usage[1][3] = 1e-4
    wg[1, 0] = wg[1, 1] = 1
    w = D.write_allocation_weights(usage, wg, Wn)
    assert w.min() >= 0 and w.max() <= 1
    np.testing.assert_allclose(w.sum(2), np.ones((B, Wn)), atol=1e-3)
    assert np.abs(w[0, 0] - w[0, 1]).max() > 0.1
    np.testing.assert_array_equal(w[0, 1], w[0, 2])
    assert np.abs(w[0, 2] - w[0, 3]).max() > 0.1
    np.testing.assert_array_equal(w[0, 3], w[0, 4])
    np.testing.assert_allclose(w[1][0], D.one_hot(N, 4), atol=1e-3)
    np.testing.assert_allclose(w[1][1], D.one_hot(N, 3), atol=1e-3)


def test_allocation_argmin_argmax_duality():               # addressing_test.py:387-401
    usage = np.random.default_rng(9).random((7, 13))
    a = D.allocation(usage)
    np.testing.assert_array_equal(np.argmin(usage, 1), np.argmax(a, 1))
    np.testing.assert_array_equal(np.argmax(usage, 1), np.argmin(a, 1))
    np.testing.assert_allclose(a.sum(1), np.ones(7), rtol=0.01)


def test_access_write_weights_planted():                   # access_test.py:77-111
    rng = np.random.default_rng(10)
    B, N, W, R, Wn = 2, 20, 6, 2, 3
    cfg = D.AccessConfig(N, W, R, Wn)
    memory = 10 * (rng.random((B, N, W)) - 0.5)
    usage = rng.random((B, N))
    ag, wg = rng.random((B, Wn)), rng.random((B, Wn))
    usage[:, 3] = 0
    ag[:, 0] = 1
    wg[:, 0] = 1
    inputs = {"allocation_gate": ag, "write_gate": wg, "write_content_keys": rng.random((B, Wn, W)),
              "write_content_strengths": rng.random((B, Wn))}
    w = D.write_weights(cfg, inputs, memory, usage)
    np.testing.assert_allclose(w.sum(2), wg, atol=5e-2)
    np.testing.assert_allclose(w[0, 0], D.one_hot(N, 3), atol=1e-3)


def test_access_read_weights_planted():                    # access_test.py:113-143
    rng = np.random.default_rng(11)
    B, N, W, R, Wn = 2, 20, 6, 2, 3
    cfg = D.AccessConfig(N, W, R, Wn)
    memory = 10 * (rng.random((B, N, W)) - 0.5)
    prw = rng.random((B, R, N))
    prw /= prw.sum(2, keepdims=True) + 1
    link = rng.random((B, Wn, N, N))
    link /= np.maximum(link.sum(2, keepdims=True), 1)
    link /= np.maximum(link.sum(3, keepdims=True), 1)
    keys = rng.random((B, R, W))
    keys[0, 0] = memory[0, 3]
    rm = rng.random((B, R, 1 + 2 * Wn))
    rm[0, 0, :] = D.one_hot(1 + 2 * Wn, 2 * Wn)
    inputs = {"read_content_keys": keys, "read_content_strengths": np.full((B, R), 100.0), "read_mode": rm}
    rw = D.read_weights(cfg, inputs, memory, prw, link)
    np.testing.assert_allclose(rw[0, 0], D.one_hot(N, 3), atol=1e-3)


def test_read_mode_is_a_distribution_and_core_runs():      # access_test.py:62-75, :44-60
    cfg = D.DNCConfig(10, 2, memory_size=20, word_size=6, num_reads=2, num_writes=3, hidden_size=16, clip_value=20)
    rng = np.random.default_rng(12)
    p = D.init_params(cfg, rng, dtype=np.float64)
    x = rng.standard_normal((4, 2, 10))
    st = D.dnc_initial_state(cfg, 2, np.float64)
    y, st2, inp = D.dnc_step(cfg, p, x[0], st)
    np.testing.assert_allclose(inp["read_mode"].sum(2), np.ones((2, 2)))
    assert inp["read_mode"].min() >= 0
    ys, fin = D.run_model(cfg, p, x)
    assert ys.shape == (4, 2, 2) and np.isfinite(ys).all()
    assert np.abs(ys).max() <= 20
    assert fin.access_state.linkage.link.shape == (2, 3, 20, 20)

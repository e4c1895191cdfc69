"""GPU: the reference's OWN DNC module tests (dnc/addressing_test.py), restated against the HIP module kernels:
the same planted one-hot cases and properties that pin the oracle (tests/test_oracle_dnc.py)."""
import numpy as np
import pytest
import torch

from oracle import dnc_oracle as D

pytestmark = pytest.mark.gpu


def test_cosine_weights_values_and_zero_memory(cuda):            # addressing_test.py:72-145
    from ntmtrack.dnc import CosineWeights
    rng = np.random.default_rng(2)
    B, H, N, W = 5, 4, 10, 2
    mem = rng.standard_normal((B, N, W)).astype(np.float32)
    mem[0, 0], mem[0, 1], mem[0, 2] = [1, 2], [3, 4], [5, 6]
    keys = rng.standard_normal((B, H, W)).astype(np.float32)
    keys[0, 0], keys[0, 1], keys[0, 2], keys[0, 3] = [5, 6], [1, 2], [5, 6], [3, 4]
    strengths = rng.standard_normal((B, H)).astype(np.float32)
    res = CosineWeights(H, W, device=cuda)(mem, keys, strengths).cpu().numpy()
    sp = np.log(1 + np.exp(strengths.astype(np.float64)))
    for b in range(B):
        for h in range(H):
            sim = np.array([np.dot(keys[b, h], mem[b, m]) / (np.linalg.norm(keys[b, h]) * np.linalg.norm(mem[b, m])) for m in range(N)])
            sim = np.exp(sim * sp[b, h]); sim /= sim.sum()
            np.testing.assert_allclose(res[b, h], sim, atol=1e-4, rtol=1e-4)
    z = CosineWeights(H, W, device=cuda)(np.zeros((B, N, W), np.float32), keys, strengths).cpu().numpy()
    assert np.isfinite(z).all()
    np.testing.assert_allclose(res, D.cosine_weights(mem, keys, strengths), atol=1e-5)


def test_temporal_linkage_planted_transitions(cuda):              # addressing_test.py:150-236
    from ntmtrack.dnc import TemporalLinkage, TemporalLinkageState
    rng = np.random.default_rng(5)
    B, N, R, Wn = 7, 4, 11, 5
    mod = TemporalLinkage(N, Wn, device=cuda)
    state = TemporalLinkageState(torch.zeros((B, Wn, N, N), device=cuda), torch.zeros((B, Wn, N), device=cuda))
    for i in range(5):
        ww = rng.random((B, Wn, N)); ww /= ww.sum(2, keepdims=True) + 1
        if i == 3:
            ww[0, 0, :] = D.one_hot(N, 0); ww[0, 1, :] = D.one_hot(N, 3)
        elif i == 4:
            ww[0, 0, :] = D.one_hot(N, 1); ww[0, 1, :] = D.one_hot(N, 2)
        state = mod(ww.astype(np.float32), state)
    link = state.link.cpu().numpy()
    assert link.min() >= 0 and link.max() <= 1
    assert not link[:, :, range(N), range(N)].any()
    assert link.sum(2).max() <= 1 + 1e-6 and link.sum(3).max() <= 1 + 1e-6
    np.testing.assert_array_equal(link[0, 0, :, 0], D.one_hot(N, 1))
    np.testing.assert_array_equal(link[0, 1, :, 3], D.one_hot(N, 2))
    prw = rng.random((B, R, N)).astype(np.float32)
    prw[0, 5, :] = D.one_hot(N, 0); prw[0, 6, :] = D.one_hot(N, 2)
    fwd = mod.directional_read_weights(state.link, prw, True).cpu().numpy()
    bwd = mod.directional_read_weights(state.link, prw, False).cpu().numpy()
    np.testing.assert_array_equal(fwd[0, 5, 0, :], D.one_hot(N, 1))
    np.testing.assert_array_equal(bwd[0, 6, 1, :], D.one_hot(N, 3))
    np.testing.assert_allclose(fwd, D.directional_read_weights(link.astype(np.float64), prw.astype(np.float64), True), atol=1e-6)


def test_freeness_and_allocation_cases(cuda):                      # addressing_test.py:277-401
    from ntmtrack.dnc import Freeness
    rng = np.random.default_rng(7)
    B, N, R, Wn = 5, 11, 3, 7
    fg = rng.random((B, R)); prw = rng.random((B, R, N)); prw[1, :, 3] = 0; prw /= prw.sum(2, keepdims=True)
    pww = rng.random((B, Wn, N)); pww /= pww.sum(2, keepdims=True); pu = rng.random((B, N))
    pww[1, 2, 3] = 1; prw[2, 0, 4] = 1; fg[2, 0] = 1
    u = Freeness(N, device=cuda)(pww, fg, prw, pu).cpu().numpy()
    assert u.min() >= 0 and u.max() <= 1 and u[1][3] == 1 and u[2][4] == 0
    # write_allocation_weights (:316-366)
    B, N, Wn = 7, 23, 5
    usage = rng.random((B, N)); wg = rng.random((B, Wn))
    wg[0, 1] = wg[0, 3] = 0; wg[0, 0] = wg[0, 2] = 1
    usage[1] = usage[1] * 0.9 + 0.1; usage[1][4] = 0; usage[1][3] = 1e-4; wg[1, 0] = wg[1, 1] = 1
    w = Freeness(N, device=cuda).write_allocation_weights(usage, wg, Wn).cpu().numpy()
    assert w.min() >= 0 and w.max() <= 1
    np.testing.assert_allclose(w.sum(2), np.ones((B, Wn)), atol=1e-3)
    assert np.abs(w[0, 0] - w[0, 1]).max() > 0.1
    np.testing.assert_array_equal(w[0, 1], w[0, 2])
    np.testing.assert_array_equal(w[0, 3], w[0, 4])
    np.testing.assert_allclose(w[1][0], D.one_hot(N, 4), atol=1e-3)
    np.testing.assert_allclose(w[1][1], D.one_hot(N, 3), atol=1e-3)
    # _allocation argmin/argmax duality (:387-401)
    usage = rng.random((7, 13)).astype(np.float32)
    a = Freeness(13, device=cuda)._allocation(usage).cpu().numpy()
    np.testing.assert_array_equal(np.argmin(usage, 1), np.argmax(a, 1))
    np.testing.assert_array_equal(np.argmax(usage, 1), np.argmin(a, 1))
    np.testing.assert_allclose(a.sum(1), np.ones(7), rtol=0.01)
    np.testing.assert_allclose(a, D.allocation(usage), atol=1e-6)

"""CPU: validate oracle/dnc_oracle_torch.py -- the source of every DNC GRADIENT reference of the GPU tests -- on its own:

* its forward pass equals the numpy restatement (oracle/dnc_oracle.py, pinned by tests/test_oracle_dnc.py against the
  reference's own module tests) to float64 rounding, on the reference's DNC test shape (dnc/access_test.py:26-32:
  memory 20 x 6, 2 reads, 3 writes) and on a step of BASELINE configs[2]'s shape (256 x 64, 4 reads, 1 write, hidden 200);
* its autograd gradients agree with central finite differences inside the bounds the reference's own gradient
  checks use: tf.test.compute_gradient_error(delta=1e-5) < 0.01 for Freeness.write_allocation_weights and
  Freeness._allocation (dnc/addressing_test.py:368-385, :403-416) and < 0.1 for one MemoryAccess step w.r.t. inputs,
  memory, read weights, precedence and link (dnc/access_test.py:145-159).  compute_gradient_error = max |J_analytic -
  J_numeric| over the whole Jacobian; the same quantity is computed here in float64.
"""
import numpy as np
import pytest
import torch

from oracle import dnc_oracle as D
from oracle import dnc_oracle_torch as DT


def _t(x):
    return torch.tensor(np.asarray(x), dtype=torch.float64)


def _tparams(p):
    return {k: _t(v) for k, v in p.items()}


def _flat_state_np(st):
    a = st.access_state
    return [st.access_output, a.memory, a.read_weights, a.write_weights, a.linkage.link, a.linkage.precedence_weights,
            a.usage, st.controller_state.hidden, st.controller_state.cell]


def _flat_state_t(st):
    return [v.detach().numpy() for v in _flat_state_np(st)]


@pytest.mark.parametrize("shape", ["reference_20x6x2x3", "config3_step"])
def test_torch_restatement_equals_numpy_restatement_float64(shape):
    rng = np.random.default_rng(7)
    if shape == "reference_20x6x2x3":
        cfg = D.DNCConfig(10, 2, memory_size=20, word_size=6, num_reads=2, num_writes=3, hidden_size=16, clip_value=20)
        S, B = 6, 2
    else:
        cfg = D.DNCConfig(514, 2, memory_size=256, word_size=64, num_reads=4, num_writes=1, hidden_size=200, clip_value=20)
        S, B = 3, 1
    p = D.init_params(cfg, rng, dtype=np.float64)
    x = rng.standard_normal((S, B, cfg.D))
    ys, fin = D.run_model(cfg, p, x)
    yt, fint = DT.run_model(cfg, _tparams(p), _t(x))
    np.testing.assert_allclose(yt.numpy(), ys, rtol=0, atol=1e-10)
    for a, b in zip(_flat_state_t(fint), _flat_state_np(fin)):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-10)


def _gradient_error(fn, xs, delta=1e-5):
    """max |J_analytic - J_numeric| (central differences) of fn(*xs) -> tensor, over every input in xs:
    what tf.test.compute_gradient_error reports."""
    xs = [x.clone().requires_grad_(True) for x in xs]
    y = fn(*xs).reshape(-1)
    err = 0.0
    jac = [torch.zeros((x.numel(), y.numel()), dtype=torch.float64) for x in xs]
    for j in range(y.numel()):
        gs = torch.autograd.grad(y[j], xs, retain_graph=True, allow_unused=True)
        for k, g_ in enumerate(gs):
            if g_ is not None:
                jac[k][:, j] = g_.reshape(-1)
    with torch.no_grad():
        for k, x in enumerate(xs):
            base = [v.detach().clone() for v in xs]
            flat = base[k].reshape(-1)
            for i in range(flat.numel()):
                old = float(flat[i])
                flat[i] = old + delta
                yp = fn(*base).reshape(-1)
                flat[i] = old - delta
                ym = fn(*base).reshape(-1)
                flat[i] = old
                num = (yp - ym) / (2 * delta)
                err = max(err, float((jac[k][i] - num).abs().max()))
    return err


def test_write_allocation_weights_gradient_bound():       # dnc/addressing_test.py:368-385
    rng = np.random.default_rng(11)
    usage, gates = _t(rng.random((7, 5))), _t(rng.random((7, 3)))
    err = _gradient_error(lambda u, g: DT.write_allocation_weights(u, g, 3), [usage, gates])
    assert err < 0.01, err


def test_allocation_gradient_bound():                      # dnc/addressing_test.py:403-416
    rng = np.random.default_rng(12)
    err = _gradient_error(DT.allocation, [_t(rng.random((1, 5)))])
    assert err < 0.01, err


def test_cosine_weights_gradient_bound():                  # CosineWeights (addressing.py:59-105) under the same bound
    rng = np.random.default_rng(13)
    mem, keys, strengths = _t(rng.standard_normal((2, 5, 3))), _t(rng.standard_normal((2, 2, 3))), _t(rng.standard_normal((2, 2)))
    err = _gradient_error(DT.cosine_weights, [mem, keys, strengths])
    assert err < 0.01, err


def test_memory_access_step_gradient_bound():              # dnc/access_test.py:145-159
    rng = np.random.default_rng(14)
    B, N, W, R, Wn, Din = 2, 20, 6, 2, 3, 10
    acfg = D.AccessConfig(N, W, R, Wn)
    p = {}
    for name, width in acfg.interface:
        p["memory_access/%s/w" % name] = _t(np.clip(rng.standard_normal((Din, width)), -2, 2) / np.sqrt(Din))
        p["memory_access/%s/b" % name] = torch.zeros(width, dtype=torch.float64)
    inputs = _t(rng.standard_normal((B, Din)))
    memory = _t(rng.standard_normal((B, N, W)) * 0.5)
    rw = _t(rng.random((B, R, N)) / N)
    prec = _t(rng.random((B, Wn, N)) / N)
    link = _t(rng.random((B, Wn, N, N)) / N)
    ww0, usage0 = _t(rng.random((B, Wn, N)) / N), _t(rng.random((B, N)))

    def loss(inp, mem, rwv, pr, lk):
        st = DT.AccessState(mem, rwv, ww0, DT.TemporalLinkageState(lk, pr), usage0)
        reads, _ = DT.access_step(acfg, p, inp, st)
        return reads.sum().reshape(1)
    err = _gradient_error(loss, [inputs, memory, rw, prec, link])
    assert err < 0.1, err
    assert err < 1e-6, err        # float64 autograd vs central differences: far inside the reference's float32 bound


def test_sequence_gradient_matches_finite_differences_on_parameters():
    """A whole (short) sequence through dnc_step: d loss / d every parameter vs central differences along random
    directions.  The reference cuts one edge on purpose -- tf.stop_gradient(write_weights) in the usage update
    (dnc/addressing.py:302) -- so the finite differences are taken on a function with that edge cut the same way: the
    previous write weights a step's usage update sees are the UNPERTURBED run's (they enter nothing else)."""
    rng = np.random.default_rng(15)
    cfg = D.DNCConfig(5, 2, memory_size=8, word_size=4, num_reads=2, num_writes=1, hidden_size=6, clip_value=20)
    p = {k: _t(v).requires_grad_(True) for k, v in D.init_params(cfg, rng, dtype=np.float64).items()}
    x = _t(rng.standard_normal((5, 2, 5)))
    tgt = _t(rng.standard_normal((5, 2, 2)))

    def f(params, frozen_ww=None, keep=None):
        st = DT.initial_state(cfg, 2, torch.float64)
        loss = 0.0
        for t in range(x.shape[0]):
            if frozen_ww is not None and t > 0:
                st = st._replace(access_state=st.access_state._replace(write_weights=frozen_ww[t - 1]))
            y, st = DT.dnc_step(cfg, params, x[t], st)
            if keep is not None:
                keep.append(st.access_state.write_weights.detach().clone())
            loss = loss + 0.5 * ((torch.tanh(y) - tgt[t]) ** 2).sum()
        return loss
    base_ww = []
    loss = f(p, keep=base_ww)
    grads = torch.autograd.grad(loss, list(p.values()))
    for (name, v), g_ in zip(p.items(), grads):
        d = _t(rng.standard_normal(tuple(v.shape)))
        d = d / d.norm()
        eps = 1e-6
        with torch.no_grad():
            lp = f({k: (w + eps * d if k == name else w) for k, w in p.items()}, frozen_ww=base_ww)
            lm = f({k: (w - eps * d if k == name else w) for k, w in p.items()}, frozen_ww=base_ww)
        num = float((lp - lm) / (2 * eps))
        ana = float((g_ * d).sum())
        assert abs(num - ana) <= 1e-6 * max(1.0, abs(ana)) + 1e-8, (name, num, ana)

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


# ---------------------------------------------------------------------------------------------------------
# dnc/access_test.py restated: MemoryAccess module (BATCH 2, MEMORY 20, WORD 6, READS 2, WRITES 3, INPUT 10)
# ---------------------------------------------------------------------------------------------------------
AB, AN, AW, AR, AWn, AD = 2, 20, 6, 2, 3, 10


def test_memory_access_valid_read_mode(cuda):                      # access_test.py:62-75
    from ntmtrack.dnc import MemoryAccess
    mod = MemoryAccess(AN, AW, AR, AWn, device=cuda)
    inputs = mod._read_inputs(np.random.default_rng(0).standard_normal((AB, AD)).astype(np.float32))
    rm = inputs["read_mode"].cpu().numpy()
    np.testing.assert_allclose(rm.sum(2), np.ones((AB, AR)), atol=1e-6)
    assert rm.min() >= 0
    assert inputs["erase_vectors"].shape == (AB, AWn, AW) and inputs["read_content_keys"].shape == (AB, AR, AW)


def test_memory_access_write_weights_planted(cuda):                # access_test.py:77-111
    from ntmtrack.dnc import MemoryAccess
    rng = np.random.default_rng(1)
    memory = 10 * (rng.random((AB, AN, AW)) - 0.5)
    usage = rng.random((AB, AN))
    ag, wg = rng.random((AB, AWn)), rng.random((AB, AWn))
    keys, strengths = rng.random((AB, AWn, AW)), rng.random((AB, AWn))
    usage[:, 3] = 0; ag[:, 0] = 1; wg[:, 0] = 1
    inputs = {"allocation_gate": ag, "write_gate": wg, "write_content_keys": keys, "write_content_strengths": strengths}
    w = MemoryAccess(AN, AW, AR, AWn, device=cuda)._write_weights(inputs, memory, usage).cpu().numpy()
    np.testing.assert_allclose(w.sum(2), wg, atol=5e-2)
    np.testing.assert_allclose(w[0, 0], D.one_hot(AN, 3), atol=1e-3)
    ref = D.write_weights(D.AccessConfig(AN, AW, AR, AWn), {k: v.astype(np.float32) for k, v in inputs.items()},
                          memory.astype(np.float32), usage.astype(np.float32))
    np.testing.assert_allclose(w, ref, atol=2e-6)


def test_memory_access_read_weights_planted(cuda):                 # access_test.py:113-143
    from ntmtrack.dnc import MemoryAccess
    rng = np.random.default_rng(2)
    memory = 10 * (rng.random((AB, AN, AW)) - 0.5)
    prw = rng.random((AB, AR, AN)); prw /= prw.sum(2, keepdims=True) + 1
    link = rng.random((AB, AWn, AN, AN))
    link /= np.maximum(link.sum(2, keepdims=True), 1); link /= np.maximum(link.sum(3, keepdims=True), 1)
    keys = rng.random((AB, AR, AW)); keys[0, 0] = memory[0, 3]
    strengths = np.full((AB, AR), 100.0)
    mode = rng.random((AB, AR, 1 + 2 * AWn)); mode[0, 0, :] = D.one_hot(1 + 2 * AWn, 2 * AWn)
    inputs = {"read_content_keys": keys, "read_content_strengths": strengths, "read_mode": mode}
    rw = MemoryAccess(AN, AW, AR, AWn, device=cuda)._read_weights(inputs, memory, prw, link).cpu().numpy()
    np.testing.assert_allclose(rw[0, 0, :], D.one_hot(AN, 3), atol=1e-3)
    f32 = lambda v: v.astype(np.float32)
    ref = D.read_weights(D.AccessConfig(AN, AW, AR, AWn), {k: f32(v) for k, v in inputs.items()}, f32(memory), f32(prw), f32(link))
    np.testing.assert_allclose(rw, ref, atol=2e-5)


def test_memory_access_steps_match_oracle(cuda):                   # access_test.py:44-60 (forward part), 4 time steps
    from ntmtrack import dnc as G
    rng = np.random.default_rng(3)
    mod = G.MemoryAccess(AN, AW, AR, AWn, input_dim=AD, device=cuda, seed=5)
    sd = {k: v.numpy() * (3.0 if k.endswith("/w") else 1.0) for k, v in mod.state_dict().items()}
    mod.load_state_dict(sd)
    cfg = D.AccessConfig(AN, AW, AR, AWn)
    st, ost = mod.initial_state(AB), D.access_initial_state(cfg, AB)
    assert mod.output_size == (AR, AW) and mod.state_size.linkage.link == (AWn, AN, AN)
    for t in range(4):
        x = rng.standard_normal((AB, AD)).astype(np.float32)
        reads, st = mod(x, st)
        oreads, ost, _ = D.access_step(cfg, sd, x, ost)
        np.testing.assert_allclose(reads.cpu().numpy(), oreads, atol=2e-5)
        np.testing.assert_allclose(st.memory.cpu().numpy(), ost.memory, atol=2e-5)
        np.testing.assert_allclose(st.usage.cpu().numpy(), ost.usage, atol=2e-5)
        np.testing.assert_allclose(st.write_weights.cpu().numpy(), ost.write_weights, atol=2e-5)
        np.testing.assert_allclose(st.read_weights.cpu().numpy(), ost.read_weights, atol=2e-5)
        np.testing.assert_allclose(st.linkage.link.cpu().numpy(), ost.linkage.link, atol=2e-5)
        np.testing.assert_allclose(st.linkage.precedence_weights.cpu().numpy(), ost.linkage.precedence_weights, atol=2e-5)


# ---------------------------------------------------------------------------------------------------------
# step-granular C-ABI entry points (SURVEY 8b minimum export set)
# ---------------------------------------------------------------------------------------------------------
def test_lstm_step_and_maxpool_and_split_loss(cuda):
    from ntmtrack import _lib
    from oracle import ntm_oracle as O
    L, P, st = _lib.lib(), _lib.ptr, _lib.stream()
    rng = np.random.default_rng(4)
    B, hid = 3, 20
    pre = rng.standard_normal((B, 4 * hid)).astype(np.float32)
    c0 = rng.standard_normal((B, hid)).astype(np.float32)
    tp, tc0 = torch.from_numpy(pre).to(cuda), torch.from_numpy(c0).to(cuda)
    c, h, act = (torch.empty((B, hid), device=cuda), torch.empty((B, hid), device=cuda), torch.empty((B, 4 * hid), device=cuda))
    _lib.check(L.ntk_lstm_step_fwd(P(tp), P(tc0), 0.0, P(c), P(h), P(act), B, hid, st), "ntk_lstm_step_fwd")
    # float64 autograd restatement of BasicLSTMCell's pointwise part (i, j, f, o blocks; forget_bias 0)
    p64 = torch.tensor(pre, dtype=torch.float64, requires_grad=True)
    c64 = torch.tensor(c0, dtype=torch.float64, requires_grad=True)
    i, j, f, o = p64.split(hid, dim=1)
    cr = c64 * torch.sigmoid(f) + torch.sigmoid(i) * torch.tanh(j)
    hr = torch.tanh(cr) * torch.sigmoid(o)
    np.testing.assert_allclose(c.cpu().numpy(), cr.detach().numpy(), atol=1e-6)
    np.testing.assert_allclose(h.cpu().numpy(), hr.detach().numpy(), atol=1e-6)
    dh = rng.standard_normal((B, hid)).astype(np.float32)
    dc = rng.standard_normal((B, hid)).astype(np.float32)
    ((hr * torch.tensor(dh, dtype=torch.float64)).sum() + (cr * torch.tensor(dc, dtype=torch.float64)).sum()).backward()
    dpre, dc0 = torch.empty((B, 4 * hid), device=cuda), torch.empty((B, hid), device=cuda)
    tdh, tdc = torch.from_numpy(dh).to(cuda), torch.from_numpy(dc).to(cuda)
    _lib.check(L.ntk_lstm_step_bwd(P(act), P(tc0), P(c), P(tdh), P(tdc), P(dpre), P(dc0), B, hid, st), "ntk_lstm_step_bwd")
    np.testing.assert_allclose(dpre.cpu().numpy(), p64.grad.numpy(), atol=2e-6)
    np.testing.assert_allclose(dc0.cpu().numpy(), c64.grad.numpy(), atol=2e-6)
    # max pool: bit-exact vs numpy
    x = rng.standard_normal((2, 6, 10, 8)).astype(np.float32)
    out = torch.empty((2, 3, 5, 8), device=cuda)
    tx = torch.from_numpy(x).to(cuda)
    _lib.check(L.ntk_maxpool2x2(P(tx), P(out), 2, 6, 10, 8, st), "ntk_maxpool2x2")
    np.testing.assert_array_equal(out.cpu().numpy(), O.maxpool2x2(x))
    # split loss = fused loss
    Bq, T, NF, Oq = 2, 3, 64, 2
    logits = torch.from_numpy(rng.standard_normal((Bq, T * (NF + 1), Oq)).astype(np.float32)).to(cuda)
    offs = torch.from_numpy(rng.uniform(-.5, .5, (Bq, T, Oq)).astype(np.float32)).to(cuda)
    pred1, loss1, dl1 = torch.empty((Bq, T - 1, Oq), device=cuda), torch.empty(1, device=cuda), torch.empty_like(logits)
    pred2, loss2, dl2 = torch.empty_like(pred1), torch.empty(1, device=cuda), torch.empty_like(logits)
    _lib.check(L.ntk_offset_loss(P(logits), P(offs), P(pred1), P(loss1), P(dl1), Bq, T, NF, Oq, st), "ntk_offset_loss")
    _lib.check(L.ntk_offset_loss_fwd(P(logits), P(offs), P(pred2), P(loss2), Bq, T, NF, Oq, st), "ntk_offset_loss_fwd")
    _lib.check(L.ntk_offset_loss_bwd(P(logits), P(offs), P(dl2), Bq, T, NF, Oq, st), "ntk_offset_loss_bwd")
    assert torch.equal(pred1, pred2) and torch.equal(loss1, loss2) and torch.equal(dl1, dl2)


def test_ntm_step_entry_point_equals_sequence_kernel(cuda):
    """NTMCell.__call__ (ntk_ntm_step_fwd) chained 3 times == ntk_ntm_seq_fwd over 3 steps (same kernel, S = 1)."""
    from ntmtrack import ntm as G
    cell = G.NTMCell(2, mem_size=64, mem_dim=8, controller_hidden_size=12, controller_num_layers=1, write_head_size=1,
                     read_head_size=2, input_dim=10, device=cuda, seed=3)
    B, S = 2, 3
    x = torch.randn((B, S, 10), generator=torch.Generator().manual_seed(1)).to(cuda)
    logits_seq, _o, new_seq, _r = cell.run_sequence(cell._pad_inputs(x), cell.zero_state(B))
    state, outs = cell.zero_state(B), []
    for t in range(S):
        r = cell(x[:, t], state)
        outs.append(r[1])
        state = r[2]
    np.testing.assert_array_equal(torch.stack(outs, 1).cpu().numpy(), logits_seq.cpu().numpy())
    np.testing.assert_array_equal(state["M"].cpu().numpy(), new_seq["M"].cpu().numpy())


# ---------------------------------------------------------------------------------------------------------
# ops.py module functions (ops_test.py)
# ---------------------------------------------------------------------------------------------------------
def test_ops_cosine_similarity_golden_vectors(cuda):
    """ops_test.py:20-34 pins TRUE smooth cosine (opt-in mode); the as-coded values (quirk Q1, the mode NTMCell uses) are
    the hand-evaluated vector of tests/golden/reference_vectors.json -- the reference's own test FAILS against its code."""
    import json, os
    from ntmtrack import ops
    from oracle import ntm_oracle as O
    vec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))
    t7 = [v for k, v in vec.items() if k.startswith("ops_test.py")][0]
    coded = [v for k, v in vec.items() if k.startswith("ops.py:147")][0]
    mem, keys = np.array(t7["memory"], np.float32), np.array(t7["keys"], np.float32)
    got_true = ops.batched_smooth_cosine_similarity(mem, keys, similarity="smooth_cosine", device=cuda).cpu().numpy()
    np.testing.assert_allclose(got_true, np.array(t7["expected"]), atol=1e-4)
    got = ops.batched_smooth_cosine_similarity(mem, keys, device=cuda).cpu().numpy()
    np.testing.assert_allclose(got, np.array(coded["expected"]), atol=1e-4)
    assert np.abs(got - np.array(t7["expected"])).max() > 0.05          # Q1: the shipped code is not the tested function
    rng = np.random.default_rng(0)
    mem, keys = rng.standard_normal((3, 128, 20)).astype(np.float32), rng.standard_normal((3, 5, 20)).astype(np.float32)
    for mode in ("as_coded", "smooth_cosine"):
        got = ops.batched_smooth_cosine_similarity(mem, keys, similarity=mode, device=cuda).cpu().numpy()
        np.testing.assert_allclose(got, O.batched_smooth_cosine_similarity(mem, keys, mode), atol=2e-6)


def test_ops_circular_convolution_and_shift(cuda):
    from ntmtrack import ops
    from oracle import ntm_oracle as O
    rng = np.random.default_rng(1)
    w = rng.random((2, 5, 128)).astype(np.float32)
    for SS in (3, 5, 1):
        s = rng.random((2, 5, SS)).astype(np.float32)
        got = ops.batched_circular_convolution(w, s, device=cuda).cpu().numpy()
        np.testing.assert_allclose(got, O.batched_circular_convolution(w, s), atol=1e-6)
    # Q2: a one-hot kernel on the LAST tap is the identity (taps are -2,-1,0, not -1,0,1)
    ident = np.zeros((2, 5, 3), np.float32); ident[..., 2] = 1
    np.testing.assert_array_equal(ops.batched_circular_convolution(w, ident, device=cuda).cpu().numpy(), w)
    for shift in (-2, -1, 0, 1, 3):
        np.testing.assert_array_equal(ops.circular_shift(w, shift, device=cuda).cpu().numpy(), O.circular_shift(w, shift))


# ---------------------------------------------------------------------------------------------------------------------
# module-granular backward: dnc/access_test.py:145-159 (testGradients) differentiates sum(read words) of ONE MemoryAccess
# step w.r.t. inputs, memory, read_weights, precedence and link.  Here: the analytic gradients of the HIP path
# (ntk_dnc_access_step_bwd) against torch autograd on the float64 restatement, for the reference's own module shape
# (memory 20, word 6 -> padded to 8, 2 reads, 3 writes) from a NON-degenerate state, and from the all-zero initial state
# the reference's test uses.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("zero_state", [False, True], ids=["random_state", "initial_state"])
@pytest.mark.parametrize("N,W,R,Wn", [(AN, AW, AR, AWn), (32, 8, 2, 1)], ids=["reference_shape_3_writes", "one_write"])
def test_memory_access_step_gradients_match_autograd(cuda, N, W, R, Wn, zero_state):
    from ntmtrack import dnc as G
    from oracle import dnc_oracle_torch as DT
    rng = np.random.default_rng(11)
    B, Din = 2, 10
    mod = G.MemoryAccess(N, W, R, Wn, input_dim=Din, device=cuda, seed=5)
    sd = {k: v.numpy() * (3.0 if k.endswith("/w") else 1.0) for k, v in mod.state_dict().items()}
    for k in sd:
        if k.endswith("/b"):
            sd[k] = rng.uniform(-0.3, 0.3, size=sd[k].shape).astype(np.float32)
    mod.load_state_dict(sd)
    cfg = D.AccessConfig(N, W, R, Wn)
    f = lambda *s: rng.random(s).astype(np.float32)
    if zero_state:
        st = D.access_initial_state(cfg, B)
    else:
        usage = np.stack([rng.permutation(N) for _ in range(B)]).astype(np.float32) / N * 0.8 + 0.1     # no near-ties for the sort
        rw = f(B, R, N); rw /= rw.sum(2, keepdims=True) + 1
        ww = f(B, Wn, N); ww /= ww.sum(2, keepdims=True) + 1
        prec = f(B, Wn, N); prec /= prec.sum(2, keepdims=True) + 1
        link = f(B, Wn, N, N)
        link /= np.maximum(link.sum(2, keepdims=True), 1); link /= np.maximum(link.sum(3, keepdims=True), 1)
        link[:, :, np.arange(N), np.arange(N)] = 0
        st = D.AccessState((f(B, N, W) - 0.5).astype(np.float32), rw, ww, D.TemporalLinkageState(link.astype(np.float32), prec), usage)
    x = rng.standard_normal((B, Din)).astype(np.float32)
    Gr = rng.standard_normal((B, R, W)).astype(np.float32)                # d loss / d read words (the reference's test: all ones)
    # oracle: autograd through one access step.  float64, except from the all-zero state: there every usage is tied and
    # the simulated usages of the later write heads are products of 1e-6 factors that underflow in float32 but not in
    # float64 -- the two precisions allocate DIFFERENT slots (as the reference's float32 TF graph would), so that case is
    # compared with the float32 restatement
    odt = torch.float32 if zero_state else torch.float64
    t64 = lambda v: torch.tensor(np.asarray(v), dtype=odt, requires_grad=True)
    pt = {k: t64(v) for k, v in sd.items()}
    xt = t64(x)
    leaves = dict(memory=t64(st.memory), read_weights=t64(st.read_weights), link=t64(st.linkage.link),
                  precedence_weights=t64(st.linkage.precedence_weights), usage=t64(st.usage))
    ost = DT.AccessState(leaves["memory"], leaves["read_weights"], torch.tensor(st.write_weights, dtype=odt),
                         DT.TemporalLinkageState(leaves["link"], leaves["precedence_weights"]), leaves["usage"])
    reads, _new = DT.access_step(cfg, pt, xt, ost)
    (reads * torch.tensor(Gr, dtype=odt)).sum().backward()
    # HIP
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(cuda)
    gst = G.AccessState(t(st.memory), t(st.read_weights), t(st.write_weights),
                        G.TemporalLinkageState(t(st.linkage.link), t(st.linkage.precedence_weights)), t(st.usage))
    out, _ = mod(t(x), gst)
    np.testing.assert_allclose(out.cpu().numpy(), reads.detach().double().numpy(), atol=2e-5)
    g = mod.step_gradients(t(x), gst, t(Gr))
    torch.cuda.synchronize()
    ref = {"inputs": xt.grad}
    ref.update({k: v.grad for k, v in leaves.items()})
    ref.update({k: v.grad for k, v in pt.items()})
    scale = max(float(v.abs().max()) for v in ref.values() if v is not None)
    for k in sorted(ref):
        r = ref[k].double().numpy() if ref[k] is not None else np.zeros(tuple(g[k].shape))
        got = g[k].cpu().numpy()
        assert got.shape == r.shape, (k, got.shape, r.shape)
        err = np.max(np.abs(got - r)) / max(np.max(np.abs(r)), 1e-3 * scale)
        assert err < 3e-3, (k, err)

"""GPU parity: DNC core sequence kernel vs the numpy oracle (oracle/dnc_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import dnc_oracle as D

pytestmark = pytest.mark.gpu

CASES = [
    # name, D, O, N, W, R, Wn, hid, clip, S, B, start from the all-zero initial_state?
    ("small_multiwrite", 10, 3, 16, 8, 2, 3, 16, 20.0, 6, 2, False),
    ("c3_shape", 514, 2, 256, 64, 4, 1, 200, 20.0, 8, 2, True),
    ("no_clip_odd", 12, 2, 40, 12, 3, 2, 24, 0.0, 5, 1, False),
    ("c5_shape_short", 514, 2, 512, 128, 4, 1, 200, 20.0, 3, 1, True),
    ("c3_shape_random_state", 514, 2, 256, 64, 4, 1, 200, 20.0, 4, 2, False),
    # the reference's own DNC test shape (dnc/access_test.py:28-34: memory 20, word_size 6, 2 reads, 3 writes): word_size is
    # zero padded to 8 inside the core, the state and the variables keep the reference's shapes
    ("reference_test_shape_w6", 10, 3, 20, 6, 2, 3, 16, 20.0, 6, 2, False),
    ("word_size_5_zero_state", 7, 2, 24, 5, 1, 1, 12, 0.0, 5, 3, True),
]


def _random_state(cfg, B, rng):
    """A valid, NON-DEGENERATE access state: distinct usages (no near-ties for the allocation sort: with
    several write heads and exactly tied usages the winner of the sort hinges on the last fp32 bit in any
    implementation -- the reference's own tests plant distinct usages for the same reason,
    addressing_test.py:328-333), sub-stochastic weights and link."""
    a = cfg.access
    N, W, R, Wn = a.N, a.W, a.R, a.Wn
    f = lambda *s: rng.random(s).astype(np.float32)
    usage = np.stack([rng.permutation(N) for _ in range(B)]).astype(np.float32) / N * 0.8 + 0.1
    rw = f(B, R, N); rw /= rw.sum(2, keepdims=True) + 1
    ww = f(B, Wn, N); ww /= ww.sum(2, keepdims=True) + 1
    prec = f(B, Wn, N); prec /= prec.sum(2, keepdims=True) + 1
    link = f(B, Wn, N, N)
    link /= np.maximum(link.sum(2, keepdims=True), 1)
    link /= np.maximum(link.sum(3, keepdims=True), 1)
    link[:, :, np.arange(N), np.arange(N)] = 0
    mem = (f(B, N, W) - 0.5).astype(np.float32)
    acc = D.AccessState(mem, rw, ww, D.TemporalLinkageState(link.astype(np.float32), prec), usage)
    reads = (rw @ mem).astype(np.float32)
    h, c = (f(B, cfg.hid) - 0.5), (f(B, cfg.hid) - 0.5)
    return D.DNCState(reads, acc, D.LSTMState(h, c))


def _with_forms(cases):
    """Every kernel family a shape can run on gets its own oracle comparison: "seq" = one workgroup per sequence
    (ntk_dnc_seq_*), "lds" = the LDS-resident cluster form (ntk_dnc_cluster_*), "mp" = the memory-partitioned cluster
    form (ntk_dnc_mp_*).  Shapes outside the cluster kernels' range (several write heads, memory not a multiple of 64)
    run once, on whatever the planner picks ("auto" = the seq kernels there)."""
    out = []
    for c in cases:
        N, Wn = c[3], c[6]
        forms = ("seq", "lds", "mp") if (Wn == 1 and N % 64 == 0) else ("auto",)
        for f in forms:
            if f == "lds" and N * N // 8 * 4 + N * (c[4] + 4) * 4 > 150 * 1024:
                continue                                   # link slice + replicated memory cannot fit 160 KiB of LDS (config 5's shape)
            out.append(pytest.param(*c, f, id="%s-%s" % (c[0], f)))
    return out


def _select_form(core, form):
    if form == "seq":
        core.cluster_k = 0
    elif form in ("lds", "mp"):
        core.cluster_form = form


def _assert_form(core, form, bwd=False):
    """The kernel family that RAN is the one the test names (no accidental fallback)."""
    k = core.last_cluster_bwd_k if bwd else core.last_cluster_k
    f = core.last_cluster_bwd_form if bwd else core.last_cluster_form
    if form == "seq":
        assert k == 1 and f is None, (k, f)
    elif form in ("lds", "mp"):
        assert k > 1 and f == form, (form, k, f)
        core.check_cluster()


@pytest.mark.parametrize("name,Din,O,N,W,R,Wn,hid,clip,S,B,zero,form", _with_forms(CASES))
def test_dnc_sequence_matches_oracle(cuda, name, Din, O, N, W, R, Wn, hid, clip, S, B, zero, form):
    from ntmtrack.dnc import DNC
    cfg = D.DNCConfig(Din, O, memory_size=N, word_size=W, num_reads=R, num_writes=Wn, hidden_size=hid, clip_value=clip)
    rng = np.random.default_rng(5)
    p = D.init_params(cfg, rng)
    for k in p:                                      # non-zero biases; stronger interface so gates/keys are not all ~0.5
        if k.endswith("/b") or k.endswith("b_gates"):
            p[k] = rng.uniform(-0.3, 0.3, size=p[k].shape).astype(np.float32)
        if k.startswith("memory_access/") and k.endswith("/w"):
            p[k] = (p[k] * 6).astype(np.float32)
    x = rng.standard_normal((S, B, Din)).astype(np.float32)
    # The oracle runs in float32 here on purpose: the allocation weighting sorts usage, and slots that were
    # never specifically written carry mathematically tied usages (they differ at 1e-12 relative in float64 and
    # are exact ties in float32, broken by index as tf.nn.top_k does).  A float64 oracle would order those
    # near-ties differently from ANY float32 implementation, the reference's TF graph included.
    st0 = None if zero else _random_state(cfg, B, rng)
    ys, fin = D.run_model(cfg, p, x, state=st0)

    from ntmtrack import dnc as G
    core = DNC({"memory_size": N, "word_size": W, "num_reads": R, "num_writes": Wn}, {"hidden_size": hid}, O, clip, device=cuda)
    core.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    gst = None
    if st0 is not None:
        t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(cuda)
        a0 = st0.access_state
        gst = G.DNCState(t(st0.access_output), G.AccessState(t(a0.memory), t(a0.read_weights), t(a0.write_weights),
                         G.TemporalLinkageState(t(a0.linkage.link), t(a0.linkage.precedence_weights)), t(a0.usage)),
                         G.LSTMState(t(st0.controller_state.hidden), t(st0.controller_state.cell)))
    _select_form(core, form)
    out, st = core.run_sequence(torch.from_numpy(x).to(cuda), gst)
    torch.cuda.synchronize()
    _assert_form(core, form)
    tol = dict(atol=5e-5, rtol=0)
    np.testing.assert_allclose(out.cpu().numpy(), ys, **tol)
    acc = st.access_state
    np.testing.assert_allclose(acc.memory.cpu().numpy(), fin.access_state.memory, err_msg="memory", **tol)
    np.testing.assert_allclose(acc.usage.cpu().numpy(), fin.access_state.usage, err_msg="usage", **tol)
    np.testing.assert_allclose(acc.write_weights.cpu().numpy(), fin.access_state.write_weights, err_msg="ww", **tol)
    np.testing.assert_allclose(acc.read_weights.cpu().numpy(), fin.access_state.read_weights, err_msg="rw", **tol)
    np.testing.assert_allclose(acc.linkage.link.cpu().numpy(), fin.access_state.linkage.link, err_msg="link", **tol)
    np.testing.assert_allclose(acc.linkage.precedence_weights.cpu().numpy(), fin.access_state.linkage.precedence_weights, err_msg="prec", **tol)
    np.testing.assert_allclose(st.access_output.cpu().numpy(), fin.access_output, err_msg="reads", **tol)
    np.testing.assert_allclose(st.controller_state.hidden.cpu().numpy(), fin.controller_state.hidden, err_msg="h", **tol)
    np.testing.assert_allclose(st.controller_state.cell.cpu().numpy(), fin.controller_state.cell, err_msg="c", **tol)
    # structural properties the reference tests assert (addressing_test.py:208-216)
    link = acc.linkage.link.cpu().numpy()
    assert link.min() >= -1e-6 and link.max() <= 1 + 1e-6
    assert np.abs(link[:, :, range(N), range(N)]).max() == 0


def test_dnc_step_api_and_state_chaining(cuda):
    """core(inputs, prev_state) one step at a time == one launch over the sequence; initial_state is all zeros."""
    from ntmtrack.dnc import DNC, DNCState
    core = DNC({"memory_size": 32, "word_size": 8, "num_reads": 2, "num_writes": 1}, {"hidden_size": 32}, 2, 20, input_dim=10,
               device=cuda, seed=3)
    st = core.initial_state(2)
    assert isinstance(st, DNCState) and not st.access_state.memory.any() and not st.access_state.linkage.link.any()
    x = torch.randn((4, 2, 10), generator=torch.Generator().manual_seed(0)).to(cuda)
    full, fin = core.run_sequence(x)
    outs = []
    for t in range(4):
        y, st = core(x[t], st)
        outs.append(y)
    torch.cuda.synchronize()
    assert torch.allclose(torch.stack(outs, 0), full, atol=1e-6)
    assert torch.allclose(st.access_state.memory, fin.access_state.memory, atol=1e-6)


def test_dnc_unsupported_shapes_fail_loudly(cuda):
    from ntmtrack.dnc import DNC
    from ntmtrack._lib import NtkError
    core = DNC({"memory_size": 18, "word_size": 6, "num_reads": 2, "num_writes": 3}, {"hidden_size": 16}, 2, 20, input_dim=10, device=cuda)
    with pytest.raises(NtkError):                    # memory_size must be a multiple of 4 (word_size is padded inside the core)
        core.run_sequence(torch.zeros((2, 1, 10), device=cuda))
    core = DNC({"memory_size": 20, "word_size": 6, "num_reads": 5, "num_writes": 1}, {"hidden_size": 16}, 2, 20, input_dim=10, device=cuda)
    with pytest.raises(NtkError):                    # at most 4 read heads
        core.run_sequence(torch.zeros((2, 1, 10), device=cuda))


def test_dnc_offset_tracker_pipeline(cuda):
    """direct_offset_output_with_dnc forward: conv4_3 map -> gather/serialise -> DNC(clip 20) -> tanh at the
    delimiter steps, vs the oracle chain (serialize_inputs -> run_model time-major -> offset gather)."""
    from oracle import ntm_oracle as O
    from ntmtrack import tracker
    B, T = 2, 2
    rng = np.random.default_rng(4)
    trk = tracker.DNCOffsetTracker(B, T, vgg_weights=None, mem_size=64, mem_dim=16, hidden_size=32, read_head_size=2,
                                   write_head_size=1, clip_value=20, device=cuda, seed=9)
    sd = {k: v.numpy() for k, v in trk.core.state_dict().items()}
    cfg = D.DNCConfig(514, 2, memory_size=64, word_size=16, num_reads=2, num_writes=1, hidden_size=32, clip_value=20)
    fmap = np.maximum(rng.standard_normal((B * T, 28, 28, 512)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
    x = O.serialize_inputs(O.extract_features(fmap).reshape(B, T, 64, 512), gts)          # [B,S,514]
    ys, _ = D.run_model(cfg, sd, np.ascontiguousarray(np.transpose(x, (1, 0, 2))))       # time-major
    logits_ref = np.transpose(ys, (1, 0, 2))
    _, pred_ref = O.offset_loss(logits_ref, np.zeros((B, T, 2), np.float32))
    logits, _ = trk.forward_features(torch.from_numpy(fmap).to(cuda), torch.from_numpy(gts[:, 0].copy()).to(cuda))
    torch.cuda.synchronize()
    np.testing.assert_allclose(logits.cpu().numpy(), logits_ref, atol=5e-5)
    offs = torch.zeros((B, T, 2), device=cuda)
    _l, pred, _ = tracker.offset_loss(logits, offs, T, want_grad=False)
    np.testing.assert_allclose(pred.cpu().numpy(), pred_ref, atol=5e-5)


BWD_CASES = [
    # name, D, O, N, W, R, Wn, hid, clip, S, B, zero initial state?
    ("small_random_state", 10, 3, 16, 8, 2, 1, 16, 20.0, 5, 2, False),
    ("small_zero_state", 12, 2, 32, 12, 3, 1, 24, 20.0, 6, 2, True),
    ("tight_clip", 10, 2, 16, 8, 2, 1, 16, 0.4, 4, 2, False),
    ("c3_shape", 514, 2, 256, 64, 4, 1, 200, 20.0, 4, 1, False),
    ("c5_shape", 514, 2, 512, 128, 4, 1, 200, 20.0, 3, 1, False),
    # several write heads (dnc_seq_bwd_mw.hip): the reference's own DNC test shape (memory 20, word 6 -> padded to 8,
    # 2 reads, 3 writes; dnc/access_test.py:28-34), two heads, four heads on a wider memory
    ("reference_test_shape_3_writes", 10, 3, 20, 6, 2, 3, 16, 20.0, 6, 2, False),
    ("two_writes_zero_state", 12, 2, 32, 12, 3, 2, 24, 20.0, 6, 2, True),
    ("four_writes", 9, 2, 64, 16, 4, 4, 32, 20.0, 5, 2, False),
    ("three_writes_c3_memory", 514, 2, 256, 64, 4, 3, 200, 20.0, 3, 1, False),
]


@pytest.mark.parametrize("name,Din,O,N,W,R,Wn,hid,clip,S,B,zero,form", _with_forms(BWD_CASES))
def test_dnc_bptt_gradients_match_autograd_oracle(cuda, name, Din, O, N, W, R, Wn, hid, clip, S, B, zero, form):
    """d(sum(y * G)) / d(params) through S recorded steps vs torch-autograd on the float64 restatement, on every kernel
    family the shape can run on (the family that ran is asserted).  Inputs are chosen without near-tied usages (see
    _random_state) so the allocation sort is well conditioned.
    Bound per tensor, relative to the tensor's largest gradient: 1e-4 -- or, for the tensors whose gradient is a sum with
    heavy cancellation (keys / strengths: the float32 evaluation of the SAME restatement is itself further than that from
    float64), no further from float64 than 3x what that float32 evaluation is."""
    from oracle import dnc_oracle_torch as DT
    from ntmtrack import dnc as G
    cfg = D.DNCConfig(Din, O, memory_size=N, word_size=W, num_reads=R, num_writes=Wn, hidden_size=hid, clip_value=clip)
    rng = np.random.default_rng(17)
    p = D.init_params(cfg, rng)
    for k in p:
        if k.endswith("/b") or k.endswith("b_gates"):
            p[k] = rng.uniform(-0.3, 0.3, size=p[k].shape).astype(np.float32)
        if k.startswith("memory_access/") and k.endswith("/w"):
            p[k] = (p[k] * 4).astype(np.float32)
    x = rng.standard_normal((S, B, Din)).astype(np.float32)
    Gy = rng.standard_normal((S, B, O)).astype(np.float32)
    st0 = None if zero else _random_state(cfg, B, rng)

    # oracle (float64 autograd)
    t64 = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64)
    pt = {k: t64(v).requires_grad_(True) for k, v in p.items()}
    ost = None
    if st0 is not None:
        a0 = st0.access_state
        ost = DT.DNCState(t64(st0.access_output), DT.AccessState(t64(a0.memory), t64(a0.read_weights), t64(a0.write_weights),
                          DT.TemporalLinkageState(t64(a0.linkage.link), t64(a0.linkage.precedence_weights)), t64(a0.usage)),
                          DT.LSTMState(t64(st0.controller_state.hidden), t64(st0.controller_state.cell)))
    ys, _ = DT.run_model(cfg, pt, t64(x), ost)
    (ys * t64(Gy)).sum().backward()
    # the same restatement evaluated in float32: how far a float32 evaluation of these sums is from float64
    t32 = lambda v: torch.tensor(np.asarray(v), dtype=torch.float32)
    pt32 = {k: t32(v).requires_grad_(True) for k, v in p.items()}
    ost32 = None
    if st0 is not None:
        a0 = st0.access_state
        ost32 = DT.DNCState(t32(st0.access_output), DT.AccessState(t32(a0.memory), t32(a0.read_weights), t32(a0.write_weights),
                            DT.TemporalLinkageState(t32(a0.linkage.link), t32(a0.linkage.precedence_weights)), t32(a0.usage)),
                            DT.LSTMState(t32(st0.controller_state.hidden), t32(st0.controller_state.cell)))
    ys32, _ = DT.run_model(cfg, pt32, t32(x), ost32)
    (ys32 * t32(Gy)).sum().backward()

    core = G.DNC({"memory_size": N, "word_size": W, "num_reads": R, "num_writes": Wn}, {"hidden_size": hid}, O, clip, device=cuda)
    core.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    _select_form(core, form)
    gst = None
    if st0 is not None:
        t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(cuda)
        a0 = st0.access_state
        gst = G.DNCState(t(st0.access_output), G.AccessState(t(a0.memory), t(a0.read_weights), t(a0.write_weights),
                         G.TemporalLinkageState(t(a0.linkage.link), t(a0.linkage.precedence_weights)), t(a0.usage)),
                         G.LSTMState(t(st0.controller_state.hidden), t(st0.controller_state.cell)))
    out, _st = core.run_sequence(torch.from_numpy(x).to(cuda), gst, record=True)
    np.testing.assert_allclose(out.cpu().numpy(), ys.detach().numpy(), atol=5e-5)
    assert {k: tuple(v.shape) for k, v in core.state_dict().items()} == {k: v.shape for k, v in p.items()}
    dout = torch.from_numpy(np.ascontiguousarray(np.transpose(Gy, (1, 0, 2)))).to(cuda)      # [B,S,O]
    _assert_form(core, form)
    grads = core.backward_sequence(core.last_X, dout)
    torch.cuda.synchronize()
    _assert_form(core, form, bwd=True)
    worst, bad = {}, {}
    for k in sorted(p):
        ref = pt[k].grad.numpy()
        got = grads[k].cpu().numpy()
        scale = np.max(np.abs(ref)) + 1e-30
        err = float(np.max(np.abs(got - ref)) / scale)
        err32 = float(np.max(np.abs(pt32[k].grad.double().numpy() - ref)) / scale)
        worst[k] = (err, err32)
        if err > max(1e-4, 3 * err32):
            bad[k] = (err, err32)
    print("%s/%s relative gradient error (HIP, float32 oracle) vs float64: %s" % (name, form, {k: ("%.1e" % a, "%.1e" % b) for k, (a, b) in worst.items()}))
    # the training step's form: no re-layout, nothing returned, the same gradients in params.grad (packed layout; its padding is
    # not defined, so the comparison goes through the re-layout)
    core.run_sequence(torch.from_numpy(x).to(cuda), gst, record=True)
    assert core.backward_sequence(core.last_X, dout, unpack=False) is None
    again = core._unpack(grad=True)
    # (the one-workgroup kernels' reductions are not fixed-order: equal to rounding, not to the bit.  Rounding of WHAT: a strength /
    # gate gradient of 1e-6 is a cancelled sum of terms the size of the largest gradients, so its run-to-run difference scales
    # with those: 1e-5 of the tensor's own scale + 1e-7 of the largest entry of any tensor)
    gmax_all = max(float(grads[k].abs().max()) for k in grads)
    for k in grads:
        assert float((again[k] - grads[k]).abs().max()) <= 1e-5 * float(grads[k].abs().max()) + 1e-7 * gmax_all + 1e-12, k
    assert not bad, bad


@pytest.mark.parametrize("Wn,W", [(1, 12), (2, 10)], ids=["one_write", "two_writes_w10"])
def test_dnc_segmented_bptt_equals_whole_sequence(cuda, Wn, W):
    """Long-horizon path (config 5): BPTT in re-recorded segments from state checkpoints gives the gradients of the
    single recorded pass (same kernels, same order inside a step; only the LDS-atomic column sums may reorder).
    The two-head case carries one precedence gradient per head between segments and a padded word size."""
    from ntmtrack import dnc as G
    rng = np.random.default_rng(23)
    N, R, hid, O, Din, S, B = 32, 3, 24, 2, 20, 13, 3
    cfg = D.DNCConfig(Din, O, memory_size=N, word_size=W, num_reads=R, num_writes=Wn, hidden_size=hid, clip_value=20.0)
    p = D.init_params(cfg, rng)
    for k in p:
        if k.startswith("memory_access/") and k.endswith("/w"):
            p[k] = (p[k] * 4).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((S, B, Din)).astype(np.float32)).to(cuda)
    dout = torch.from_numpy(rng.standard_normal((B, S, O)).astype(np.float32)).to(cuda)
    res = {}
    for seg in (None, 5, 1):
        core = G.DNC({"memory_size": N, "word_size": W, "num_reads": R, "num_writes": Wn}, {"hidden_size": hid}, O, 20.0, device=cuda)
        core.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
        core.bptt_segment = seg
        out, st = core.run_sequence(x, None, record=True)
        assert (core.last_segments is not None) == (seg is not None)
        grads = core.backward_sequence(core.last_X, dout)
        res[seg] = (out.cpu().numpy(), st.access_state.memory.cpu().numpy(), {k: v.cpu().numpy() for k, v in grads.items()})
    for seg in (5, 1):
        np.testing.assert_array_equal(res[seg][0], res[None][0])
        np.testing.assert_array_equal(res[seg][1], res[None][1])
        for k, g in res[None][2].items():
            scale = np.max(np.abs(g)) + 1e-30
            assert np.max(np.abs(res[seg][2][k] - g)) <= 2e-5 * scale, (seg, k)


def test_dnc_offset_tracker_training_step(cuda):
    """One optimiser step of the DNC tracker head (no VGG): loss and the clipped TF-RMSProp update equal the
    oracle chain (torch-autograd gradients -> clip_by_global_norm(50) -> RMSProp(decay .9, momentum 0, eps 1e-10))."""
    from oracle import ntm_oracle as O
    from oracle import dnc_oracle_torch as DT
    from ntmtrack import tracker
    B, T = 2, 2
    rng = np.random.default_rng(8)
    trk = tracker.DNCOffsetTracker(B, T, vgg_weights=None, mem_size=32, mem_dim=16, hidden_size=32, read_head_size=2,
                                   write_head_size=1, clip_value=20, device=cuda, seed=11)
    sd = {k: v.numpy() for k, v in trk.core.state_dict().items()}
    cfg = D.DNCConfig(514, 2, memory_size=32, word_size=16, num_reads=2, num_writes=1, hidden_size=32, clip_value=20)
    fmap = np.maximum(rng.standard_normal((B * T, 28, 28, 512)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
    offs = rng.uniform(-.5, .5, size=(B, T, 2)).astype(np.float32)
    x = O.serialize_inputs(O.extract_features(fmap).reshape(B, T, 64, 512), gts)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sd.items()}
    ys, _ = DT.run_model(cfg, pt, torch.tensor(np.ascontiguousarray(np.transpose(x, (1, 0, 2))), dtype=torch.float64))
    logits = ys.permute(1, 0, 2)
    from oracle import ntm_oracle_torch as OT
    loss_ref, _ = OT.offset_loss(logits, torch.tensor(offs, dtype=torch.float64))
    loss_ref.backward()
    names = sorted(sd)
    clipped, gn = O.clip_by_global_norm([pt[k].grad.numpy() for k in names], 50.0)
    loss, _ = trk.loss_and_grads(torch.from_numpy(fmap).to(cuda), torch.from_numpy(gts[:, 0].copy()).to(cuda),
                                 torch.from_numpy(offs).to(cuda))
    trk.opt.step()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), float(loss_ref.detach()), rtol=1e-4)
    np.testing.assert_allclose(float(trk.opt.gnorm.cpu()), gn, rtol=3e-3)
    new = trk.core.state_dict()
    for k, g in zip(names, clipped):
        ref_new, _, _ = O.rmsprop_step(sd[k].astype(np.float64), g, np.ones_like(g), np.zeros_like(g), lr=1e-4, decay=0.9, momentum=0.0, eps=1e-10)
        step = np.max(np.abs(ref_new - sd[k]))
        err = np.max(np.abs(new[k].numpy().astype(np.float64) - ref_new))
        assert err <= 6e-8 + 5e-3 * step, (k, err, step)


def test_dnc_full_length_bptt_gradients_match_autograd_oracle(cuda):
    """BASELINE config 3's cell (DNC 256x64, 4 read heads, hidden 200, clip 20) over its FULL horizon, S = 1300 strictly
    sequential steps (20 serialised frames) of the tracking task: loss and every gradient tensor against the float64
    torch-autograd restatement (direct_offset_output_with_dnc.py:534-541, :615-620).

    Two bounds.  (1) Every tensor, on the scale of the whole gradient bucket (what clip_by_global_norm and the RMSProp step
    see): max|g_hip - g_f64| <= 3e-4 x the largest gradient entry of ANY tensor.  Measured: 1.1e-4 at B = 1 and 4.6e-5 at B = 2,
    both on output_linear/b, where a float32 evaluation of the SAME restatement is 7.8e-5 off float64 (B = 2): that gradient is
    the sum of d loss / d logit over the 19 delimiter steps, and the logits themselves carry the forward pass's float32 drift
    over 1300 steps (asserted <= 1e-4 above), so 1e-5 is not a bound float32 arithmetic can meet over this horizon.  (2) The tensors that carry the bucket (own largest entry
    >= 1 % of the bucket's): within 2e-3 of float64 on their OWN scale (measured <= 6.4e-4).  The key / strength gradients are
    1e-5 .. 1e-6 of the largest gradient with Sonnet's default initialisation and are sums with heavy cancellation: on their own
    scale both HIP and the float32 oracle are 20 - 100 % off float64 there (measured in round 3 at S = 650 and in this round at
    S = 1300 with both oracles: HIP never further from float64 than the float32 oracle); bound (1) is what holds them: an error of
    180 % on entries of 1e-6 would pass (1) only because it IS 1e-6 of what the optimiser works with.  All columns are printed."""
    from oracle import ntm_oracle as O
    from oracle import ntm_oracle_torch as OT
    from oracle import dnc_oracle_torch as DT
    from ntmtrack import dnc as G
    from ntmtrack import tracker
    B, T = 1, 20
    S = T * 65
    cfg = D.DNCConfig(514, 2, memory_size=256, word_size=64, num_reads=4, num_writes=1, hidden_size=200, clip_value=20)
    rng = np.random.default_rng(23)
    p = D.init_params(cfg, rng)
    feats = np.maximum(rng.standard_normal((B, T, 64, 512)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
    x = O.serialize_inputs(feats, gts)                                  # [B,S,514]
    offs = rng.uniform(-.5, .5, size=(B, T, 2)).astype(np.float32)
    x_tm = np.ascontiguousarray(np.transpose(x, (1, 0, 2)))
    tt = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64)
    pt = {k: tt(v).requires_grad_(True) for k, v in p.items()}
    ys, _ = DT.run_model(cfg, pt, tt(x_tm))
    loss_o, _ = OT.offset_loss(ys.permute(1, 0, 2), tt(offs))
    loss_o.backward()
    loss_ref, ys_ref, g64 = float(loss_o.detach()), ys.detach().numpy(), {k: v.grad.numpy() for k, v in pt.items()}

    core = G.DNC({"memory_size": 256, "word_size": 64, "num_reads": 4, "num_writes": 1}, {"hidden_size": 200}, 2, 20, device=cuda)
    core.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    out, _st = core.run_sequence(torch.from_numpy(x_tm).to(cuda), None, record=True)
    assert out.shape == (S, B, 2)
    logits = out.transpose(0, 1).contiguous()
    np.testing.assert_allclose(logits.cpu().numpy(), np.transpose(ys_ref, (1, 0, 2)), atol=1e-4)
    loss, _pred, dlogits = tracker.offset_loss(logits, torch.from_numpy(offs).to(cuda), T)
    grads = core.backward_sequence(core.last_X, dlogits)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), loss_ref, rtol=1e-4)
    gmax = max(float(np.abs(g64[k]).max()) for k in p)                  # the largest gradient entry of the whole bucket
    print("full-length (S=%d, B=%d) DNC gradients vs float64 autograd; largest entry of all tensors %.3e" % (S, B, gmax))
    print("  tensor, max|ref|, HIP error / own scale, HIP error / bucket scale")
    bad = {}
    for k in sorted(p):
        gh = grads[k].cpu().numpy()
        own = float(np.abs(g64[k]).max())
        err = float(np.max(np.abs(gh - g64[k])))
        e_own, e_glob = err / (own + 1e-30), err / gmax
        print("  %-36s %.3e  %.3e  %.3e" % (k, own, e_own, e_glob))
        if e_glob > 3e-4 or (own >= 1e-2 * gmax and e_own > 2e-3):
            bad[k] = (e_own, e_glob)
    assert not bad, bad


CLUSTER_CASES = [
    # name, N, W, R, hid, S, B, cluster sizes to try
    ("c3_shape", 256, 64, 4, 200, 6, 2, (2, 4, 8)),
    ("small_64x16", 64, 16, 2, 24, 7, 3, (2, 4, 8)),
    ("r1_128x32", 128, 32, 1, 40, 5, 1, (4,)),
    ("r3_odd_hidden", 128, 20, 3, 36, 5, 2, (8,)),
    # batches that are a multiple of 8: the clusters are laid out one XCD each and, where the launch-time handshake
    # confirms it, hand off through plain stores kept in that XCD's L2 (csrc/dnc_cluster.h)
    ("c3_shape_b8_same_xcd", 256, 64, 4, 200, 6, 8, (8,)),
    ("small_64x16_b16_same_xcd", 64, 16, 2, 24, 7, 16, (4, 8)),
]


@pytest.mark.parametrize("form", ["lds", "mp"])
@pytest.mark.parametrize("name,N,W,R,hid,S,B,ks", CLUSTER_CASES, ids=[c[0] for c in CLUSTER_CASES])
def test_dnc_cluster_forward_equals_single_workgroup_kernel(cuda, name, N, W, R, hid, S, B, ks, form):
    """The cluster form (k workgroups per sequence, LDS-resident link rows, two mailbox exchanges per step) against the
    one-workgroup-per-sequence kernel (itself checked against the oracle above): outputs, final state and EVERY BPTT
    record, from a random non-degenerate state, for several cluster sizes."""
    from ntmtrack import dnc as G
    Din, O = 12, 2
    cfg = D.DNCConfig(Din, O, memory_size=N, word_size=W, num_reads=R, num_writes=1, hidden_size=hid, clip_value=20.0)
    rng = np.random.default_rng(31)
    p = D.init_params(cfg, rng)
    for kk in p:
        if kk.endswith("/b") or kk.endswith("b_gates"):
            p[kk] = rng.uniform(-0.3, 0.3, size=p[kk].shape).astype(np.float32)
        if kk.startswith("memory_access/") and kk.endswith("/w"):
            p[kk] = (p[kk] * 6).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((S, B, Din)).astype(np.float32)).to(cuda)
    st0 = _random_state(cfg, B, rng)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(cuda)
    a0 = st0.access_state
    gst = G.DNCState(t(st0.access_output), G.AccessState(t(a0.memory), t(a0.read_weights), t(a0.write_weights),
                     G.TemporalLinkageState(t(a0.linkage.link), t(a0.linkage.precedence_weights)), t(a0.usage)),
                     G.LSTMState(t(st0.controller_state.hidden), t(st0.controller_state.cell)))

    def run(k):
        core = G.DNC({"memory_size": N, "word_size": W, "num_reads": R, "num_writes": 1}, {"hidden_size": hid}, O, 20.0, device=cuda)
        core.load_state_dict({kk: torch.from_numpy(v) for kk, v in p.items()})
        core.cluster_k = k
        core.cluster_form = form if k else None
        out, st = core.run_sequence(x, gst, record=True)
        core.check_cluster()
        torch.cuda.synchronize()
        assert k == 0 or core.last_cluster_k == 1 or core.last_cluster_form == form
        return out, st, core.last_record, core.last_cluster_k

    ref_out, ref_st, ref_rec, used = run(0)
    assert used == 1
    tried = 0
    for k in ks:
        out, st, rec, used = run(k)
        if used != k:                 # this cluster size does not fit (e.g. 128 link rows x 256 do not fit LDS beside the memory)
            assert used == 1
            continue
        tried += 1
        tol = dict(atol=2e-5, rtol=0)
        np.testing.assert_allclose(out.cpu().numpy(), ref_out.cpu().numpy(), err_msg="out k=%d" % k, **tol)
        for nm, a_, b_ in (("memory", st.access_state.memory, ref_st.access_state.memory),
                           ("link", st.access_state.linkage.link, ref_st.access_state.linkage.link),
                           ("usage", st.access_state.usage, ref_st.access_state.usage),
                           ("rw", st.access_state.read_weights, ref_st.access_state.read_weights),
                           ("ww", st.access_state.write_weights, ref_st.access_state.write_weights),
                           ("prec", st.access_state.linkage.precedence_weights, ref_st.access_state.linkage.precedence_weights),
                           ("reads", st.access_output, ref_st.access_output),
                           ("h", st.controller_state.hidden, ref_st.controller_state.hidden),
                           ("c", st.controller_state.cell, ref_st.controller_state.cell)):
            np.testing.assert_allclose(a_.cpu().numpy(), b_.cpu().numpy(), err_msg="%s k=%d" % (nm, k), **tol)
        for nm in G.DNC.REC_NAMES:
            np.testing.assert_allclose(rec[nm].cpu().numpy(), ref_rec[nm].cpu().numpy(), err_msg="record %s k=%d" % (nm, k), **tol)
    assert tried >= 1, "no cluster size of %s was usable" % (ks,)


def test_dnc_cluster_forward_full_length_is_deterministic(cuda):
    """Config 3 at full size (B 32, S 1300, k = 8: all 256 CUs): two launches are bitwise identical, no hand-off times
    out, and sequence 5 inside the batch equals sequence 5 alone (run as its own cluster launch)."""
    from ntmtrack import dnc as G
    B, S = 32, 1300
    core = G.DNC({"memory_size": 256, "word_size": 64, "num_reads": 4, "num_writes": 1}, {"hidden_size": 200}, 2, 20.0,
                 input_dim=514, device=cuda, seed=4)
    g = torch.Generator().manual_seed(8)
    x = torch.randn((S, B, 514), generator=g).to(cuda) * 0.5
    o1, s1 = core.run_sequence(x)
    core.check_cluster()
    assert core.last_cluster_k == 8
    (same, total), = core.cluster_placement()
    print("clusters that ran the same-XCD hand-off form: %d of %d" % (same, total))
    assert total == B and 0 <= same <= B          # placement is observed, never promised: a diagnostic, not a requirement
    o2, s2 = core.run_sequence(x)
    core.check_cluster()
    assert torch.equal(o1, o2) and torch.equal(s1.access_state.linkage.link, s2.access_state.linkage.link)
    assert torch.isfinite(o1).all()
    o5, _ = core.run_sequence(x[:, 5:6].contiguous())
    core.check_cluster()
    assert torch.equal(o5[:, 0], o1[:, 5])


@pytest.mark.parametrize("form", ["lds", "mp"])
@pytest.mark.parametrize("name,N,W,R,hid,S,B,ks", CLUSTER_CASES, ids=[c[0] for c in CLUSTER_CASES])
def test_dnc_cluster_bptt_equals_single_workgroup_kernel(cuda, name, N, W, R, hid, S, B, ks, form):
    """Cluster BPTT (d(link) rows LDS resident and split k ways, d(memory) in registers, two exchanges per step) against
    the one-workgroup-per-sequence BPTT kernel (itself checked against torch autograd above) on the same recorded
    sequence: every gradient tensor to 1e-4 of its largest entry, bitwise identical across two runs (no float
    atomics), also when the sequence is cut into BPTT segments (gradients carried between launches)."""
    from ntmtrack import dnc as G
    Din, O = 12, 2
    cfg = D.DNCConfig(Din, O, memory_size=N, word_size=W, num_reads=R, num_writes=1, hidden_size=hid, clip_value=20.0)
    rng = np.random.default_rng(41)
    p = D.init_params(cfg, rng)
    for kk in p:
        if kk.endswith("/b") or kk.endswith("b_gates"):
            p[kk] = rng.uniform(-0.3, 0.3, size=p[kk].shape).astype(np.float32)
        if kk.startswith("memory_access/") and kk.endswith("/w"):
            p[kk] = (p[kk] * 4).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((S, B, Din)).astype(np.float32)).to(cuda)
    dout = torch.from_numpy(rng.standard_normal((B, S, O)).astype(np.float32)).to(cuda)
    st0 = _random_state(cfg, B, rng)
    t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(cuda)
    a0 = st0.access_state
    gst = G.DNCState(t(st0.access_output), G.AccessState(t(a0.memory), t(a0.read_weights), t(a0.write_weights),
                     G.TemporalLinkageState(t(a0.linkage.link), t(a0.linkage.precedence_weights)), t(a0.usage)),
                     G.LSTMState(t(st0.controller_state.hidden), t(st0.controller_state.cell)))

    def run(k, segment=None):
        core = G.DNC({"memory_size": N, "word_size": W, "num_reads": R, "num_writes": 1}, {"hidden_size": hid}, O, 20.0, device=cuda)
        core.load_state_dict({kk: torch.from_numpy(v) for kk, v in p.items()})
        core.cluster_k = k
        core.cluster_form = form if k else None
        core.bptt_segment = segment
        core.run_sequence(x, gst, record=True)
        grads = core.backward_sequence(core.last_X, dout)
        core.check_cluster()
        torch.cuda.synchronize()
        assert k == 0 or core.last_cluster_bwd_k == 1 or core.last_cluster_bwd_form == form
        return {kk: v.clone() for kk, v in grads.items()}, core.last_cluster_bwd_k

    ref, used = run(0)
    assert used == 1
    tried = 0
    for k in ks:
        got, used = run(k)
        if used != k:
            assert used == 1
            continue
        tried += 1
        for kk in sorted(ref):
            a_, b_ = got[kk].cpu().numpy(), ref[kk].cpu().numpy()
            err = float(np.max(np.abs(a_ - b_)) / (np.max(np.abs(b_)) + 1e-30))
            assert err < 1e-4, "%s k=%d: %.3e" % (kk, k, err)
        again, _ = run(k)
        assert all(torch.equal(again[kk], got[kk]) for kk in got), "cluster BPTT is not bitwise reproducible (k=%d)" % k
        seg, _ = run(k, segment=max(2, S // 2))
        for kk in sorted(ref):
            a_, b_ = seg[kk].cpu().numpy(), got[kk].cpu().numpy()
            err = float(np.max(np.abs(a_ - b_)) / (np.max(np.abs(b_)) + 1e-30))
            assert err < 2e-5, "segmented %s k=%d: %.3e" % (kk, k, err)
    assert tried >= 1


@pytest.mark.parametrize("form", ["lds", "mp"])
def test_dnc_segmented_bptt_does_not_overlap_two_cooperative_grids(cuda, form):
    """Segmented BPTT re-records segment s - 1 while segment s is back-propagated -- on a side stream only when both
    grids fit the device together.  The cluster kernels are cooperative (every workgroup of a launch must be resident
    before any hand-off completes): at B 32 x k 8 the re-recording forward and the backward each want all 256 CUs, side
    by side each could be half-dispatched and spin until the bounded waits abort both.  Here: the passes are serialised
    (last_rerecord_overlapped False), no hand-off times out, and the gradients equal the unsegmented run's."""
    from ntmtrack import dnc as G
    N, W, R, hid, O, Din, S, B, k = 64, 16, 2, 32, 2, 20, 12, 32, 8
    g = torch.Generator().manual_seed(5)
    x = torch.randn((S, B, Din), generator=g).to(cuda)
    dout = torch.randn((B, S, O), generator=g).to(cuda)
    res = {}
    for seg in (None, S // 3):
        core = G.DNC({"memory_size": N, "word_size": W, "num_reads": R, "num_writes": 1}, {"hidden_size": hid}, O, 20.0,
                     input_dim=Din, device=cuda, seed=6)
        core.cluster_k, core.cluster_form, core.bptt_segment = k, form, seg
        out, _ = core.run_sequence(x, None, record=True)
        grads = core.backward_sequence(core.last_X, dout)
        core.check_cluster()
        torch.cuda.synchronize()
        assert core.last_cluster_k == k and core.last_cluster_bwd_k == k and core.last_cluster_form == form
        if seg is not None:
            assert core.last_segments is not None and core.last_rerecord_overlapped is False
        res[seg] = (out.clone(), {kk: v.clone() for kk, v in grads.items()})
    assert torch.equal(res[None][0], res[S // 3][0])
    for kk, gref in res[None][1].items():
        err = float((res[S // 3][1][kk] - gref).abs().max() / (gref.abs().max() + 1e-30))
        assert err <= 2e-5, (kk, err)


def test_dnc_cluster_abort_reaches_the_loss_without_a_sync(cuda):
    """An aborted cluster launch must not feed the optimiser silently: with the workspace's sticky error word set (what a
    timed-out hand-off does), DNC.guard turns the loss AND the gradient into NaN on the device (NaN survives the
    data-parallel SUM all-reduce; zeros would not), the checked optimiser step behind it leaves parameters and slots
    untouched and counts the skipped step, and check_cluster raises -- and clears the word."""
    from ntmtrack import dnc as G
    from ntmtrack._lib import NtkError
    from ntmtrack.tracker import RMSPropClip
    for form in ("lds", "mp"):
        core = G.DNC({"memory_size": 64, "word_size": 16, "num_reads": 2, "num_writes": 1}, {"hidden_size": 32}, 2, 20.0,
                     input_dim=20, device=cuda, seed=6)
        core.cluster_k, core.cluster_form = 4, form
        x = torch.randn((5, 2, 20), generator=torch.Generator().manual_seed(1)).to(cuda)
        core.run_sequence(x, None, record=True)
        core.backward_sequence(core.last_X, torch.ones((2, 5, 2), device=cuda))
        loss, grad = torch.ones(1, device=cuda), core.params.grad
        core.guard(loss, grad)
        torch.cuda.synchronize()
        assert float(loss) == 1.0 and float(grad.abs().max()) > 0          # clean run: untouched
        opt = RMSPropClip(core.params, 1e-2, 0.9, 0.0, 1e-10, 50.0)
        before = core.params.flat.clone()
        opt.step(loss)
        torch.cuda.synchronize()
        assert float(loss) == 1.0 and int(opt.skipped) == 0 and not torch.equal(core.params.flat, before)
        opt.check()
        # plant what mp_wait / the latch kernel write on a timeout: the sticky word of the forward workspace
        assert core.inject_abort(2)
        before, ms0 = core.params.flat.clone(), opt.ms.clone()
        core.guard(loss, grad)
        torch.cuda.synchronize()
        assert torch.isnan(loss).all() and torch.isnan(grad).all()
        loss.fill_(1.0)
        opt.step(loss)                                                      # NaN norm: nothing applied, loss poisoned, step counted
        torch.cuda.synchronize()
        assert torch.equal(core.params.flat, before) and torch.equal(opt.ms, ms0)
        assert torch.isnan(loss).all() and int(opt.skipped) == 1
        with pytest.raises(NtkError):
            opt.check()
        opt.check()                                                         # the counter was cleared
        with pytest.raises(NtkError):
            core.check_cluster()
        core.check_cluster()                                                # the word was cleared

"""GPU parity: NTM sequence kernel + tracking head vs the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import ntm_oracle as O

pytestmark = pytest.mark.gpu


def _mk(cfg_kwargs, D, O_dim, seed, scale=0.05):
    cfg = O.NTMConfig(D, O_dim, **cfg_kwargs)
    rng = np.random.default_rng(seed)
    params = O.init_params(cfg, rng, scale=scale)
    # non-zero biases so the bias path is exercised
    for k in params:
        if k.endswith("biases"):
            params[k] = rng.uniform(-scale, scale, size=params[k].shape).astype(np.float32)
    return cfg, params, rng


def _cell(cfg, params, cuda):
    from ntmtrack.ntm import NTMCell
    cell = NTMCell(cfg.output_dim, mem_size=cfg.mem_size, mem_dim=cfg.mem_dim, shift_range=cfg.shift_range,
                   controller_hidden_size=cfg.hidden, controller_num_layers=1,
                   write_head_size=cfg.write_heads, read_head_size=cfg.read_heads,
                   write_first=cfg.write_first, device=cuda)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, input_dim=cfg.input_dim)
    return cell


CASES = [
    # (name, cfg kwargs, D, S, B)
    ("c2_r4w1", dict(mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=200, controller_num_layers=1,
                     write_head_size=1, read_head_size=4), 514, 12, 3),
    ("c1_copy_r1w1", dict(mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=100, controller_num_layers=1,
                          write_head_size=1, read_head_size=1), 4, 41, 2),
    ("write_first_multiwrite", dict(mem_size=64, mem_dim=8, shift_range=2, controller_hidden_size=96,
                                    controller_num_layers=1, write_head_size=2, read_head_size=3, write_first=True), 10, 7, 2),
    ("batch1_odd_dims", dict(mem_size=192, mem_dim=13, shift_range=1, controller_hidden_size=77, controller_num_layers=1,
                             write_head_size=1, read_head_size=2), 9, 5, 1),
    ("shift_range_3", dict(mem_size=64, mem_dim=8, shift_range=3, controller_hidden_size=48, controller_num_layers=1,
                           write_head_size=1, read_head_size=2), 10, 6, 2),
    ("shift_range_4", dict(mem_size=128, mem_dim=6, shift_range=4, controller_hidden_size=40, controller_num_layers=1,
                           write_head_size=2, read_head_size=1, write_first=True), 7, 5, 2),      # nine taps: the kernels' widest
]


@pytest.mark.parametrize("name,kw,D,S,B", CASES, ids=[c[0] for c in CASES])
def test_sequence_matches_oracle(cuda, name, kw, D, S, B):
    cfg, params, rng = _mk(kw, D, 2, seed=11)
    x = rng.standard_normal((B, S, D)).astype(np.float32)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    outs, logits, fin, states = O.loop_ntm_tracker(cfg, p64, x.astype(np.float64), return_states=True)

    from ntmtrack.ntm import LoopNTMTracker
    cell = _cell(cfg, params, cuda)
    trk = LoopNTMTracker.__new__(LoopNTMTracker)
    trk.cell, trk.initializer, trk.sequence_length = cell, None, S
    o_gpu, l_gpu = trk(torch.from_numpy(x).to(cuda), record=True)
    torch.cuda.synchronize()
    # north_star tolerance 1e-4 (fp32); observed error is ~1e-6
    np.testing.assert_allclose(l_gpu.cpu().numpy(), logits, atol=2e-5, rtol=0)
    np.testing.assert_allclose(o_gpu.cpu().numpy(), outs, atol=2e-5, rtol=0)
    for key in ("M", "w", "read", "controller_state"):
        np.testing.assert_allclose(trk.last_state[key].cpu().numpy(), fin[key], atol=2e-5, rtol=0, err_msg=key)
    # per-step records (what the reference's TensorArrays Ms/ws/reads hold, ntm_tracker_new.py:59-61)
    rec = trk.last_record
    for t in (0, S // 2, S - 1):
        np.testing.assert_allclose(rec["M"][:, t].cpu().numpy(), states[t]["M"], atol=2e-5, rtol=0)
        np.testing.assert_allclose(rec["w"][:, t].cpu().numpy(), states[t]["w"], atol=2e-5, rtol=0)
        np.testing.assert_allclose(rec["read"][:, t].cpu().numpy(), states[t]["read"], atol=2e-5, rtol=0)


def test_step_api_8tuple_and_debug(cuda):
    """cell(inputs, prev_state) with a state dict (the test_tracker.py:340-341 form) and with
    keyword state (the LoopNTMTracker form, ntm_tracker_new.py:54-56) agree with the oracle step."""
    kw = CASES[0][1]
    cfg, params, rng = _mk(kw, 514, 2, seed=5)
    B = 2
    x = rng.standard_normal((B, 514)).astype(np.float32)
    st = O.zero_state(cfg, params, B)
    # perturb the state so it is not the broadcast initial one
    st = {k: (v + rng.uniform(0, 0.05, size=v.shape)).astype(np.float32) for k, v in st.items()}
    out, logit, new, dbg = O.ntm_step(cfg, params, x, st)
    cell = _cell(cfg, params, cuda)
    tst = {k: torch.from_numpy(v).to(cuda) for k, v in st.items()}
    res = cell(torch.from_numpy(x).to(cuda), tst)
    assert len(res) == 8
    o, l, state, debug, M, w, read, cs = res
    res2 = cell(torch.from_numpy(x).to(cuda), None, M_prev=tst["M"], w_prev=tst["w"], read_prev=tst["read"],
                controller_state=tst["controller_state"])
    torch.cuda.synchronize()
    for a, b_ in zip((o, l, M, w, read, cs), (res2[0], res2[1], res2[4], res2[5], res2[6], res2[7])):
        assert torch.equal(a, b_)
    np.testing.assert_allclose(l.cpu().numpy(), logit, atol=1e-5)
    np.testing.assert_allclose(o.cpu().numpy(), out, atol=1e-5)
    np.testing.assert_allclose(M.cpu().numpy(), new["M"], atol=1e-5)
    np.testing.assert_allclose(w.cpu().numpy(), new["w"], atol=1e-5)
    np.testing.assert_allclose(read.cpu().numpy(), new["read"], atol=1e-5)
    np.testing.assert_allclose(cs.cpu().numpy(), new["controller_state"], atol=1e-5)
    assert set(state.keys()) == {"M", "w", "read", "controller_state"}
    # every one of the 19 tensors of the reference's debug dict (ntm_cell.py:230-250), against the oracle's
    assert set(debug.keys()) == {"k", "gamma", "add", "erase", "bega", "g", "sw", "similarity", "w_content_focused", "w_gated",
                                 "w_conv", "w_conv_powed", "w", "w_read", "w_write", "M", "M_prev", "M_write", "M_erase"}
    R = cfg.read_heads
    ref = dict(dbg)
    ref.update(beta=dbg["beta"], w_read=dbg["w"][:, :R], w_write=dbg["w"][:, R:], M=new["M"], M_prev=st["M"])
    for key, okey in (("k", "k"), ("bega", "beta"), ("g", "g"), ("gamma", "gamma"), ("erase", "erase"),
                      ("add", "add"), ("sw", "sw"), ("similarity", "similarity"), ("w_content_focused", "w_content_focused"),
                      ("w_gated", "w_gated"), ("w_conv", "w_conv"), ("w_conv_powed", "w_conv_powed"), ("w", "w"),
                      ("w_read", "w_read"), ("w_write", "w_write"), ("M", "M"), ("M_prev", "M_prev"), ("M_write", "M_write"),
                      ("M_erase", "M_erase")):
        got, want = debug[key].cpu().numpy(), np.asarray(ref[okey])
        assert got.shape == want.shape, (key, got.shape, want.shape)
        np.testing.assert_allclose(got, want, atol=1e-5, err_msg=key)


def test_zero_state_matches_oracle(cuda):
    kw = CASES[0][1]
    cfg, params, rng = _mk(kw, 514, 2, seed=3)
    cell = _cell(cfg, params, cuda)
    st = cell.zero_state(4)
    ref = O.zero_state(cfg, params, 4)
    torch.cuda.synchronize()
    for k in ref:
        np.testing.assert_allclose(st[k].cpu().numpy(), ref[k], atol=1e-6, err_msg=k)


def test_unsupported_configs_fail_loudly(cuda):
    from ntmtrack.ntm import NTMCell
    from ntmtrack._lib import NtkError
    cell = NTMCell(2, mem_size=100, mem_dim=20, controller_hidden_size=64, controller_num_layers=1,
                   write_head_size=1, read_head_size=1, input_dim=8, device=cuda)
    with pytest.raises(NtkError):                                   # mem_size must be a multiple of 64
        st = cell.zero_state(1)
        cell(torch.zeros((1, 8), device=cuda), st)
    # limits of the fused kernel's decomposition are refused, never computed wrongly: shift_range 5
    # (11 taps > the 9-tap register array; up to 4 is served since round 4), 16 heads (one wave per head), hidden 1000 (> 960)
    for kw in (dict(shift_range=5, write_head_size=1, read_head_size=1, controller_hidden_size=64),
               dict(shift_range=1, write_head_size=8, read_head_size=8, controller_hidden_size=64, mem_dim=4),
               dict(shift_range=1, write_head_size=1, read_head_size=1, controller_hidden_size=1000)):
        c = NTMCell(2, mem_size=64, mem_dim=kw.pop("mem_dim", 8), controller_num_layers=1, input_dim=8, device=cuda, **kw)
        with pytest.raises(NtkError):
            c(torch.zeros((1, 8), device=cuda), c.zero_state(1))


def test_full_length_sequence_drift(cuda):
    """BASELINE config-2 length (T=20 frames -> S=1300 strictly sequential steps): fp32 rounding must not
    drift past the north_star tolerance (1e-4) against the float64 oracle on the quantity the tracker
    consumes, tanh(logit), nor on the final memory."""
    kw = CASES[0][1]
    cfg, params, rng = _mk(kw, 514, 2, seed=123)
    B, T = 2, 20
    feats = np.maximum(rng.standard_normal((B, T, 64, 512)), 0).astype(np.float32)
    x = O.serialize_inputs(feats, rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32))
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    _, logits, fin = O.loop_ntm_tracker(cfg, p64, x.astype(np.float64))
    from ntmtrack.ntm import LoopNTMTracker
    cell = _cell(cfg, params, cuda)
    trk = LoopNTMTracker.__new__(LoopNTMTracker)
    trk.cell, trk.initializer, trk.sequence_length = cell, None, T * 65
    _o, l_gpu = trk(torch.from_numpy(x).to(cuda))
    torch.cuda.synchronize()
    err = np.max(np.abs(np.tanh(l_gpu.cpu().numpy()) - np.tanh(logits)))
    assert err < 1e-4, err
    assert np.max(np.abs(trk.last_state["M"].cpu().numpy() - fin["M"])) < 1e-4


@pytest.mark.parametrize("layers", [2, 3])
def test_multilayer_controller_steps_match_oracle(cuda, layers):
    """MultiRNNCell controller with L > 1 BasicLSTMCell layers (ntm_cell.py:45-50, :101-105; the constructor default is
    10): StackedNTMCell steps -- lower layers as LSTM steps, top layer + addressing in the fused kernel -- against the
    oracle, chained over 5 steps (state layout [c_0, h_0, c_1, h_1, ...])."""
    from ntmtrack.ntm import NTMCell, StackedNTMCell
    rng = np.random.default_rng(31)
    cfg = O.NTMConfig(11, 3, mem_size=64, mem_dim=12, shift_range=1, controller_hidden_size=24, controller_num_layers=layers,
                      write_head_size=2, read_head_size=2)
    params = O.init_params(cfg, rng, scale=0.3)
    for k in params:
        if k.endswith("biases"):
            params[k] = rng.uniform(-0.2, 0.2, size=params[k].shape).astype(np.float32)
    cell = NTMCell(3, mem_size=64, mem_dim=12, shift_range=1, controller_hidden_size=24, controller_num_layers=layers,
                   write_head_size=2, read_head_size=2, device=cuda)
    assert isinstance(cell, StackedNTMCell)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    B = 3
    st = cell.zero_state(B)
    ost = O.zero_state(cfg, params, B)
    assert st["controller_state"].shape == (B, 2 * 24 * layers)
    for t in range(5):
        x = rng.standard_normal((B, 11)).astype(np.float32)
        out, logit, st, _dbg, M, w, read, cs = cell(torch.from_numpy(x).to(cuda), st)
        oout, ologit, ost, _ = O.ntm_step(cfg, params, x, ost)
        np.testing.assert_allclose(logit.cpu().numpy(), ologit, atol=2e-5)
        np.testing.assert_allclose(out.cpu().numpy(), oout, atol=2e-5)
        np.testing.assert_allclose(cs.cpu().numpy(), ost["controller_state"], atol=2e-5)
        np.testing.assert_allclose(M.cpu().numpy(), ost["M"], atol=2e-5)
        np.testing.assert_allclose(w.cpu().numpy(), ost["w"], atol=2e-5)
        np.testing.assert_allclose(read.cpu().numpy(), ost["read"], atol=2e-5)
    sd = cell.state_dict()
    assert set(sd) == set(params) and all(np.array_equal(sd[k].numpy(), params[k]) for k in params)
    # LoopNTMTracker over a deep cell = Python loop over step()
    from ntmtrack.ntm import LoopNTMTracker
    trk = LoopNTMTracker(4, 3, None, mem_size=64, mem_dim=12, shift_range=1, controller_hidden_size=24,
                         controller_num_layers=layers, write_head_size=2, read_head_size=2, device=cuda)
    trk.cell.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    xs = rng.standard_normal((B, 4, 11)).astype(np.float32)
    outs, logits = trk(torch.from_numpy(xs).to(cuda))
    oouts, ologits, _ = O.loop_ntm_tracker(cfg, params, xs)
    np.testing.assert_allclose(logits.cpu().numpy(), ologits, atol=3e-5)
    np.testing.assert_allclose(outs.cpu().numpy(), oouts, atol=3e-5)


def test_static_unroll_trackers_on_a_deep_controller_return_every_state(cuda):
    """PlainNTMTracker / NTMTracker (ntm_tracker_new.py:66-195) on a cell with a 2-layer MultiRNNCell controller (the
    constructor default is 10 layers, ntm_cell.py:18-20): `states` holds the initial state and the state after every step
    (S + 1 dicts, :95-100), controller_state = [c_0, h_0, c_1, h_1] as MultiRNNCell packs it -- against the oracle."""
    from ntmtrack.ntm import PlainNTMTracker, NTMTracker, StackedNTMCell
    rng = np.random.default_rng(8)
    kw = dict(mem_size=64, mem_dim=12, shift_range=1, controller_hidden_size=24, controller_num_layers=2,
              write_head_size=1, read_head_size=2)
    cfg = O.NTMConfig(11, 3, **kw)
    params = O.init_params(cfg, rng, scale=0.3)
    B, S = 2, 6
    plain = PlainNTMTracker(S, 3, device=cuda, **kw)
    assert isinstance(plain.cell, StackedNTMCell)
    plain.cell.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    xs = rng.standard_normal((B, S, 11)).astype(np.float32)
    outs, logits, states, debugs = plain(torch.from_numpy(xs).to(cuda))
    oouts, ologits, _fin, ostates = O.loop_ntm_tracker(cfg, params, xs, return_states=True)
    torch.cuda.synchronize()
    assert len(states) == S + 1
    np.testing.assert_allclose(logits.cpu().numpy(), ologits, atol=3e-5)
    ost0 = O.zero_state(cfg, params, B)
    for key in ("M", "w", "read", "controller_state"):
        np.testing.assert_allclose(states[0][key].cpu().numpy(), ost0[key], atol=1e-6, err_msg=key)
        for t in range(S):
            assert states[t + 1][key].shape == ostates[t][key].shape, (key, t)
            np.testing.assert_allclose(states[t + 1][key].cpu().numpy(), ostates[t][key], atol=3e-5, err_msg="%s step %d" % (key, t))
    assert states[1]["controller_state"].shape == (B, 2 * 24 * 2)
    assert debugs["w"].shape == (B, S, 3, 64) and debugs["M"].shape == (B, S, 64, 12)
    # NTMTracker: same cell behind the target-indicator serialisation
    F = 5
    trk = NTMTracker(S, B, 3, device=cuda, **kw)
    _o, l2, st2, _d = trk(torch.from_numpy(xs).to(cuda), torch.rand((B, F), generator=torch.Generator().manual_seed(1)).to(cuda))
    assert len(st2) == S + 1 and l2.shape == (B, S, 3) and st2[-1]["controller_state"].shape == (B, 2 * 24 * 2)


@pytest.mark.parametrize("S", [1, 2, 7])
def test_wave_specialised_kernels_equal_the_resident_form_on_short_sequences(cuda, monkeypatch, S):
    """The benchmark shape runs the wave-specialised sequence kernels (csrc/ntm_seq_fwd_ws.hip, the WS form of ntm_seq_bwd.hip):
    stream waves carry the h part of the recurrent products from one step into the next, primed before the first step and
    drained after the last.  Their edges -- a one-step sequence (what NTMCell.__call__ launches), a non-trivial initial state,
    a gradient of the FINAL state -- against round 2's kernels (NTK_NTM_*_FORM=res) on the same inputs: every output, every
    record, the state gradient and the weight gradients agree to float32 rounding (the summation order of a gate differs)."""
    kw = CASES[0][1]
    cfg, params, rng = _mk(kw, 514, 2, seed=41, scale=0.2)
    B = 3
    cell = _cell(cfg, params, cuda)
    x = torch.from_numpy(rng.standard_normal((B, S, 514)).astype(np.float32)).to(cuda)
    X = cell._pad_inputs(x)
    st = {k: torch.from_numpy((v + rng.uniform(0, 0.1, size=v.shape)).astype(np.float32)).to(cuda) for k, v in O.zero_state(cfg, params, B).items()}
    dlog = torch.from_numpy(rng.standard_normal((B, S, 2)).astype(np.float32)).to(cuda)
    dfin = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape)).astype(np.float32) * 0.1).to(cuda) for k, v in st.items()}
    res = {}
    for form in ("res", "ws"):
        monkeypatch.setenv("NTK_NTM_FWD_FORM", form)
        monkeypatch.setenv("NTK_NTM_BWD_FORM", form)
        logits, outs, new, rec = cell.run_sequence(X, st, record=True)
        g0 = cell.backward_sequence(X, st, rec, dlog, dfinal=dfin)
        torch.cuda.synchronize()
        res[form] = dict(logits=logits.clone(), outs=outs.clone(), grad=cell.params.grad.clone(),
                         **{"new_" + k: v.clone() for k, v in new.items()}, **{"g0_" + k: v.clone() for k, v in g0.items()},
                         **{"rec_" + k: rec[k].clone() for k in ("gates", "c", "u", "w", "M", "read")})
    for k in res["res"]:
        a, b = res["res"][k], res["ws"][k]
        scale = float(a.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 2e-6 * scale + 1e-7, (k, float((a - b).abs().max()), scale)

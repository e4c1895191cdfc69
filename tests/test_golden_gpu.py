"""GPU: HIP path against the committed golden vectors (tests/golden/*.npz, generated from the oracle
by tests/golden/make_golden.py) -- runs on the GPU box without the reference tree."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import ntm_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _cell(cfg, params, cuda):
    from ntmtrack.ntm import NTMCell
    cell = NTMCell(cfg.output_dim, mem_size=cfg.mem_size, mem_dim=cfg.mem_dim, shift_range=cfg.shift_range,
                   controller_hidden_size=cfg.hidden, controller_num_layers=1, write_head_size=cfg.write_heads,
                   read_head_size=cfg.read_heads, write_first=cfg.write_first, device=cuda)
    cell.load_state_dict({k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in params.items()}, input_dim=cfg.input_dim)
    return cell


def test_step_c2_golden(cuda):
    g = np.load(os.path.join(GOLD, "ntm_step_c2.npz"))
    cfg, p = _mg().c2_params(int(g["seed"]))
    cell = _cell(cfg, p, cuda)
    st = {"M": g["M"], "w": g["w"], "read": g["read"], "controller_state": g["cs"]}
    res = cell(torch.from_numpy(g["x"]).to(cuda), {k: torch.from_numpy(v).to(cuda) for k, v in st.items()})
    torch.cuda.synchronize()
    o, l, state, debug, M, w, read, cs = res
    for got, key in ((l, "logit"), (o, "out"), (M, "M_new"), (w, "w_new"), (read, "read_new"), (cs, "cs_new"),
                     (debug["k"], "k"), (debug["w_content_focused"], "wc"), (debug["w_conv"], "wv")):
        np.testing.assert_allclose(got.cpu().numpy(), g[key], atol=1e-5, err_msg=key)


def test_small_sequence_golden(cuda):
    g = np.load(os.path.join(GOLD, "ntm_seq_small.npz"))
    cfg = _mg().small_cfg()
    p = {k[len("param:"):]: g[k] for k in g.files if k.startswith("param:")}
    from ntmtrack.ntm import LoopNTMTracker
    trk = LoopNTMTracker.__new__(LoopNTMTracker)
    trk.cell, trk.initializer, trk.sequence_length = _cell(cfg, p, cuda), None, g["x"].shape[1]
    outs, logits = trk(torch.from_numpy(g["x"]).to(cuda), record=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], atol=2e-5)
    np.testing.assert_allclose(outs.cpu().numpy(), g["outputs"], atol=2e-5)
    np.testing.assert_allclose(trk.last_record["w"].cpu().numpy(), g["w_steps"], atol=2e-5)
    np.testing.assert_allclose(trk.last_state["M"].cpu().numpy(), g["M_final"], atol=2e-5)


def test_vgg_golden(cuda):
    from ntmtrack import vgg
    v = np.load(os.path.join(GOLD, "vgg_small.npz"))
    ws = O.init_vgg_weights(np.random.default_rng(int(v["seed"])))
    net = vgg.VGG16Conv43(ws, device=cuda)
    x = torch.from_numpy(v["frame"]).to(cuda)
    got = net(x).cpu().numpy()
    np.testing.assert_allclose(got, v["conv4_3"], rtol=0, atol=1e-4 * np.abs(v["conv4_3"]).max())
    p1 = net.forward_chunk(x, upto="conv1_2")      # un-pooled conv1_2; pool on the host for the check
    np.testing.assert_allclose(O.maxpool2x2(p1.cpu().numpy()), v["pool1"], rtol=0, atol=1e-4 * np.abs(v["pool1"]).max())


def test_gradients_golden(cuda):
    from ntmtrack import tracker
    g = np.load(os.path.join(GOLD, "ntm_grads_small.npz"))
    cfg = O.NTMConfig(514, 2, mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=16,
                      controller_num_layers=1, write_head_size=1, read_head_size=2)
    rng = np.random.default_rng(int(g["seed"]))
    p = O.init_params(cfg, rng, scale=0.2)
    feats = np.maximum(rng.standard_normal((2, 2, 64, 512)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(2, 2, 64)).astype(np.float32)
    offs = rng.uniform(-.5, .5, size=(2, 2, 2)).astype(np.float32)
    x = O.serialize_inputs(feats, gts)
    cell = _cell(cfg, p, cuda)
    X = cell._pad_inputs(torch.from_numpy(x).to(cuda))
    st0 = cell.zero_state(2)
    logits, _o, _n, rec = cell.run_sequence(X, st0, record=True)
    loss, pred, dlog = tracker.offset_loss(logits, torch.from_numpy(offs).to(cuda), 2)
    g0 = cell.backward_sequence(X, st0, rec, dlog)
    cell.init_state_backward(g0, 2)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss.cpu()), float(g["loss"]), rtol=1e-4)
    got = cell.params.to_tf(grad=True)
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            err = np.max(np.abs(got[k[5:]].numpy() - ref)) / (np.max(np.abs(ref)) + 1e-30)
            assert err < 2e-3, (k, err)
    W = got["lstm/cell_0/weights"].numpy()
    np.testing.assert_allclose(W.sum(axis=0), g["grad_lstm_w_rowsum"], rtol=0, atol=2e-3 * np.abs(g["grad_lstm_w_rowsum"]).max())
    np.testing.assert_allclose(W[514:], g["grad_lstm_w_tail"], rtol=0, atol=2e-3 * np.abs(g["grad_lstm_w_tail"]).max())

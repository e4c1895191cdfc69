#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ from the CPU oracle (oracle/).

Provenance: the reference (TF1/Python-2/Sonnet) cannot run in the build image, so these vectors
are outputs of the *oracle restatement* on seeded inputs -- regression fixtures that freeze the
oracle and travel to the GPU box.  The vectors that come from the reference's OWN tests are kept
separately in tests/golden/reference_vectors.json (ops_test.py:20-34, dnc/util_test.py:51-53,
the planted cases of dnc/addressing_test.py / access_test.py) and are what pins the oracle.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ntm_oracle as O          # noqa: E402
from oracle import ntm_oracle_torch as OT   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def small_cfg():
    return O.NTMConfig(10, 2, mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=32,
                       controller_num_layers=1, write_head_size=1, read_head_size=2)


def c2_cfg():
    return O.NTMConfig(514, 2, mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=200,
                       controller_num_layers=1, write_head_size=1, read_head_size=4)


def c2_params(seed=2024, scale=0.05):
    """Parameters of the config-2 cell are regenerated from the seed (2.7 MB would not be a small fixture)."""
    cfg = c2_cfg()
    rng = np.random.default_rng(seed)
    p = O.init_params(cfg, rng, scale=scale)
    for k in sorted(p):
        if k.endswith("biases"):
            p[k] = rng.uniform(-scale, scale, size=p[k].shape).astype(np.float32)
    return cfg, p


def main():
    # (1) small sequence, float64 reference values
    cfg = small_cfg()
    rng = np.random.default_rng(7)
    p = O.init_params(cfg, rng, scale=0.3)
    x = rng.standard_normal((2, 6, 10)).astype(np.float32)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    outs, logits, fin, states = O.loop_ntm_tracker(cfg, p64, x.astype(np.float64), return_states=True)
    np.savez(os.path.join(HERE, "ntm_seq_small.npz"), x=x, outputs=outs, logits=logits,
             M_final=fin["M"], w_final=fin["w"], read_final=fin["read"], cs_final=fin["controller_state"],
             w_steps=np.stack([s["w"] for s in states], 1), **{"param:" + k: v for k, v in p.items()})

    # (2) one config-2-shaped step from a perturbed state
    cfg2, p2 = c2_params()
    rng = np.random.default_rng(11)
    B = 2
    x2 = rng.standard_normal((B, 514)).astype(np.float32)
    st = O.zero_state(cfg2, p2, B)
    st = {k: (v + rng.uniform(0, 0.05, size=v.shape)).astype(np.float32) for k, v in st.items()}
    p2_64 = {k: v.astype(np.float64) for k, v in p2.items()}
    st64 = {k: v.astype(np.float64) for k, v in st.items()}
    out, logit, new, dbg = O.ntm_step(cfg2, p2_64, x2.astype(np.float64), st64)
    np.savez(os.path.join(HERE, "ntm_step_c2.npz"), seed=2024, x=x2, M=st["M"], w=st["w"], read=st["read"],
             cs=st["controller_state"], out=out, logit=logit, M_new=new["M"], w_new=new["w"], read_new=new["read"],
             cs_new=new["controller_state"], k=dbg["k"], beta=dbg["beta"], g=dbg["g"], sw=dbg["sw"],
             gamma=dbg["gamma"], wc=dbg["w_content_focused"], wv=dbg["w_conv"])

    # (3) tracking head: serialiser + loss
    rng = np.random.default_rng(13)
    feats = rng.standard_normal((2, 3, 64, 512)).astype(np.float32)
    gts = rng.uniform(0, 1, size=(2, 3, 64)).astype(np.float32)
    X = O.serialize_inputs(feats, gts)
    lg = rng.standard_normal((2, 3 * 65, 2)).astype(np.float32)
    offs = rng.uniform(-.5, .5, size=(2, 3, 2)).astype(np.float32)
    loss, pred = O.offset_loss(lg.astype(np.float64), offs.astype(np.float64))
    # store only what is needed to re-derive X cheaply: checksum rows + the loss case
    np.savez(os.path.join(HERE, "tracking_head.npz"), feats_seed=13, X_rowsum=X.sum(axis=2), X_delim=X[:, :, 512],
             X_target=X[:, :, 513], logits=lg, offsets=offs, loss=loss, pred=pred)

    # (4) VGG trunk on one 32x32 frame (weights from the seed; the HIP conv needs H, W multiples of 4 at conv4)
    rng = np.random.default_rng(17)
    ws = O.init_vgg_weights(rng)
    frame = (rng.uniform(0, 255, size=(1, 32, 32, 3)).astype(np.float32) - O.VGG_MEAN)
    ws64 = {k: (w.astype(np.float64), b.astype(np.float64)) for k, (w, b) in ws.items()}
    f43 = O.vgg16_conv43(frame.astype(np.float64), ws64)
    f12 = O.maxpool2x2(O.vgg16_conv43(frame.astype(np.float64), ws64, upto="conv1_2"))
    np.savez(os.path.join(HERE, "vgg_small.npz"), seed=17, frame=frame, conv4_3=f43, pool1=f12)

    # (5) BPTT gradients of the tracking loss, tiny cell, T=2 frames (torch autograd restatement, float64)
    cfg = O.NTMConfig(514, 2, mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=16,
                      controller_num_layers=1, write_head_size=1, read_head_size=2)
    rng = np.random.default_rng(19)
    p = O.init_params(cfg, rng, scale=0.2)
    feats = np.maximum(rng.standard_normal((2, 2, 64, 512)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(2, 2, 64)).astype(np.float32)
    offs = rng.uniform(-.5, .5, size=(2, 2, 2)).astype(np.float32)
    x = O.serialize_inputs(feats, gts)
    loss, grads, logits, pred = OT.loss_and_grads(cfg, p, x, offs)
    np.savez(os.path.join(HERE, "ntm_grads_small.npz"), seed=19, loss=loss, logits=logits,
             **{"grad:" + k: v.astype(np.float32) for k, v in grads.items() if not k.startswith("lstm/cell_0/weights")},
             grad_lstm_w_rowsum=grads["lstm/cell_0/weights"].sum(axis=0),
             grad_lstm_w_tail=grads["lstm/cell_0/weights"][514:])
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()

"""NumPy restatement of the reference's DNC core.  TEST INFRASTRUCTURE ONLY.

Follows (file:line into the reference tree):
  dnc/util.py:25-45          batch_invert_permutation, batch_gather, one_hot
  dnc/addressing.py:34-56    _vector_norms, weighted_softmax
  dnc/addressing.py:83-105   CosineWeights._build            -> cosine_weights
  dnc/addressing.py:133-249  TemporalLinkage                 -> link_update, precedence_weights,
                                                                directional_read_weights
  dnc/addressing.py:279-405  Freeness                        -> freeness, write_allocation_weights,
                                                                allocation
  dnc/access.py:32-63        _erase_and_write
  dnc/access.py:113-303      MemoryAccess._build / _read_inputs / _write_weights / _read_weights
  dnc/dnc.py:84-134          DNC._build, initial_state
  direct_offset_output_with_dnc.py:66-88   run_model (time-major dynamic_rnn)

Pinned by the reference's own tests (restated in tests/test_oracle_dnc.py):
dnc/util_test.py:51-53 (fixed batch_gather vector), the planted one-hot cases of
dnc/addressing_test.py:180-236, :253-272, :294-314, :335-366 and
dnc/access_test.py:86-111, :123-143, the explicit numpy definition of cosine
weights addressing_test.py:103-118, and the properties listed there.

Third-party arithmetic not in the reference tree (dm-sonnet v1, TensorFlow 1.x,
versions unpinned): snt.LSTM (gates i,j,f,o, forget_bias 1.0, state (hidden,
cell)), snt.Linear (x @ w + b), tf.nn.top_k (descending, ties by lower index),
tf.cumprod(exclusive=True), tf.matrix_set_diag.  They follow the libraries'
documented formulas: **parity unpinned** beyond what the module tests above pin.
"""
from __future__ import annotations

import collections

import numpy as np

EPS = 1e-6   # dnc/addressing.py:28

TemporalLinkageState = collections.namedtuple("TemporalLinkageState", ("link", "precedence_weights"))
AccessState = collections.namedtuple("AccessState", ("memory", "read_weights", "write_weights", "linkage", "usage"))
DNCState = collections.namedtuple("DNCState", ("access_output", "access_state", "controller_state"))
LSTMState = collections.namedtuple("LSTMState", ("hidden", "cell"))


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def softplus(x):
    return np.logaddexp(np.zeros((), x.dtype), x).astype(x.dtype)


def softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / np.sum(e, axis=axis, keepdims=True)).astype(x.dtype)


# ---- dnc/util.py
def batch_invert_permutation(perm):
    inv = np.empty_like(perm)
    rows = np.arange(perm.shape[0])[:, None]
    inv[rows, perm] = np.arange(perm.shape[1])[None, :]
    return inv


def batch_gather(values, indices):
    return np.take_along_axis(values, indices, axis=1)


def one_hot(length, index):
    r = np.zeros(length)
    r[index] = 1
    return r


# ---- dnc/addressing.py
def vector_norms(m):
    return np.sqrt(np.sum(m * m, axis=2, keepdims=True) + m.dtype.type(EPS))


def weighted_softmax(activations, strengths, strengths_op=softplus):
    return softmax(activations * strengths_op(strengths)[..., None], axis=2)


def cosine_weights(memory, keys, strengths, strength_op=softplus):
    """[B,N,W], [B,H,W], [B,H] -> [B,H,N]  (addressing.py:83-105)."""
    dot = keys @ np.transpose(memory, (0, 2, 1))
    norm = vector_norms(keys) @ np.transpose(vector_norms(memory), (0, 2, 1))
    sim = dot / (norm + memory.dtype.type(EPS))
    return weighted_softmax(sim, strengths, strength_op)


def link_update(prev_link, prev_precedence, ww):
    """addressing.py:183-218.  link [B,Wn,N,N], precedence [B,Wn,N], ww [B,Wn,N]."""
    wi = ww[:, :, :, None]
    wj = ww[:, :, None, :]
    pj = prev_precedence[:, :, None, :]
    link = (1 - wi - wj) * prev_link + wi * pj
    n = link.shape[-1]
    link[:, :, np.arange(n), np.arange(n)] = 0
    return link.astype(ww.dtype)


def precedence_weights(prev_precedence, ww):
    """addressing.py:220-240."""
    ws = np.sum(ww, axis=2, keepdims=True)
    return ((1 - ws) * prev_precedence + ww).astype(ww.dtype)


def directional_read_weights(link, prev_read_weights, forward):
    """addressing.py:155-181 -> [B,R,Wn,N]: forward = r @ L^T, backward = r @ L."""
    Wn = link.shape[1]
    r = np.stack([prev_read_weights] * Wn, axis=1)                   # [B,Wn,R,N]
    L = np.transpose(link, (0, 1, 3, 2)) if forward else link
    return np.transpose(r @ L, (0, 2, 1, 3)).astype(link.dtype)


def usage_after_write(prev_usage, ww):
    w = 1 - np.prod(1 - ww, axis=1)
    return (prev_usage + (1 - prev_usage) * w).astype(prev_usage.dtype)


def usage_after_read(prev_usage, free_gate, rw):
    phi = np.prod(1 - free_gate[..., None] * rw, axis=1)
    return (prev_usage * phi).astype(prev_usage.dtype)


def freeness(ww, free_gate, rw, prev_usage):
    """addressing.py:279-305 (write weights enter under stop_gradient)."""
    return usage_after_read(usage_after_write(prev_usage, ww), free_gate, rw)


def allocation(usage):
    """addressing.py:376-405: sort by usage ascending (top_k of non-usage, ties by lower index)."""
    dt = usage.dtype
    u = dt.type(EPS) + (1 - dt.type(EPS)) * usage
    nonusage = 1 - u
    idx = np.argsort(-nonusage, axis=1, kind="stable")
    sorted_nonusage = np.take_along_axis(nonusage, idx, axis=1)
    sorted_usage = 1 - sorted_nonusage
    prod = np.cumprod(sorted_usage, axis=1)
    prod = np.concatenate([np.ones_like(prod[:, :1]), prod[:, :-1]], axis=1)     # exclusive
    sorted_alloc = sorted_nonusage * prod
    return batch_gather(sorted_alloc, batch_invert_permutation(idx)).astype(dt)


def write_allocation_weights(usage, write_gates, num_writes):
    """addressing.py:307-340."""
    usage = usage.copy()
    out = []
    for i in range(num_writes):
        a = allocation(usage)
        out.append(a)
        usage = usage + (1 - usage) * write_gates[:, i:i + 1] * a
    return np.stack(out, axis=1)


# ---- dnc/access.py
def erase_and_write(memory, address, reset_weights, values):
    """access.py:32-63."""
    reset_gate = np.prod(1 - address[:, :, :, None] * reset_weights[:, :, None, :], axis=1)
    return (memory * reset_gate + np.transpose(address, (0, 2, 1)) @ values).astype(memory.dtype)


class AccessConfig(object):
    def __init__(self, memory_size=128, word_size=20, num_reads=1, num_writes=1):
        self.N, self.W, self.R, self.Wn = memory_size, word_size, num_reads, num_writes

    @property
    def interface(self):
        """(name, width, reshape) of the 10 snt.Linear modules in creation order (access.py:170-204)."""
        N, W, R, Wn = self.N, self.W, self.R, self.Wn
        return [
            ("write_vectors", Wn * W), ("erase_vectors", Wn * W), ("free_gate", R), ("allocation_gate", Wn),
            ("write_gate", Wn), ("read_mode", R * (1 + 2 * Wn)), ("write_keys", Wn * W), ("write_strengths", Wn),
            ("read_keys", R * W), ("read_strengths", R),
        ]


def read_inputs(cfg, params, h):
    """access.py:160-218: the ten linears and their activations."""
    B = h.shape[0]
    lin = lambda name: (h @ params["memory_access/%s/w" % name] + params["memory_access/%s/b" % name]).astype(h.dtype)
    W, R, Wn = cfg.W, cfg.R, cfg.Wn
    return {
        "write_vectors": lin("write_vectors").reshape(B, Wn, W),
        "erase_vectors": sigmoid(lin("erase_vectors")).reshape(B, Wn, W),
        "free_gate": sigmoid(lin("free_gate")),
        "allocation_gate": sigmoid(lin("allocation_gate")),
        "write_gate": sigmoid(lin("write_gate")),
        "read_mode": softmax(lin("read_mode").reshape(B, R, 1 + 2 * Wn), axis=2),
        "write_content_keys": lin("write_keys").reshape(B, Wn, W),
        "write_content_strengths": lin("write_strengths"),
        "read_content_keys": lin("read_keys").reshape(B, R, W),
        "read_content_strengths": lin("read_strengths"),
    }


def write_weights(cfg, inputs, memory, usage):
    """access.py:220-257."""
    cw = cosine_weights(memory, inputs["write_content_keys"], inputs["write_content_strengths"])
    aw = write_allocation_weights(usage, inputs["allocation_gate"] * inputs["write_gate"], cfg.Wn)
    ag = inputs["allocation_gate"][..., None]
    wg = inputs["write_gate"][..., None]
    return (wg * (ag * aw + (1 - ag) * cw)).astype(memory.dtype)


def read_weights(cfg, inputs, memory, prev_read_weights, link):
    """access.py:259-303."""
    Wn = cfg.Wn
    cw = cosine_weights(memory, inputs["read_content_keys"], inputs["read_content_strengths"])
    fw = directional_read_weights(link, prev_read_weights, True)
    bw = directional_read_weights(link, prev_read_weights, False)
    rm = inputs["read_mode"]
    backward_mode, forward_mode, content_mode = rm[:, :, :Wn], rm[:, :, Wn:2 * Wn], rm[:, :, 2 * Wn]
    return (content_mode[..., None] * cw + np.sum(forward_mode[..., None] * fw, axis=2)
            + np.sum(backward_mode[..., None] * bw, axis=2)).astype(memory.dtype)


def access_step(cfg, params, h, prev):
    """MemoryAccess._build (access.py:113-158)."""
    inp = read_inputs(cfg, params, h)
    usage = freeness(prev.write_weights, inp["free_gate"], prev.read_weights, prev.usage)
    ww = write_weights(cfg, inp, prev.memory, usage)
    memory = erase_and_write(prev.memory, ww, inp["erase_vectors"], inp["write_vectors"])
    link = link_update(prev.linkage.link, prev.linkage.precedence_weights, ww)
    prec = precedence_weights(prev.linkage.precedence_weights, ww)
    rw = read_weights(cfg, inp, memory, prev.read_weights, link)
    reads = (rw @ memory).astype(memory.dtype)
    return reads, AccessState(memory, rw, ww, TemporalLinkageState(link, prec), usage), inp


def access_initial_state(cfg, B, dtype=np.float32):
    z = lambda *s: np.zeros(s, dtype)
    return AccessState(z(B, cfg.N, cfg.W), z(B, cfg.R, cfg.N), z(B, cfg.Wn, cfg.N),
                       TemporalLinkageState(z(B, cfg.Wn, cfg.N, cfg.N), z(B, cfg.Wn, cfg.N)), z(B, cfg.N))


# ---- dnc/dnc.py
class DNCConfig(object):
    def __init__(self, input_dim, output_size, memory_size=128, word_size=20, num_reads=1, num_writes=1,
                 hidden_size=200, clip_value=0):
        self.D, self.O, self.hid = input_dim, output_size, hidden_size
        self.access = AccessConfig(memory_size, word_size, num_reads, num_writes)
        self.clip = clip_value or 0


def init_params(cfg, rng, dtype=np.float32):
    """Sonnet v1 defaults: truncated-normal(stddev 1/sqrt(fan_in)) weights, zero biases
    (un-vendored library, unpinned; only the shapes and names matter for parity tests)."""
    a = cfg.access

    def tn(fan_in, *shape):
        v = rng.standard_normal(shape)
        v = np.clip(v, -2, 2) / np.sqrt(fan_in)
        return v.astype(dtype)

    in_dim = cfg.D + a.R * a.W + cfg.hid
    p = {"lstm/w_gates": tn(in_dim, in_dim, 4 * cfg.hid), "lstm/b_gates": np.zeros(4 * cfg.hid, dtype)}
    for name, width in a.interface:
        p["memory_access/%s/w" % name] = tn(cfg.hid, cfg.hid, width)
        p["memory_access/%s/b" % name] = np.zeros(width, dtype)
    p["output_linear/w"] = tn(cfg.hid + a.R * a.W, cfg.hid + a.R * a.W, cfg.O)
    p["output_linear/b"] = np.zeros(cfg.O, dtype)
    return p


def clip(cfg, x):
    return np.clip(x, -cfg.clip, cfg.clip).astype(x.dtype) if cfg.clip > 0 else x


def dnc_initial_state(cfg, B, dtype=np.float32):
    a = cfg.access
    return DNCState(np.zeros((B, a.R, a.W), dtype), access_initial_state(a, B, dtype),
                    LSTMState(np.zeros((B, cfg.hid), dtype), np.zeros((B, cfg.hid), dtype)))


def sonnet_lstm(x, state, W, b, forget_bias=1.0):
    """snt.LSTM (Sonnet v1): gates = [x, h] @ w_gates + b_gates; i, j, f, o = split;
    c' = sigmoid(f + forget_bias) * c + sigmoid(i) * tanh(j); h' = tanh(c') * sigmoid(o)."""
    hid = W.shape[1] // 4
    g = (np.concatenate([x, state.hidden], axis=1) @ W + b).astype(x.dtype)
    i, j, f, o = g[:, :hid], g[:, hid:2 * hid], g[:, 2 * hid:3 * hid], g[:, 3 * hid:]
    c2 = sigmoid(f + x.dtype.type(forget_bias)) * state.cell + sigmoid(i) * np.tanh(j)
    h2 = np.tanh(c2) * sigmoid(o)
    return h2.astype(x.dtype), LSTMState(h2.astype(x.dtype), c2.astype(x.dtype))


def dnc_step(cfg, params, x, prev):
    """DNC._build (dnc.py:84-127)."""
    B = x.shape[0]
    a = cfg.access
    ci = np.concatenate([x.reshape(B, -1), prev.access_output.reshape(B, -1)], axis=1)
    h, cs = sonnet_lstm(ci, prev.controller_state, params["lstm/w_gates"], params["lstm/b_gates"])
    h = clip(cfg, h)
    cs = LSTMState(clip(cfg, cs.hidden), clip(cfg, cs.cell))
    reads, acc, inp = access_step(a, params, h, prev.access_state)
    y = np.concatenate([h, reads.reshape(B, -1)], axis=1) @ params["output_linear/w"] + params["output_linear/b"]
    y = clip(cfg, y.astype(x.dtype))
    return y, DNCState(reads, acc, cs), inp


def run_model(cfg, params, inputs_tm, state=None):
    """direct_offset_output_with_dnc.py:66-88: dynamic_rnn over TIME-MAJOR inputs [S,B,D] -> [S,B,O]."""
    S, B, _ = inputs_tm.shape
    st = state or dnc_initial_state(cfg, B, inputs_tm.dtype)
    ys = []
    for t in range(S):
        y, st, _ = dnc_step(cfg, params, inputs_tm[t], st)
        ys.append(y)
    return np.stack(ys, 0), st

"""torch-CPU autograd restatement of the reference's DNC core.  TEST INFRASTRUCTURE ONLY.

Independent second restatement of oracle/dnc_oracle.py (same reference citations) on torch tensors so
autograd supplies what tf.gradients computes through tf.nn.dynamic_rnn over dnc.DNC
(direct_offset_output_with_dnc.py:66-88, :615-620).  Explicit non-differentiable edges of the reference
(SURVEY A.4): tf.stop_gradient(write_weights) in the usage update (dnc/addressing.py:302); tf.nn.top_k passes
gradient to the selected values only (:396-397); batch_invert_permutation is integer-only; clip_by_value
passes gradient inside the interval only (dnc/dnc.py:78-82).
"""
import collections

import torch

EPS = 1e-6

TemporalLinkageState = collections.namedtuple("TemporalLinkageState", ("link", "precedence_weights"))
AccessState = collections.namedtuple("AccessState", ("memory", "read_weights", "write_weights", "linkage", "usage"))
DNCState = collections.namedtuple("DNCState", ("access_output", "access_state", "controller_state"))
LSTMState = collections.namedtuple("LSTMState", ("hidden", "cell"))


def vector_norms(m):
    return torch.sqrt((m * m).sum(2, keepdim=True) + EPS)


def cosine_weights(memory, keys, strengths):
    dot = keys @ memory.transpose(1, 2)
    norm = vector_norms(keys) @ vector_norms(memory).transpose(1, 2)
    sim = dot / (norm + EPS)
    return torch.softmax(sim * torch.nn.functional.softplus(strengths).unsqueeze(-1), dim=2)


def link_update(prev_link, prev_prec, ww):
    wi, wj, pj = ww.unsqueeze(3), ww.unsqueeze(2), prev_prec.unsqueeze(2)
    link = (1 - wi - wj) * prev_link + wi * pj
    n = link.shape[-1]
    mask = 1 - torch.eye(n, dtype=link.dtype)
    return link * mask


def precedence(prev_prec, ww):
    return (1 - ww.sum(2, keepdim=True)) * prev_prec + ww


def directional(link, prw, forward):
    Wn = link.shape[1]
    r = torch.stack([prw] * Wn, 1)
    L = link.transpose(2, 3) if forward else link
    return (r @ L).permute(0, 2, 1, 3)


def allocation(usage):
    u = EPS + (1 - EPS) * usage
    nonusage = 1 - u
    sorted_nonusage, idx = torch.sort(nonusage, dim=1, descending=True, stable=True)
    sorted_usage = 1 - sorted_nonusage
    prod = torch.cumprod(sorted_usage, dim=1)
    prod = torch.cat([torch.ones_like(prod[:, :1]), prod[:, :-1]], dim=1)
    sorted_alloc = sorted_nonusage * prod
    inv = torch.argsort(idx, dim=1)
    return torch.gather(sorted_alloc, 1, inv)


def write_allocation_weights(usage, write_gates, Wn):
    out = []
    for i in range(Wn):
        a = allocation(usage)
        out.append(a)
        usage = usage + (1 - usage) * write_gates[:, i:i + 1] * a
    return torch.stack(out, 1)


def access_step(cfg, p, h, prev):
    B = h.shape[0]
    N, W, R, Wn = cfg.N, cfg.W, cfg.R, cfg.Wn
    lin = lambda name: h @ p["memory_access/%s/w" % name] + p["memory_access/%s/b" % name]
    v = lin("write_vectors").reshape(B, Wn, W)
    e = torch.sigmoid(lin("erase_vectors")).reshape(B, Wn, W)
    fg = torch.sigmoid(lin("free_gate"))
    ag = torch.sigmoid(lin("allocation_gate"))
    wg = torch.sigmoid(lin("write_gate"))
    rm = torch.softmax(lin("read_mode").reshape(B, R, 1 + 2 * Wn), dim=2)
    kw = lin("write_keys").reshape(B, Wn, W)
    bw = lin("write_strengths")
    kr = lin("read_keys").reshape(B, R, W)
    br = lin("read_strengths")
    # Freeness (addressing.py:279-305), write weights under stop_gradient
    wwp = prev.write_weights.detach()
    usage = prev.usage + (1 - prev.usage) * (1 - torch.prod(1 - wwp, dim=1))
    usage = usage * torch.prod(1 - fg.unsqueeze(-1) * prev.read_weights, dim=1)
    cw = cosine_weights(prev.memory, kw, bw)
    aw = write_allocation_weights(usage, ag * wg, Wn)
    ww = wg.unsqueeze(-1) * (ag.unsqueeze(-1) * aw + (1 - ag.unsqueeze(-1)) * cw)
    reset = torch.prod(1 - ww.unsqueeze(3) * e.unsqueeze(2), dim=1)
    memory = prev.memory * reset + ww.transpose(1, 2) @ v
    link = link_update(prev.linkage.link, prev.linkage.precedence_weights, ww)
    prec = precedence(prev.linkage.precedence_weights, ww)
    cr = cosine_weights(memory, kr, br)
    fw = directional(link, prev.read_weights, True)
    bwd = directional(link, prev.read_weights, False)
    rw = (rm[:, :, 2 * Wn].unsqueeze(-1) * cr + (rm[:, :, Wn:2 * Wn].unsqueeze(-1) * fw).sum(2)
          + (rm[:, :, :Wn].unsqueeze(-1) * bwd).sum(2))
    reads = rw @ memory
    return reads, AccessState(memory, rw, ww, TemporalLinkageState(link, prec), usage)


def initial_state(cfg, B, dtype=torch.float64):
    a = cfg.access
    z = lambda *s: torch.zeros(s, dtype=dtype)
    return DNCState(z(B, a.R, a.W), AccessState(z(B, a.N, a.W), z(B, a.R, a.N), z(B, a.Wn, a.N),
                                                  TemporalLinkageState(z(B, a.Wn, a.N, a.N), z(B, a.Wn, a.N)), z(B, a.N)),
                    LSTMState(z(B, cfg.hid), z(B, cfg.hid)))


def clip(cfg, x):
    return torch.clamp(x, -cfg.clip, cfg.clip) if cfg.clip > 0 else x


def dnc_step(cfg, p, x, prev):
    B = x.shape[0]
    hid = cfg.hid
    ci = torch.cat([x.reshape(B, -1), prev.access_output.reshape(B, -1)], 1)
    g = torch.cat([ci, prev.controller_state.hidden], 1) @ p["lstm/w_gates"] + p["lstm/b_gates"]
    i, j, f, o = g[:, :hid], g[:, hid:2 * hid], g[:, 2 * hid:3 * hid], g[:, 3 * hid:]
    c2 = torch.sigmoid(f + 1.0) * prev.controller_state.cell + torch.sigmoid(i) * torch.tanh(j)
    h2 = torch.tanh(c2) * torch.sigmoid(o)
    h = clip(cfg, h2)
    cs = LSTMState(clip(cfg, h2), clip(cfg, c2))
    reads, acc = access_step(cfg.access, p, h, prev.access_state)
    y = torch.cat([h, reads.reshape(B, -1)], 1) @ p["output_linear/w"] + p["output_linear/b"]
    return clip(cfg, y), DNCState(reads, acc, cs)


def run_model(cfg, p, inputs_tm, state=None):
    S, B, _ = inputs_tm.shape
    st = state or initial_state(cfg, B, inputs_tm.dtype)
    ys = []
    for t in range(S):
        y, st = dnc_step(cfg, p, inputs_tm[t], st)
        ys.append(y)
    return torch.stack(ys, 0), st

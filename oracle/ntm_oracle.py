"""NumPy restatement of the reference NTM tracking path.  TEST INFRASTRUCTURE ONLY.

Follows (file:line into the reference tree):
  ntm_cell.py:53-253     NTMCell.__call__          -> ntm_step
  ntm_cell.py:284-315    NTMCell.zero_state        -> zero_state
  ops.py:135-158         batched_smooth_cosine_similarity (quirk Q1)
  ops.py:180-242         batched_circular_convolution / circular_shift (Q2)
  ntm_tracker_new.py:13-64  LoopNTMTracker         -> loop_ntm_tracker
  direct_offset_output.py:392-399   extract_features
  direct_offset_output.py:439-500   input serialiser -> serialize_inputs
  direct_offset_output.py:581-606   output gather + tanh + l2 loss -> offset_loss
  direct_offset_output.py:611-626   clip_by_global_norm + RMSProp
  vgg.py:155-161 (+ :49-63)         conv1_1..conv4_3 stack -> vgg16_conv43
  receptive_field_sizes.py:135-143  conv43Points

Third-party arithmetic not present in the reference tree (TensorFlow 1.x,
version unpinned): BasicLSTMCell (gate order i,j,f,o; forget_bias added to
f before the sigmoid), tf.nn.l2_normalize (x * rsqrt(max(sum x^2, 1e-12))),
tf.nn.softplus / softmax / pow, conv2d SAME, max_pool 2x2/2 VALID,
clip_by_global_norm, RMSPropOptimizer (ms slot initialised to ONE, epsilon
inside the sqrt).  These follow the libraries' documented formulas:
**parity unpinned** for them (no reference test covers them).

All functions compute in ``dtype`` (float32 mirrors the TF graph; float64 is
used by tests to bound rounding).
"""
from __future__ import annotations

import numpy as np

# receptive_field_sizes.py:135-143 -- 64 (y, x) sample points on the 28x28
# conv4_3 map: y, x in {6, 8, ..., 20}, y-major.
CONV43_POINTS = [(y, x) for y in range(6, 22, 2) for x in range(6, 22, 2)]

# direct_offset_output.py:58-59
VGG_MEAN = np.array([123.68, 116.78, 103.94], dtype=np.float32)

# vgg.py:155-160 -- (name, Cin, Cout, pool_after)
VGG_LAYERS = [
    ("conv1_1", 3, 64, False), ("conv1_2", 64, 64, True),
    ("conv2_1", 64, 128, False), ("conv2_2", 128, 128, True),
    ("conv3_1", 128, 256, False), ("conv3_2", 256, 256, False), ("conv3_3", 256, 256, True),
    ("conv4_1", 256, 512, False), ("conv4_2", 512, 512, False), ("conv4_3", 512, 512, False),
]


# --------------------------------------------------------------------------
# elementwise helpers (TF semantics)
# --------------------------------------------------------------------------
def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def softplus(x):
    # tf.nn.softplus = log(exp(x) + 1)
    return np.logaddexp(np.zeros((), dtype=x.dtype), x).astype(x.dtype)


def softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return (e / np.sum(e, axis=axis, keepdims=True)).astype(x.dtype)


def l2_normalize(x, axis, eps=1e-12):
    # tf.nn.l2_normalize: x * rsqrt(maximum(reduce_sum(x^2, axis), eps))
    ss = np.sum(x * x, axis=axis, keepdims=True)
    return (x / np.sqrt(np.maximum(ss, x.dtype.type(eps)))).astype(x.dtype)


# --------------------------------------------------------------------------
# ops.py
# --------------------------------------------------------------------------
def batched_smooth_cosine_similarity(memory, keys, mode="as_coded"):
    """ops.py:135-158.  memory [B,N,M], keys [B,H,M] -> [B,H,N].

    mode="as_coded" (default, quirk Q1): the code transposes memory to
    [B,M,N] and l2-normalises along axis 2, i.e. each *feature column* is
    normalised over the N slots (ops.py:147-150); keys are normalised per
    head over M (:152); similarity = keys_hat @ memory_hat (:156).

    mode="smooth_cosine": what ops_test.py:20-34 expects (Torch7
    nn.SmoothCosineSimilarity): dot / (|m| |k| + 1e-3) per memory row.
    """
    if mode == "as_coded":
        mt = np.transpose(memory, (0, 2, 1))          # [B,M,N]
        mt = l2_normalize(mt, 2)
        kh = l2_normalize(keys, 2)
        return np.matmul(kh, mt).astype(memory.dtype)
    if mode == "smooth_cosine":
        dot = np.matmul(keys, np.transpose(memory, (0, 2, 1)))
        mn = np.sqrt(np.sum(memory * memory, axis=2))[:, None, :]
        kn = np.sqrt(np.sum(keys * keys, axis=2))[:, :, None]
        return (dot / (mn * kn + memory.dtype.type(1e-3))).astype(memory.dtype)
    raise ValueError(mode)


def circular_shift(x, shift):
    """ops.py:216-242: result[..., i] = x[..., (i + shift) mod N]."""
    n = x.shape[-1]
    sp = n + shift if shift < 0 else shift
    assert 0 <= sp < n
    return np.concatenate([x[..., sp:], x[..., :sp]], axis=-1)


def shift_offsets(shift_space):
    """ops.py:203-209 under Python-2 integer division (quirk Q2):
    start = -shift_space/2 floors, so 3 -> -2 and the taps are (-2,-1,0)."""
    start = (-shift_space) // 2
    return list(range(start, shift_space + start))


def batched_circular_convolution(w, kernel):
    """ops.py:180-214.  w [B,H,N], kernel [B,H,2r+1] -> [B,H,N].

    out[i] = sum_j kernel[j] * w[(i + off_j) mod N], off = shift_offsets().
    (Q3: the reference tf.squeeze()s every unit dim of the result; later
    broadcasting restores them, so the values are those returned here.)
    """
    offs = shift_offsets(kernel.shape[-1])
    out = np.zeros_like(w)
    for j, s in enumerate(offs):
        out = out + circular_shift(w, s) * kernel[..., j:j + 1]
    return out.astype(w.dtype)


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
class NTMConfig(object):
    """Constructor arguments of NTMCell (ntm_cell.py:18-20) + input width."""

    def __init__(self, input_dim, output_dim, mem_size=128, mem_dim=20, shift_range=1,
                 controller_hidden_size=100, controller_num_layers=10,
                 write_head_size=3, read_head_size=3, write_first=False,
                 similarity="as_coded"):
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.mem_size = mem_size
        self.mem_dim = mem_dim
        self.shift_range = shift_range
        self.hidden = controller_hidden_size
        self.layers = controller_num_layers
        self.write_heads = write_head_size
        self.read_heads = read_head_size
        self.write_first = write_first
        self.similarity = similarity

    @property
    def heads(self):
        return self.read_heads + self.write_heads

    @property
    def shift_space(self):
        return 2 * self.shift_range + 1

    @property
    def control_sizes(self):
        """ntm_cell.py:113-121 split order: k, beta, g, sw, gamma, erase, add."""
        H, M, Wh = self.heads, self.mem_dim, self.write_heads
        return [M * H, H, H, self.shift_space * H, H, M * Wh, M * Wh]

    @property
    def control_dim(self):
        return sum(self.control_sizes)


def init_params(cfg, rng, scale=0.05, dtype=np.float32):
    """U(-scale, scale) like tf.random_uniform_initializer
    (direct_offset_output.py:528); linear biases start at 0
    (ntm_cell.py:366-369); BasicLSTMCell bias 0."""
    def u(*shape):
        return rng.uniform(-scale, scale, size=shape).astype(dtype)

    p = {
        "init_state/M": u(cfg.mem_size, cfg.mem_dim),
        "init_state/w": u(cfg.heads, cfg.mem_size),
        "init_state/read": u(cfg.read_heads, cfg.mem_dim),
        "addressing/weights": u(cfg.hidden, cfg.control_dim),
        "addressing/biases": np.zeros(cfg.control_dim, dtype),
        "output/weights": u(cfg.hidden, cfg.output_dim),
        "output/biases": np.zeros(cfg.output_dim, dtype),
    }
    in_dim = cfg.input_dim + cfg.read_heads * cfg.mem_dim
    for l in range(cfg.layers):
        p["lstm/cell_%d/weights" % l] = u(in_dim + cfg.hidden, 4 * cfg.hidden)
        p["lstm/cell_%d/biases" % l] = np.zeros(4 * cfg.hidden, dtype)
        in_dim = cfg.hidden
    return p


def zero_state(cfg, params, batch):
    """ntm_cell.py:284-315 (quirk Q5: trainable, w0 = sigmoid un-normalised)."""
    dt = params["init_state/M"].dtype
    M = np.tanh(params["init_state/M"])
    w = sigmoid(params["init_state/w"])
    r = np.tanh(params["init_state/read"])
    tile = lambda a: np.ascontiguousarray(np.broadcast_to(a, (batch,) + a.shape))
    return {
        "M": tile(M), "w": tile(w), "read": tile(r),
        "controller_state": np.zeros((batch, 2 * cfg.hidden * cfg.layers), dt),
    }


def basic_lstm_cell(x, state, W, b, forget_bias=0.0):
    """TF1 BasicLSTMCell, state_is_tuple=False: state = [c, h];
    [i, j, f, o] = split([x, h] @ W + b); c' = c*sig(f + fb) + sig(i)*tanh(j);
    h' = tanh(c') * sig(o).  (call site ntm_cell.py:45-50, forget_bias=0.0)"""
    hid = W.shape[1] // 4
    c, h = state[:, :hid], state[:, hid:]
    g = (np.concatenate([x, h], axis=1) @ W + b).astype(x.dtype)
    i, j, f, o = g[:, :hid], g[:, hid:2 * hid], g[:, 2 * hid:3 * hid], g[:, 3 * hid:]
    c2 = c * sigmoid(f + x.dtype.type(forget_bias)) + sigmoid(i) * np.tanh(j)
    h2 = np.tanh(c2) * sigmoid(o)
    return h2.astype(x.dtype), np.concatenate([c2, h2], axis=1).astype(x.dtype)


def ntm_step(cfg, params, x, state):
    """One NTMCell step (ntm_cell.py:53-253).

    x [B,D]; state dict(M [B,N,M], w [B,H,N], read [B,R,M],
    controller_state [B, 2*hid*L]).  Returns (output, logit, new_state, debug).
    """
    dt = x.dtype
    B = x.shape[0]
    H, R, Wh, Md, N = cfg.heads, cfg.read_heads, cfg.write_heads, cfg.mem_dim, cfg.mem_size
    M_prev, w_prev, read_prev = state["M"], state["w"], state["read"]
    cs = state["controller_state"]

    # :101-105 controller (MultiRNNCell over BasicLSTMCell, forget_bias 0, Q7)
    inp = np.concatenate([x, read_prev.reshape(B, R * Md)], axis=1)
    new_cs = []
    hid = cfg.hidden
    for l in range(cfg.layers):
        st = cs[:, 2 * hid * l:2 * hid * (l + 1)]
        inp, st2 = basic_lstm_cell(inp, st, params["lstm/cell_%d/weights" % l],
                                   params["lstm/cell_%d/biases" % l], 0.0)
        new_cs.append(st2)
    h = inp
    new_cs = np.concatenate(new_cs, axis=1)

    # :124-130 unpack
    u = (h @ params["addressing/weights"] + params["addressing/biases"]).astype(dt)
    sizes = cfg.control_sizes
    offs = np.cumsum([0] + sizes)
    k, beta, g, sw, gamma, erase, add = [u[:, offs[i]:offs[i + 1]] for i in range(7)]

    k = np.tanh(k.reshape(B, H, Md))                                   # :133
    sim = batched_smooth_cosine_similarity(M_prev, k, cfg.similarity)  # :136
    beta = softplus(beta)[:, :, None]                                  # :140
    wc = softmax(sim * beta, axis=2)                                   # :142
    g = sigmoid(g)[:, :, None]                                         # :151
    wg = (wc * g + w_prev * (dt.type(1.0) - g)).astype(dt)             # :153-156
    sw = softmax(sw.reshape(B, H, cfg.shift_space), axis=2)            # :161
    wv = batched_circular_convolution(wg, sw)                          # :165
    gamma = (softplus(gamma) + dt.type(1.0))[:, :, None]               # :169-170
    pw = np.power(wv, gamma).astype(dt)                                # :173
    w = (pw / (np.sum(pw, axis=2, keepdims=True) + dt.type(1e-3))).astype(dt)  # :175 (Q4)

    w_read, w_write = w[:, :R], w[:, R:]                               # :181-184
    erase = sigmoid(erase.reshape(B, Wh, Md))                          # :193
    add = np.tanh(add.reshape(B, Wh, Md))                              # :195
    # :202-210
    M_erase = np.prod(dt.type(1.0) - w_write[:, :, :, None] * erase[:, :, None, :], axis=1)
    M_write = np.sum(w_write[:, :, :, None] * add[:, :, None, :], axis=1)
    M = (M_prev * M_erase + M_write).astype(dt)
    read = np.matmul(w_read, M if cfg.write_first else M_prev).astype(dt)  # :212-215 (Q6)

    logit = (h @ params["output/weights"] + params["output/biases"]).astype(dt)  # :220
    out = softmax(logit, axis=1)                                       # :221
    new_state = {"M": M, "w": w, "read": read, "controller_state": new_cs}
    debug = {"k": k, "beta": beta, "g": g, "sw": sw, "gamma": gamma, "erase": erase,
             "add": add, "similarity": sim, "w_content_focused": wc, "w_gated": wg,
             "w_conv": wv, "w_conv_powed": pw, "w": w, "M_write": M_write.astype(dt), "M_erase": M_erase.astype(dt), "h": h, "u": u}
    return out, logit, new_state, debug


def loop_ntm_tracker(cfg, params, inputs, state=None, return_states=False):
    """ntm_tracker_new.py:13-64.  inputs [B,S,D] -> (outputs, logits) [B,S,O]."""
    B, S, _ = inputs.shape
    state = state or zero_state(cfg, params, B)
    outs, logits, states = [], [], []
    for t in range(S):
        o, l, state, _ = ntm_step(cfg, params, inputs[:, t], state)
        outs.append(o)
        logits.append(l)
        if return_states:
            states.append(state)
    res = (np.stack(outs, 1), np.stack(logits, 1))
    if return_states:
        return res + (state, states)
    return res + (state,)


# --------------------------------------------------------------------------
# tracking head (direct_offset_output.py)
# --------------------------------------------------------------------------
def extract_features(fmap, points=CONV43_POINTS):
    """direct_offset_output.py:392-399.  [F,28,28,C] -> [F,64,C]."""
    return np.stack([fmap[:, y, x, :] for (y, x) in points], axis=1)


def serialize_inputs(features, gts):
    """direct_offset_output.py:439-500.

    features [B,T,F,C] (F=64 sampled points), gts [B,T,F] heat-maps.
    Returns [B, T*(F+1), C+2]: per frame F rows [feat, 0, tgt] then one
    delimiter row [0..0, 1, 0]; tgt = gts[:,0,:] on the first F steps only.
    """
    B, T, F, C = features.shape
    dt = features.dtype
    padded = np.concatenate([features, np.zeros((B, T, F, 1), dt)], axis=3)      # :463
    delim = np.zeros((B, T, 1, C + 1), dt)
    delim[..., C] = 1.0                                                          # :469-477
    padded = np.concatenate([padded, delim], axis=2)                             # :480
    padded = padded.reshape(B, T * (F + 1), C + 1)                               # :483
    target = np.concatenate([gts[:, 0, :].astype(dt),
                             np.zeros((B, (T - 1) * (F + 1) + 1), dt)], axis=1)  # :492-496
    return np.concatenate([padded, target[:, :, None]], axis=2)                  # :498


def serialize_sequential(features, gts):
    """main.py:1701-1775 (ntm_sevenbyseven; the same layout in :979-1291).  features [B,T,F,C], gts [B,T,F].
    Returns [B, F + (T-1)(2F+1), C+3]: columns [feat, feature delimiter, frame delimiter, target]."""
    B, T, F, C = features.shape
    dt = features.dtype
    padded = np.concatenate([features, np.zeros((B, T, F, 2), dt)], axis=3)                     # :1718-1719
    rest = padded[:, 1:]                                                                        # :1721
    frame_delim = np.zeros((B, T - 1, 1, C + 2), dt); frame_delim[..., C + 1] = 1.0             # :1725-1734
    feat_delim = np.zeros((B, T - 1, F, C + 2), dt); feat_delim[..., C] = 1.0                   # :1735-1744
    rest = np.concatenate([rest, feat_delim], axis=3).reshape(B, T - 1, 2 * F, C + 2)           # :1746-1750
    rest = np.concatenate([frame_delim, rest], axis=2).reshape(B, (T - 1) * (2 * F + 1), C + 2)  # :1752-1760
    x = np.concatenate([padded[:, 0], rest], axis=1)                                            # :1764-1767
    target = np.concatenate([gts[:, 0, :].astype(dt), np.zeros((B, (T - 1) * (2 * F + 1)), dt)], axis=1)   # :1768-1772
    return np.concatenate([x, target[:, :, None]], axis=2)                                      # :1774-1776


def heatmap_gather(logits, T, F):
    """main.py:1880-1897: logits [B,S,1] -> scores [B,T-1,F] (the feature-delimiter step of every feature)."""
    B = logits.shape[0]
    g = logits.reshape(B, -1)[:, F:]
    g = g.reshape(B, T - 1, 2 * F + 1)[:, :, 1:]
    return g.reshape(B, T - 1, F, 2)[:, :, :, 1]


def heatmap_ce_loss(logits, gt, T):
    """main.py:1919-1923: sum of softmax_cross_entropy_with_logits(scores, gt) / (T-1).  Returns (loss, softmax)."""
    F = gt.shape[2]
    z = heatmap_gather(logits, T, F).astype(np.float64)
    z = z - z.max(axis=2, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=2, keepdims=True))
    return float(-(gt * (z - lse)).sum() / (T - 1)), np.exp(z - lse)


def offset_loss(logits, offsets, num_features=64):
    """direct_offset_output.py:581-606.  logits [B,S,2], offsets [B,T,2].
    Returns (loss, pred [B,T-1,2])."""
    B, S, O = logits.shape
    F1 = num_features + 1
    T = S // F1
    g = logits[:, F1:, :].reshape(B, T - 1, F1, O)[:, :, num_features, :]
    pred = np.tanh(g)
    d = pred - offsets[:, 1:, :]
    return logits.dtype.type(0.5) * np.sum(d * d, dtype=logits.dtype), pred


def discrete_gauss(center=(.5, .5), shape=(8, 8), sigma=1.0):
    """preprocess.py:205-221 (heat-map used as the frame-0 target)."""
    cx, cy = [a * b for a, b in zip(center, shape)]
    w, h = shape
    y, x = np.ogrid[-cy + .5:h - cy + .5, -cx + .5:w - cx + .5]
    hm = np.exp(-(x * x + y * y) / (2. * sigma * sigma))
    hm[hm < np.finfo(hm.dtype).eps * hm.max()] = 0
    s = hm.sum()
    if s != 0:
        hm /= s
    return hm


# --------------------------------------------------------------------------
# VGG-16 conv1_1 .. conv4_3 (vgg.py:155-161), NHWC, HWIO weights
# --------------------------------------------------------------------------
def conv3x3_same_relu(x, w, b):
    """x [F,H,W,Cin], w [3,3,Cin,Cout] (TF HWIO), b [Cout]; SAME, stride 1, ReLU."""
    F, H, W, Cin = x.shape
    xp = np.zeros((F, H + 2, W + 2, Cin), x.dtype)
    xp[:, 1:-1, 1:-1, :] = x
    out = np.zeros((F, H, W, w.shape[3]), x.dtype)
    for ky in range(3):
        for kx in range(3):
            out += xp[:, ky:ky + H, kx:kx + W, :] @ w[ky, kx]
    out += b
    return np.maximum(out, 0).astype(x.dtype)


def maxpool2x2(x):
    F, H, W, C = x.shape
    return x.reshape(F, H // 2, 2, W // 2, 2, C).max(axis=(2, 4))


def vgg16_conv43(frames, weights, upto="conv4_3"):
    """frames [F,H,W,3] mean-subtracted; weights {name: (w, b)}.  -> conv4_3 ReLU."""
    x = frames
    for name, _cin, _cout, pool in VGG_LAYERS:
        w, b = weights[name]
        x = conv3x3_same_relu(x, w, b)
        if name == upto:
            return x
        if pool:
            x = maxpool2x2(x)
    return x


def init_vgg_weights(rng, dtype=np.float32):
    """Seeded He-normal N(0, 2/(9 Cin)), zero bias (the real
    checkpoint is a download and not available; conv arithmetic is
    weight-agnostic)."""
    ws = {}
    for name, cin, cout, _ in VGG_LAYERS:
        std = np.sqrt(2.0 / (9 * cin))
        ws[name] = ((rng.standard_normal((3, 3, cin, cout)) * std).astype(dtype),
                    np.zeros(cout, dtype))
    return ws


# --------------------------------------------------------------------------
# optimiser (direct_offset_output.py:620-626)
# --------------------------------------------------------------------------
def clip_by_global_norm(grads, clip):
    """tf.clip_by_global_norm: g * clip / max(global_norm, clip)."""
    dt = grads[0].dtype
    gn = np.sqrt(sum(np.sum(g.astype(dt) ** 2, dtype=dt) for g in grads))
    s = dt.type(clip) / np.maximum(gn, dt.type(clip))
    return [g * s for g in grads], gn


def rmsprop_step(param, grad, ms, mom, lr=1e-4, decay=0.95, momentum=0.9, eps=1e-10):
    """tf.train.RMSPropOptimizer (non-centered) dense update:
    ms <- decay*ms + (1-decay)*g^2   (ms slot initialised to ONES)
    mom <- momentum*mom + lr*g/sqrt(ms + eps);  param <- param - mom."""
    dt = param.dtype
    ms = dt.type(decay) * ms + dt.type(1 - decay) * grad * grad
    mom = dt.type(momentum) * mom + dt.type(lr) * grad / np.sqrt(ms + dt.type(eps))
    return (param - mom).astype(dt), ms.astype(dt), mom.astype(dt)


# --------------------------------------------------------------------------
# bf16-operand variant of the trunk (BASELINE config 5: bf16 MFMA conv, fp32 accumulate)
# --------------------------------------------------------------------------
def bf16_round(x):
    """Round fp32 -> bf16 (round to nearest even) and return as fp32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def vgg16_conv43_bf16(frames, weights):
    """conv1_1 in fp32 on the fp32 frames (output rounded to bf16); every later layer multiplies bf16
    activations by bf16-rounded weights, accumulates wide, adds the fp32 bias, ReLU (+pool), and rounds the
    stored activation to bf16 -- except conv4_3, which stays fp32 for the memory cell."""
    x = frames.astype(np.float32)
    for name, _cin, _cout, pool in VGG_LAYERS:
        w, b = weights[name]
        if name == "conv1_1":
            y = conv3x3_same_relu(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
        else:
            y = conv3x3_same_relu(x.astype(np.float64), bf16_round(w).astype(np.float64), b.astype(np.float64))
        if name == "conv4_3":
            return y.astype(np.float32)
        if pool:
            y = maxpool2x2(y)
        x = bf16_round(y.astype(np.float32))
    return x


# ---- two-step presentation (main.py:862-977; ntm_tracker_new.py:112-195 with two_step=True)
def two_step_inputs(feat, target):
    """feat [B, T, D] (flattened, optionally compressed feature map of every frame), target [B, F] -> X [B, 2T-1, 1+D+F]:
    step 0 = [0, feat_0, target]; frame t >= 1 = [0, feat_t, 0] then the query step [1, 0, 0]
    (ntm_tracker_new.py:150-181: concat([switch, inputs, target]))."""
    B, T, D = feat.shape
    F = target.shape[1]
    X = np.zeros((B, 2 * T - 1, 1 + D + F), feat.dtype)
    X[:, 0, 1:1 + D] = feat[:, 0]
    X[:, 0, 1 + D:] = target
    for t in range(1, T):
        X[:, 2 * t - 1, 1:1 + D] = feat[:, t]
        X[:, 2 * t, 0] = 1
    return X


def two_step_labels(gt):
    """gt [B, T, F] -> labels [B, 2T-1, F+1] (main.py:903-934): the first frame and every presentation step carry the
    background row [0..0, 1]; the query step of frame t >= 1 carries [gt_t, 0]."""
    B, T, F = gt.shape
    lab = np.zeros((B, 2 * T - 1, F + 1), gt.dtype)
    lab[:, 0, F] = 1
    for t in range(1, T):
        lab[:, 2 * t - 1, F] = 1
        lab[:, 2 * t, :F] = gt[:, t]
    return lab


def two_step_ce_loss(logits, gt):
    """main.py:943-947: sum softmax_cross_entropy_with_logits(logits, SOFTMAX(labels)) / ((2T-1) B) -- the labels pass
    through tf.nn.softmax as coded.  Returns (loss, probs, dlogits)."""
    B, S, K = logits.shape
    lab = two_step_labels(gt)
    q = np.exp(lab - lab.max(-1, keepdims=True)); q /= q.sum(-1, keepdims=True)
    z = logits - logits.max(-1, keepdims=True)
    lp = z - np.log(np.exp(z).sum(-1, keepdims=True))
    loss = -(q * lp).sum() / (S * B)
    p = np.exp(lp)
    return loss, p, (p - q) / (S * B)


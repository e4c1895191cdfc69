"""torch-CPU autograd restatement of the NTM tracking path.  TEST INFRASTRUCTURE ONLY.

A second, independent restatement of the arithmetic in oracle/ntm_oracle.py
(same reference citations) written on torch tensors so that autograd supplies
the gradients tf.gradients would (direct_offset_output.py:611-621): full BPTT
through every step, trainable initial state summed over the batch
(ntm_cell.py:296-306).  Gradient conventions of the un-vendored TF ops it
stands in for (parity unpinned): pow's gradient w.r.t. the
exponent uses log(x) -> 0 for x <= 0; l2_normalize differentiates through
rsqrt(max(sum x^2, eps)).
"""
import numpy as np
import torch

from . import ntm_oracle as O


def _l2n(x, dim, eps=1e-12):
    ss = (x * x).sum(dim=dim, keepdim=True)
    return x * torch.rsqrt(torch.clamp(ss, min=eps))


class _PowTF(torch.autograd.Function):
    """tf.pow gradient: dx = g*y*x^(y-1); dy = g*x^y*log(x) with log(x)=0 where x<=0."""

    @staticmethod
    def forward(ctx, x, y):
        z = torch.pow(x, y)
        ctx.save_for_backward(x, y, z)
        return z

    @staticmethod
    def backward(ctx, g):
        x, y, z = ctx.saved_tensors
        dx = g * y * torch.pow(x, y - 1)
        logx = torch.where(x > 0, torch.log(torch.clamp(x, min=1e-300)), torch.zeros_like(x))
        dy = (g * z * logx)
        # reduce dy to y's broadcast shape
        while dy.dim() > y.dim():
            dy = dy.sum(0)
        for i, (a, b) in enumerate(zip(dy.shape, y.shape)):
            if b == 1 and a != 1:
                dy = dy.sum(i, keepdim=True)
        return dx, dy


def similarity(memory, keys, mode="as_coded"):
    if mode == "as_coded":            # ops.py:147-156 (Q1)
        mt = _l2n(memory.transpose(1, 2), 2)
        kh = _l2n(keys, 2)
        return kh @ mt
    dot = keys @ memory.transpose(1, 2)
    mn = memory.pow(2).sum(2).sqrt()[:, None, :]
    kn = keys.pow(2).sum(2).sqrt()[:, :, None]
    return dot / (mn * kn + 1e-3)


def circular_conv(w, kernel):
    offs = O.shift_offsets(kernel.shape[-1])
    out = torch.zeros_like(w)
    for j, s in enumerate(offs):
        out = out + torch.roll(w, shifts=-s, dims=-1) * kernel[..., j:j + 1]   # out[i] = w[(i+s) mod N]
    return out


def zero_state(cfg, p, B):
    tile = lambda a: a.unsqueeze(0).expand((B,) + tuple(a.shape))
    return {
        "M": tile(torch.tanh(p["init_state/M"])), "w": tile(torch.sigmoid(p["init_state/w"])),
        "read": tile(torch.tanh(p["init_state/read"])),
        "controller_state": torch.zeros((B, 2 * cfg.hidden * cfg.layers), dtype=p["init_state/M"].dtype),
    }


def ntm_step(cfg, p, x, st):
    B = x.shape[0]
    H, R, Wh, Md = cfg.heads, cfg.read_heads, cfg.write_heads, cfg.mem_dim
    hid = cfg.hidden
    inp = torch.cat([x, st["read"].reshape(B, R * Md)], 1)
    cs, new_cs = st["controller_state"], []
    for l in range(cfg.layers):
        c, h = cs[:, 2 * hid * l:2 * hid * l + hid], cs[:, 2 * hid * l + hid:2 * hid * (l + 1)]
        g = torch.cat([inp, h], 1) @ p["lstm/cell_%d/weights" % l] + p["lstm/cell_%d/biases" % l]
        i, j, f, o = g[:, :hid], g[:, hid:2 * hid], g[:, 2 * hid:3 * hid], g[:, 3 * hid:]
        c2 = c * torch.sigmoid(f) + torch.sigmoid(i) * torch.tanh(j)
        h2 = torch.tanh(c2) * torch.sigmoid(o)
        new_cs += [c2, h2]
        inp = h2
    h = inp
    u = h @ p["addressing/weights"] + p["addressing/biases"]
    k, beta, g, sw, gamma, erase, add = torch.split(u, cfg.control_sizes, dim=1)
    k = torch.tanh(k.reshape(B, H, Md))
    sim = similarity(st["M"], k, cfg.similarity)
    beta = torch.nn.functional.softplus(beta).unsqueeze(-1)
    wc = torch.softmax(sim * beta, dim=2)
    g = torch.sigmoid(g).unsqueeze(-1)
    wg = wc * g + st["w"] * (1 - g)
    sw = torch.softmax(sw.reshape(B, H, cfg.shift_space), dim=2)
    wv = circular_conv(wg, sw)
    gamma = (torch.nn.functional.softplus(gamma) + 1.0).unsqueeze(-1)
    pw = _PowTF.apply(wv, gamma)
    w = pw / (pw.sum(2, keepdim=True) + 1e-3)
    w_read, w_write = w[:, :R], w[:, R:]
    erase = torch.sigmoid(erase.reshape(B, Wh, Md))
    add = torch.tanh(add.reshape(B, Wh, Md))
    M_erase = torch.prod(1 - w_write.unsqueeze(3) * erase.unsqueeze(2), dim=1)
    M_write = torch.sum(w_write.unsqueeze(3) * add.unsqueeze(2), dim=1)
    M = st["M"] * M_erase + M_write
    read = w_read @ (M if cfg.write_first else st["M"])
    logit = h @ p["output/weights"] + p["output/biases"]
    return logit, {"M": M, "w": w, "read": read, "controller_state": torch.cat(new_cs, 1)}


def loop(cfg, p, inputs, state=None):
    B, S, _ = inputs.shape
    st = state or zero_state(cfg, p, B)
    logits = []
    for t in range(S):
        l, st = ntm_step(cfg, p, inputs[:, t], st)
        logits.append(l)
    return torch.stack(logits, 1), st


def offset_loss(logits, offsets, num_features=64):
    B, S, Od = logits.shape
    F1 = num_features + 1
    T = S // F1
    g = logits[:, F1:, :].reshape(B, T - 1, F1, Od)[:, :, num_features, :]
    pred = torch.tanh(g)
    return 0.5 * ((pred - offsets[:, 1:, :]) ** 2).sum(), pred


def loss_and_grads(cfg, params_np, inputs_np, offsets_np, num_features=64, dtype=torch.float64):
    """Returns (loss, {name: grad}) with grads in the TF variable layout."""
    p = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in params_np.items()}
    x = torch.tensor(inputs_np, dtype=dtype)
    off = torch.tensor(offsets_np, dtype=dtype)
    logits, _ = loop(cfg, p, x)
    loss, pred = offset_loss(logits, off, num_features)
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in p.items()}, logits.detach().numpy(), pred.detach().numpy()


def vgg16_conv43(frames, weights):
    """VGG conv1_1..conv4_3 on torch-CPU ops (conv2d SAME + bias + ReLU, 2x2/2 max-pool):
    the op granularity TF-CPU would execute for the frozen graph
    (direct_offset_output.py:417-422, vgg.py:155-161).  frames [F,H,W,3] NHWC,
    weights {name: (w HWIO, b)} numpy.  Returns [F,H/8,W/8,512] numpy."""
    x = torch.as_tensor(frames).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        for name, _cin, _cout, pool in O.VGG_LAYERS:
            w, b = weights[name]
            wt = torch.as_tensor(w).permute(3, 2, 0, 1).contiguous()      # HWIO -> OIHW
            x = torch.relu(torch.nn.functional.conv2d(x, wt, torch.as_tensor(b), padding=1))
            if name == "conv4_3":
                break
            if pool:
                x = torch.nn.functional.max_pool2d(x, 2, 2)
    return x.permute(0, 2, 3, 1).contiguous().numpy()


def vgg16_conv43_bf16(frames, weights):
    """The bf16 trunk of BASELINE config 5 restated on torch-CPU float64 convolutions, so that a 224x224 frame
    finishes in seconds (oracle/ntm_oracle.py:vgg16_conv43_bf16 is the numpy restatement of the same thing and
    tests/test_oracle_ntm.py holds the two together): conv1_1 multiplies the fp32 frames by fp32 weights; every
    later layer multiplies bf16-rounded activations by bf16-rounded weights (exact products in float64), adds the
    fp32 bias, ReLU (+ pool) and rounds the stored activation to bf16 -- except conv4_3, which stays fp32."""
    x = torch.as_tensor(np.asarray(frames, dtype=np.float32)).double().permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        for name, _cin, _cout, pool in O.VGG_LAYERS:
            w, b = weights[name]
            wq = np.asarray(w, np.float32) if name == "conv1_1" else O.bf16_round(np.asarray(w, np.float32))
            wt = torch.as_tensor(wq).double().permute(3, 2, 0, 1).contiguous()
            y = torch.relu(torch.nn.functional.conv2d(x, wt, torch.as_tensor(np.asarray(b, np.float32)).double(), padding=1))
            if name == "conv4_3":
                return y.permute(0, 2, 3, 1).contiguous().float().numpy()
            if pool:
                y = torch.nn.functional.max_pool2d(y, 2, 2)
            x = torch.as_tensor(O.bf16_round(y.float().numpy())).double()
    raise AssertionError("conv4_3 not reached")

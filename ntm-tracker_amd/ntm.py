"""NTMCell / LoopNTMTracker on the HIP sequence kernels.

Mirrors the reference operator interface:
  * ``NTMCell(output_dim, mem_size, mem_dim, shift_range, controller_hidden_size,
    controller_num_layers, write_head_size, read_head_size, write_first)``
    (ntm_cell.py:18-20), ``cell(inputs, prev_state, M_prev=..., w_prev=...,
    read_prev=..., controller_state=...)`` returning the reference's 8-tuple
    (ntm_cell.py:252-253), ``zero_state`` (:284-315), ``state_placeholder``
    (:255-282);
  * ``LoopNTMTracker(sequence_length, output_dim, initializer, **cell_kwargs)
    (inputs[B,S,D], state=None) -> (outputs, output_logits)``
    (ntm_tracker_new.py:5-49).
State lives in device tensors; every arithmetic step runs in
libntmtrack_hip.so (no torch math on the path, no CPU fallback).
"""
import ctypes

import torch

from . import _lib

_P = _lib.ptr


def _np(t):
    return None if t is None else _lib.ptr(t)


class NTMDims(object):
    def __init__(self, input_dim, output_dim, mem_size, mem_dim, shift_range, hidden, read_heads, write_heads):
        self.D = int(input_dim)
        self.O = int(output_dim)
        self.N, self.Md = int(mem_size), int(mem_dim)
        self.R, self.Wh = int(read_heads), int(write_heads)
        self.H = self.R + self.Wh
        self.hid = int(hidden)
        self.shift_range = int(shift_range)
        self.SS = 2 * self.shift_range + 1
        vals = [ctypes.c_int() for _ in range(5)]
        _lib.check(_lib.lib().ntk_ntm_padded_dims(self.N, self.Md, self.R, self.Wh, self.hid, self.shift_range,
                                                  self.O, *[ctypes.byref(v) for v in vals]), "ntk_ntm_padded_dims")
        self.P, self.PP, self.K, self.ldz, self.ldh = [v.value for v in vals]
        self.ldx = (self.D + 3) // 4 * 4
        self.RM = self.R * self.Md
        # control offsets (ntm_cell.py:128-130)
        self.oK = 0
        self.oB = self.H * self.Md
        self.oG = self.oB + self.H
        self.oS = self.oG + self.H
        self.oY = self.oS + self.H * self.SS
        self.oE = self.oY + self.H
        self.oA = self.oE + self.Wh * self.Md


class PackedParams(object):
    """Flat fp32 master copy of the trainable parameters in kernel layout
    (see csrc/ntm_common.h) with named views, plus a same-shaped gradient."""

    @staticmethod
    def shapes_for(d):
        return [
            ("WxT", (4 * d.hid, d.ldx)),
            ("Wr", (d.ldz, 4 * d.hid)),
            ("Wa", (d.ldh, d.PP)),
            ("V_M", (d.N, d.Md)),
            ("V_w", (d.H, d.N)),
            ("V_r", (d.R, d.Md)),
        ]

    def __init__(self, dims, device):
        d = dims
        self.dims = d
        self.shapes = self.shapes_for(d)
        n = 0
        self.offsets = {}
        for name, shp in self.shapes:
            sz = 1
            for s in shp:
                sz *= s
            n = (n + 3) // 4 * 4          # keep every view 16-byte aligned
            self.offsets[name] = (n, sz, shp)
            n += sz
        self.numel = (n + 3) // 4 * 4
        self.flat = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=device, dtype=torch.float32)

    def view(self, name, grad=False):
        o, sz, shp = self.offsets[name]
        return (self.grad if grad else self.flat)[o:o + sz].view(shp)

    # ---- conversion from / to the reference's TF variable layout
    def load_tf(self, sd):
        d = self.dims
        W = torch.as_tensor(sd["lstm/cell_0/weights"], dtype=torch.float32)
        b = torch.as_tensor(sd["lstm/cell_0/biases"], dtype=torch.float32)
        assert tuple(W.shape) == (d.D + d.RM + d.hid, 4 * d.hid), W.shape
        # gate-major columns g*hid+j  ->  unit-major n' = j*4+g
        perm = torch.arange(4 * d.hid).view(4, d.hid).t().reshape(-1)
        Wp = W[:, perm]
        bp = b[perm]
        WxT = torch.zeros((4 * d.hid, d.ldx))
        WxT[:, :d.D] = Wp[:d.D].t()
        Wr = torch.zeros((d.ldz, 4 * d.hid))
        Wr[:d.K] = Wp[d.D:]
        Wr[d.K] = bp
        Wa = torch.zeros((d.ldh, d.PP))
        Wa[:d.hid, :d.P] = torch.as_tensor(sd["addressing/weights"], dtype=torch.float32)
        Wa[:d.hid, d.P:d.P + d.O] = torch.as_tensor(sd["output/weights"], dtype=torch.float32)
        Wa[d.hid, :d.P] = torch.as_tensor(sd["addressing/biases"], dtype=torch.float32)
        Wa[d.hid, d.P:d.P + d.O] = torch.as_tensor(sd["output/biases"], dtype=torch.float32)
        dev = self.flat.device
        self.view("WxT").copy_(WxT.to(dev))
        self.view("Wr").copy_(Wr.to(dev))
        self.view("Wa").copy_(Wa.to(dev))
        self.view("V_M").copy_(torch.as_tensor(sd["init_state/M"], dtype=torch.float32).to(dev))
        self.view("V_w").copy_(torch.as_tensor(sd["init_state/w"], dtype=torch.float32).to(dev))
        self.view("V_r").copy_(torch.as_tensor(sd["init_state/read"], dtype=torch.float32).to(dev))

    def to_tf(self, grad=False):
        d = self.dims
        WxT = self.view("WxT", grad).cpu()
        Wr = self.view("Wr", grad).cpu()
        Wa = self.view("Wa", grad).cpu()
        inv = torch.arange(4 * d.hid).view(d.hid, 4).t().reshape(-1)   # gate-major col g*hid+j <- n' = j*4+g
        Wp = torch.cat([WxT[:, :d.D].t(), Wr[:d.K]], dim=0)
        return {
            "lstm/cell_0/weights": Wp[:, inv].contiguous(),
            "lstm/cell_0/biases": Wr[d.K][inv].contiguous(),
            "addressing/weights": Wa[:d.hid, :d.P].contiguous(),
            "addressing/biases": Wa[d.hid, :d.P].contiguous(),
            "output/weights": Wa[:d.hid, d.P:d.P + d.O].contiguous(),
            "output/biases": Wa[d.hid, d.P:d.P + d.O].contiguous(),
            "init_state/M": self.view("V_M", grad).cpu().clone(),
            "init_state/w": self.view("V_w", grad).cpu().clone(),
            "init_state/read": self.view("V_r", grad).cpu().clone(),
        }


def gemm_nt(A, B, bias=None, out=None):
    """out[M,N] = A[M,K] @ B[N,K]^T (+bias) on the fp32 MFMA kernel."""
    M, K = A.shape
    N, K2 = B.shape
    assert K == K2
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=torch.float32)
    _lib.check(_lib.lib().ntk_gemm_nt_f32(_P(A), A.stride(0), _P(B), B.stride(0), _np(bias), _P(out), out.stride(0),
                                          M, N, K, _lib.stream()), "ntk_gemm_nt_f32")
    return out


_WS = {}   # per-device split-K slab workspace, grown to the largest request


def gemm_tn(A, B, out, accumulate=False, splits=None, workspace=None):
    """out[M,N] (+)= A[K,M]^T @ B[K,N] on the fp32 MFMA kernel (fixed-order split-K)."""
    K, M = A.shape
    K2, N = B.shape
    assert K == K2 and tuple(out.shape) == (M, N)
    if splits is None:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        splits = max(1, min((K + 255) // 256, (1024 + tiles - 1) // tiles))
    L = _lib.lib()
    need = L.ntk_gemm_tn_workspace_bytes(M, N, splits) // 4
    if workspace is None or workspace.numel() < need:
        workspace = _WS.get(A.device)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(need, device=A.device, dtype=torch.float32)
            _WS[A.device] = workspace
    _lib.check(L.ntk_gemm_tn_f32(_P(A), A.stride(0), _P(B), B.stride(0), _P(out), out.stride(0), M, N, K, splits,
                                 1 if accumulate else 0, _P(workspace), _lib.stream()), "ntk_gemm_tn_f32")
    return out


class NTMCell(object):
    """The NTM recurrent cell (ntm_cell.py:17-315) on the HIP path.

    ``controller_num_layers == 1`` (what every reference script runs, direct_offset_output.py:24) is the fused
    persistent kernel with forward, BPTT and training.  A deeper MultiRNNCell controller (the constructor's default
    is 10) returns a ``StackedNTMCell``: forward / step() only, lower layers as separate LSTM steps."""

    def __new__(cls, output_dim=None, mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=100,
                controller_num_layers=10, *args, **kwargs):
        if cls is NTMCell and controller_num_layers != 1:
            return object.__new__(StackedNTMCell)
        return object.__new__(cls)

    def __init__(self, output_dim, mem_size=128, mem_dim=20, shift_range=1,
                 controller_hidden_size=100, controller_num_layers=10,
                 write_head_size=3, read_head_size=3, write_first=False,
                 input_dim=None, device="cuda", init_scale=0.1, seed=None):
        if controller_num_layers != 1:
            raise _lib.NtkError("controller_num_layers=%d reached the single-layer cell" % controller_num_layers)
        self.mem_size = mem_size
        self.mem_dim = mem_dim
        self.controller_hidden_size = controller_hidden_size
        self.controller_num_layers = controller_num_layers
        self.write_head_size = write_head_size
        self.read_head_size = read_head_size
        self.shift_range = shift_range
        self.output_dim = output_dim
        self.write_first = bool(write_first)
        self.device = torch.device(device)
        self.init_scale = init_scale
        self.seed = seed
        self.dims = None
        self.params = None
        if input_dim is not None:
            self._build(input_dim)

    # ---- parameters
    def _build(self, input_dim, tf_state_dict=None):
        self.dims = NTMDims(input_dim, self.output_dim, self.mem_size, self.mem_dim, self.shift_range,
                            self.controller_hidden_size, self.read_head_size, self.write_head_size)
        if self.params is not None and self.params.shapes == PackedParams.shapes_for(self.dims):
            # same layout: load INTO the existing flat buffer, so views, gradient buffer and any optimiser bound to
            # this PackedParams (tracker.RMSPropClip holds flat / grad / ms / mom by reference) stay valid
            self.params.dims = self.dims
        else:
            self.params = PackedParams(self.dims, self.device)
        if tf_state_dict is None:
            d = self.dims
            g = torch.Generator().manual_seed(0 if self.seed is None else int(self.seed))
            s = self.init_scale
            u = lambda *shape: (torch.rand(shape, generator=g) * 2 - 1) * s
            tf_state_dict = {
                "lstm/cell_0/weights": u(d.D + d.RM + d.hid, 4 * d.hid),
                "lstm/cell_0/biases": torch.zeros(4 * d.hid),
                "addressing/weights": u(d.hid, d.P), "addressing/biases": torch.zeros(d.P),
                "output/weights": u(d.hid, d.O), "output/biases": torch.zeros(d.O),
                "init_state/M": u(d.N, d.Md), "init_state/w": u(d.H, d.N), "init_state/read": u(d.R, d.Md),
            }
        self.params.load_tf(tf_state_dict)

    def load_state_dict(self, sd, input_dim=None):
        if input_dim is None:
            input_dim = sd["lstm/cell_0/weights"].shape[0] - self.read_head_size * self.mem_dim - self.controller_hidden_size
        self._build(int(input_dim), sd)

    def state_dict(self, grad=False):
        return self.params.to_tf(grad)

    #: leading dimension of the serialised input rows the cell consumes (features zero padded to a multiple of 4)
    input_ldx = property(lambda self: self.dims.ldx)

    # ---- state helpers
    def zero_state(self, batch_size, initializer=None):
        """ntm_cell.py:284-315: M0=tanh(V_M), w0=sigmoid(V_w) (un-normalised), read0=tanh(V_r), LSTM state 0."""
        d, dev, L = self.dims, self.device, _lib.lib()
        if d is None:
            raise _lib.NtkError("zero_state before the cell has parameters: pass input_dim= or load_state_dict()")
        st = self.state_placeholder(batch_size)
        for name, key, act in (("V_M", "M", 0), ("V_w", "w", 1), ("V_r", "read", 0)):
            v = self.params.view(name)
            _lib.check(L.ntk_ntm_init_state(_P(v), _P(st[key]), v.numel(), batch_size, act, _lib.stream()),
                       "ntk_ntm_init_state")
        st["controller_state"].zero_()
        return st

    def state_placeholder(self, batch_size):
        d, dev = self.dims, self.device
        return {
            "M": torch.empty((batch_size, d.N, d.Md), device=dev),
            "w": torch.empty((batch_size, d.H, d.N), device=dev),
            "read": torch.empty((batch_size, d.R, d.Md), device=dev),
            "controller_state": torch.empty((batch_size, 2 * d.hid), device=dev),
        }

    # ---- sequence kernel
    def _pad_inputs(self, inputs):
        d = self.dims
        B, S, D = inputs.shape
        if D == d.ldx and inputs.is_contiguous():
            return inputs                      # already in kernel layout (serialiser output, zero padded)
        if D != d.D:
            raise _lib.NtkError("inputs have %d features, the cell was built for %d" % (D, d.D))
        X = torch.zeros((B, S, d.ldx), device=self.device, dtype=torch.float32)
        X[:, :, :D] = inputs
        return X

    def run_sequence(self, X, state, record=False, want_outputs=True, after_projection=None):
        """X [B,S,ldx] (zero padded beyond D).  Returns (logits, outputs, new_state, record dict).
        after_projection: optional callable invoked once the input projection GEMM has been enqueued (the
        two-stream pipeline records an event there, see tracker._TwoStreamPipeline)."""
        d, dev = self.dims, self.device
        B, S, ldx = X.shape
        assert ldx == d.ldx
        xproj = gemm_nt(X.view(B * S, ldx), self.params.view("WxT"))
        if after_projection is not None:
            after_projection()
        logits = torch.empty((B, S, d.O), device=dev)
        outputs = torch.empty((B, S, d.O), device=dev) if want_outputs else None
        new = self.state_placeholder(B)
        rec = {}
        if record:
            rec = {
                "z": torch.empty((B, S, d.ldz), device=dev), "gates": torch.empty((B, S, d.hid, 4), device=dev),
                "c": torch.empty((B, S, d.hid), device=dev), "h": torch.empty((B, S, d.ldh), device=dev),
                "u": torch.empty((B, S, d.PP), device=dev), "wc": torch.empty((B, S, d.H, d.N), device=dev),
                "wv": torch.empty((B, S, d.H, d.N), device=dev), "w": torch.empty((B, S, d.H, d.N), device=dev),
                "M": torch.empty((B, S, d.N, d.Md), device=dev), "read": torch.empty((B, S, d.R, d.Md), device=dev),
            }
        g = lambda k: _np(rec.get(k))
        keep = []                                    # contiguous copies stay referenced until the launch is queued

        def cp(t):
            keep.append(t.contiguous())
            return _P(keep[-1])
        # a single step binds the step entry point of the C ABI (same kernel, S = 1)
        fn, name, lead = ((_lib.lib().ntk_ntm_step_fwd, "ntk_ntm_step_fwd", (B,)) if S == 1 else
                          (_lib.lib().ntk_ntm_seq_fwd, "ntk_ntm_seq_fwd", (B, S)))
        _lib.check(fn(
            *lead, d.N, d.Md, d.R, d.Wh, d.hid, d.shift_range, d.O, 1 if self.write_first else 0,
            _P(xproj), _P(self.params.view("Wr")), _P(self.params.view("Wa")),
            cp(state["M"]), cp(state["w"]), cp(state["read"]), cp(state["controller_state"]),
            _P(logits), _np(outputs), _P(new["M"]), _P(new["w"]), _P(new["read"]), _P(new["controller_state"]),
            g("z"), g("gates"), g("c"), g("h"), g("u"), g("wc"), g("wv"), g("w"), g("M"), g("read"),
            _lib.stream()), name)
        rec["xproj"] = xproj
        return logits, outputs, new, rec

    want_input_grad = False
    last_dX = None

    def backward_sequence(self, X, state0, rec, dlogits, dfinal=None, workspace=None):
        """BPTT through a recorded sequence: fills ``self.params.grad`` (kernel layout) with
        d loss / d params given ``dlogits`` [B,S,O] (and optionally gradients w.r.t. the final
        state).  Returns the gradient w.r.t. the initial state tensors."""
        d, dev, L = self.dims, self.device, _lib.lib()
        B, S, _ = X.shape
        P = self.params
        ldkT, ldhT = (d.K + 3) // 4 * 4, (d.hid + 3) // 4 * 4
        WrT = torch.empty((4 * d.hid, ldkT), device=dev)
        WaT = torch.empty((d.PP, ldhT), device=dev)
        st = _lib.stream()
        _lib.check(L.ntk_transpose_pad(_P(P.view("Wr")), 4 * d.hid, _P(WrT), ldkT, d.K, 4 * d.hid, st), "ntk_transpose_pad")
        _lib.check(L.ntk_transpose_pad(_P(P.view("Wa")), d.PP, _P(WaT), ldhT, d.hid, d.PP, st), "ntk_transpose_pad")
        dgates = torch.empty((B, S, 4 * d.hid), device=dev)
        du = torch.empty((B, S, d.PP), device=dev)
        g0 = self.state_placeholder(B)
        df = dfinal or {}
        keep = []

        def cp(t):
            keep.append(t.contiguous())
            return _P(keep[-1])
        _lib.check(L.ntk_ntm_seq_bwd(
            B, S, d.N, d.Md, d.R, d.Wh, d.hid, d.shift_range, d.O, 1 if self.write_first else 0,
            _P(WrT), ldkT, _P(WaT), ldhT,
            cp(state0["M"]), cp(state0["w"]), cp(state0["controller_state"]),
            _P(rec["gates"]), _P(rec["c"]), _P(rec["u"]), _P(rec["wc"]), _P(rec["wv"]), _P(rec["w"]), _P(rec["M"]),
            cp(dlogits),
            _np(df.get("M")), _np(df.get("w")), _np(df.get("read")), _np(df.get("controller_state")),
            _P(dgates), _P(du), _P(g0["M"]), _P(g0["w"]), _P(g0["read"]), _P(g0["controller_state"]), st),
            "ntk_ntm_seq_bwd")
        BS = B * S
        # weight gradients: three k-major contractions over all B*S recorded rows
        gemm_tn(dgates.view(BS, 4 * d.hid), X.view(BS, d.ldx), P.view("WxT", grad=True), workspace=workspace)
        gemm_tn(rec["z"].view(BS, d.ldz), dgates.view(BS, 4 * d.hid), P.view("Wr", grad=True), workspace=workspace)
        gemm_tn(rec["h"].view(BS, d.ldh), du.view(BS, d.PP), P.view("Wa", grad=True), workspace=workspace)
        #: gradient w.r.t. the (padded) input rows, for callers with a trainable layer in front of the cell (the input
        #: compressor of main.py's trackers): dX = dgates . Wx, one more GEMM -- only when asked for
        self.last_dX = None
        if self.want_input_grad:
            Wx = torch.empty((d.ldx, 4 * d.hid), device=dev)
            _lib.check(L.ntk_transpose_pad(_P(P.view("WxT")), d.ldx, _P(Wx), 4 * d.hid, 4 * d.hid, d.ldx, st), "ntk_transpose_pad")
            self.last_dX = gemm_nt(dgates.view(BS, 4 * d.hid), Wx).view(B, S, d.ldx)
        return g0

    def init_state_backward(self, g0, batch_size):
        """Gradient of the trainable initial state (ntm_cell.py:292-306): summed over the batch."""
        L = _lib.lib()
        for name, key, act in (("V_M", "M", 0), ("V_w", "w", 1), ("V_r", "read", 0)):
            v = self.params.view(name)
            _lib.check(L.ntk_ntm_init_state_bwd(_P(v), _P(g0[key]), _P(self.params.view(name, grad=True)),
                                                v.numel(), batch_size, act, 0, _lib.stream()), "ntk_ntm_init_state_bwd")

    # ---- the reference step() API
    def __call__(self, inputs, prev_state, M_prev=None, w_prev=None, read_prev=None,
                 controller_state=None, scope=None):
        """One step; returns (ntm_output, ntm_output_logit, state, debug, M, w, read, controller_state)
        exactly as ntm_cell.py:252-253."""
        if self.dims is None:
            self._build(inputs.shape[1])
        d = self.dims
        if prev_state is not None:
            M_prev, w_prev = prev_state["M"], prev_state["w"]
            read_prev, controller_state = prev_state["read"], prev_state["controller_state"]
        st = {"M": M_prev, "w": w_prev, "read": read_prev, "controller_state": controller_state}
        X = self._pad_inputs(inputs.unsqueeze(1))
        logits, outputs, new, rec = self.run_sequence(X, st, record=True)
        u = rec["u"][:, 0]
        B = inputs.shape[0]
        H, Md, R = d.H, d.Md, d.R
        w = rec["w"][:, 0]
        k = u[:, d.oK:d.oB].reshape(B, H, Md)
        # the six tensors the fused step keeps in registers, from HIP kernels on what the step recorded: the similarity is
        # ops.batched_smooth_cosine_similarity as coded (ntm_cell.py:136, quirk Q1), the other four one elementwise kernel
        from . import ops as _ops
        Mp, wp = M_prev.contiguous().float(), w_prev.contiguous().float()
        similarity = _ops.batched_smooth_cosine_similarity(Mp, k.contiguous(), device=self.device)
        wc, wv, wcur = rec["wc"][:, 0].contiguous(), rec["wv"][:, 0].contiguous(), w.contiguous()
        uc = u.contiguous()
        w_gated, powed = torch.empty_like(wcur), torch.empty_like(wcur)
        M_write, M_erase = torch.empty_like(Mp), torch.empty_like(Mp)
        sw = torch.empty((B, H, d.SS), device=self.device)
        _lib.check(_lib.lib().ntk_ntm_step_debug(_P(uc), uc.shape[1], d.oG, d.oS, d.SS, d.oY, d.oE, d.oA, _P(wc), _P(wv), _P(wcur), _P(wp),
                                                 _P(sw), _P(w_gated), _P(powed), _P(M_write), _P(M_erase), B, d.N, Md, R, d.Wh,
                                                 _lib.stream()), "ntk_ntm_step_debug")
        debug = {   # ntm_cell.py:230-250: all 19 tensors (the reference's key 'bega' is kept)
            "k": k, "bega": u[:, d.oB:d.oG].unsqueeze(-1),
            "g": u[:, d.oG:d.oS].unsqueeze(-1), "gamma": u[:, d.oY:d.oE].unsqueeze(-1),
            "erase": u[:, d.oE:d.oA].reshape(B, d.Wh, Md), "add": u[:, d.oA:d.P].reshape(B, d.Wh, Md),
            "sw": sw, "similarity": similarity,
            "w_content_focused": rec["wc"][:, 0], "w_gated": w_gated, "w_conv": rec["wv"][:, 0], "w_conv_powed": powed,
            "w": w, "w_read": w[:, :R], "w_write": w[:, R:], "M": new["M"], "M_prev": M_prev,
            "M_write": M_write, "M_erase": M_erase,
        }
        state = {"M": new["M"], "w": new["w"], "read": new["read"], "controller_state": new["controller_state"]}
        return (outputs[:, 0], logits[:, 0], state, debug, new["M"], new["w"], new["read"], new["controller_state"])

    step = __call__


class _StackedFlat(object):
    """One flat fp32 buffer (+ same-shaped gradient) for a deep controller: [packed top cell | lower layer matrices].
    The top cell's PackedParams is re-pointed at its slice, so the optimiser and the all-reduce stay flat passes."""

    def __init__(self, top_params, lower_shapes, device):
        n = top_params.numel
        self.lower_off = []
        for shp in lower_shapes:
            n = (n + 3) // 4 * 4
            self.lower_off.append((n, shp))
            n += shp[0] * shp[1]
        self.numel = (n + 3) // 4 * 4
        self.flat = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.flat[:top_params.numel].copy_(top_params.flat)
        top_params.flat = self.flat[:top_params.numel]
        top_params.grad = self.grad[:top_params.numel]
        self.top = top_params

    def lower(self, k, grad=False):
        o, shp = self.lower_off[k]
        return (self.grad if grad else self.flat)[o:o + shp[0] * shp[1]].view(shp)

    def view(self, name, grad=False):
        return self.top.view(name, grad)


class StackedNTMCell(NTMCell):
    """NTMCell with a MultiRNNCell controller of L > 1 BasicLSTMCell layers (ntm_cell.py:45-50, :101-105): layer 0 reads
    concat(x, read_prev), layer k reads h_{k-1}, the top layer's h drives the heads.  Layers 0 .. L-2 run as separate
    LSTM steps (ntk_gemm_nt_f32 + ntk_lstm_step_fwd/bwd); the top layer, the addressing and the memory update run in
    the fused cell kernel one step at a time (ntk_ntm_step_fwd/bwd; its read_prev rows of the recurrent matrix are
    zero here, because read_prev enters at layer 0).  Forward, BPTT (tf.gradients through every layer, direct_offset_
    output.py:611-621) and training are step-wise launches from Python: functional parity for the reference's deep
    constructor default, not a tuned path (every reference script runs one layer).
    controller_state layout = [c_0, h_0, c_1, h_1, ...] (state_is_tuple=False, MultiRNNCell concatenation).
    Lower layer k is stored as WT_k [4*hid][ld_k]: columns [input | h_prev | bias (the matching input column is 1)]."""

    def __init__(self, output_dim, mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=100,
                 controller_num_layers=10, write_head_size=3, read_head_size=3, write_first=False,
                 input_dim=None, device="cuda", init_scale=0.1, seed=None):
        self.L = int(controller_num_layers)
        self.output_dim, self.mem_size, self.mem_dim, self.shift_range = output_dim, mem_size, mem_dim, shift_range
        self.controller_hidden_size, self.controller_num_layers = controller_hidden_size, self.L
        self.write_head_size, self.read_head_size, self.write_first = write_head_size, read_head_size, bool(write_first)
        self.device, self.init_scale, self.seed = torch.device(device), init_scale, seed
        self.top = NTMCell(output_dim, mem_size, mem_dim, shift_range, controller_hidden_size, 1, write_head_size,
                           read_head_size, write_first, input_dim=None, device=device, init_scale=init_scale, seed=seed)
        self.D = None
        self.params = None
        self.lower = []
        if input_dim is not None:
            self._build(int(input_dim))

    dims = property(lambda self: self.top.dims)
    input_ldx = property(lambda self: (self.D + 3) // 4 * 4)

    def _lower_shapes(self):
        hid, RM = self.controller_hidden_size, self.read_head_size * self.mem_dim
        shapes, in_dim = [], self.D + RM
        for _k in range(self.L - 1):
            shapes.append((4 * hid, (in_dim + hid + 1 + 3) // 4 * 4))
            in_dim = hid
        return shapes

    def _build(self, input_dim, sd=None):
        hid, RM, L = self.controller_hidden_size, self.read_head_size * self.mem_dim, self.L
        self.D = int(input_dim)
        if sd is None:
            g = torch.Generator().manual_seed(0 if self.seed is None else int(self.seed))
            u = lambda *shape: (torch.rand(shape, generator=g) * 2 - 1) * self.init_scale
            tmp = NTMDims(hid, self.output_dim, self.mem_size, self.mem_dim, self.shift_range, hid, self.read_head_size,
                          self.write_head_size)
            sd = {"addressing/weights": u(hid, tmp.P), "addressing/biases": torch.zeros(tmp.P),
                  "output/weights": u(hid, tmp.O), "output/biases": torch.zeros(tmp.O),
                  "init_state/M": u(tmp.N, tmp.Md), "init_state/w": u(tmp.H, tmp.N), "init_state/read": u(tmp.R, tmp.Md)}
            in_dim = self.D + RM
            for k in range(L):
                sd["lstm/cell_%d/weights" % k] = u(in_dim + hid, 4 * hid)
                sd["lstm/cell_%d/biases" % k] = torch.zeros(4 * hid)
                in_dim = hid
        t = lambda v: torch.as_tensor(v, dtype=torch.float32)
        Wt, bt = t(sd["lstm/cell_%d/weights" % (L - 1)]), t(sd["lstm/cell_%d/biases" % (L - 1)])
        assert tuple(Wt.shape) == (2 * hid, 4 * hid), Wt.shape
        top_sd = {k: sd[k] for k in ("addressing/weights", "addressing/biases", "output/weights", "output/biases",
                                      "init_state/M", "init_state/w", "init_state/read")}
        top_sd["lstm/cell_0/weights"] = torch.cat([Wt[:hid], torch.zeros((RM, 4 * hid)), Wt[hid:]], dim=0)
        top_sd["lstm/cell_0/biases"] = bt
        keep = self.params is not None and self.top.params is not None and self.params.lower_off and \
            [shp for _o, shp in self.params.lower_off] == self._lower_shapes()
        self.top._build(hid, top_sd)           # same layout -> loads into the existing (shared) flat buffer
        if not keep:
            self.params = _StackedFlat(self.top.params, self._lower_shapes(), self.device)
        self.lower = []
        in_dim = self.D + RM
        for k in range(L - 1):
            W, b = t(sd["lstm/cell_%d/weights" % k]), t(sd["lstm/cell_%d/biases" % k])
            assert tuple(W.shape) == (in_dim + hid, 4 * hid), W.shape
            WT = self.params.lower(k)
            ld = WT.shape[1]
            host = torch.zeros((4 * hid, ld))
            host[:, :in_dim + hid] = W.t()
            host[:, in_dim + hid] = b
            WT.copy_(host.to(self.device))
            self.lower.append((WT, in_dim, ld))
            in_dim = hid

    def load_state_dict(self, sd, input_dim=None):
        if input_dim is None:
            input_dim = sd["lstm/cell_0/weights"].shape[0] - self.read_head_size * self.mem_dim - self.controller_hidden_size
        self._build(int(input_dim), sd)

    def state_dict(self, grad=False):
        """Variables under the reference's names (lstm/cell_k/{weights,biases}, ...); grad=True: their gradients."""
        hid, RM, L = self.controller_hidden_size, self.read_head_size * self.mem_dim, self.L
        top = self.top.params.to_tf(grad)
        out = {k: v for k, v in top.items() if not k.startswith("lstm/")}
        Wt = top["lstm/cell_0/weights"]
        out["lstm/cell_%d/weights" % (L - 1)] = torch.cat([Wt[:hid], Wt[hid + RM:]], dim=0).contiguous()
        out["lstm/cell_%d/biases" % (L - 1)] = top["lstm/cell_0/biases"]
        for k, (_WT, in_dim, _ld) in enumerate(self.lower):
            WT = self.params.lower(k, grad).cpu()
            out["lstm/cell_%d/weights" % k] = WT[:, :in_dim + hid].t().contiguous()
            out["lstm/cell_%d/biases" % k] = WT[:, in_dim + hid].contiguous()
        return out

    def state_placeholder(self, batch_size):
        st = self.top.state_placeholder(batch_size)
        st["controller_state"] = torch.empty((batch_size, 2 * self.controller_hidden_size * self.L), device=self.device)
        return st

    def zero_state(self, batch_size, initializer=None):
        st = self.top.zero_state(batch_size)
        st["controller_state"] = torch.zeros((batch_size, 2 * self.controller_hidden_size * self.L), device=self.device)
        return st

    # ---- one step (optionally recorded for BPTT)
    def _step(self, x, state, record):
        hid, L = self.controller_hidden_size, self.L
        B = x.shape[0]
        lib, stream = _lib.lib(), _lib.stream()
        cs = state["controller_state"].contiguous()
        inp = torch.cat([x.to(self.device, torch.float32)[:, :self.D], state["read"].reshape(B, -1)], dim=1)
        new_cs, lrec = [], []
        for k, (WT, in_dim, ld) in enumerate(self.lower):
            buf = torch.zeros((B, ld), device=self.device)
            buf[:, :in_dim] = inp
            buf[:, in_dim:in_dim + hid] = cs[:, 2 * hid * k + hid:2 * hid * (k + 1)]
            buf[:, in_dim + hid] = 1.0                                    # bias column
            pre = gemm_nt(buf, WT)
            c_prev = cs[:, 2 * hid * k:2 * hid * k + hid].contiguous()
            c, h = torch.empty((B, hid), device=self.device), torch.empty((B, hid), device=self.device)
            act = torch.empty((B, 4 * hid), device=self.device) if record else None
            _lib.check(lib.ntk_lstm_step_fwd(_P(pre), _P(c_prev), 0.0, _P(c), _P(h), _np(act), B, hid, stream), "ntk_lstm_step_fwd")
            new_cs += [c, h]
            if record:
                lrec.append({"buf": buf, "act": act, "c_prev": c_prev, "c": c})
            inp = h
        top_state = {"M": state["M"], "w": state["w"], "read": state["read"],
                     "controller_state": cs[:, 2 * hid * (L - 1):].contiguous()}
        Xt = self.top._pad_inputs(inp.unsqueeze(1))
        logits, outputs, new, rec = self.top.run_sequence(Xt, top_state, record=True)
        full_cs = torch.cat(new_cs + [new["controller_state"]], dim=1)
        new_state = {"M": new["M"], "w": new["w"], "read": new["read"], "controller_state": full_cs}
        srec = {"lower": lrec, "top": rec, "Xtop": Xt, "top_state": top_state} if record else None
        return outputs[:, 0], logits[:, 0], new_state, rec, srec

    def __call__(self, inputs, prev_state, M_prev=None, w_prev=None, read_prev=None, controller_state=None, scope=None):
        if self.D is None:
            self._build(inputs.shape[1])
        if prev_state is not None:
            M_prev, w_prev = prev_state["M"], prev_state["w"]
            read_prev, controller_state = prev_state["read"], prev_state["controller_state"]
        st = {"M": M_prev, "w": w_prev, "read": read_prev, "controller_state": controller_state}
        out, logit, new, rec, _ = self._step(inputs, st, False)
        d = self.dims
        B, H, Md, R = inputs.shape[0], d.H, d.Md, d.R
        u, w = rec["u"][:, 0], rec["w"][:, 0]
        debug = {"k": u[:, d.oK:d.oB].reshape(B, H, Md), "bega": u[:, d.oB:d.oG].unsqueeze(-1),
                 "g": u[:, d.oG:d.oS].unsqueeze(-1), "gamma": u[:, d.oY:d.oE].unsqueeze(-1),
                 "erase": u[:, d.oE:d.oA].reshape(B, d.Wh, Md), "add": u[:, d.oA:d.P].reshape(B, d.Wh, Md),
                 "w_content_focused": rec["wc"][:, 0], "w_conv": rec["wv"][:, 0], "w": w, "w_read": w[:, :R],
                 "w_write": w[:, R:], "M": new["M"], "M_prev": M_prev}
        return (out, logit, new, debug, new["M"], new["w"], new["read"], new["controller_state"])

    step = __call__

    def run_sequence(self, X, state, record=False, want_outputs=True, after_projection=None):
        """Python loop over steps (the LoopNTMTracker fallback for deep controllers).  record=True keeps what
        backward_sequence needs."""
        B, S, _ = X.shape
        logits, outs, steps, states = [], [], [], []
        for t in range(S):
            o, l, state, _rec, srec = self._step(X[:, t], state, record)
            outs.append(o); logits.append(l)
            if record:
                steps.append(srec)
                states.append(state)
        # "states": the full state after every step (controller_state = [c_0, h_0, c_1, h_1, ...] as MultiRNNCell packs it),
        # what the static-unroll trackers hand back (ntm_tracker_new.py:95-100)
        return torch.stack(logits, 1), (torch.stack(outs, 1) if want_outputs else None), state, \
            ({"steps": steps, "states": states} if record else {})

    def _pad_inputs(self, inputs):
        return inputs

    def backward_sequence(self, X, state0, rec, dlogits, dfinal=None, workspace=None):
        """BPTT through a recorded sequence of the deep controller: fills ``self.params.grad`` (top cell in kernel
        layout, lower layers as WT_k) and returns the gradient w.r.t. the initial state tensors."""
        steps = rec["steps"]
        S, B = len(steps), X.shape[0]
        hid, L, dev = self.controller_hidden_size, self.L, self.device
        d, lib, stream = self.dims, _lib.lib(), _lib.stream()
        TP = self.top.params
        ldkT, ldhT = (d.K + 3) // 4 * 4, (d.hid + 3) // 4 * 4
        WrT = torch.empty((4 * hid, ldkT), device=dev)
        WaT = torch.empty((d.PP, ldhT), device=dev)
        _lib.check(lib.ntk_transpose_pad(_P(TP.view("Wr")), 4 * hid, _P(WrT), ldkT, d.K, 4 * hid, stream), "ntk_transpose_pad")
        _lib.check(lib.ntk_transpose_pad(_P(TP.view("Wa")), d.PP, _P(WaT), ldhT, hid, d.PP, stream), "ntk_transpose_pad")
        Wx_top = TP.view("WxT").t().contiguous()                       # [ldx_top][4*hid]: d(top input) = dgates @ WxT
        W_low = [WT.t().contiguous() for WT, _i, _l in self.lower]      # [ld_k][4*hid]
        z = lambda *s_: torch.zeros(s_, device=dev)
        df = dfinal or {}
        dM = df.get("M", z(B, d.N, d.Md)).contiguous()
        dw = df.get("w", z(B, d.H, d.N)).contiguous()
        dread = df.get("read", z(B, d.R, d.Md)).contiguous()
        dcs_full = df.get("controller_state", z(B, 2 * hid * L))
        dcs_top = dcs_full[:, 2 * hid * (L - 1):].contiguous()
        dc_low = [dcs_full[:, 2 * hid * k:2 * hid * k + hid].contiguous() for k in range(L - 1)]
        dh_low = [dcs_full[:, 2 * hid * k + hid:2 * hid * (k + 1)].contiguous() for k in range(L - 1)]
        dgates_all = torch.empty((B, S, 4 * hid), device=dev)
        du_all = torch.empty((B, S, d.PP), device=dev)
        dpre_all = [torch.empty((B, S, 4 * hid), device=dev) for _ in range(L - 1)]
        dl = dlogits.contiguous()
        for t in range(S - 1, -1, -1):
            st, r, ts = steps[t], steps[t]["top"], steps[t]["top_state"]
            dgates, du = torch.empty((B, 4 * hid), device=dev), torch.empty((B, d.PP), device=dev)
            g0 = self.top.state_placeholder(B)
            dlt = dl[:, t].contiguous()
            _lib.check(lib.ntk_ntm_step_bwd(
                B, d.N, d.Md, d.R, d.Wh, d.hid, d.shift_range, d.O, 1 if self.write_first else 0,
                _P(WrT), ldkT, _P(WaT), ldhT, _P(ts["M"].contiguous()), _P(ts["w"].contiguous()), _P(ts["controller_state"]),
                _P(r["gates"]), _P(r["c"]), _P(r["u"]), _P(r["wc"]), _P(r["wv"]), _P(r["w"]), _P(r["M"]), _P(dlt),
                _P(dM), _P(dw), _P(dread), _P(dcs_top),
                _P(dgates), _P(du), _P(g0["M"]), _P(g0["w"]), _P(g0["read"]), _P(g0["controller_state"]), stream),
                "ntk_ntm_step_bwd")
            dgates_all[:, t], du_all[:, t] = dgates, du
            dM, dw, dcs_top = g0["M"], g0["w"], g0["controller_state"]
            dread_prev = g0["read"]                                        # zero-weight path of the fused cell (kept for exactness)
            dx = gemm_nt(dgates, Wx_top)[:, :hid]                          # gradient of the top layer's input h_{L-2}(t)
            for k in range(L - 2, -1, -1):
                WT, in_dim, ld = self.lower[k]
                lr = st["lower"][k]
                dh = (dx + dh_low[k]).contiguous()
                dpre, dc_prev = torch.empty((B, 4 * hid), device=dev), torch.empty((B, hid), device=dev)
                _lib.check(lib.ntk_lstm_step_bwd(_P(lr["act"]), _P(lr["c_prev"]), _P(lr["c"]), _P(dh), _P(dc_low[k]), _P(dpre), _P(dc_prev),
                                                 B, hid, stream), "ntk_lstm_step_bwd")
                dpre_all[k][:, t] = dpre
                dbuf = gemm_nt(dpre, W_low[k])                              # [B, ld_k]
                dc_low[k] = dc_prev
                dh_low[k] = dbuf[:, in_dim:in_dim + hid].contiguous()
                dx = dbuf[:, :in_dim]
            dread = (dread_prev.reshape(B, -1) + dx[:, self.D:]).reshape(B, d.R, d.Md).contiguous()   # layer 0 reads read_{t-1}
        BS = B * S
        # weight gradients: k-major contractions over all recorded rows (bias = the ones column of the recorded inputs)
        Xtop = torch.stack([st["Xtop"][:, 0] for st in steps], 1).contiguous()
        cat = lambda key: torch.stack([st["top"][key][:, 0] for st in steps], 1).contiguous()
        gemm_tn(dgates_all.view(BS, 4 * hid), Xtop.view(BS, d.ldx), TP.view("WxT", grad=True), workspace=workspace)
        gemm_tn(cat("z").view(BS, d.ldz), dgates_all.view(BS, 4 * hid), TP.view("Wr", grad=True), workspace=workspace)
        gemm_tn(cat("h").view(BS, d.ldh), du_all.view(BS, d.PP), TP.view("Wa", grad=True), workspace=workspace)
        # the read_prev rows of the fused cell's recurrent matrix are structural zeros here (read_prev enters at layer 0)
        TP.view("Wr", grad=True)[:d.RM].zero_()
        for k, (WT, in_dim, ld) in enumerate(self.lower):
            bufs = torch.stack([st["lower"][k]["buf"] for st in steps], 1).contiguous()
            gemm_tn(dpre_all[k].view(BS, 4 * hid), bufs.view(BS, ld), self.params.lower(k, grad=True), workspace=workspace)
        g0 = {"M": dM, "w": dw, "read": dread,
              "controller_state": torch.cat([t_ for k in range(L - 1) for t_ in (dc_low[k], dh_low[k])] + [dcs_top], dim=1)}
        return g0

    def init_state_backward(self, g0, batch_size):
        self.top.init_state_backward(g0, batch_size)


class LoopNTMTracker(object):
    """ntm_tracker_new.py:4-64: unroll the cell over [B,S,D] inputs.  The
    tf.while_loop becomes one persistent kernel launch."""

    def __init__(self, sequence_length, output_dim, initializer=None, **kwargs):
        self.cell = NTMCell(output_dim, **kwargs)
        self.initializer = initializer
        self.sequence_length = sequence_length

    def __call__(self, inputs, state=None, scope=None, record=False):
        B, S, D = inputs.shape
        if S != self.sequence_length:
            raise _lib.NtkError("inputs have %d steps, tracker was built for %d" % (S, self.sequence_length))
        if self.cell.dims is None:
            self.cell._build(D)
        X = self.cell._pad_inputs(inputs)
        state = state or self.cell.zero_state(B, self.initializer)
        logits, outputs, new, rec = self.cell.run_sequence(X, state, record=record)
        self.last_state, self.last_record = new, rec
        return outputs, logits


class PlainNTMTracker(object):
    """ntm_tracker_new.py:66-110: the cell statically unrolled ``model_length`` times over [B, model_length, D] inputs,
    nothing assumed about the inputs.  Same call signature; the unrolled graph becomes one persistent kernel launch.
    Returns (outputs, output_logits, states, debugs) as the reference: ``states`` = the initial state followed by the state
    after EVERY step (model_length + 1 dicts, :95-100; views of the tensors the launch recorded), ``debugs`` = the recorded
    per-step tensors [B,S,...] (u, wc, wv, w, M, read: the reference's per-step debug dicts side by side)."""

    def __init__(self, model_length, output_dim, initializer=None, **kwargs):
        self.model_length = model_length
        self.cell = NTMCell(output_dim, **kwargs)
        self.initializer = initializer

    def __call__(self, inputs, state=None, scope=None):
        B, S, D = inputs.shape
        if S != self.model_length:
            raise _lib.NtkError("inputs have %d steps, tracker was built for %d" % (S, self.model_length))
        if self.cell.dims is None:
            self.cell._build(D)
        X = self.cell._pad_inputs(inputs)
        state = state or self.cell.zero_state(B, self.initializer)
        logits, outputs, new, rec = self.cell.run_sequence(X, state, record=True)
        return outputs, logits, [state] + per_step_states(self.cell, rec, new), per_step_debugs(rec)


def per_step_debugs(rec):
    """The recorded per-step tensors [B,S,...] of a launch (u, wc, wv, w, M, read: the reference's per-step debug dicts side by
    side).  A deep controller's step-wise records (StackedNTMCell) are stacked into the same [B,S,...] form."""
    keys = ("u", "wc", "wv", "w", "M", "read")
    if "steps" in rec:
        return {k: torch.cat([st["top"][k] for st in rec["steps"]], dim=1) for k in keys if rec["steps"] and k in rec["steps"][0]["top"]}
    return {k: rec[k] for k in keys if k in rec}


def per_step_states(cell, rec, final):
    """The state after every step of a recorded launch, as the reference's state dicts (ntm_cell.py:223-228): M, w, read and
    controller_state = [c, h] (BasicLSTMCell, state_is_tuple=False; the recorded cell is the step's c, the recorded h its
    output).  Views / copies of the records only; the last entry is the launch's final state itself.  A deep controller
    (StackedNTMCell, controller_num_layers > 1) runs step-wise and records each step's full state, controller_state =
    [c_0, h_0, c_1, h_1, ...] as MultiRNNCell concatenates it (ntm_cell.py:45-50)."""
    if "states" in rec:
        return list(rec["states"][:-1]) + [final]
    hid = cell.dims.hid
    S = rec["M"].shape[1]
    out = []
    for t in range(S - 1):
        out.append({"M": rec["M"][:, t], "w": rec["w"][:, t], "read": rec["read"][:, t],
                    "controller_state": torch.cat([rec["c"][:, t], rec["h"][:, t, :hid]], dim=1)})
    out.append(final)
    return out


class NTMTracker(object):
    """ntm_tracker_new.py:112-195: one step per frame, the target indicator appended to the frame's features: step 0 sees
    [inputs_0, target], later steps [inputs_t, 0]; with ``two_step`` every later frame takes a presentation step
    [0, inputs_t, 0] and a query step [1, 0, 0] (2T - 1 steps, see ntmtrack.twostep).  Same constructor and call
    signature: ``tracker(inputs [B,T,D], target [B,F]) -> (outputs, output_logits, states, debugs)``."""

    def __init__(self, sequence_length, batch_size, output_dim, initializer=None, two_step=False, **kwargs):
        self.sequence_length, self.batch_size, self.output_dim = sequence_length, batch_size, output_dim
        self.cell = NTMCell(output_dim, **kwargs)
        self.initializer = initializer
        self.two_step = two_step

    def __call__(self, inputs, target, scope=None):
        B, T, D = inputs.shape
        F = target.shape[1]
        if T != self.sequence_length or B != self.batch_size:
            raise _lib.NtkError("inputs [%d,%d,..] do not match the tracker (batch %d, length %d)" % (B, T, self.batch_size, self.sequence_length))
        if self.two_step:
            from .twostep import serialize_two_step
            if self.cell.dims is None:
                self.cell._build(1 + D + F)
            X = serialize_two_step(inputs, target, self.cell.input_ldx)
        else:
            if self.cell.dims is None:
                self.cell._build(D + F)
            X = torch.zeros((B, T, self.cell.input_ldx), device=inputs.device)
            X[:, :, :D] = inputs
            X[:, 0, D:D + F] = target                                    # the indicator: target at the first frame, zeros after
        state = self.cell.zero_state(B, self.initializer)
        logits, outputs, new, rec = self.cell.run_sequence(X, state, record=True)
        return outputs, logits, [state] + per_step_states(self.cell, rec, new), per_step_debugs(rec)


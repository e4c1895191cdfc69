"""DNC core on the HIP sequence kernel.

Mirrors the reference operator interface (dnc/dnc.py:36-142, vendored DeepMind DNC):
  ``DNC(access_config={'memory_size','word_size','num_reads','num_writes'},
        controller_config={'hidden_size'}, output_size, clip_value=None)``,
  ``core(inputs, prev_state) -> (output, DNCState)``, ``initial_state(batch_size)``,
  ``state_size`` / ``output_size``; and ``run_model(input_sequence[S,B,D], output_size)``
  of direct_offset_output_with_dnc.py:66-88 (time-major dynamic_rnn).
State tuples are the reference's namedtuples holding device tensors.
"""
import collections
import ctypes

import torch

from . import _lib
from .ntm import gemm_nt, gemm_tn

_P = _lib.ptr

DNCState = collections.namedtuple("DNCState", ("access_output", "access_state", "controller_state"))
AccessState = collections.namedtuple("AccessState", ("memory", "read_weights", "write_weights", "linkage", "usage"))
TemporalLinkageState = collections.namedtuple("TemporalLinkageState", ("link", "precedence_weights"))
LSTMState = collections.namedtuple("LSTMState", ("hidden", "cell"))

# order of the ten interface linears inside the packed Wi matrix (csrc/dnc_seq_fwd.hip: DncDims)
INTERFACE = ("write_vectors", "erase_vectors", "free_gate", "allocation_gate", "write_gate", "read_mode",
             "write_keys", "write_strengths", "read_keys", "read_strengths")


class FlatParams(object):
    """Flat fp32 parameter buffer with named 16-byte-aligned views and a same-shaped gradient."""

    def __init__(self, shapes, device):
        n = 0
        self.shapes = list(shapes)
        self.offsets = {}
        for name, shp in shapes:
            sz = 1
            for s_ in shp:
                sz *= s_
            n = (n + 3) // 4 * 4
            self.offsets[name] = (n, sz, shp)
            n += sz
        self.numel = (n + 3) // 4 * 4
        self.flat = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=device, dtype=torch.float32)

    def view(self, name, grad=False):
        o, sz, shp = self.offsets[name]
        return (self.grad if grad else self.flat)[o:o + sz].view(shp)


class DNC(object):
    def __init__(self, access_config, controller_config, output_size, clip_value=None, input_dim=None,
                 device="cuda", seed=0):
        self.N = int(access_config.get("memory_size", 128))
        #: word_size as the caller sees it; the kernels work on words zero padded to a multiple of 4 floats (self.W):
        #: padded interface columns have zero weights (write / key entries 0; erase sigmoid(0) acts on zeros only), so
        #: the padded memory columns stay zero and every result is unchanged (the reference's own DNC test shape is
        #: word_size 6, dnc/access_test.py:28-34)
        self.word_size = int(access_config.get("word_size", 20))
        self.W = (self.word_size + 3) // 4 * 4
        self.R = int(access_config.get("num_reads", 1))
        self.Wn = int(access_config.get("num_writes", 1))
        self.hid = int(controller_config["hidden_size"])
        self.O = int(output_size)
        self.clip_value = float(clip_value or 0)
        self.device = torch.device(device)
        self.seed = seed
        vals = [ctypes.c_int() for _ in range(8)]
        _lib.check(_lib.lib().ntk_dnc_padded_dims(self.N, self.W, self.R, self.Wn, self.hid, self.O,
                                                  *[ctypes.byref(v) for v in vals]), "ntk_dnc_padded_dims")
        self.I, self.IP, self.K, self.ldz, self.ldh, self.Ky, self.ldy, self.OP = [v.value for v in vals]
        self.D = None
        if input_dim is not None:
            self._build(int(input_dim))

    # ---- parameters (Sonnet v1 names)
    def interface_widths(self, logical=False):
        N, W, R, Wn = self.N, (self.word_size if logical else self.W), self.R, self.Wn
        return dict(write_vectors=Wn * W, erase_vectors=Wn * W, free_gate=R, allocation_gate=Wn, write_gate=Wn,
                    read_mode=R * (1 + 2 * Wn), write_keys=Wn * W, write_strengths=Wn, read_keys=R * W, read_strengths=R)

    def _build(self, input_dim, sd=None):
        self.D = input_dim
        self.ldx = (input_dim + 3) // 4 * 4
        if sd is None:
            g = torch.Generator().manual_seed(int(self.seed))
            tn = lambda fan_in, *shape: torch.clamp(torch.randn(shape, generator=g), -2, 2) / fan_in ** 0.5
            in_dim = input_dim + self.R * self.word_size + self.hid
            sd = {"lstm/w_gates": tn(in_dim, in_dim, 4 * self.hid), "lstm/b_gates": torch.zeros(4 * self.hid)}
            for name, width in self.interface_widths(logical=True).items():
                sd["memory_access/%s/w" % name] = tn(self.hid, self.hid, width)
                sd["memory_access/%s/b" % name] = torch.zeros(width)
            ky = self.hid + self.R * self.word_size
            sd["output_linear/w"] = tn(ky, ky, self.O)
            sd["output_linear/b"] = torch.zeros(self.O)
        self.load_state_dict(sd, input_dim)

    # ---- word padding (word_size -> multiple of 4) of the reference-layout variables
    _WORD_FIELDS = {"write_vectors": "Wn", "erase_vectors": "Wn", "write_keys": "Wn", "read_keys": "R"}

    def _words_cols(self, M, groups, pad):
        """[..., groups * W_from] -> [..., groups * W_to] along the last axis (zero fill / strip)."""
        Wl, Wp = self.word_size, self.W
        if Wl == Wp:
            return M
        a, b = (Wl, Wp) if pad else (Wp, Wl)
        out = M.new_zeros(M.shape[:-1] + (groups * b,))
        out.view(M.shape[:-1] + (groups, b))[..., :min(a, b)] = M.reshape(M.shape[:-1] + (groups, a))[..., :min(a, b)]
        return out

    def _pad_sd(self, sd):
        t = lambda v: torch.as_tensor(v, dtype=torch.float32)
        if self.word_size == self.W:
            return {k: t(v) for k, v in sd.items()}
        hid, R = self.hid, self.R
        out = {k: t(v) for k, v in sd.items()}
        Wg = out["lstm/w_gates"]
        D = Wg.shape[0] - R * self.word_size - hid
        out["lstm/w_gates"] = torch.cat([Wg[:D], self._words_cols(Wg[D:D + R * self.word_size].t(), R, True).t(), Wg[D + R * self.word_size:]], 0)
        for name, grp in self._WORD_FIELDS.items():
            g = self.Wn if grp == "Wn" else R
            out["memory_access/%s/w" % name] = self._words_cols(out["memory_access/%s/w" % name], g, True)
            out["memory_access/%s/b" % name] = self._words_cols(out["memory_access/%s/b" % name], g, True)
        Wo = out["output_linear/w"]
        out["output_linear/w"] = torch.cat([Wo[:hid], self._words_cols(Wo[hid:].t(), R, True).t()], 0)
        return out

    def _strip_sd(self, sd):
        if self.word_size == self.W:
            return sd
        hid, R = self.hid, self.R
        out = dict(sd)
        Wg = out["lstm/w_gates"]
        D = Wg.shape[0] - R * self.W - hid
        out["lstm/w_gates"] = torch.cat([Wg[:D], self._words_cols(Wg[D:D + R * self.W].t().contiguous(), R, False).t(), Wg[D + R * self.W:]], 0).contiguous()
        for name, grp in self._WORD_FIELDS.items():
            g = self.Wn if grp == "Wn" else R
            out["memory_access/%s/w" % name] = self._words_cols(out["memory_access/%s/w" % name], g, False)
            out["memory_access/%s/b" % name] = self._words_cols(out["memory_access/%s/b" % name], g, False)
        Wo = out["output_linear/w"]
        out["output_linear/w"] = torch.cat([Wo[:hid], self._words_cols(Wo[hid:].t().contiguous(), R, False).t()], 0).contiguous()
        return out

    def _pad_state(self, st):
        if st is None or self.word_size == self.W:
            return st
        acc = st.access_state
        return DNCState(self._words_cols(st.access_output, 1, True),
                        AccessState(self._words_cols(acc.memory, 1, True), acc.read_weights, acc.write_weights, acc.linkage, acc.usage),
                        st.controller_state)

    def _strip_state(self, st):
        if self.word_size == self.W:
            return st
        acc = st.access_state
        return DNCState(self._words_cols(st.access_output, 1, False),
                        AccessState(self._words_cols(acc.memory, 1, False), acc.read_weights, acc.write_weights, acc.linkage, acc.usage),
                        st.controller_state)

    def load_state_dict(self, sd, input_dim=None):
        t = lambda v: torch.as_tensor(v, dtype=torch.float32)
        sd = self._pad_sd(sd)
        W = t(sd["lstm/w_gates"])
        if input_dim is None:
            input_dim = W.shape[0] - self.R * self.W - self.hid
        self.D, self.ldx = int(input_dim), (int(input_dim) + 3) // 4 * 4
        hid, dev = self.hid, self.device
        assert tuple(W.shape) == (self.D + self.K, 4 * hid), W.shape
        perm = torch.arange(4 * hid).view(4, hid).t().reshape(-1)          # gate-major -> unit-major columns
        Wp, bp = W[:, perm], t(sd["lstm/b_gates"])[perm]
        WxT = torch.zeros((4 * hid, self.ldx))
        WxT[:, :self.D] = Wp[:self.D].t()
        Wr = torch.zeros((self.ldz, 4 * hid))
        Wr[:self.K] = Wp[self.D:]
        Wr[self.K] = bp
        Wi = torch.zeros((self.ldh, self.IP))
        o = 0
        for name in INTERFACE:
            w, b_ = t(sd["memory_access/%s/w" % name]), t(sd["memory_access/%s/b" % name])
            Wi[:hid, o:o + w.shape[1]] = w
            Wi[hid, o:o + w.shape[1]] = b_
            o += w.shape[1]
        assert o == self.I
        Wy = torch.zeros((self.ldy, self.OP))
        Wy[:self.Ky, :self.O] = t(sd["output_linear/w"])
        Wy[self.Ky, :self.O] = t(sd["output_linear/b"])
        # flat fp32 master copy (kernel layout) + same-shaped gradient: the optimiser and the all-reduce are flat passes
        shapes = [("WxT", WxT), ("Wr", Wr), ("Wi", Wi), ("Wy", Wy)]
        layout = [(n, tuple(v.shape)) for n, v in shapes]
        if getattr(self, "params", None) is None or self.params.shapes != layout:
            self.params = FlatParams(layout, dev)       # else: load into the existing buffer (an optimiser may hold it)
        for n, v in shapes:
            self.params.view(n).copy_(v.to(dev))

    WxT = property(lambda self: self.params.view("WxT"))
    Wr = property(lambda self: self.params.view("Wr"))
    Wi = property(lambda self: self.params.view("Wi"))
    Wy = property(lambda self: self.params.view("Wy"))

    def _unpack(self, grad=False):
        """Packed kernel layout -> the reference's Sonnet variable layout (device tensors)."""
        hid, dev = self.hid, self.device
        gWxT, gWr = self.params.view("WxT", grad), self.params.view("Wr", grad)
        gWi, gWy = self.params.view("Wi", grad), self.params.view("Wy", grad)
        inv = torch.arange(4 * hid, device=dev).view(hid, 4).t().reshape(-1)
        Wg = torch.cat([gWxT[:, :self.D].t(), gWr[:self.K]], dim=0)
        out = {"lstm/w_gates": Wg[:, inv].contiguous(), "lstm/b_gates": gWr[self.K][inv].contiguous()}
        o = 0
        for name in INTERFACE:
            wd = self.interface_widths()[name]
            out["memory_access/%s/w" % name] = gWi[:hid, o:o + wd].contiguous()
            out["memory_access/%s/b" % name] = gWi[hid, o:o + wd].contiguous()
            o += wd
        out["output_linear/w"] = gWy[:self.Ky, :self.O].contiguous()
        out["output_linear/b"] = gWy[self.Ky, :self.O].contiguous()
        return self._strip_sd(out)

    def state_dict(self):
        return {k: v.cpu() for k, v in self._unpack().items()}

    # ---- state
    def initial_state(self, batch_size, dtype=torch.float32):
        """dnc.py:129-134: all zeros (not trainable)."""
        z = lambda *s: torch.zeros(s, device=self.device, dtype=torch.float32)
        B, N, W, R, Wn = batch_size, self.N, self.word_size, self.R, self.Wn
        return DNCState(
            access_output=z(B, R, W),
            access_state=AccessState(z(B, N, W), z(B, R, N), z(B, Wn, N), TemporalLinkageState(z(B, Wn, N, N), z(B, Wn, N)), z(B, N)),
            controller_state=LSTMState(z(B, self.hid), z(B, self.hid)))

    @property
    def state_size(self):
        N, W, R, Wn = self.N, self.word_size, self.R, self.Wn
        return DNCState(R * W, AccessState((N, W), (R, N), (Wn, N), TemporalLinkageState((Wn, N, N), (Wn, N)), (N,)),
                        LSTMState((self.hid,), (self.hid,)))

    @property
    def output_size(self):
        return (self.O,)

    # ---- sequence kernel: inputs time-major [S,B,D] like dynamic_rnn(time_major=True)
    def run_sequence(self, inputs_tm, prev_state=None, record=False):
        S, B, D = inputs_tm.shape
        if self.D is None:
            self._build(D)
        if D != self.D:
            raise _lib.NtkError("inputs have %d features, the core was built for %d" % (D, self.D))
        dev = self.device
        X = torch.zeros((B, S, self.ldx), device=dev)
        X[:, :, :D] = inputs_tm.transpose(0, 1)
        xproj = gemm_nt(X.view(B * S, self.ldx), self.WxT)
        self.last_X = X
        return self.run_projected(xproj, B, S, prev_state, record=record)

    REC_NAMES = ("z", "gates", "c", "hc", "yin", "ifc", "u", "ww", "rw", "cw", "cr", "al", "p", "fwd", "bwd", "M", "L", "ypre")

    def _alloc_records(self, B, S, cap=None):
        """Record tensors [B, S, ...].  cap >= S: allocate room for cap steps and return the leading [B, S, ...] part -- segmented
        BPTT asks for its (equal) segment length every time, so that a freed set's blocks fit the next set exactly (a 70 GB
        block does not serve a 73.5 GB request, and config 5 has no room for a third link record)."""
        cap = max(S, cap or S)

        def e(*s):
            n = 1
            for v in s:
                n *= v
            return torch.empty((B * cap * n,), device=self.device)[:B * S * n].view((B, S) + s)
        N, W, R, Wn, hid = self.N, self.W, self.R, self.Wn, self.hid
        return {"z": e(self.ldz), "gates": e(hid, 4), "c": e(hid), "hc": e(self.ldh), "yin": e(self.ldy), "ifc": e(self.IP),
                "u": e(N), "ww": e(Wn, N), "rw": e(R, N), "cw": e(Wn, N), "cr": e(R, N), "al": e(Wn, N), "p": e(Wn, N),
                "fwd": e(R, Wn, N), "bwd": e(R, Wn, N), "M": e(N, W), "L": e(Wn, N, N), "ypre": e(self.O)}

    #: per-sequence-step record budget above which a recorded pass switches to segmented BPTT (bytes).  Config 5
    #: (N=512: 1 MiB of link per step and sequence, 3250 steps) cannot keep every step: the forward pass then keeps
    #: one state checkpoint per segment and backward_sequence re-records segment by segment, last to first.
    record_budget_bytes = 96 << 30      # a third of an MI355X's 288 GB: two record sets may be alive at a time (allocator caching)
    #: steps per BPTT segment; None = derive from record_budget_bytes (whole sequence when it fits)
    bptt_segment = None
    last_record = None
    last_segments = None
    #: how many trailing segments the forward pass of a segmented run records itself (2 = as many as may be alive at a time)
    recorded_tail_segments = 2
    _rerec_stream = None
    last_initial = None

    def _record_floats_per_step(self):
        N, W, R, Wn, hid = self.N, self.W, self.R, self.Wn, self.hid
        return (self.ldz + 5 * hid + self.ldh + self.ldy + self.IP + N + 4 * Wn * N + 2 * R * N + 2 * R * Wn * N +
                N * W + Wn * N * N + self.O)

    def _segment_len(self, B, S):
        if self.bptt_segment is not None:
            return max(1, min(S, int(self.bptt_segment)))
        per_step = 4 * B * self._record_floats_per_step()
        seg = max(1, min(S, self.record_budget_bytes // per_step))
        n = -(-S // seg)                                 # that many segments are needed: make them even, so that a record set is
        return -(-S // n)                                # no larger than it has to be (config 5: 1 084 steps, not 1 101)

    #: cluster size of the multi-CU sequence kernels: None = automatic (largest that fits), 0 = never (one workgroup
    #: per sequence, the ntk_dnc_seq_* kernels), k = exactly k or fall back.  NTK_DNC_CLUSTER_K overrides the default.
    cluster_k = None
    _cluster = None

    #: which cluster form may run: None = automatic (the LDS-resident form when the shape fits it, else the memory-partitioned
    #: form), "lds" = only ntk_dnc_cluster_*, "mp" = only ntk_dnc_mp_* (csrc/dnc_mp.h).  NTK_DNC_CLUSTER_FORM overrides the default.
    cluster_form = None

    def _want(self):
        import os
        want = self.cluster_k
        if want is None and os.environ.get("NTK_DNC_CLUSTER_K"):
            want = int(os.environ["NTK_DNC_CLUSTER_K"])
        form = self.cluster_form or os.environ.get("NTK_DNC_CLUSTER_FORM") or None
        return want, form

    def _plan(self, B, bwd):
        """(form, k, workspace tensor, workspace bytes) of the cluster kernels at batch B, or None when neither form takes
        the shape (then the one-workgroup-per-sequence kernels run)."""
        want, form = self._want()
        if want == 0:
            return None
        key = (B, want, form)
        slot = "_cluster_b" if bwd else "_cluster"
        cur = getattr(self, slot)
        if cur is None or cur[0] != key:
            L = _lib.lib()
            found = None
            for f, fn in (("lds", L.ntk_dnc_cluster_bwd_plan if bwd else L.ntk_dnc_cluster_plan),
                          ("mp", getattr(L, "ntk_dnc_mp_bwd_plan", None) if bwd else L.ntk_dnc_mp_plan)):
                if form not in (None, f) or fn is None:
                    continue
                k, nbytes = ctypes.c_int(0), ctypes.c_size_t(0)
                rc = fn(B, self.N, self.W, self.R, self.Wn, self.hid, self.O, int(want or 0), ctypes.byref(k), ctypes.byref(nbytes))
                if rc == 0 and k.value > 1:
                    # zeroed ONCE here: the mp form keeps a sticky error word in the workspace that no launch clears
                    ws = torch.zeros((nbytes.value + 3) // 4, device=self.device, dtype=torch.float32)
                    found = (f, k.value, ws, nbytes.value)
                    break
            setattr(self, slot, (key, found))
            cur = getattr(self, slot)
        return cur[1]

    def _cluster_plan(self, B):
        return self._plan(B, False)

    _cluster_b = None

    def _cluster_bwd_plan(self, B):
        return self._plan(B, True)

    def check_cluster(self, clear=True):
        """Synchronise and raise if a hand-off of the cluster launches timed out (the last launch, or -- sticky word -- any
        launch since the last check).  `DNCOffsetTracker.check_step()` calls it (the place a training script reads the loss on
        the host); bench.py calls that after its timed loop.  `clear` applies to the memory-partitioned form only: the
        LDS-resident form's status call always reads AND clears its sticky word."""
        for c in (self._cluster, self._cluster_b):
            if c is not None and c[1] is not None:
                form, k, ws, nbytes = c[1]
                if form == "lds":
                    _lib.check(_lib.lib().ntk_dnc_cluster_status(_P(ws), c[0][0], k, _lib.stream()), "ntk_dnc_cluster_status")
                else:
                    _lib.check(_lib.lib().ntk_dnc_mp_status(_P(ws), nbytes, c[0][0], k, 1 if clear else 0, _lib.stream()), "ntk_dnc_mp_status")

    def guard(self, loss=None, grad=None):
        """Device-side propagation of an aborted cluster launch (no synchronisation): if a hand-off of any cluster launch
        since the last check_cluster() timed out, `loss` and every element of `grad` become NaN (ntk_dnc_cluster_guard).
        NaN rather than zero so that the data-parallel SUM all-reduce carries the failure to every rank: the optimiser
        (RMSPropClip.step -> ntk_rmsprop_clip_step_checked) then skips the update everywhere."""
        for c in (self._cluster, self._cluster_b):
            if c is not None and c[1] is not None:
                form, k, ws, nbytes = c[1]
                _lib.check(_lib.lib().ntk_dnc_cluster_guard(_P(ws), nbytes, 1 if form == "mp" else 0, c[0][0], k,
                                                            _P(loss) if loss is not None else None,
                                                            _P(grad) if grad is not None else None,
                                                            grad.numel() if grad is not None else 0, _lib.stream()),
                           "ntk_dnc_cluster_guard")

    def inject_abort(self, B, bwd=False):
        """Fault injection (tests): set the sticky error word of the forward (or BPTT) cluster workspace at batch B exactly
        as a timed-out hand-off does.  Returns False when no cluster form runs at this shape."""
        plan = self._cluster_bwd_plan(B) if bwd else self._cluster_plan(B)
        if not plan:
            return False
        form, k, ws, nbytes = plan
        _lib.check(_lib.lib().ntk_dnc_cluster_inject_abort(_P(ws), nbytes, 1 if form == "mp" else 0, B, k, _lib.stream()),
                   "ntk_dnc_cluster_inject_abort")
        return True

    def _may_overlap(self, B):
        """May the re-recording forward pass of segment s - 1 run on a side stream WHILE segment s is back-propagated?
        The cluster kernels are cooperative: every one of their B * k workgroups must be resident (one per CU, their LDS
        footprint allows no second one) before any hand-off completes.  Two such launches side by side that together want
        more CUs than the device has can each be dispatched in part and spin on peers that never arrive, until the bounded
        waits abort both.  So: overlap only when both grids fit the device together."""
        pf, pb = self._cluster_plan(B), self._cluster_bwd_plan(B)
        if not pf and not pb:
            return True                                   # one workgroup per sequence on both sides: nothing waits on a peer
        need = B * (pf[1] if pf else 1) + B * (pb[1] if pb else 1)
        return need <= _lib.lib().ntk_cu_count()

    def cluster_placement(self):
        """(clusters that ran the same-XCD form of the hand-offs, clusters) of the last forward / BPTT cluster launches;
        a speed diagnostic only (csrc/dnc_cluster.h).  Synchronises."""
        out = []
        for c in (self._cluster, self._cluster_b):
            if c is not None and c[1] is not None:
                form, k, ws, _nb = c[1]
                n = ctypes.c_int(0)
                fn = _lib.lib().ntk_dnc_cluster_placement if form == "lds" else _lib.lib().ntk_dnc_mp_placement
                _lib.check(fn(_P(ws), c[0][0], k, ctypes.byref(n), _lib.stream()), "ntk_dnc_cluster_placement")
                out.append((n.value, c[0][0]))
        return out

    def _launch_fwd(self, xproj, B, S, st, rec):
        """One sequence-kernel launch over contiguous xproj [B*S, 4*hid] starting from state `st` (not modified):
        the cluster kernel (k CUs per sequence) when the shape allows, else one workgroup per sequence."""
        acc = st.access_state
        # the kernel updates the state in place: work on private copies
        mem, link = acc.memory.clone().contiguous(), acc.linkage.link.clone().contiguous()
        usage, rw, ww = acc.usage.clone().contiguous(), acc.read_weights.clone().contiguous(), acc.write_weights.clone().contiguous()
        prec, reads = acc.linkage.precedence_weights.clone().contiguous(), st.access_output.clone().contiguous()
        hc = torch.cat([st.controller_state.hidden, st.controller_state.cell], dim=1).contiguous()
        out = torch.empty((B, S, self.O), device=self.device)
        recp = [(_P(rec[k]) if rec else None) for k in self.REC_NAMES]
        plan = self._cluster_plan(B)
        self.last_cluster_k = plan[1] if plan else 1
        self.last_cluster_form = plan[0] if plan else None
        if plan:
            fn = _lib.lib().ntk_dnc_cluster_fwd if plan[0] == "lds" else _lib.lib().ntk_dnc_mp_fwd
            _lib.check(fn(B, S, self.N, self.W, self.R, self.Wn, self.hid, self.O, self.clip_value, plan[1],
                          _P(xproj), _P(self.Wr), _P(self.Wi), _P(self.Wy), _P(mem), _P(link), _P(usage),
                          _P(rw), _P(ww), _P(prec), _P(reads), _P(hc), _P(out), *recp, _P(plan[2]),
                          _lib.stream()), "ntk_dnc_cluster_fwd" if plan[0] == "lds" else "ntk_dnc_mp_fwd")
        else:
            _lib.check(_lib.lib().ntk_dnc_seq_fwd(B, S, self.N, self.W, self.R, self.Wn, self.hid, self.O, self.clip_value,
                                                  _P(xproj), _P(self.Wr), _P(self.Wi), _P(self.Wy), _P(mem), _P(link), _P(usage),
                                                  _P(rw), _P(ww), _P(prec), _P(reads), _P(hc), _P(out), *recp, _lib.stream()),
                       "ntk_dnc_seq_fwd")
        new = DNCState(reads, AccessState(mem, rw, ww, TemporalLinkageState(link, prec), usage),
                       LSTMState(hc[:, :self.hid].contiguous(), hc[:, self.hid:].contiguous()))
        return out, new

    def run_projected(self, xproj, B, S, prev_state=None, record=False):
        """Sequence kernel on an already projected input (xproj [B*S, 4*hid] = X WxT^T)."""
        st = self._pad_state(prev_state or self.initial_state(B))
        seg = self._segment_len(B, S) if record else S
        self.last_initial = st
        self.last_segments = None
        if seg >= S:
            rec = self._alloc_records(B, S) if record else {}
            out, new = self._launch_fwd(xproj, B, S, st, rec)
            self.last_record = rec
            return out.transpose(0, 1), self._strip_state(new)          # time-major [S,B,O]
        # segmented: forward without records, one state checkpoint per segment
        xp = xproj.view(B, S, 4 * self.hid)
        out = torch.empty((B, S, self.O), device=self.device)
        ckpt, bounds = [], []
        starts = list(range(0, S, seg))
        # The LAST TWO segments are recorded right here: BPTT walks the segments last to first and two record sets may be
        # alive at a time (record_budget_bytes), so neither of them needs a second forward pass -- the forward work of a
        # training step is S + (n - 2) / n * S steps instead of S + (n - 1) / n * S (three segments at config 5: 1.33 S, not 1.67 S)
        keep_from = max(0, len(starts) - self.recorded_tail_segments)
        recs = {}
        for si, s0 in enumerate(starts):
            s1 = min(S, s0 + seg)
            ckpt.append(st)
            bounds.append((s0, s1))
            rec = self._alloc_records(B, s1 - s0, cap=seg) if si >= keep_from else {}
            o, st = self._launch_fwd(xp[:, s0:s1].contiguous().view(B * (s1 - s0), 4 * self.hid), B, s1 - s0, st, rec)
            out[:, s0:s1] = o
            if rec:
                recs[si] = rec
        self.last_record = {}
        self.last_segments = (xp, ckpt, bounds, recs, seg)
        return out.transpose(0, 1), self._strip_state(st)

    def _launch_bwd(self, B, S, st0, rec, dout, WrT, ldkT, WiT, ldhT, gM, gL, gcarry, carry_in):
        dev, hid = self.device, self.hid
        acc = st0.access_state
        hc0 = torch.cat([st0.controller_state.hidden, st0.controller_state.cell], dim=1).contiguous()
        dgates = torch.empty((B, S, 4 * hid), device=dev)
        dxi = torch.empty((B, S, self.IP), device=dev)
        dypre = torch.empty((B, S, self.OP), device=dev)
        keep = []                                    # contiguous copies stay referenced until the launch is queued

        def c(t):
            keep.append(t.contiguous())
            return _P(keep[-1])
        recs = (_P(rec["gates"]), _P(rec["c"]), _P(rec["ifc"]), _P(rec["u"]), _P(rec["ww"]), _P(rec["rw"]), _P(rec["cw"]),
                _P(rec["cr"]), _P(rec["al"]), _P(rec["p"]), _P(rec["fwd"]), _P(rec["bwd"]), _P(rec["M"]), _P(rec["L"]),
                _P(rec["ypre"]), _P(dout), _P(gM), _P(gL), _P(dgates), _P(dxi), _P(dypre),
                _P(gcarry) if gcarry is not None else None, 1 if carry_in else 0)
        state0 = (c(acc.memory), c(acc.linkage.link), c(acc.usage), c(acc.read_weights), c(acc.write_weights),
                  c(acc.linkage.precedence_weights), _P(hc0))
        plan = self._cluster_bwd_plan(B)
        self.last_cluster_bwd_k = plan[1] if plan else 1
        self.last_cluster_bwd_form = plan[0] if plan else None
        if plan:
            fn = _lib.lib().ntk_dnc_cluster_bwd if plan[0] == "lds" else _lib.lib().ntk_dnc_mp_bwd
            _lib.check(fn(B, S, self.N, self.W, self.R, self.Wn, hid, self.O, self.clip_value, plan[1],
                          _P(WrT), ldkT, _P(self.Wi), _P(self.Wy), *state0, *recs, _P(plan[2]), _lib.stream()),
                       "ntk_dnc_cluster_bwd" if plan[0] == "lds" else "ntk_dnc_mp_bwd")
        else:
            _lib.check(_lib.lib().ntk_dnc_seq_bwd(
                B, S, self.N, self.W, self.R, self.Wn, hid, self.O, self.clip_value,
                _P(WrT), ldkT, _P(WiT), ldhT, _P(self.Wy), *state0, *recs, _lib.stream()), "ntk_dnc_seq_bwd")
        return dgates, dxi, dypre

    def _weight_grads(self, X2, rec, dgates, dxi, dypre, BS, accumulate):
        P, hid = self.params, self.hid
        gemm_tn(dgates.view(BS, 4 * hid), X2, P.view("WxT", grad=True), accumulate=accumulate)
        gemm_tn(rec["z"].view(BS, self.ldz), dgates.view(BS, 4 * hid), P.view("Wr", grad=True), accumulate=accumulate)
        gemm_tn(rec["hc"].view(BS, self.ldh), dxi.view(BS, self.IP), P.view("Wi", grad=True), accumulate=accumulate)
        gemm_tn(rec["yin"].view(BS, self.ldy), dypre.view(BS, self.OP), P.view("Wy", grad=True), accumulate=accumulate)

    def backward_sequence(self, X, dout, unpack=True):
        """BPTT through the last recorded sequence (run_projected(..., record=True)).
        X [B,S,ldx] serialised inputs, dout [B,S,O] = d loss / d output.  Returns the gradients in the
        reference's Sonnet variable layout ({name: tensor on device}) -- or nothing with unpack=False: the gradients are in
        `params.grad` (packed kernel layout) either way, which is all a training step reads; the re-layout is a dozen small
        launches at the tail of the serial chain (1.3 ms of configs[2]'s step).  When the forward pass was segmented
        (see record_budget_bytes) each segment is re-recorded from its checkpoint, last segment first; the
        state gradients flow between segments through gM / gL / gcarry."""
        if not 1 <= self.Wn <= 4:
            raise _lib.NtkError("DNC BPTT on the HIP path implements 1..4 write heads (got %d)" % self.Wn)
        rec, st0 = self.last_record, self.last_initial
        if not rec and not self.last_segments:
            raise _lib.NtkError("backward_sequence needs a recorded forward pass (record=True)")
        B, S, _ = X.shape
        dev, L, stream = self.device, _lib.lib(), _lib.stream()
        hid = self.hid
        ldkT, ldhT = (self.K + 3) // 4 * 4, (hid + 3) // 4 * 4
        WrT = torch.empty((4 * hid, ldkT), device=dev)
        WiT = torch.empty((self.IP, ldhT), device=dev)
        _lib.check(L.ntk_transpose_pad(_P(self.Wr), 4 * hid, _P(WrT), ldkT, self.K, 4 * hid, stream), "ntk_transpose_pad")
        _lib.check(L.ntk_transpose_pad(_P(self.Wi), self.IP, _P(WiT), ldhT, hid, self.IP, stream), "ntk_transpose_pad")
        gM = torch.zeros((B, self.N, self.W), device=dev)
        gL = torch.zeros((B, self.Wn, self.N, self.N), device=dev)
        dout = dout.contiguous()
        if not self.last_segments:
            dgates, dxi, dypre = self._launch_bwd(B, S, st0, rec, dout, WrT, ldkT, WiT, ldhT, gM, gL, None, False)
            self._weight_grads(X.view(B * S, self.ldx), rec, dgates, dxi, dypre, B * S, False)
            return self._unpack(grad=True) if unpack else None
        xp, ckpt, bounds, fwd_recs, seg_cap = self.last_segments
        nseg = len(bounds)
        gcarry = torch.zeros((B, (self.Wn + 1) * self.N + self.R * self.N + ldkT + hid), device=dev)
        # Last segment first.  The re-recording forward pass of segment s - 1 depends only on its checkpoint, so it runs on a
        # side stream (other CUs: these kernels hold one CU per sequence) WHILE segment s is back-propagated: the second forward
        # pass disappears behind the longer backward pass.  Two record sets are alive at a time (record_budget_bytes).
        cur = torch.cuda.current_stream(dev)
        if self._rerec_stream is None:
            self._rerec_stream = torch.cuda.Stream(device=dev)
        side = self._rerec_stream if self._may_overlap(B) else cur      # cooperative grids that do not fit together: serialise
        self.last_rerecord_overlapped = side is not cur
        segs = list(zip(reversed(ckpt), reversed(bounds)))

        def rerecord(k):
            si = nseg - 1 - k                                      # segs is last-to-first
            if si in fwd_recs:                                     # recorded by the forward pass itself: consumed here
                return fwd_recs.pop(si), None                      # (no second reference: the set must die with its segment)
            st_k, (a0, a1) = segs[k]
            # records come from THIS stream's allocator pool (a freed set is reused two segments later; pools are per stream:
            # allocating on the side stream kept three sets alive and ran out of HBM at config 5); the side stream orders
            # itself after everything enqueued here, which includes the last reader of the memory being reused
            r = self._alloc_records(B, a1 - a0, cap=seg_cap)
            if side is cur:
                self._launch_fwd(xp[:, a0:a1].contiguous().view(B * (a1 - a0), 4 * hid), B, a1 - a0, st_k, r)
                return r, None
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._launch_fwd(xp[:, a0:a1].contiguous().view(B * (a1 - a0), 4 * hid), B, a1 - a0, st_k, r)
                ev = torch.cuda.Event()
                ev.record(side)
            return r, ev

        pending = rerecord(0)
        first = True
        for k, (st, (s0, s1)) in enumerate(segs):
            n = s1 - s0
            rec, ev = pending
            pending = rerecord(k + 1) if k + 1 < len(segs) else None
            if ev is not None:
                cur.wait_event(ev)
            dgates, dxi, dypre = self._launch_bwd(B, n, st, rec, dout[:, s0:s1].contiguous(), WrT, ldkT, WiT, ldhT, gM, gL,
                                                  gcarry, not first)
            self._weight_grads(X[:, s0:s1].contiguous().view(B * n, self.ldx), rec, dgates, dxi, dypre, B * n, not first)
            first = False
            del rec                                                # freed in this stream's order: after its last reader
        return self._unpack(grad=True) if unpack else None

    def __call__(self, inputs, prev_state):
        """One step of the core: (output [B,O], DNCState), dnc.py:84-127."""
        y, new = self.run_sequence(inputs.unsqueeze(0), prev_state)
        return y[0], new


def run_model(input_sequence, output_size, core=None, **flags):
    """direct_offset_output_with_dnc.py:66-88 with FLAGS passed as keyword arguments:
    mem_size, mem_dim, read_head_size, write_head_size, hidden_size, clip_value."""
    if core is None:
        core = DNC({"memory_size": flags.get("mem_size", 128), "word_size": flags.get("mem_dim", 20),
                    "num_reads": flags.get("read_head_size", 4), "num_writes": flags.get("write_head_size", 1)},
                   {"hidden_size": flags.get("hidden_size", 200)}, output_size, flags.get("clip_value", 20),
                   device=input_sequence.device)
    out, _ = core.run_sequence(input_sequence, core.initial_state(input_sequence.shape[1]))
    return out


# ---------------------------------------------------------------------------------------------------------
# Stand-alone addressing modules (dnc/addressing.py) with the reference's constructor arguments and call
# signatures; forward only (training goes through the fused core above).
# ---------------------------------------------------------------------------------------------------------
def _f32(t, dev):
    return torch.as_tensor(t, dtype=torch.float32).to(dev).contiguous()


class CosineWeights(object):
    """addressing.py:59-105: CosineWeights(num_heads, word_size)(memory [B,N,W], keys [B,H,W], strengths [B,H]) -> [B,H,N]
    (strength_op is softplus, the reference default)."""

    def __init__(self, num_heads, word_size, name="cosine_weights", device="cuda"):
        self._num_heads, self._word_size, self.device = num_heads, word_size, torch.device(device)

    def __call__(self, memory, keys, strengths):
        m, k, s = _f32(memory, self.device), _f32(keys, self.device), _f32(strengths, self.device)
        B, N, W = m.shape
        H = k.shape[1]
        out = torch.empty((B, H, N), device=self.device)
        _lib.check(_lib.lib().ntk_dnc_cosine_weights(_P(m), _P(k), _P(s), _P(out), B, N, W, H, _lib.stream()), "ntk_dnc_cosine_weights")
        return out


class TemporalLinkage(object):
    """addressing.py:108-249: TemporalLinkage(memory_size, num_writes)(write_weights, prev_state) -> TemporalLinkageState;
    directional_read_weights(link, prev_read_weights, forward) -> [B,R,Wn,N]."""

    def __init__(self, memory_size, num_writes, name="temporal_linkage", device="cuda"):
        self._memory_size, self._num_writes, self.device = memory_size, num_writes, torch.device(device)

    def __call__(self, write_weights, prev_state):
        ww = _f32(write_weights, self.device)
        pl, pp = _f32(prev_state.link, self.device), _f32(prev_state.precedence_weights, self.device)
        B, Wn, N = ww.shape
        link, prec = torch.empty_like(pl), torch.empty_like(pp)
        _lib.check(_lib.lib().ntk_dnc_linkage(_P(pl), _P(pp), _P(ww), _P(link), _P(prec), B, N, Wn, _lib.stream()), "ntk_dnc_linkage")
        return TemporalLinkageState(link=link, precedence_weights=prec)

    def directional_read_weights(self, link, prev_read_weights, forward):
        L, rw = _f32(link, self.device), _f32(prev_read_weights, self.device)
        B, Wn, N, _ = L.shape
        R = rw.shape[1]
        out = torch.empty((B, R, Wn, N), device=self.device)
        _lib.check(_lib.lib().ntk_dnc_directional_read_weights(_P(L), _P(rw), _P(out), B, N, Wn, R, 1 if forward else 0,
                                                               _lib.stream()), "ntk_dnc_directional_read_weights")
        return out

    @property
    def state_size(self):
        return TemporalLinkageState((self._num_writes, self._memory_size, self._memory_size), (self._num_writes, self._memory_size))


class Freeness(object):
    """addressing.py:252-410: Freeness(memory_size)(write_weights, free_gate, read_weights, prev_usage) -> usage;
    write_allocation_weights(usage, write_gates, num_writes) -> [B,Wn,N]."""

    def __init__(self, memory_size, name="freeness", device="cuda"):
        self._memory_size, self.device = memory_size, torch.device(device)

    def __call__(self, write_weights, free_gate, read_weights, prev_usage):
        ww, fg = _f32(write_weights, self.device), _f32(free_gate, self.device)
        rw, pu = _f32(read_weights, self.device), _f32(prev_usage, self.device)
        B, Wn, N = ww.shape
        usage = torch.empty_like(pu)
        _lib.check(_lib.lib().ntk_dnc_freeness(_P(ww), _P(fg), _P(rw), _P(pu), _P(usage), B, N, Wn, rw.shape[1], _lib.stream()),
                   "ntk_dnc_freeness")
        return usage

    def write_allocation_weights(self, usage, write_gates, num_writes):
        u, g = _f32(usage, self.device), _f32(write_gates, self.device)
        B, N = u.shape
        out = torch.empty((B, num_writes, N), device=self.device)
        _lib.check(_lib.lib().ntk_dnc_write_allocation_weights(_P(u), _P(g), _P(out), B, N, num_writes, _lib.stream()),
                   "ntk_dnc_write_allocation_weights")
        return out

    def _allocation(self, usage):
        u = _f32(usage, self.device)
        B, N = u.shape
        out = torch.empty((B, 1, N), device=self.device)
        _lib.check(_lib.lib().ntk_dnc_write_allocation_weights(_P(u), None, _P(out), B, N, 1, _lib.stream()),
                   "ntk_dnc_write_allocation_weights")
        return out[:, 0]

    @property
    def state_size(self):
        return (self._memory_size,)


class MemoryAccess(object):
    """dnc/access.py:66-303 -- MemoryAccess(memory_size=128, word_size=20, num_reads=1, num_writes=1), callable on its
    own: ``module(inputs [B,D], AccessState) -> (read_words [B,R,W], AccessState)``; the reference's tests also call
    ``_read_inputs``, ``_write_weights`` and ``_read_weights`` directly.  Forward through the module kernels, module-granular gradients through ``step_gradients``; parameters carry the Sonnet
    names ``memory_access/<linear>/{w,b}``.  (Training runs through the fused DNC core.)"""

    def __init__(self, memory_size=128, word_size=20, num_reads=1, num_writes=1, name="memory_access", input_dim=None,
                 device="cuda", seed=0):
        self.N, self.W, self.R, self.Wn = int(memory_size), int(word_size), int(num_reads), int(num_writes)
        self.device, self.seed, self.D = torch.device(device), seed, None
        N, W, R, Wn = self.N, self.W, self.R, self.Wn
        self.widths = [("write_vectors", Wn * W), ("erase_vectors", Wn * W), ("free_gate", R), ("allocation_gate", Wn),
                       ("write_gate", Wn), ("read_mode", R * (1 + 2 * Wn)), ("write_keys", Wn * W), ("write_strengths", Wn),
                       ("read_keys", R * W), ("read_strengths", R)]
        self.I = sum(w for _, w in self.widths)
        self.IP = (self.I + 3) // 4 * 4
        if input_dim is not None:
            self._build_params(int(input_dim))

    # ---- parameters: the ten snt.Linear modules packed as one [IP][ldx] matrix (row = interface column) + bias
    def _build_params(self, D, sd=None):
        self.D, self.ldx = D, (D + 3) // 4 * 4
        if sd is None:
            g = torch.Generator().manual_seed(int(self.seed))
            sd = {}
            for name, width in self.widths:
                sd["memory_access/%s/w" % name] = torch.clamp(torch.randn((D, width), generator=g), -2, 2) / D ** 0.5
                sd["memory_access/%s/b" % name] = torch.zeros(width)
        self.load_state_dict(sd)

    def load_state_dict(self, sd):
        t = lambda v: torch.as_tensor(v, dtype=torch.float32)
        D = t(sd["memory_access/write_vectors/w"]).shape[0]
        self.D, self.ldx = D, (D + 3) // 4 * 4
        WT, bias = torch.zeros((self.IP, self.ldx)), torch.zeros(self.IP)
        o = 0
        for name, width in self.widths:
            WT[o:o + width, :D] = t(sd["memory_access/%s/w" % name]).t()
            bias[o:o + width] = t(sd["memory_access/%s/b" % name])
            o += width
        self.WT, self.bias = WT.to(self.device), bias.to(self.device)

    def state_dict(self):
        out, o = {}, 0
        for name, width in self.widths:
            out["memory_access/%s/w" % name] = self.WT[o:o + width, :self.D].t().contiguous().cpu()
            out["memory_access/%s/b" % name] = self.bias[o:o + width].cpu()
            o += width
        return out

    def initial_state(self, batch_size, dtype=torch.float32):
        z = lambda *s: torch.zeros(s, device=self.device)
        B, N, W, R, Wn = batch_size, self.N, self.W, self.R, self.Wn
        return AccessState(z(B, N, W), z(B, R, N), z(B, Wn, N), TemporalLinkageState(z(B, Wn, N, N), z(B, Wn, N)), z(B, N))

    @property
    def state_size(self):
        N, W, R, Wn = self.N, self.W, self.R, self.Wn
        return AccessState((N, W), (R, N), (Wn, N), TemporalLinkageState((Wn, N, N), (Wn, N)), (N,))

    @property
    def output_size(self):
        return (self.R, self.W)

    def _raw_interface(self, inputs):
        x = _f32(inputs, self.device)
        B, D = x.shape
        if self.D is None:
            self._build_params(D)
        if D != self.D:
            raise _lib.NtkError("inputs have %d features, the module was built for %d" % (D, self.D))
        X = torch.zeros((B, self.ldx), device=self.device)
        X[:, :D] = x
        raw = torch.empty((B, self.IP), device=self.device)
        _lib.check(_lib.lib().ntk_gemm_nt_f32(_P(X), self.ldx, _P(self.WT), self.ldx, _P(self.bias), _P(raw), self.IP,
                                              B, self.IP, self.ldx, _lib.stream()), "ntk_gemm_nt_f32")
        return raw, B

    def _read_inputs(self, inputs):
        """access.py:160-218 -> dict of activated interface tensors (reference key names)."""
        raw, B = self._raw_interface(inputs)
        act = torch.empty(B * self.IP, device=self.device)
        _lib.check(_lib.lib().ntk_dnc_interface_activations(_P(raw), self.IP, _P(act), B, self.N, self.W, self.R, self.Wn,
                                                            _lib.stream()), "ntk_dnc_interface_activations")
        N, W, R, Wn = self.N, self.W, self.R, self.Wn
        shapes = {"write_vectors": (B, Wn, W), "erase_vectors": (B, Wn, W), "free_gate": (B, R), "allocation_gate": (B, Wn),
                  "write_gate": (B, Wn), "read_mode": (B, R, 1 + 2 * Wn), "write_keys": (B, Wn, W), "write_strengths": (B, Wn),
                  "read_keys": (B, R, W), "read_strengths": (B, R)}
        rename = {"write_keys": "write_content_keys", "write_strengths": "write_content_strengths",
                  "read_keys": "read_content_keys", "read_strengths": "read_content_strengths"}
        out, o = {}, 0
        for name, width in self.widths:
            out[rename.get(name, name)] = act[B * o:B * (o + width)].view(shapes[name])
            o += width
        return out

    def _write_weights(self, inputs, memory, usage):
        """access.py:220-257."""
        dev = self.device
        m, u = _f32(memory, dev), _f32(usage, dev)
        B, N, W = m.shape
        Wn = self.Wn
        ws = torch.empty(2 * B * Wn * N + B * Wn, device=dev)
        out = torch.empty((B, Wn, N), device=dev)
        # device copies stay referenced until the launch is queued (a temporary's block would be reused by the next one)
        k, s_ = _f32(inputs["write_content_keys"], dev), _f32(inputs["write_content_strengths"], dev)
        ag, wg = _f32(inputs["allocation_gate"], dev), _f32(inputs["write_gate"], dev)
        _lib.check(_lib.lib().ntk_dnc_write_weights(_P(m), _P(u), _P(k), _P(s_), _P(ag), _P(wg), _P(out), _P(ws), B, N, W, Wn,
                                                    _lib.stream()), "ntk_dnc_write_weights")
        return out

    def _read_weights(self, inputs, memory, prev_read_weights, link):
        """access.py:259-303."""
        dev = self.device
        m, prw, L = _f32(memory, dev), _f32(prev_read_weights, dev), _f32(link, dev)
        B, N, W = m.shape
        R, Wn = self.R, self.Wn
        ws = torch.empty(B * R * N * (1 + 2 * Wn), device=dev)
        out = torch.empty((B, R, N), device=dev)
        k, s_, rm = (_f32(inputs["read_content_keys"], dev), _f32(inputs["read_content_strengths"], dev),
                     _f32(inputs["read_mode"], dev))
        _lib.check(_lib.lib().ntk_dnc_read_weights(_P(m), _P(prw), _P(L), _P(k), _P(s_), _P(rm), _P(out), _P(ws), B, N, W, R, Wn,
                                                   _lib.stream()), "ntk_dnc_read_weights")
        return out

    # ---- module-granular backward (what tf.gradients gives dnc/access_test.py:145-159)
    def _word_fields(self):
        """(offset, groups) of the interface fields that hold words, in the packed order."""
        out, o = [], 0
        for name, width in self.widths:
            if name in ("write_vectors", "erase_vectors", "write_keys"):
                out.append((o, self.Wn))
            elif name == "read_keys":
                out.append((o, self.R))
            o += width
        return out

    def step_gradients(self, inputs, prev_state, d_read_words, d_state=None):
        """Gradients of a scalar function of ONE step's outputs: d_read_words [B,R,W] = its gradient w.r.t. the read words,
        d_state (optional AccessState; write_weights ignored) = w.r.t. the new state's fields.  Returns a dict with the
        gradients w.r.t. 'inputs' [B,D], the previous state's 'memory', 'read_weights', 'link', 'precedence_weights',
        'usage', and the parameters ('memory_access/<linear>/w', '/b').  One C-ABI call (ntk_dnc_access_step_bwd) + the
        linear layers' GEMMs; word_size is zero padded to a multiple of 4 on the way in."""
        dev, L = self.device, _lib.lib()
        raw, B = self._raw_interface(inputs)
        N, W, R, Wn = self.N, self.W, self.R, self.Wn
        Wp = (W + 3) // 4 * 4
        c = lambda t: _f32(t, dev).contiguous()

        def padw(t):                                    # [..., W] -> [..., Wp]
            if Wp == W:
                return c(t)
            out = torch.zeros(tuple(t.shape[:-1]) + (Wp,), device=dev)
            out[..., :W] = _f32(t, dev)
            return out
        # interface rows in the padded-word layout
        vals = [ctypes.c_int() for _ in range(8)]
        _lib.check(L.ntk_dnc_padded_dims(N, Wp, R, Wn, 4, 1, *[ctypes.byref(v) for v in vals]), "ntk_dnc_padded_dims")
        IPp = vals[1].value
        if Wp == W:
            rawp = raw
        else:
            rawp = torch.zeros((B, IPp), device=dev)
            o = op = 0
            words = dict(self._word_fields())
            for _name, width in self.widths:
                if o in words:
                    g = words[o]
                    rawp[:, op:op + g * Wp].view(B, g, Wp)[:, :, :W] = raw[:, o:o + width].reshape(B, g, W)
                    op += g * Wp
                else:
                    rawp[:, op:op + width] = raw[:, o:o + width]
                    op += width
                o += width
        z = lambda *sh: torch.zeros(sh, device=dev)
        ds = d_state
        g_mem = padw(ds.memory) if ds is not None else z(B, N, Wp)
        g_rw = c(ds.read_weights).clone() if ds is not None else z(B, R, N)
        g_link = c(ds.linkage.link).clone() if ds is not None else z(B, Wn, N, N)
        g_prec = c(ds.linkage.precedence_weights).clone() if ds is not None else z(B, Wn, N)
        g_usage = c(ds.usage).clone() if ds is not None else z(B, N)
        if ds is not None and Wp == W:
            g_mem = g_mem.clone()
        mem = padw(prev_state.memory)
        rw, ww = c(prev_state.read_weights), c(prev_state.write_weights)
        link, prec, usage = c(prev_state.linkage.link), c(prev_state.linkage.precedence_weights), c(prev_state.usage)
        dr = padw(d_read_words)
        d_rawp = torch.empty((B, IPp), device=dev)
        ws = torch.empty(L.ntk_dnc_access_step_bwd_workspace_bytes(B, N, Wp, R, Wn) // 4, device=dev)
        _lib.check(L.ntk_dnc_access_step_bwd(_P(rawp), IPp, _P(mem), _P(rw), _P(ww), _P(link), _P(prec), _P(usage), _P(dr), _P(g_mem),
                                             _P(g_rw), _P(g_link), _P(g_prec), _P(g_usage), _P(d_rawp), _P(ws), B, N, Wp, R, Wn,
                                             _lib.stream()), "ntk_dnc_access_step_bwd")
        # back to the module's own interface layout
        if Wp == W:
            d_raw = d_rawp
        else:
            d_raw = torch.zeros((B, self.IP), device=dev)
            o = op = 0
            words = dict(self._word_fields())
            for _name, width in self.widths:
                if o in words:
                    g = words[o]
                    d_raw[:, o:o + width] = d_rawp[:, op:op + g * Wp].view(B, g, Wp)[:, :, :W].reshape(B, width)
                    op += g * Wp
                else:
                    d_raw[:, o:o + width] = d_rawp[:, op:op + width]
                    op += width
                o += width
        # the ten linears: d_inputs = d_raw . W, dW = d_raw^T . inputs, db = d_raw^T . 1   (fp32 MFMA GEMMs)
        X = torch.zeros((B, self.ldx), device=dev)
        X[:, :self.D] = _f32(inputs, dev)
        ones = torch.ones((B, 4), device=dev)
        Wn_t = torch.empty((self.ldx, self.IP), device=dev)
        _lib.check(L.ntk_transpose_pad(_P(self.WT), self.ldx, _P(Wn_t), self.IP, self.IP, self.ldx, _lib.stream()), "ntk_transpose_pad")
        d_in = gemm_nt(d_raw, Wn_t)
        dWT = torch.empty((self.IP, self.ldx), device=dev)
        gemm_tn(d_raw, X, dWT)
        db4 = torch.empty((self.IP, 4), device=dev)
        gemm_tn(d_raw, ones, db4)
        out = {"inputs": d_in[:, :self.D].contiguous(), "memory": g_mem[..., :W].contiguous(), "read_weights": g_rw, "link": g_link,
               "precedence_weights": g_prec, "usage": g_usage}
        o = 0
        for name, width in self.widths:
            out["memory_access/%s/w" % name] = dWT[o:o + width, :self.D].t().contiguous()
            out["memory_access/%s/b" % name] = db4[o:o + width, 0].contiguous()
            o += width
        return out

    def __call__(self, inputs, prev_state):
        """access.py:113-158 in one C-ABI call (ntk_dnc_access_step_fwd)."""
        raw, B = self._raw_interface(inputs)
        dev, L = self.device, _lib.lib()
        N, W, R, Wn = self.N, self.W, self.R, self.Wn
        c = lambda t: _f32(t, dev)
        mem, rw, ww = c(prev_state.memory), c(prev_state.read_weights), c(prev_state.write_weights)
        link, prec, usage = c(prev_state.linkage.link), c(prev_state.linkage.precedence_weights), c(prev_state.usage)
        e = lambda t: torch.empty_like(t)
        mem2, rw2, ww2, link2, prec2, usage2 = e(mem), e(rw), e(ww), e(link), e(prec), e(usage)
        reads = torch.empty((B, R, W), device=dev)
        ws = torch.empty(L.ntk_dnc_access_step_workspace_bytes(B, N, W, R, Wn) // 4, device=dev)
        _lib.check(L.ntk_dnc_access_step_fwd(_P(raw), self.IP, _P(mem), _P(rw), _P(ww), _P(link), _P(prec), _P(usage),
                                             _P(mem2), _P(rw2), _P(ww2), _P(link2), _P(prec2), _P(usage2), _P(reads), _P(ws),
                                             B, N, W, R, Wn, _lib.stream()), "ntk_dnc_access_step_fwd")
        return reads, AccessState(mem2, rw2, ww2, TemporalLinkageState(link2, prec2), usage2)

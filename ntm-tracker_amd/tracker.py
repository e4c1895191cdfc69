"""Graph assembly of the offsets tracker (ntm_offsets(), direct_offset_output.py:401-653)
on the HIP path: frames -> VGG conv4_3 -> 64-point gather + serialise -> NTM
sequence -> output gather / tanh / l2 loss -> BPTT -> clip + RMSProp.

One process drives one GPU; with ``torch.distributed`` initialised the flat
gradient bucket is summed across ranks (one RCCL all-reduce per step) before
the clip, so every rank applies the identical update -- the same result as a
single process running the global batch, because the loss is an un-normalised
sum (direct_offset_output.py:606).
"""
import os
import torch

from . import _lib
from . import parallel
from .ntm import NTMCell, _P, _np
from .vgg import VGG16Conv43

# receptive_field_sizes.py:135-143: y, x in {6, 8, ..., 20} on the 28x28 conv4_3 map
GRID_START, GRID_STEP, GRID_N = 6, 2, 8
NUM_FEATURES = GRID_N * GRID_N


def gather_serialize(fmap, gts0, B, T, ldx, out=None):
    """fmap [B*T,Hf,Wf,C] conv4_3 features, gts0 [B,64] frame-0 heat-map (or None)
    -> X [B, T*65, ldx] (direct_offset_output.py:392-399, :439-500)."""
    F, Hf, Wf, C = fmap.shape
    assert F == B * T
    S = T * (NUM_FEATURES + 1)
    if out is None:
        out = torch.empty((B, S, ldx), device=fmap.device, dtype=torch.float32)
    _lib.check(_lib.lib().ntk_gather_serialize(_P(fmap), _np(gts0), _P(out), B, T, Hf, Wf, C, ldx,
                                              GRID_START, GRID_STEP, GRID_N, _lib.stream()), "ntk_gather_serialize")
    return out


def offset_loss(logits, offsets, T, want_grad=True):
    """direct_offset_output.py:581-606.  Returns (loss[1], pred[B,T-1,O], dlogits or None)."""
    B, S, O = logits.shape
    pred = torch.empty((B, T - 1, O), device=logits.device)
    loss = torch.empty(1, device=logits.device)
    dlogits = torch.empty_like(logits) if want_grad else None
    _lib.check(_lib.lib().ntk_offset_loss(_P(logits), _P(offsets.contiguous()), _P(pred), _P(loss), _np(dlogits),
                                         B, T, NUM_FEATURES, O, _lib.stream()), "ntk_offset_loss")
    return loss, pred, dlogits


class RMSPropClip(object):
    """tf.clip_by_global_norm + tf.train.RMSPropOptimizer on one flat buffer
    (direct_offset_output.py:620-626): ms slot starts at 1, epsilon inside the sqrt."""

    def __init__(self, params, learning_rate=1e-4, decay=0.95, momentum=0.9, epsilon=1e-10, max_gradient_norm=5.0):
        self.p = params
        self.lr, self.decay, self.momentum, self.eps, self.clip = learning_rate, decay, momentum, epsilon, max_gradient_norm
        self.ms = torch.ones_like(params.flat)
        self.mom = torch.zeros_like(params.flat)
        L = _lib.lib()
        self.ws = torch.empty(max(1, L.ntk_global_norm_workspace_bytes(params.numel) // 4), device=params.flat.device)
        self.gnorm = torch.zeros(1, device=params.flat.device)
        self.skipped = torch.zeros(1, device=params.flat.device, dtype=torch.int32)     # steps refused on a non-finite norm
        self.global_step = 0

    def step(self, loss=None):
        """One clipped RMSProp update of the flat buffer from `params.grad` (after the data-parallel all-reduce, so the
        norm -- and the decision below -- is the same on every rank).  A NaN / Inf global norm (a poisoned gradient: an
        aborted cluster launch on ANY rank, DNC.guard) skips the update on the device: parameters and slots untouched,
        `loss` (the step's 1-element loss tensor, optional) becomes NaN, `self.skipped` counts it."""
        L, st = _lib.lib(), _lib.stream()
        n = self.p.numel
        _lib.check(L.ntk_global_norm(_P(self.p.grad), n, _P(self.ws), _P(self.gnorm), st), "ntk_global_norm")
        _lib.check(L.ntk_rmsprop_clip_step_checked(_P(self.p.flat), _P(self.p.grad), _P(self.ms), _P(self.mom), n,
                                                   self.lr, self.decay, self.momentum, self.eps, self.clip, _P(self.gnorm),
                                                   _np(loss), _P(self.skipped), st),
                   "ntk_rmsprop_clip_step_checked")
        self.global_step += 1

    def check(self):
        """Synchronises; raises if any step since the last check was skipped on a non-finite gradient norm."""
        n = int(self.skipped.item())
        if n:
            self.skipped.zero_()
            raise _lib.NtkError("%d optimiser step(s) were skipped: the (all-reduced) gradient norm was not finite -- an aborted "
                                "cluster launch on some rank, or an overflow" % n)


class _Checkpointing(object):
    """Save / restore of parameters + optimiser slots + global_step (the role of tf.train.Saver in the reference,
    direct_offset_output.py:260-267, :329-333: save at validation intervals, restore from --ckpt_path after init).
    TF checkpoints themselves are not readable here (none ship with the reference); parameters can be imported
    from / exported to the reference's variable names with load_state_dict() / state_dict()."""

    def _ckpt_params(self):
        raise NotImplementedError

    def _core(self):
        raise NotImplementedError

    def load_state_dict(self, sd, reset_optimizer=True):
        """Import parameters under the reference's variable names into the tracker.  The core
        loads into its existing flat buffer when the layout is unchanged; if it had to re-allocate, the optimiser
        is re-bound to the new buffer.  Slots (ms = 1, mom = 0) and global_step restart unless reset_optimizer=False
        and the buffer was kept."""
        core = self._core()
        before = core.params
        core.load_state_dict(sd)
        if core.params is not before:
            o = self.opt
            self.opt = RMSPropClip(core.params, o.lr, o.decay, o.momentum, o.eps, o.clip)
        elif reset_optimizer:
            self.opt.ms.fill_(1.0)
            self.opt.mom.zero_()
            self.opt.global_step = 0

    def state_dict(self):
        return self._core().state_dict()

    def save_checkpoint(self, path):
        P = self._ckpt_params()
        torch.save({"format": "ntmtrack-ckpt-1", "kind": type(self).__name__, "numel": P.numel,
                    "params": P.flat.detach().cpu(), "ms": self.opt.ms.detach().cpu(), "mom": self.opt.mom.detach().cpu(),
                    "global_step": int(self.opt.global_step)}, path)
        return path

    def load_checkpoint(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        P = self._ckpt_params()
        if ck.get("format") != "ntmtrack-ckpt-1" or ck.get("kind") != type(self).__name__ or ck.get("numel") != P.numel:
            raise _lib.NtkError("checkpoint %s does not match this tracker (%s, %d parameters)" % (path, type(self).__name__, P.numel))
        P.flat.copy_(ck["params"].to(P.flat.device))
        self.opt.ms.copy_(ck["ms"].to(P.flat.device))
        self.opt.mom.copy_(ck["mom"].to(P.flat.device))
        self.opt.global_step = int(ck["global_step"])


class _TwoStreamPipeline(object):
    """Two-stage software pipeline shared by the trackers: the frozen VGG trunk of batch i+1 runs on its own
    HIP stream while the recurrent core's forward / BPTT / optimiser of batch i runs on a high-priority stream."""

    def add_pipeline(self):
        self._s_vgg = None
        self._s_ntm = None
        self._slots = []
        self._pending = []
        self._next_slot = 0
        self._proj_done = None      # event: input projection of the batch being trained is enqueued/done
        self.trunk_waits_for_projection = os.environ.get("NTK_TRUNK_WAITS_FOR_PROJECTION", "1") != "0"

    def _mark_projection(self):
        """Called by the core's forward pass right after the input-projection GEMM.  With
        `trunk_waits_for_projection` (the default; NTK_TRUNK_WAITS_FOR_PROJECTION=0 turns it off) the next trunk pass waits
        for this point, so the tail of the serial recurrent chain -- weight-gradient GEMMs, optimiser, gather, the input
        projection -- runs on an idle chip instead of queueing behind the first layers of the next trunk pass (stream
        priority does not help: a trunk workgroup holds its CU until it exits).  Which way this pays depends on which
        stream bounds the step: round 1 had it on; round 2 off (the trunk stream was the bound and the 0.5 ms it idled
        cost more than the projection gained: 64.62 -> 64.51 ms per step); since round 3's trunk kernel the recurrent
        chain is the bound and it is on again: configs[1] 58.14 -> 57.26 ms per step (the projection 2.1 -> 0.39 ms, the
        gather 0.43 -> 0.03); configs[2] 107.7 -> 107.6 with the four-wave trunk kernel but 110.8 -> 104.6 with the eight-wave one,
        which is what its tracker runs when the trunk waits (DNCOffsetTracker)."""
        if self._s_ntm is not None:
            self._proj_done = torch.cuda.Event()
            self._proj_done.record(torch.cuda.current_stream(self.device))

    #: True: the trunk pass and the core pass share ONE stream (strictly alternating).  Set by trackers whose core runs as a
    #: cooperative cluster launch over (nearly) every CU: such a launch cannot start until all its workgroups are resident,
    #: so a trunk pass beside it does not overlap anything -- its short workgroups keep taking the CUs the cluster is waiting
    #: for while the already resident cluster workgroups spin (measured at BASELINE configs[4]: 1 110 ms per step with two
    #: streams, trunk 136 ms + core 703 ms alone).
    serial_trunk = False

    def _streams(self):
        if self._s_vgg is None:
            self._s_ntm = torch.cuda.Stream(device=self.device, priority=-1)
            self._s_vgg = self._s_ntm if self.serial_trunk else torch.cuda.Stream(device=self.device)
        return self._s_vgg, self._s_ntm

    def submit_features(self, frames, beside="train"):
        """Enqueue the VGG trunk for `frames` on the feature stream (returns immediately).
        At most two submissions may be outstanding.  `beside`: what the core stream runs meanwhile, "train" (a training pass of the
        previous batch: the default) or "infer" (its forward only) -- the DNC tracker picks the trunk's form by it."""
        s_vgg, _ = self._streams()
        if len(self._pending) >= 2:
            raise _lib.NtkError("submit_features: two feature batches already outstanding")
        F = frames.shape[0]
        if not self._slots:
            alloc = torch.zeros if getattr(self, "features_roi", False) else torch.empty      # outside the window the map stays zero
            self._slots = [dict(buf=alloc((F, frames.shape[1] // 8, frames.shape[2] // 8, 512), device=self.device),
                                free=None) for _ in range(2)]
        slot = self._slots[self._next_slot]          # strict alternation: never the buffer the core pass may still be reading
        self._next_slot ^= 1
        s_vgg.wait_stream(torch.cuda.current_stream(self.device))       # frames were produced on the caller's stream
        if self._proj_done is not None:
            if self.trunk_waits_for_projection:
                s_vgg.wait_event(self._proj_done)                        # see _mark_projection (no-op if already passed)
            self._proj_done = None
        if slot["free"] is not None:
            s_vgg.wait_event(slot["free"])                               # core pass that last read this buffer is done
        with torch.cuda.stream(s_vgg):
            keep = getattr(self.vgg, "split3", False)
            self.vgg.split3 = keep and (beside != "train" or getattr(self, "pipeline_trunk_split3", True))    # (DNCOffsetTracker.__init__)
            try:
                self.vgg(frames, out=slot["buf"])
            finally:
                self.vgg.split3 = keep
            done = torch.cuda.Event()
            done.record(s_vgg)
        self._pending.append((slot, done))

    def train_on_submitted(self, gts0, offsets):
        """Core forward + BPTT + (all-reduce) + optimiser on the oldest submitted feature batch.
        Call it BEFORE submit_features() of the following batch: both only enqueue work, and the next trunk pass
        then starts right after this batch's input projection (see _mark_projection)."""
        _, s_ntm = self._streams()
        slot, done = self._pending.pop(0)
        s_ntm.wait_stream(torch.cuda.current_stream(self.device))
        s_ntm.wait_event(done)
        with torch.cuda.stream(s_ntm):
            loss, _pred = self.loss_and_grads(slot["buf"], gts0, offsets)
            parallel.allreduce_gradients(self._flat_grad())
            self.opt.step(loss)
            slot["free"] = torch.cuda.Event()
            slot["free"].record(s_ntm)
        return loss

    def check_step(self):
        """Call where the host reads a step's loss (it synchronises): raises NtkError when a cluster launch of this rank
        aborted since the last check (DNC.check_cluster: the sticky error word, read and cleared) or when an optimiser step
        was skipped because the all-reduced gradient was not finite -- which is how an abort on ANOTHER rank shows here.
        Training scripts exit non-zero on it; nothing was applied for the failed step(s)."""
        self.join()
        err = None
        core = getattr(self, "core", None)
        if core is not None and hasattr(core, "check_cluster"):
            try:
                core.check_cluster()
            except _lib.NtkError as e:
                err = e
        try:
            self.opt.check()
        except _lib.NtkError as e:
            err = err or e
        if err is not None:
            raise err

    def join(self):
        """Make the caller's stream wait for everything enqueued on the pipeline streams."""
        cur = torch.cuda.current_stream(self.device)
        if self._s_vgg is not None:
            cur.wait_stream(self._s_vgg)
            cur.wait_stream(self._s_ntm)


class NTMOffsetTracker(_TwoStreamPipeline, _Checkpointing):
    """VGG-16 conv4_3 + NTMCell offsets tracker, defaults from direct_offset_output.py:21-42."""

    def __init__(self, batch_size, sequence_length, vgg_weights=None, mem_size=128, mem_dim=20, hidden_size=200,
                 num_layers=1, read_head_size=4, write_head_size=1, write_first=False, init_scale=0.05,
                 learning_rate=1e-4, decay=0.95, momentum=0.9, max_gradient_norm=5.0, feature_channels=512,
                 device="cuda", seed=42, vgg_chunk_frames=1024, conv_dtype="f32", conv_algo=None, features_roi=False):
        self.B, self.T = int(batch_size), int(sequence_length)
        self.S = self.T * (NUM_FEATURES + 1)
        self.device = torch.device(device)
        self.vgg = VGG16Conv43(vgg_weights, device=self.device, chunk_frames=vgg_chunk_frames, dtype=conv_dtype, algo=conv_algo) if vgg_weights else None
        self.features_roi = bool(features_roi) and self.vgg is not None and conv_dtype == "f32" and self.vgg.algo == "winograd"
        if self.features_roi:          # conv4_3 only where extract_features reads it (GRID_START .. GRID_START + (GRID_N - 1) * GRID_STEP), whole 4x4 tiles
            lo = (GRID_START // 4) * 4
            hi = ((GRID_START + (GRID_N - 1) * GRID_STEP) // 4 + 1) * 4
            self.vgg.features_window = (lo, lo, hi, hi)
        self.cell = NTMCell(2, mem_size=mem_size, mem_dim=mem_dim, controller_hidden_size=hidden_size,
                            controller_num_layers=num_layers, write_head_size=write_head_size,
                            read_head_size=read_head_size, write_first=write_first,
                            input_dim=feature_channels + 2, device=self.device, init_scale=init_scale, seed=seed)
        self.opt = RMSPropClip(self.cell.params, learning_rate, decay, momentum, 1e-10, max_gradient_norm)
        self.add_pipeline()

    # ---- forward pieces
    def features(self, frames):
        if self.vgg is None:
            raise _lib.NtkError("tracker was built without VGG weights")
        return self.vgg(frames)

    def serialize(self, fmap, gts0):
        return gather_serialize(fmap, gts0, self.B, self.T, self.cell.input_ldx)

    def forward_features(self, fmap, gts0, record=False):
        X = self.serialize(fmap, gts0)
        st0 = self.cell.zero_state(self.B)
        logits, _outs, new, rec = self.cell.run_sequence(X, st0, record=record, want_outputs=False,
                                                         after_projection=self._mark_projection)
        return X, st0, logits, rec

    def infer(self, frames, gts0):
        """-> predicted offsets [B,T-1,2] (tanh of the logits at each frame's delimiter step)."""
        _X, _st0, logits, _ = self.forward_features(self.features(frames), gts0)
        offs = torch.zeros((self.B, self.T, 2), device=self.device)
        _loss, pred, _ = offset_loss(logits, offs, self.T, want_grad=False)
        return pred

    # ---- one optimiser step
    def loss_and_grads(self, fmap, gts0, offsets):
        X, st0, logits, rec = self.forward_features(fmap, gts0, record=True)
        loss, pred, dlogits = offset_loss(logits, offsets, self.T)
        g0 = self.cell.backward_sequence(X, st0, rec, dlogits)
        self.cell.init_state_backward(g0, self.B)
        return loss, pred

    def _flat_grad(self):
        return self.cell.params.grad

    def _ckpt_params(self):
        return self.cell.params

    def _core(self):
        return self.cell

    def train_step(self, frames, gts0, offsets):
        """VGG forward, NTM forward + BPTT, gradient all-reduce (if distributed), clip + RMSProp.
        Returns the (local) loss as a 1-element device tensor."""
        fmap = self.features(frames)
        loss, _pred = self.loss_and_grads(fmap, gts0, offsets)
        parallel.allreduce_gradients(self.cell.params.grad)
        self.opt.step(loss)
        return loss


class DNCOffsetTracker(_TwoStreamPipeline, _Checkpointing):
    """VGG-16 conv4_3 + DNC core offsets tracker (direct_offset_output_with_dnc.py:408-648), forward path:
    frames -> VGG -> 64-point gather + serialise -> time-major dynamic_rnn over dnc.DNC (clip_value 20)
    -> output gather at the delimiter steps -> tanh.  Defaults from :22-43 (mem 128x20, R4/W1, hidden 200)."""

    def __init__(self, batch_size, sequence_length, vgg_weights=None, mem_size=128, mem_dim=20, hidden_size=200,
                 read_head_size=4, write_head_size=1, clip_value=20, feature_channels=512, device="cuda", seed=42,
                 vgg_chunk_frames=1024, learning_rate=1e-4, optimizer_epsilon=1e-10, max_gradient_norm=50.0,
                 conv_dtype="f32", conv_algo=None, features_roi=False):
        from .dnc import DNC
        self.B, self.T = int(batch_size), int(sequence_length)
        self.S = self.T * (NUM_FEATURES + 1)
        self.device = torch.device(device)
        # The pipelined DNC training step is bound by its cluster kernels, which share the chip (and its clock) with the trunk pass of
        # the next batch: beside the split-form trunk (the fp16 matrix pipe at 1.6 - 1.75 GHz) they run 2 % slower than beside the Winograd
        # trunk, and the trunk's own time is hidden either way (configs[2]: 6 020 against 6 145 frames/s, same box).  A trunk pass that
        # runs ALONE (infer, train_step) or beside the forward only is 13 % shorter in the split form (pipelined DNC inference: 11 170
        # against 9 760 frames/s).  So, unless a form was asked for: the split form, except for the trunk passes submit_features() puts
        # beside a TRAINING pass.
        self.pipeline_trunk_split3 = not (conv_algo is None and "NTK_TRUNK_ALGO" not in os.environ)
        self.vgg = VGG16Conv43(vgg_weights, device=self.device, chunk_frames=vgg_chunk_frames, dtype=conv_dtype, algo=conv_algo) if vgg_weights else None
        self.features_roi = bool(features_roi) and self.vgg is not None and conv_dtype == "f32" and self.vgg.algo == "winograd"
        if self.features_roi:          # conv4_3 only where extract_features reads it (GRID_START .. GRID_START + (GRID_N - 1) * GRID_STEP), whole 4x4 tiles
            lo = (GRID_START // 4) * 4
            hi = ((GRID_START + (GRID_N - 1) * GRID_STEP) // 4 + 1) * 4
            self.vgg.features_window = (lo, lo, hi, hi)
        self.core = DNC({"memory_size": mem_size, "word_size": mem_dim, "num_reads": read_head_size,
                         "num_writes": write_head_size}, {"hidden_size": hidden_size}, 2, clip_value,
                        input_dim=feature_channels + 2, device=self.device, seed=seed)
        # _with_dnc.py:615-620: clip_by_global_norm(50), RMSPropOptimizer(lr, epsilon=1e-10) -> decay 0.9, momentum 0
        self.opt = RMSPropClip(self.core.params, learning_rate, 0.9, 0.0, optimizer_epsilon, max_gradient_norm)
        self.add_pipeline()
        # Which cluster form, and does the trunk run beside it?  A cooperative core over every CU takes the chip alone
        # (serial_trunk).  Where the memory-partitioned form has a compile-time instantiation at a cluster size that leaves
        # HALF the chip free (2 B k <= CUs), that form runs instead and the trunk pass of the next batch overlaps it on the
        # other half: BASELINE configs[2] (256 x 64, B 32) at k = 4 on 128 CUs: 124.1 -> 107.3 ms per step, although the core
        # alone is slower there (89 ms on half the chip vs 69 ms on all of it).
        L, c = _lib.lib(), self.core
        #: what the constructor chose for TRAINING steps with a trunk to overlap, (form, k) or None = the core's own automatic
        #: choice (bench.py prints it); infer() on a tracker used for nothing else may set core.cluster_form / cluster_k back
        #: to None: with no trunk pass beside it the whole-chip LDS-resident form is the faster core (69 vs 89 ms at configs[2])
        self.cluster_choice = None
        if self.vgg is not None and c.cluster_form is None and c.cluster_k is None and not os.environ.get("NTK_DNC_CLUSTER_FORM") \
                and not os.environ.get("NTK_DNC_CLUSTER_K"):
            for k in (4, 2):        # preference order: k = 4 is the measured one (configs[2]); first match wins
                if 2 * self.B * k <= L.ntk_cu_count() and L.ntk_dnc_mp_compiled_shape(c.N, c.W, c.R, c.Wn, c.hid, c.O, k) > 0:
                    c.cluster_form, c.cluster_k = "mp", k
                    self.cluster_choice = ("mp", k)
                    # on half a chip the trunk's two half-batch streams only fight each other: one stream
                    # (configs[2]: 111.4 -> 107.3 ms per step; four parts: 113.6)
                    if not os.environ.get("NTK_TRUNK_SPLIT"):
                        self.vgg.split_streams = 1
                    break
        if os.environ.get("NTK_DNC_SERIAL_TRUNK"):
            self.serial_trunk = os.environ["NTK_DNC_SERIAL_TRUNK"] != "0"
        else:
            plan = self.core._cluster_plan(self.B)
            self.serial_trunk = bool(plan) and 2 * self.B * plan[1] > L.ntk_cu_count()
        # The eight-wave form of the F(4x4) trunk kernel beside the HBM-streaming cluster kernels (configs[2], ms per step): with the
        # next trunk pass waiting for the input projection (the default) 104.6 against 107.8 with round 2's four-wave form; WITHOUT
        # that wait it loses what it gains alone (110.8 against 107.7: its workgroups then sit on the CUs the chain's tail is
        # queueing for), so a tracker whose trunk does not wait keeps the four-wave kernel.
        if self.vgg is not None and not self.serial_trunk and not self.trunk_waits_for_projection and self.vgg.wino_waves is None:
            self.vgg.wino_waves = 4

    def _flat_grad(self):
        return self.core.params.grad

    def _ckpt_params(self):
        return self.core.params

    def _core(self):
        return self.core

    def forward_features(self, fmap, gts0, record=False):
        """-> logits [B,S,2] (batch-major view of the time-major core output, _with_dnc.py:534-541)."""
        from .ntm import gemm_nt
        X = gather_serialize(fmap, gts0, self.B, self.T, self.core.ldx)
        xproj = gemm_nt(X.view(self.B * self.S, self.core.ldx), self.core.WxT)
        self._mark_projection()
        out_tm, state = self.core.run_projected(xproj, self.B, self.S, record=record)
        self._X = X
        return out_tm.transpose(0, 1).contiguous(), state

    def loss_and_grads(self, fmap, gts0, offsets):
        logits, _state = self.forward_features(fmap, gts0, record=True)
        loss, pred, dlogits = offset_loss(logits, offsets, self.T)
        self.core.backward_sequence(self._X, dlogits, unpack=False)      # the optimiser reads params.grad (packed layout)
        # a cluster launch that aborted (a hand-off timed out) must not feed the optimiser, on ANY rank: loss -> NaN and
        # gradient -> NaN on the device, without a synchronisation.  The NaN survives the SUM all-reduce, every rank's global
        # norm is NaN and RMSPropClip.step skips the update everywhere (and turns every rank's loss into NaN);
        # check_step() raises where the caller reads the loss
        self.core.guard(loss, self.core.params.grad)
        return loss, pred

    def train_step(self, frames, gts0, offsets):
        """VGG forward, DNC forward + BPTT, gradient all-reduce (if distributed), clip(50) + RMSProp."""
        if self.vgg is None:
            raise _lib.NtkError("tracker was built without VGG weights")
        loss, _ = self.loss_and_grads(self.vgg(frames), gts0, offsets)
        parallel.allreduce_gradients(self.core.params.grad)
        self.opt.step(loss)
        return loss

    def infer(self, frames, gts0):
        if self.vgg is None:
            raise _lib.NtkError("tracker was built without VGG weights")
        logits, _ = self.forward_features(self.vgg(frames), gts0)
        offs = torch.zeros((self.B, self.T, 2), device=self.device)
        _loss, pred, _ = offset_loss(logits, offs, self.T, want_grad=False)
        return pred

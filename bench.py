#!/usr/bin/env python
"""bench.py -- frames/sec of the VGG16 + NTM(128x20) offsets tracker training step.

Contract (see the task statement):  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
one rank per GPU over RCCL.  Rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM:
VGG-16 conv1_1..conv4_3 on B*T frames -> 64-point gather + serialise -> NTM forward
(T*65 steps) -> offsets loss -> BPTT -> gradient all-reduce (N > 1) -> clip + RMSProp.
Workload = BASELINE.json configs[1]: batch 32 sequences / GPU, seq_len 20, 224x224 frames,
NTMCell(128x20, hidden 200, 4 read / 1 write heads), fp32 (weak scaling: per-GPU work fixed).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

# HBM-side bytes of one VGG trunk pass over 640 frames (10 conv launches): rocprofv3 --pmc FETCH_SIZE (x2: gfx950
# counts 128-B requests at 64 B) + --pmc WRITE_SIZE, separate passes -- profiles/r01_vgg_trunk_hbm_traffic_pmc.csv
# (direct kernels) and profiles/r01_vgg_trunk_winograd_hbm_traffic_pmc.csv (default trunk).
# Algorithmic bytes (inputs + weights + outputs of the ten layers) are 4.563e10.
TRUNK_TRAFFIC_BYTES_640_FRAMES = {"direct": 5.4737e10 + 2.3121e10, "winograd2": 8.3752e10 + 2.3121e10, "winograd": 9.0467e10 + 2.3860e10,
                                  "split3": 5.4769e10 + 2.3734e10}
TRUNK_TRAFFIC_PROFILE = {"direct": "profiles/r01_vgg_trunk_hbm_traffic_pmc.csv", "winograd2": "profiles/r01_vgg_trunk_winograd_hbm_traffic_pmc.csv",
                         "winograd": "profiles/r04_vgg_trunk_blocked_hbm_traffic_pmc.csv",
                         "split3": "profiles/r04_vgg_trunk_split3_hbm_traffic_pmc.csv"}
# fraction of the direct-convolution multiplies the Winograd layers execute on the MFMA pipe: F(2x2,3x3) 16 per 2x2 tile
# where the direct form has 36; F(4x4,3x3) 36 per 4x4 tile where it has 144
WINO_EXECUTED_FRACTION = {"winograd": 36.0 / 144.0, "winograd2": 16.0 / 36.0}
NTM_FWD_TRAFFIC_BYTES_B32_S1300 = 1.448e8 + 1.036e9     # profiles/r04_ntm_seq_hbm_traffic_pmc.csv (FETCH_SIZE KB x 1024 x 2 + WRITE_SIZE KB x 1024)
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0     # dense fp16 / bf16 peak (v_mfma_f32_32x32x16_f16: 1024 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz)
# The split form (csrc/conv_bf16p.hip X3) executes THREE fp16 MFMA products per fp32 product of conv1_2 .. conv4_3 (conv1_1 stays on
# the fp32 pipe: 0.6 % of the flops)
SPLIT3_PRODUCTS = 3.0
HBM_PEAK_GBS = 8000.0


def log(msg):
    print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def synth_inputs(B, T, device, seed, first_sequence=0):
    """SURVEY 8(d) synthetic inputs: frames U[0,255) - VGG_MEAN; frame-0 heat-map =
    discrete_gauss((.5,.5),(8,8),1); offsets U(-.5,.5), frame 0 = 0.
    Every SEQUENCE of the global batch has its own generator, seeded by (seed, global sequence index): rank r of a
    data-parallel run, which owns sequences parallel.shard_range(world * B, r, world), generates exactly the rows a single
    process would hold at those indices."""
    mean = torch.tensor([123.68, 116.78, 103.94])
    frames = torch.empty((B * T, 224, 224, 3), dtype=torch.float32)
    offs = torch.empty((B, T, 2), dtype=torch.float32)
    for b in range(B):
        g = torch.Generator(device="cpu").manual_seed(seed * 1000003 + first_sequence + b)
        frames[b * T:(b + 1) * T] = torch.rand((T, 224, 224, 3), generator=g) * 255.0 - mean
        offs[b] = torch.rand((T, 2), generator=g) - 0.5
    from ntmtrack.geometry import discrete_gauss
    hm = discrete_gauss((.5, .5), (8, 8), 1.0)
    gts0 = torch.from_numpy(np.tile(hm.reshape(1, 64), (B, 1)).astype(np.float32))
    offs[:, 0, :] = 0
    return frames.to(device), gts0.to(device), offs.to(device)


def vgg_weights(seed):
    rng = np.random.default_rng(seed)
    from ntmtrack.vgg import VGG_LAYERS
    ws = {}
    for name, cin, cout, _ in VGG_LAYERS:
        ws[name] = ((rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32),
                    np.zeros(cout, np.float32))
    return ws


def host_cpu_info():
    """CPU model, sockets and physical cores of this box from /proc/cpuinfo."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name":
                    model = v
                elif k == "physical id":
                    phys = v
                elif k == "core id":
                    core = v
                elif not k and phys is not None:
                    cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return {"cpu_model": model, "sockets": len({p for p, _ in cores}) or None, "physical_cores": len(cores) or None,
            "logical_cpus": os.cpu_count()}


def _cgroup_cpu_quota():
    """CPUs' worth of time this process's cgroup may use (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p_ = f.read().split()[:2]
            if q != "max":
                return float(q) / float(p_)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p_ = float(f.read())
        if q > 0:
            return q / p_
    except (OSError, ValueError):
        pass
    return None


def _cpu_sample(ws, threads, n_vgg_frames, T, O, OT, model="ntm", dnc_shape=(256, 64)):
    torch.set_num_threads(threads)
    rng = np.random.default_rng(42)
    frames = (rng.uniform(0, 255, size=(n_vgg_frames, 224, 224, 3)).astype(np.float32) - O.VGG_MEAN)
    OT.vgg16_conv43(frames[:1], ws)            # warm the conv primitives
    t0 = time.perf_counter()
    OT.vgg16_conv43(frames, ws)
    t_vgg = (time.perf_counter() - t0) / n_vgg_frames
    feats = np.maximum(rng.standard_normal((1, T, 64, 512)), 0).astype(np.float32)
    x = O.serialize_inputs(feats, rng.uniform(0, 1, size=(1, T, 64)).astype(np.float32))
    offs = rng.uniform(-.5, .5, size=(1, T, 2)).astype(np.float32)
    if model == "dnc":
        # the DNC core of the benchmarked configuration: forward + BPTT of one sequence on the torch-CPU autograd restatement
        from oracle import dnc_oracle as D
        from oracle import dnc_oracle_torch as DTo
        cfg = D.DNCConfig(514, 2, memory_size=dnc_shape[0], word_size=dnc_shape[1], num_reads=4, num_writes=1, hidden_size=200, clip_value=20)
        params = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in D.init_params(cfg, rng).items()}
        xt = torch.tensor(np.ascontiguousarray(np.transpose(x, (1, 0, 2))), dtype=torch.float32)          # time-major
        t0 = time.perf_counter()
        ys, _ = DTo.run_model(cfg, params, xt)
        pred = torch.tanh(ys[65:].reshape(T - 1, 65, 1, 2)[:, 64])
        (0.5 * ((pred - torch.tensor(offs[:, 1:].transpose(1, 0, 2))) ** 2).sum()).backward()
        t_ntm = time.perf_counter() - t0
    else:
        cfg = O.NTMConfig(514, 2, mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=200,
                          controller_num_layers=1, write_head_size=1, read_head_size=4)
        params = O.init_params(cfg, rng)
        t0 = time.perf_counter()
        OT.loss_and_grads(cfg, params, x, offs, dtype=torch.float32)
        t_ntm = time.perf_counter() - t0
    log("cpu_baseline: %d thread(s): VGG %.3f s/frame, %s fwd+BPTT %.2f s/sequence" % (threads, t_vgg, model.upper(), t_ntm))
    return T / (T * t_vgg + t_ntm), t_vgg, t_ntm


def cpu_baseline(ws, n_vgg_frames=8, T=20, model="ntm", dnc_shape=(256, 64)):
    """The CPU restatement (oracle/, kind "port") timed on this box's host cores on a bounded sample of the same
    workload: VGG trunk on `n_vgg_frames` frames (torch-CPU conv2d, the op granularity TF-CPU would run) + NTM
    forward + BPTT of ONE sequence of T frames (torch-CPU autograd restatement), combined as frames/s of one
    sequence: T / (T * t_vgg_per_frame + t_ntm_per_sequence).  Run with all usable cores and once single-threaded."""
    from oracle import ntm_oracle as O
    from oracle import ntm_oracle_torch as OT
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = _cgroup_cpu_quota()
    # "all cores" = every core this process may really use: its affinity mask, cut by the cgroup's CPU quota where one is
    # set (threads beyond a quota only fight over it).  Where no quota is readable the thread count is CALIBRATED: the
    # trunk of one frame is timed with 16 threads (the box's CPU share for one GPU) and with the whole mask, and the
    # faster of the two is what the baseline runs with -- so an over-subscribed mask can never make the sample unbounded.
    cores = max(1, min(affinity, int(quota) if quota else affinity))
    calib = None
    if cores > 16 and not quota:
        frames1 = (np.random.default_rng(0).uniform(0, 255, size=(1, 224, 224, 3)).astype(np.float32) - O.VGG_MEAN)
        calib = {}
        for th in (16, cores):
            torch.set_num_threads(th)
            OT.vgg16_conv43(frames1, ws)
            t0 = time.perf_counter()
            OT.vgg16_conv43(frames1, ws)
            calib[th] = time.perf_counter() - t0
        log("cpu_baseline: calibration, VGG trunk of 1 frame: %s" % ", ".join("%d threads %.2f s" % kv for kv in calib.items()))
        if calib[16] <= calib[cores]:
            cores = 16
    fps, t_vgg, t_ntm = _cpu_sample(ws, cores, n_vgg_frames, T, O, OT, model, dnc_shape)
    fps1, t_vgg1, t_ntm1 = _cpu_sample(ws, 1, 2, T, O, OT, model, dnc_shape)
    torch.set_num_threads(min(cores, 16))
    out = {"value": round(fps, 3), "unit": "frames/sec", "cores": cores, "kind": "port",
           "sample": "VGG conv1_1..conv4_3 on %d frames (torch-CPU conv2d, %.3f s/frame) + %s fwd+BPTT of 1 "
                     "sequence x %d frames (torch-CPU autograd restatement, %.2f s); frames/s of one sequence"
                     % (n_vgg_frames, t_vgg, "NTM" if model == "ntm" else "DNC(%dx%d)" % dnc_shape, T, t_ntm),
           "threads_note": "threads = min(affinity mask %d, cgroup CPU quota %s)%s" % (
               affinity, ("%.1f" % quota) if quota else "none readable",
               "" if calib is None else "; calibrated on the trunk of one frame: " + ", ".join("%d threads %.2f s" % kv for kv in calib.items())),
           "single_thread": {"value": round(fps1, 3), "unit": "frames/sec", "vgg_s_per_frame": round(t_vgg1, 3),
                             "ntm_s_per_sequence": round(t_ntm1, 2)}}
    out.update(host_cpu_info())
    return out


# HBM-side bytes of the DNC cluster kernels at configs[2] (B 32, S 1300): profiles/r02_dnc_cluster_hbm_traffic_pmc.csv
DNC_FWD_TRAFFIC_BYTES_B32_S1300 = 4.56e9        # inference-mode forward (2 x FETCH_SIZE + WRITE_SIZE)
DNC_BWD_TRAFFIC_BYTES_B32_S1300 = 2.123e10
NTM_BWD_TRAFFIC_BYTES_B32_S1300 = 0.950e9 + 0.162e9   # profiles/r04_ntm_seq_hbm_traffic_pmc.csv
# HBM-side bytes per sequence-step of the memory-partitioned DNC cluster kernels at configs[4]'s shape (512 x 128, B 64):
# profiles/r03_dnc_mp_hbm_traffic_pmc.csv (2 x FETCH_SIZE + WRITE_SIZE over B 64 x S 200): the inference-mode forward moves
# 1.00 - 1.04x the algorithmic bytes (2 650 112 per sequence-step; two collections: 2 649 848 and 2 758 376 -- write-backs
# straddle kernel boundaries); BPTT reads L_t, L_{t-1} and d(link) and rewrites d(link)
DNC_MP_FWD_TRAFFIC_BYTES_PER_SEQ_STEP = 2758376.0
DNC_MP_BWD_TRAFFIC_BYTES_PER_SEQ_STEP = 5401689.0
# ... and at configs[2]'s shape (256 x 64, B 32, k = 4; same file, second block): the inference forward's 8 MB of link state
# stays in L2 / Infinity Cache, so fewer bytes than the algorithmic 669 696 reach HBM
DNC_MP_C3_FWD_TRAFFIC_BYTES_PER_SEQ_STEP = 311360.0
DNC_MP_C3_BWD_TRAFFIC_BYTES_PER_SEQ_STEP = 1254740.0


def _median_ms(fn, n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = []
    for _ in range(n):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    return float(np.median(times))


def memory_step_probe(trk, model, gts0, offs, B, T):
    """The memory cell's recurrent passes alone (outside the timed region): algorithmic state bytes per sequence-step
    (state read once + written once per step) x B x S / time, against the 8 TB/s HBM peak, for the forward
    pass and for BPTT (which touches the same state plus its gradient: priced with the same per-step bytes).
    The cells keep their state on-chip (NTM: LDS of one CU per sequence; DNC: LDS / registers of a cluster of 8 CUs per
    sequence) and the S steps of a sequence are strictly dependent, so these passes are bound by the per-step
    dependency chain (and, for the DNC, by the two hand-offs per step), not by HBM: the fraction is reported as SURVEY
    defines it, next to the HBM-side bytes the kernels really move (PMC) and the per-step latency."""
    from ntmtrack import tracker as T_
    S = T * 65
    fmap = trk._slots[0]["buf"]
    torch.cuda.synchronize()
    ms = _median_ms(lambda: trk.forward_features(fmap, gts0))
    if model == "ntm":
        d = trk.cell.dims
        per_step = 2 * d.N * d.Md * 4 + 2 * d.H * d.N * 4 + 2 * d.R * d.Md * 4 + d.P * 4
        X, st0, logits, rec = trk.forward_features(fmap, gts0, record=True)
        _loss, _pred, dlogits = T_.offset_loss(logits, offs, T)
        ms_b = _median_ms(lambda: trk.cell.backward_sequence(X, st0, rec, dlogits))
        kern, kern_b = "ntm_seq_fwd_ws_kernel (benchmark shape; ntm_seq_fwd_kernel otherwise)", "ntm_seq_bwd_kernel, wave-specialised form at the benchmark shape (+ 3 weight-gradient GEMMs)"
        traffic = NTM_FWD_TRAFFIC_BYTES_B32_S1300 * (B * S) / (32.0 * 1300.0)
        traffic_b = NTM_BWD_TRAFFIC_BYTES_B32_S1300 * (B * S) / (32.0 * 1300.0)
        note = "serialise + input projection + persistent sequence kernel, one workgroup per sequence; state is LDS resident; the h part of the recurrent weight stream runs on stream waves beside the step (DESIGN.md 4.2')"
        tnote = "PMC, profiles/r04_ntm_seq_hbm_traffic_pmc.csv: input projection read + per-step BPTT records; the memory state itself never leaves LDS"
    else:
        c = trk.core
        per_step = 2 * c.N * c.W * 4 + 2 * c.Wn * c.N * c.N * 4 + 2 * (c.R + c.Wn) * c.N * 4 + 2 * c.Wn * c.N * 4 + 2 * c.N * 4
        logits, _st = trk.forward_features(fmap, gts0, record=True)
        _loss, _pred, dlogits = T_.offset_loss(logits, offs, T)
        ms_b = _median_ms(lambda: c.backward_sequence(trk._X, dlogits))
        c.check_cluster()
        k, kb = getattr(c, "last_cluster_k", 1), getattr(c, "last_cluster_bwd_k", 1)
        form, form_b = getattr(c, "last_cluster_form", None), getattr(c, "last_cluster_bwd_form", None)
        fam = {"lds": "dnc_cluster_%s_kernel", "mp": "dnc_mp_%s_kernel"}
        kern = (fam[form] % "fwd" + " (k = %d workgroups per sequence)" % k) if k > 1 else "dnc_seq_fwd_kernel"
        kern_b = ((fam[form_b] % "bwd" + " (k = %d)" % kb) if kb > 1 else "dnc_seq_bwd_kernel") + " (+ 4 weight-gradient GEMMs)"
        if getattr(c, "last_segments", None):
            kern_b += "; %d BPTT segments: this time includes re-recording all but the last two (recorded by the forward pass) with the forward kernel" % len(c.last_segments[1])
        is_c3 = (c.N, c.W, c.R) == (256, 64, 4) and form == "lds"
        is_c5 = (c.N, c.W, c.R) == (512, 128, 4) and form == "mp"
        is_c3mp = (c.N, c.W, c.R) == (256, 64, 4) and form == "mp" and k == 4
        traffic = DNC_FWD_TRAFFIC_BYTES_B32_S1300 * (B * S) / (32.0 * 1300.0) if is_c3 else (
            DNC_MP_FWD_TRAFFIC_BYTES_PER_SEQ_STEP * B * S if is_c5 else (DNC_MP_C3_FWD_TRAFFIC_BYTES_PER_SEQ_STEP * B * S if is_c3mp else None))
        traffic_b = DNC_BWD_TRAFFIC_BYTES_B32_S1300 * (B * S) / (32.0 * 1300.0) if is_c3 else (
            DNC_MP_BWD_TRAFFIC_BYTES_PER_SEQ_STEP * B * S if is_c5 else (DNC_MP_C3_BWD_TRAFFIC_BYTES_PER_SEQ_STEP * B * S if is_c3mp else None))
        note = {"lds": "serialise + input projection + persistent cluster kernel: link rows and memory LDS resident, two mailbox hand-offs per step",
                "mp": "serialise + input projection + persistent memory-partitioned cluster kernel: the link streams through HBM once per step "
                      "(N/k rows per workgroup), memory rows LDS resident, four mailbox hand-offs per step; HBM-bound link pass",
                None: "serialise + input projection + persistent sequence kernel, one workgroup per sequence; link and memory L2 resident"}[form]
        tnote = ("PMC, profiles/r03_dnc_mp_hbm_traffic_pmc.csv (2 x FETCH_SIZE + WRITE_SIZE), inference-mode forward / BPTT kernel alone" if (is_c5 or is_c3mp)
                 else "PMC, profiles/r02_dnc_cluster_hbm_traffic_pmc.csv (2 x FETCH_SIZE + WRITE_SIZE)")
        if k > 1:
            pl = c.cluster_placement()
            note += "; clusters on one XCD handing off through that XCD's L2 (forward, BPTT launch): " + ", ".join("%d of %d" % x for x in pl)

    def entry(kernel, t_ms, tr):
        gbps = per_step * B * S / (t_ms * 1e-3) / 1e9
        return {"kernel": kernel, "bound": "hbm" if (model == "dnc" and getattr(trk.core, "last_cluster_form", None) == "mp") else "hbm (nominal); dependency-chain latency (actual)",
                "algorithmic_bytes_per_sequence_step": per_step, "sequences": B, "steps": S,
                "ms": round(t_ms, 3), "us_per_step": round(t_ms * 1e3 / S, 3),
                "achieved": round(gbps, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBS, 5),
                "traffic": tr, "traffic_note": tnote}
    fwd = entry(kern, ms, traffic)
    fwd["forward_ms"] = fwd["ms"]
    fwd["note"] = note
    return fwd, entry(kern_b, ms_b, traffic_b)


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks ourselves, as a CHILD process
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same
    arguments>`), pass its stdout (rank 0's one JSON line) and stderr through, and return its exit code.  This process has
    made no GPU call at this point (importing torch does not initialise HIP) and never replaces itself: no os.exec*."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("--gpus %d without WORLD_SIZE: launching %d ranks as a child process: %s" % (n, n, " ".join(cmd)))
    return subprocess.call(cmd, env=env)


def trunk_alone_probe(trk, frames):
    """The trunk pass alone on an idle device (outside the timed region; HIP events on the current stream, median of 3)."""
    torch.cuda.synchronize()
    return _median_ms(lambda: trk.vgg(frames, out=trk._slots[0]["buf"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--features-roi", action="store_true",
                    help="compute conv4_3 only in the 25 of 49 output tiles extract_features reads (NOT the reference's graph, "
                         "which computes the whole map: an optional optimisation, never the headline number)")
    ap.add_argument("--batch", type=int, default=32, help="sequences per GPU")
    ap.add_argument("--seq-len", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--model", default="ntm", choices=["ntm", "dnc"],
                    help="ntm = BASELINE configs[1] (the headline metric); dnc = configs[2] (DNC 256x64, 4 read heads), reported for reference")
    ap.add_argument("--wino-waves", type=int, default=0, choices=[0, 4, 8], help="form of the F(4x4) kernel (0 = library default = 8)")
    ap.add_argument("--conv-algo", default=None, choices=["split3", "winograd", "winograd2", "direct"],
                    help="fp32 trunk form (default: the tracker's own -- the split form for the NTM tracker, F(4x4) Winograd for the DNC "
                         "tracker): split3 = three fp16 MFMA products per fp32 product, winograd / winograd2 = fused F(4x4) / F(2x2) on the "
                         "fp32 MFMA pipe, direct = the implicit-GEMM kernel")
    ap.add_argument("--conv-dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = exact fp32 MFMA (configs 2-4, the headline); bf16 = bf16 operands / fp32 accumulate (config 5)")
    ap.add_argument("--mem-size", type=int, default=None)
    ap.add_argument("--mem-dim", type=int, default=None)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print("bench.py: --gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world),
              file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU; NTK_DIST_BACKEND=gloo (rehearsal on a box with fewer GPUs than ranks) lets ranks share a device
    backend = os.environ.get("NTK_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ntmtrack import tracker
    from ntmtrack.vgg import conv_flops_per_frame

    B, T = args.batch, args.seq_len
    ws = vgg_weights(42)
    if args.model == "dnc":
        trk = tracker.DNCOffsetTracker(B, T, vgg_weights=ws, device=dev, seed=42, mem_size=args.mem_size or 256,
                                       mem_dim=args.mem_dim or 64, conv_dtype=args.conv_dtype, conv_algo=args.conv_algo)
    else:
        trk = tracker.NTMOffsetTracker(B, T, vgg_weights=ws, device=dev, seed=42, conv_dtype=args.conv_dtype, conv_algo=args.conv_algo,
                                       features_roi=args.features_roi)   # same init on every rank
    if args.wino_waves:
        trk.vgg.wino_waves = args.wino_waves
    if args.conv_algo is None:                                      # what the tracker's pipeline runs (reported below)
        pipe_split3 = getattr(trk.vgg, "split3", False) and (args.mode != "train" or getattr(trk, "pipeline_trunk_split3", True))
        args.conv_algo = "split3" if pipe_split3 else trk.vgg.algo
    trk.vgg.split3 = getattr(trk.vgg, "split3", False) and args.conv_algo == "split3"     # the probes outside the pipeline run the same form
    log("tracker built; generating synthetic inputs")
    from ntmtrack import parallel
    lo, hi = parallel.shard_range(world * B, rank, world)          # this rank's sequences of the global batch
    assert hi - lo == B
    frames, gts0, offs = synth_inputs(B, T, dev, 42, first_sequence=lo)
    log("inputs resident in HBM: frames %s" % (tuple(frames.shape),))

    ev = lambda: torch.cuda.Event(enable_timing=True)
    marks_vgg, marks_ntm = [], []
    s_vgg, s_ntm = trk._streams()

    def submit(timed):
        """VGG trunk of one batch on the feature stream (events recorded on THAT stream)."""
        if timed:
            e0, e1 = ev(), ev()
            with torch.cuda.stream(s_vgg):
                e0.record()
            trk.submit_features(frames, beside=args.mode)
            with torch.cuda.stream(s_vgg):
                e1.record()
            marks_vgg.append((e0, e1))
        else:
            trk.submit_features(frames, beside=args.mode)

    def consume(timed):
        """NTM forward + BPTT + all-reduce + optimiser of the oldest submitted batch."""
        if args.mode == "train":
            if timed:
                e0, e1 = ev(), ev()
                with torch.cuda.stream(s_ntm):
                    e0.record()
                trk.train_on_submitted(gts0, offs)
                with torch.cuda.stream(s_ntm):
                    e1.record()
                marks_ntm.append((e0, e1))
            else:
                trk.train_on_submitted(gts0, offs)
        else:
            slot, done = trk._pending.pop(0)
            s_ntm.wait_event(done)
            with torch.cuda.stream(s_ntm):
                trk.forward_features(slot["buf"], gts0)
                slot["free"] = torch.cuda.Event()
                slot["free"].record(s_ntm)

    def run(k, timed):
        # K steps = K VGG passes + K NTM passes; VGG(i+1) is in flight while NTM(i) runs
        submit(timed)
        for i in range(k):
            consume(timed)              # enqueue the core pass of batch i (waits for its features on the GPU) ...
            if i + 1 < k:
                submit(timed)           # ... then the trunk of batch i+1, which starts after batch i's input projection
        trk.join()

    if args.warmup > 0:
        run(args.warmup, False)
        torch.cuda.synchronize()
        log("warmup (%d steps) done" % args.warmup)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    run(args.steps, True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if args.mode == "train":
        trk.check_step()        # raises (non-zero exit) if a cluster launch aborted or an optimiser step was skipped on a non-finite gradient
    log("timed %d steps in %.3f s; peak device memory allocated %.1f GB, reserved %.1f GB" % (
        args.steps, elapsed, torch.cuda.max_memory_allocated(dev) / 1e9, torch.cuda.max_memory_reserved(dev) / 1e9))
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        frames_total = world * B * T * args.steps
        vgg_ms = float(np.mean([a.elapsed_time(b) for a, b in marks_vgg]))
        ntm_ms = float(np.mean([a.elapsed_time(b) for a, b in marks_ntm])) if marks_ntm else 0.0
        ends = [m[1] for m in marks_ntm]
        steady_ms = round(float(np.median([ends[i].elapsed_time(ends[i + 1]) for i in range(len(ends) - 1)])), 3) if len(ends) >= 4 else None
        if len(ends) >= 2:
            log("core pass end-to-end intervals (ms): %s; first core pass %.2f ms, first trunk pass %.2f ms" % (
                " ".join("%.2f" % ends[i].elapsed_time(ends[i + 1]) for i in range(len(ends) - 1)),
                marks_ntm[0][0].elapsed_time(marks_ntm[0][1]), marks_vgg[0][0].elapsed_time(marks_vgg[0][1])))
        flops = conv_flops_per_frame() * B * T
        algorithmic = flops / (vgg_ms * 1e-3) / 1e12
        # executed MFMA flops: conv1_1 runs the direct kernel; the nine Winograd layers execute a fixed fraction of the
        # direct form's multiplies (WINO_EXECUTED_FRACTION)
        c11 = 2 * 224 * 224 * 9 * 3 * 64 * B * T
        wino = args.conv_algo in WINO_EXECUTED_FRACTION and args.conv_dtype == "f32"
        if args.features_roi and getattr(trk, "features_roi", False):      # conv4_3 in 25 of its 49 tiles only
            flops -= 2 * 28 * 28 * 9 * 512 * 512 * B * T * (24.0 / 49.0)
        executed_flops = (c11 + (flops - c11) * WINO_EXECUTED_FRACTION[args.conv_algo]) if wino else flops
        split3 = args.conv_dtype == "f32" and bool(getattr(trk.vgg, "split3", False)) and trk.vgg.split3_trunk_supported(frames.shape)
        if split3:
            executed_flops = (flops - c11) * SPLIT3_PRODUCTS                # on the fp16 pipe (conv1_1's fp32 MFMAs are not counted)
        achieved = executed_flops / (vgg_ms * 1e-3) / 1e12
        PEAK = (F16_MFMA_PEAK_TFLOPS if split3 else FP32_MFMA_PEAK_TFLOPS) if args.conv_dtype == "f32" else 2500.0     # dense bf16 MFMA peak (MI355X_MICROARCH.md)
        out = {
            "metric": ("frames/sec (whole node) VGG16+NTM(128x20) seq_len=%d" % T) if args.model == "ntm" else
                      ("frames/sec (whole node) VGG16+DNC(%dx%d) seq_len=%d" % (trk.core.N, trk.core.W, T)),
            "value": round(frames_total / elapsed, 2), "unit": "frames/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.conv_dtype == "f32" else "bf16 conv operands / f32 accumulate, f32 memory cell",
            **({"dtype_note": "fp32 values and fp32 accumulators everywhere; in conv1_2 .. conv4_3 each fp32 operand is carried as two fp16 numbers "
                              "(hi + lo = the value to 2^-22) and each product is three fp16 MFMA products (2^-20 per product): measured against "
                              "float64, 0.7e-6 - 1.7e-6 of the activation scale per layer and 3.0e-6 through the trunk (the fp32 F(4x4) Winograd "
                              "trunk: 6e-6 - 1.5e-5 and 3.5e-6; north_star allows 1e-4); DESIGN.md 4.0''"} if split3 else {}),
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: VGG-16 conv1_1..conv4_3 + NTMCell(128x20, hidden 200, R4/W1) "
                                    "direct_offset_output %s step, batch %d sequences/GPU, seq_len %d, 224x224 frames"
                                    % ("training" if args.mode == "train" else "inference", B, T)) if args.model == "ntm" else
                                   ("BASELINE configs[%d]: VGG-16 conv1_1..conv4_3 + DNC core (mem %dx%d, 4 read heads, hidden 200, clip 20) "
                                    "direct_offset_output_with_dnc %s step, batch %d sequences/GPU, seq_len %d"
                                    % (4 if trk.core.N >= 512 else 2, trk.core.N, trk.core.W, "training" if args.mode == "train" else "inference", B, T)),
                       "global_batch": world * B, "seq_len": T, "steps_per_sequence": T * 65,
                       "parallelism": "dp%d" % world, "mode": args.mode,
                       **({"features_roi": "conv4_3 computed in the 25 of 49 output tiles extract_features reads (optional; the flops "
                                           "in `roofline` are reduced accordingly; NOT the headline configuration)"}
                          if (args.features_roi and getattr(trk, "features_roi", False)) else {})},
            "roofline": {"bound": "mfma", "kernel": ("conv3x3_relu_bf16p_kernel<X3> (VGG trunk: conv1_1 on the fp32 pipe + conv1_2 .. conv4_3 in the split form)" if split3 else
                                                     ((("conv3x3_wino43d_kernel" if trk.vgg.wino_waves in (None, 8) else "conv3x3_wino43_kernel (four-wave form)")
                                                      + " (VGG trunk: conv1_1 direct + 9 fused Winograd F(4x4,3x3) layers; a 1x1x32-block layer"
                                                        " whose blocks span more than 16 MB of input falls back to the four-wave form)"
                                                      if args.conv_algo == "winograd" else
                                                      ("conv3x3_wino_kernel (VGG trunk: conv1_1 direct + 9 fused Winograd F(2x2,3x3) layers)"
                                                       if args.conv_algo == "winograd2" else "conv3x3_relu_dma_kernel (VGG trunk, 10 layers)"))
                                                     if args.conv_dtype == "f32" else "conv3x3_relu_bf16_kernel (VGG trunk, 10 layers)")),
                         "achieved": round(achieved, 2), "peak": PEAK, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK, 4),
                         "frac_algorithmic": round(algorithmic / PEAK, 4),
                         "algorithmic_over_fp32_mfma_peak": round(algorithmic / FP32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": (TRUNK_TRAFFIC_BYTES_640_FRAMES[args.conv_algo] * (B * T) / 640.0)
                         if (args.conv_dtype == "f32" and TRUNK_TRAFFIC_BYTES_640_FRAMES.get(args.conv_algo)) else None,
                         "traffic_note": "HBM-side bytes per trunk pass from PMC FETCH_SIZE*2+WRITE_SIZE (%s), scaled by frames/640; algorithmic 4.563e10 B per 640 frames" % TRUNK_TRAFFIC_PROFILE.get(args.conv_algo),
                         **({"pipe_busy_pmc": [0.61, 0.83], "clock_ghz_pmc": [1.55, 1.85],
                             "pmc_note": "profiles/r04_split3_pmc.txt (a separate --pmc pass over the nine split-form layers, 640 frames each): share of SIMD cycles "
                                         "with the fp16 MFMA pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles; conv2_1 0.61, conv1_2 0.68, the 128- to 512-channel "
                                         "layers 0.77 - 0.83) and the clock the chip holds in these kernels (GRBM_GUI_ACTIVE / 8 / duration; 2.4 GHz nominal): "
                                         "frac = busy x clock / 2.4 within a few per cent"} if split3 else {}),
                         "algorithmic_flops_per_frame": conv_flops_per_frame(),
                         "executed_flops_per_frame": executed_flops / (B * T),
                         "algorithmic_tflops": round(algorithmic, 2),
                         "note": ("achieved / frac = EXECUTED MFMA flops (conv1_1 direct + %.4f of the direct-convolution count for the "
                                  "nine Winograd layers) / trunk time measured with HIP events on the trunk's stream inside the timed "
                                  "region; algorithmic_tflops / frac_algorithmic = the direct-convolution count (SURVEY 8d: 27.92 GFLOP/frame) / "
                                  "the same time (/ peak: above 1 because F(4x4,3x3) executes a quarter of the direct form's multiplies, "
                                  "not because work is skipped)"
                                  "; the trunk pass runs as %d stream part(s): with 2, kernel durations in a rocprof --stats summary overlap "
                                  "pairwise (scripts/trace_union.py gives the union of their intervals per pass)")
                                 % (WINO_EXECUTED_FRACTION[args.conv_algo], getattr(trk.vgg, "split_streams", 1))
                                 if wino else
                                 ("achieved / frac = EXECUTED fp16 MFMA flops (3 x the direct-convolution count of conv1_2 .. conv4_3) / trunk time "
                                  "measured with HIP events on the trunk's stream inside the timed region, against the dense fp16 peak at 2.4 GHz "
                                  "(the chip holds 1.5 - 1.75 GHz in these kernels and their MFMA pipe is busy 63 - 84 %% of the cycles: "
                                  "profiles/r04_split3_pmc.txt); algorithmic_tflops = the direct-convolution count (SURVEY 8d: 27.92 GFLOP/frame) / the "
                                  "same time, algorithmic_over_fp32_mfma_peak = that against the fp32 pipe's 157.3 TFLOP/s (the peak rounds 1 - 3 priced "
                                  "the trunk against); the trunk pass runs as %d stream part(s)" % getattr(trk.vgg, "split_streams", 1))
                                 if split3 else "direct convolution: executed = algorithmic flops"},
            "breakdown_ms": {"vgg_trunk_stream": round(vgg_ms, 3), "ntm_fwd_bwd_opt_stream": round(ntm_ms, 3),
                             "steady_state_step": steady_ms,
                             "note": "two HIP streams: VGG(i+1) overlaps NTM(i); per-stream event times; steady_state_step = median interval "
                                     "between the ends of consecutive core passes (ms_per_step also carries the pipeline's fill and drain: "
                                     "the first trunk pass and the last core pass of the K timed steps run alone)"},
        }
        if args.model == "dnc":
            ch = getattr(trk, "cluster_choice", None)
            out["config"]["cluster_choice"] = ("%s, k = %d (constructor's choice for a step with a trunk to overlap)" % ch) if ch else "core's automatic choice"
        out["memory_step"], out["memory_step_bptt"] = memory_step_probe(trk, args.model, gts0, offs, B, T)
        # The step is bound by total CU-time once both streams are busy: the trunk's workgroups fill whatever CUs the
        # persistent core workgroups (one per sequence; B * k for the DNC cluster forms) do not hold.
        from ntmtrack import _lib as L_
        cus = int(L_.lib().ntk_cu_count())
        trunk_alone = trunk_alone_probe(trk, frames)
        core_wgs = B * (getattr(trk.core, "last_cluster_k", 1) if args.model == "dnc" else 1)
        # the persistent recurrent kernels alone (the probes above): what the core's workgroups hold their CUs for; the per-stream
        # event time `ntm_fwd_bwd_opt_stream` also contains the core stream's wait for its features when the trunk is the bound
        core_ms = out["memory_step"]["ms"] + (out["memory_step_bptt"]["ms"] if args.mode == "train" else 0.0)
        cu_trunk, cu_core = trunk_alone * 1e-3 * cus, core_ms * 1e-3 * min(core_wgs, cus)
        serial = bool(getattr(trk, "serial_trunk", False))          # trunk and core on ONE stream (a cooperative core over every CU): nothing overlaps
        out["breakdown_ms"]["trunk_alone"] = round(trunk_alone, 3)
        out["breakdown_ms"]["cu_seconds"] = {
            "trunk": round(cu_trunk, 3), "core": round(cu_core, 3), "compute_units": cus, "core_workgroups": core_wgs,
            "cu_time_bound_ms": None if serial else round((cu_trunk + cu_core) / cus * 1e3, 3),
            "schedule": "serial: trunk pass and core pass alternate on one stream" if serial else "overlapped: the trunk pass of batch i + 1 runs beside the core pass of batch i",
            "note": "trunk = the trunk pass alone on the idle device x every CU (its grids fill the chip); core = the recurrent forward + "
                    "BPTT kernels alone (memory_step.ms + memory_step_bptt.ms) x the CUs their persistent workgroups hold; "
                    "cu_time_bound_ms = (trunk + core) / CUs: what the step cannot beat while both streams overlap"}
        if not args.no_cpu_baseline and world == 1:
            # the DNC restatement costs ~15 ms per step on the host: a 4-frame sequence keeps the sample inside its time budget
            out["cpu_baseline"] = cpu_baseline(ws, T=20 if args.model == "ntm" else 4, model=args.model,
                                               dnc_shape=(args.mem_size or 256, args.mem_dim or 64))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

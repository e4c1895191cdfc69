"""Dev check: the memory-partitioned cluster kernels (ntk_dnc_mp_*) against the one-workgroup-per-sequence kernels
(ntk_dnc_seq_*) on the same inputs: outputs, final state and every BPTT record (forward), every gradient (BPTT)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ntmtrack.dnc import DNC

dev = torch.device("cuda:0")
SHAPES = [  # N, W, R, hid, B, S, k
    (64, 16, 2, 32, 2, 5, 2), (64, 16, 2, 32, 2, 5, 4), (64, 16, 2, 32, 3, 5, 8), (128, 32, 4, 64, 8, 6, 4),
    (256, 64, 4, 200, 8, 8, 4), (256, 64, 4, 200, 8, 8, 2), (512, 128, 4, 200, 8, 6, 4), (512, 128, 4, 200, 2, 40, 4),
]
if any("," in a for a in sys.argv[1:]):
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if "," in a]
do_bwd = "--bwd" in sys.argv or os.environ.get("MP_BWD") == "1"


def run(core, x, k, form, bwd):
    core.cluster_k, core.cluster_form = k, form
    core._cluster = core._cluster_b = None
    out, st = core.run_sequence(x, None, record=True)
    torch.cuda.synchronize()
    if k:
        core.check_cluster()
    rec = {kk: v.clone() for kk, v in core.last_record.items()}
    grads = None
    if bwd:
        g = torch.Generator().manual_seed(3)
        dout = torch.randn(tuple(out.shape), generator=g).to(dev).transpose(0, 1).contiguous()
        grads = core.backward_sequence(core.last_X, dout)
        torch.cuda.synchronize()
        if k:
            core.check_cluster()
        grads = {kk: v.clone() for kk, v in grads.items()}
    return out, st, rec, grads


worst = 0.0
for (N, W, R, hid, B, S, k) in SHAPES:
    torch.manual_seed(1)
    core = DNC({"memory_size": N, "word_size": W, "num_reads": R, "num_writes": 1}, {"hidden_size": hid}, 2, 20.0, input_dim=20, device=dev, seed=4)
    sd = core.state_dict()
    g = torch.Generator().manual_seed(2)
    for kk in sd:
        if kk.startswith("memory_access/") and kk.endswith("/w"):
            sd[kk] = sd[kk] * 6
        if kk.endswith("/b") or kk.endswith("b_gates"):
            sd[kk] = (torch.rand(sd[kk].shape, generator=g) - 0.5) * 0.6
    core.load_state_dict(sd)
    x = torch.randn((S, B, 20), generator=g).to(dev)
    o0, s0, r0, g0 = run(core, x, 0, None, do_bwd)
    t0 = time.time()
    o1, s1, r1, g1 = run(core, x, k, "mp", do_bwd)
    assert core.last_cluster_form == "mp" and core.last_cluster_k == k, (core.last_cluster_form, core.last_cluster_k)
    errs = {"out": float((o0 - o1).abs().max())}
    a0, a1 = s0.access_state, s1.access_state
    for nm, u, v in (("mem", a0.memory, a1.memory), ("rw", a0.read_weights, a1.read_weights), ("ww", a0.write_weights, a1.write_weights),
                     ("link", a0.linkage.link, a1.linkage.link), ("prec", a0.linkage.precedence_weights, a1.linkage.precedence_weights),
                     ("usage", a0.usage, a1.usage), ("reads", s0.access_output, s1.access_output),
                     ("h", s0.controller_state.hidden, s1.controller_state.hidden), ("c", s0.controller_state.cell, s1.controller_state.cell)):
        errs[nm] = float((u - v).abs().max())
    for kk in r0:
        errs["rec_" + kk] = float((r0[kk] - r1[kk]).abs().max())
    if do_bwd:
        for kk in g0:
            sc = float(g0[kk].abs().max()) + 1e-12
            errs["g_" + kk] = float((g0[kk] - g1[kk]).abs().max()) / sc
    bad = {kk: v for kk, v in errs.items() if not (v < (2e-3 if kk.startswith("g_") else 5e-6))}
    m = max(errs.values())
    worst = max(worst, m)
    print("N%d W%d R%d hid%d B%d S%d k%d: max err %.3g %s" % (N, W, R, hid, B, S, k, m, ("BAD " + str(bad)) if bad else "ok"), flush=True)
    if bad and "rec_al" in bad:
        d = (r0["al"] - r1["al"]).abs().amax(dim=(2, 3))          # [B,S]
        for bb in range(B):
            ts = torch.nonzero(d[bb] > 1e-4).flatten().tolist()
            if ts:
                t = ts[0]
                du = (r0["u"][bb, t] - r1["u"][bb, t]).abs()
                u0 = r0["u"][bb, t]
                srt = torch.sort(u0).values
                gaps = (srt[1:] - srt[:-1])
                print("  seq %d first bad step %d: max |du| %.3g, smallest usage gaps %s, n zero gaps %d, bad slots %s" % (
                    bb, t, float(du.max()), [float(v) for v in torch.sort(gaps).values[:4]], int((gaps == 0).sum()),
                    torch.nonzero((r0["al"][bb, t, 0] - r1["al"][bb, t, 0]).abs() > 1e-4).flatten().tolist()[:8]))
                if t > 0:
                    for nm in ("rw", "ww", "u"):
                        print("    step %d rec_%s max diff %.3g" % (t - 1, nm, float((r0[nm][bb, t - 1] - r1[nm][bb, t - 1]).abs().max())))
print("worst", worst)

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dnc_oracle as D
from ntmtrack.dnc import DNC
cuda = torch.device("cuda:0")
Din, O, N, W, R, Wn, hid, clip, B = 10, 3, 16, 8, 2, 3, 16, 20.0, 2
cfg = D.DNCConfig(Din, O, memory_size=N, word_size=W, num_reads=R, num_writes=Wn, hidden_size=hid, clip_value=clip)
rng = np.random.default_rng(5)
p = D.init_params(cfg, rng)
for k in p:
    if k.endswith("/b") or k.endswith("b_gates"): p[k] = rng.uniform(-0.3, 0.3, size=p[k].shape).astype(np.float32)
    if k.startswith("memory_access/") and k.endswith("/w"): p[k] = (p[k] * 6).astype(np.float32)
core = DNC({"memory_size": N, "word_size": W, "num_reads": R, "num_writes": Wn}, {"hidden_size": hid}, O, clip, device=cuda)
core.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
for S in (1, 2, 3):
    x = np.random.default_rng(9).standard_normal((S, B, Din)).astype(np.float32)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    ys, fin = D.run_model(cfg, p64, x.astype(np.float64))
    out, st = core.run_sequence(torch.from_numpy(x).to(cuda)); torch.cuda.synchronize()
    acc = st.access_state
    d = lambda a, b: float(np.abs(a.cpu().numpy() - b).max())
    print("S=%d out %.2e mem %.2e usage %.2e ww %.2e rw %.2e link %.2e prec %.2e reads %.2e h %.2e c %.2e" % (
        S, d(out, ys), d(acc.memory, fin.access_state.memory), d(acc.usage, fin.access_state.usage),
        d(acc.write_weights, fin.access_state.write_weights), d(acc.read_weights, fin.access_state.read_weights),
        d(acc.linkage.link, fin.access_state.linkage.link), d(acc.linkage.precedence_weights, fin.access_state.linkage.precedence_weights),
        d(st.access_output, fin.access_output), d(st.controller_state.hidden, fin.controller_state.hidden), d(st.controller_state.cell, fin.controller_state.cell)))
    if S == 1:
        print("ww gpu", acc.write_weights.cpu().numpy()[0, :, :6]); print("ww ref", fin.access_state.write_weights[0, :, :6])
        print("mem rows diff", np.abs(acc.memory.cpu().numpy() - fin.access_state.memory).max(axis=2))

# deeper: step 2 pieces from the oracle
S = 2
x = np.random.default_rng(9).standard_normal((S, B, Din)).astype(np.float32)
st = D.dnc_initial_state(cfg, B, np.float64)
y, st1, inp1 = D.dnc_step(cfg, p64, x[0].astype(np.float64), st)
a = cfg.access
B_ = B
ci = np.concatenate([x[1].astype(np.float64), st1.access_output.reshape(B_, -1)], axis=1)
h, cs = D.sonnet_lstm(ci, st1.controller_state, p64["lstm/w_gates"], p64["lstm/b_gates"])
h = D.clip(cfg, h)
inp = D.read_inputs(a, p64, h)
usage = D.freeness(st1.access_state.write_weights, inp["free_gate"], st1.access_state.read_weights, st1.access_state.usage)
cw = D.cosine_weights(st1.access_state.memory, inp["write_content_keys"], inp["write_content_strengths"])
aw = D.write_allocation_weights(usage, inp["allocation_gate"] * inp["write_gate"], a.Wn)
cw0 = D.cosine_weights(np.zeros_like(st1.access_state.memory), inp["write_content_keys"], inp["write_content_strengths"])
ag = inp["allocation_gate"][..., None]; wg = inp["write_gate"][..., None]
ww_ref = wg * (ag * aw + (1 - ag) * cw)
ww_stale = wg * (ag * aw + (1 - ag) * cw0)
out, stg = core.run_sequence(torch.from_numpy(x).to(cuda)); torch.cuda.synchronize()
g = stg.access_state.write_weights.cpu().numpy()
print("ww gpu vs ref", np.abs(g - ww_ref).max(), " vs stale-memory hypothesis", np.abs(g - ww_stale).max())
print("gpu b0", g[0]); print("ref b0", ww_ref[0]); print("alloc b0", aw[0]); print("cw b0", cw[0]); print("usage b0", usage[0])

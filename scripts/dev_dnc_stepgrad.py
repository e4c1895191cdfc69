"""Dev: per-step raw interface gradients (dxi) of the DNC BPTT kernel vs torch autograd with retained grads."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dnc_oracle as D, dnc_oracle_torch as DT, ntm_oracle as O
from ntmtrack import dnc as G
cuda = torch.device("cuda:0")
B, T = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = T * 65
cfg = D.DNCConfig(514, 2, memory_size=256, word_size=64, num_reads=4, num_writes=1, hidden_size=200, clip_value=20)
rng = np.random.default_rng(23)
p = D.init_params(cfg, rng)
feats = np.maximum(rng.standard_normal((B, T, 64, 512)), 0).astype(np.float32)
gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
x = O.serialize_inputs(feats, gts)
t64 = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64)
pt = {k: t64(v).requires_grad_(True) for k, v in p.items()}
# per-step additive probes on every interface linear: grad(probe[t]) = d loss / d raw interface at step t
names = [n for n, _ in cfg.access.interface]
probes = {n: [torch.zeros(w, dtype=torch.float64, requires_grad=True) for _ in range(S)] for n, w in cfg.access.interface}
xs = t64(np.ascontiguousarray(np.transpose(x, (1, 0, 2))))
st = DT.initial_state(cfg, B)
ys = []
for t in range(S):
    pp = dict(pt)
    for n in names:
        pp["memory_access/%s/b" % n] = pt["memory_access/%s/b" % n] + probes[n][t]
    y, st = DT.dnc_step(cfg, pp, xs[t], st)
    ys.append(y)
ys = torch.stack(ys, 0)
(0.5 * (ys ** 2).sum()).backward()
core = G.DNC({"memory_size": 256, "word_size": 64, "num_reads": 4, "num_writes": 1}, {"hidden_size": 200}, 2, 20, device=cuda)
core.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
out, _st = core.run_sequence(torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).to(cuda), None, record=True)
dlogits = out.transpose(0, 1).contiguous()
keep = {}
orig = core._launch_bwd
def spy(*a, **k):
    r = orig(*a, **k); keep["dxi"] = r[1]; return r
core._launch_bwd = spy
core.backward_sequence(core.last_X, dlogits)
torch.cuda.synchronize()
dxi = keep["dxi"].cpu().numpy()[0]     # [S, IP]
o = 0
for n, w in cfg.access.interface:
    ref = np.stack([probes[n][t].grad.numpy() for t in range(S)])       # [S, w]
    got = dxi[:, o:o + w]
    err = np.abs(got - ref).max(1)
    worst = int(np.argmax(err))
    print("%-18s max|ref| %.3e  max err %.3e at t=%d ; sum-over-t ref %s got %s" % (n, np.abs(ref).max(), err.max(), worst,
          ref.sum(0)[:2], got.sum(0)[:2]))
    if n == "free_gate":
        for t in range(S):
            print("   t=%2d ref %s got %s" % (t, ref[t], got[t]))
    o += w

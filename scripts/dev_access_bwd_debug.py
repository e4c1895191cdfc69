"""Dev: where does the module-level d(memory) differ from autograd?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dnc_oracle as D
from oracle import dnc_oracle_torch as DT
from ntmtrack import dnc as G
cuda = torch.device("cuda")
rng = np.random.default_rng(11)
N, W, R, Wn, B, Din = [int(v) for v in (sys.argv[1:7] + ['32', '8', '2', '1', '2', '10'][len(sys.argv) - 1:])][:6]
ZERO = len(sys.argv) > 7 and sys.argv[7] == 'zero'
mod = G.MemoryAccess(N, W, R, Wn, input_dim=Din, device=cuda, seed=5)
sd = {k: v.numpy() * (3.0 if k.endswith("/w") else 1.0) for k, v in mod.state_dict().items()}
mod.load_state_dict(sd)
cfg = D.AccessConfig(N, W, R, Wn)
f = lambda *s: rng.random(s).astype(np.float32)
usage = np.stack([rng.permutation(N) for _ in range(B)]).astype(np.float32) / N * 0.8 + 0.1
rw = f(B, R, N); rw /= rw.sum(2, keepdims=True) + 1
ww = f(B, Wn, N); ww /= ww.sum(2, keepdims=True) + 1
prec = f(B, Wn, N); prec /= prec.sum(2, keepdims=True) + 1
link = f(B, Wn, N, N); link /= np.maximum(link.sum(2, keepdims=True), 1); link /= np.maximum(link.sum(3, keepdims=True), 1)
link[:, :, np.arange(N), np.arange(N)] = 0
st = D.AccessState((f(B, N, W) - 0.5).astype(np.float32), rw, ww, D.TemporalLinkageState(link.astype(np.float32), prec), usage)
if ZERO:
    st = D.access_initial_state(cfg, B)
x = rng.standard_normal((B, Din)).astype(np.float32)
Gr = rng.standard_normal((B, R, W)).astype(np.float32)
t64 = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True)
pt = {k: t64(v) for k, v in sd.items()}
xt = t64(x)
mem = t64(st.memory)
ost = DT.AccessState(mem, torch.tensor(st.read_weights, dtype=torch.float64), torch.tensor(st.write_weights, dtype=torch.float64),
                     DT.TemporalLinkageState(torch.tensor(st.linkage.link, dtype=torch.float64), torch.tensor(st.linkage.precedence_weights, dtype=torch.float64)),
                     torch.tensor(st.usage, dtype=torch.float64))
reads, _ = DT.access_step(cfg, pt, xt, ost)
(reads * torch.tensor(Gr, dtype=torch.float64)).sum().backward()
t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(cuda)
gst = G.AccessState(t(st.memory), t(st.read_weights), t(st.write_weights), G.TemporalLinkageState(t(st.linkage.link), t(st.linkage.precedence_weights)), t(st.usage))
g = mod.step_gradients(t(x), gst, t(Gr))
torch.cuda.synchronize()
ref = mem.grad.numpy(); got = g["memory"].cpu().numpy()
d = got - ref
print("max |ref| %.4e  max |err| %.4e" % (np.abs(ref).max(), np.abs(d).max()))
b, n, w = np.unravel_index(np.argmax(np.abs(d)), d.shape)
print("worst at", b, n, w, "ref", ref[b, n, w], "got", got[b, n, w])
print("ratio got/ref row:", got[b, n] / ref[b, n])
np.set_printoptions(precision=3, linewidth=200)
print("err per slot (batch %d):" % b, np.abs(d[b]).max(1))
print("ref per slot:", np.abs(ref[b]).max(1))
print("got per slot:", np.abs(got[b]).max(1))
for k in ("inputs", "usage", "read_weights", "precedence_weights"):
    print(k, "max", float(g[k].abs().max()))

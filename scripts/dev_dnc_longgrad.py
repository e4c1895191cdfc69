"""Dev: DNC c3 gradients vs the float64 oracle at several sequence lengths (diagnostic for the full-length test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dnc_oracle as D, dnc_oracle_torch as DT, ntm_oracle as O, ntm_oracle_torch as OT
from ntmtrack import dnc as G, tracker
cuda = torch.device("cuda:0")
B = 1
for T in [int(a) for a in sys.argv[1:]] or [1, 3, 10]:
    S = T * 65
    cfg = D.DNCConfig(514, 2, memory_size=256, word_size=64, num_reads=4, num_writes=1, hidden_size=200, clip_value=20)
    rng = np.random.default_rng(23)
    p = D.init_params(cfg, rng)
    feats = np.maximum(rng.standard_normal((B, T, 64, 512)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
    x = O.serialize_inputs(feats, gts)
    offs = rng.uniform(-.5, .5, size=(B, T, 2)).astype(np.float32)
    t64 = lambda v: torch.tensor(np.asarray(v), dtype=torch.float64)
    pt = {k: t64(v).requires_grad_(True) for k, v in p.items()}
    ys, _ = DT.run_model(cfg, pt, t64(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))))
    if T > 1:
        loss_ref, _ = OT.offset_loss(ys.permute(1, 0, 2), t64(offs))
    else:
        loss_ref = (ys ** 2).sum() * 0.5
    loss_ref.backward()
    core = G.DNC({"memory_size": 256, "word_size": 64, "num_reads": 4, "num_writes": 1}, {"hidden_size": 200}, 2, 20, device=cuda)
    core.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()})
    out, _st = core.run_sequence(torch.from_numpy(np.ascontiguousarray(np.transpose(x, (1, 0, 2)))).to(cuda), None, record=True)
    logits = out.transpose(0, 1).contiguous()
    if T > 1:
        loss, _pred, dlogits = tracker.offset_loss(logits, torch.from_numpy(offs).to(cuda), T)
    else:
        dlogits = logits.clone()
    grads = core.backward_sequence(core.last_X, dlogits)
    torch.cuda.synchronize()
    print("T=%d S=%d fwd max err %.3e" % (T, S, float(np.abs(logits.cpu().numpy() - ys.permute(1, 0, 2).detach().numpy()).max())))
    gmax = max(float(np.abs(pt[k].grad.numpy()).max()) for k in p)
    for k in sorted(p):
        ref = pt[k].grad.numpy(); got = grads[k].cpu().numpy()
        print("  %-36s ref %.3e abs err %.3e rel %.3e  (vs global max %.3e)" % (k, np.abs(ref).max(), np.abs(got - ref).max(),
              np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30), np.abs(got - ref).max() / gmax))
    if "free" in " ".join(p):
        print("  free_gate/b ref", pt["memory_access/free_gate/b"].grad.numpy(), "got", grads["memory_access/free_gate/b"].cpu().numpy())

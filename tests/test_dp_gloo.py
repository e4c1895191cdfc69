"""CPU, world_size 2 over gloo: the data-parallel contract of ntmtrack.parallel.

Each rank owns a contiguous block of sequences, computes the gradient of its local (un-normalised,
summed) loss -- here with the CPU oracle standing in for the HIP kernels -- and the ONE collective
of the step, a SUM all-reduce of the flat gradient bucket, must reproduce the gradient of a single
process running the global batch; the redundant clip + RMSProp then keeps the replicas identical."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    from oracle import ntm_oracle as O
    cfg = O.NTMConfig(514, 2, mem_size=64, mem_dim=8, shift_range=1, controller_hidden_size=16,
                      controller_num_layers=1, write_head_size=1, read_head_size=2)
    rng = np.random.default_rng(3)
    p = O.init_params(cfg, rng, scale=0.2)
    B, T = 4, 2
    feats = np.maximum(rng.standard_normal((B, T, 64, 512)), 0).astype(np.float32)
    gts = rng.uniform(0, 1, size=(B, T, 64)).astype(np.float32)
    offs = rng.uniform(-.5, .5, size=(B, T, 2)).astype(np.float32)
    return cfg, p, O.serialize_inputs(feats, gts), offs


def _flatten(grads):
    return torch.cat([torch.from_numpy(np.ascontiguousarray(grads[k])).reshape(-1) for k in sorted(grads)])


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from ntmtrack import parallel
    from oracle import ntm_oracle as O
    from oracle import ntm_oracle_torch as OT
    torch.set_num_threads(1)
    assert parallel.init_from_env(backend="gloo") == (rank, world)
    cfg, p, x, offs = _case()
    lo, hi = parallel.shard_range(x.shape[0])
    loss, grads, _, _ = OT.loss_and_grads(cfg, p, x[lo:hi], offs[lo:hi])
    flat = _flatten(grads)
    parallel.allreduce_gradients(flat)                         # the step's only collective
    # replicas stay identical: same clipped RMSProp update everywhere
    names = sorted(grads)
    sizes = [grads[k].size for k in names]
    parts = torch.split(flat, sizes)
    g_list = [q.numpy().reshape(grads[k].shape) for q, k in zip(parts, names)]
    clipped, gn = O.clip_by_global_norm(g_list, 5.0)
    newp = [O.rmsprop_step(p[k].astype(np.float64), g, np.ones_like(g), np.zeros_like(g))[0] for k, g in zip(names, clipped)]
    chk = torch.tensor([float(sum(float(np.sum(q)) for q in newp))], dtype=torch.float64)
    gathered = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(gathered, chk)
    if rank == 0:
        torch.save({"flat": flat, "loss": loss, "checks": torch.cat(gathered)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_global_batch(tmp_path):
    from oracle import ntm_oracle_torch as OT
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    cfg, p, x, offs = _case()
    loss, grads, _, _ = OT.loss_and_grads(cfg, p, x, offs)     # single process, global batch
    ref = _flatten(grads)
    np.testing.assert_allclose(res["flat"].numpy(), ref.numpy(), rtol=1e-9, atol=1e-12)
    assert res["checks"][0] == res["checks"][1]                # bit-identical replicas after the update

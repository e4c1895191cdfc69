"""Data parallelism over independent sequences.

The reference has no multi-GPU path.  The recurrence couples every step of one
sequence, so a sequence (all its frames) lives on exactly one GPU; sequences are
independent in forward and backward.  One process drives one GPU
(``torch.distributed``, backend "nccl" = RCCL over xGMI; "gloo" in CPU tests):

  * rank r takes sequences [r*B_local, (r+1)*B_local) of the global batch;
  * parameters are replicated (same seed, or broadcast from rank 0);
  * exactly ONE collective per optimiser step: a SUM all-reduce of the flat fp32
    gradient bucket (2.7 MB for the NTM tracker).  The loss is an un-normalised
    sum over sequences (direct_offset_output.py:606), so summing the per-rank
    gradients reproduces a single process running the global batch; the clip and
    RMSProp then run redundantly and identically on every rank.
"""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend="nccl", device=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or dist.is_initialized():
        return world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=ws, **kw)
    return world()


def shard_range(global_batch, rank=None, world_size=None):
    """Contiguous block of sequence indices owned by `rank`; the global batch must divide evenly
    (every rank runs the same number of strictly sequential steps)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    if global_batch % world_size:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world_size))
    b = global_batch // world_size
    return rank * b, (rank + 1) * b


def allreduce_gradients(flat_grad):
    """SUM all-reduce of the flat gradient bucket (no-op for a single process)."""
    _, ws = world()
    if ws > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def broadcast_parameters(flat_params, src=0):
    _, ws = world()
    if ws > 1:
        dist.broadcast(flat_params, src=src)
    return flat_params

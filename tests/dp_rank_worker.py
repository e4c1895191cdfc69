"""One data-parallel rank of the HIP training path, started as a fresh child process by tests/test_dp_gpu.py:
    python tests/dp_rank_worker.py <out.pt> <global_batch> <T> <steps> <model>
RANK / WORLD_SIZE / MASTER_* come from the environment (torchrun contract).  The ranks share cuda:0 and talk
over gloo (NTK_DIST_BACKEND=gloo is how bench.py rehearses N > 1 on a one-GPU box); the code under test is the
product path: NTMOffsetTracker.submit_features -> train_on_submitted (HIP trunk, HIP BPTT, the all-reduce issued
inside the side stream, HIP clip + RMSProp)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_inputs(B, T, seed=5):
    g = torch.Generator().manual_seed(seed)
    frames = torch.rand((B * T, 224, 224, 3), generator=g) * 255.0 - torch.tensor([123.68, 116.78, 103.94])
    gts0 = torch.rand((B, 64), generator=g)
    offs = torch.rand((B, T, 2), generator=g) - 0.5
    return frames, gts0, offs


def make_tracker(model, B, T, dev):
    from ntmtrack import tracker
    from ntmtrack.vgg import VGG_LAYERS
    rng = np.random.default_rng(1)
    ws = {n: ((rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32), np.zeros(co, np.float32))
          for n, ci, co, _ in VGG_LAYERS}
    if model == "dnc":
        return tracker.DNCOffsetTracker(B, T, vgg_weights=ws, mem_size=64, mem_dim=16, hidden_size=64, device=dev, seed=3,
                                        learning_rate=1e-2)
    return tracker.NTMOffsetTracker(B, T, vgg_weights=ws, device=dev, seed=3, learning_rate=1e-2)


def run_steps(trk, frames, gts0, offs, steps, before_step=None):
    losses = []
    trk.submit_features(frames)
    for i in range(steps):
        if before_step is not None:
            before_step(i)
        losses.append(trk.train_on_submitted(gts0, offs))
        if i + 1 < steps:
            trk.submit_features(frames)
    trk.join()
    torch.cuda.synchronize()
    return [float(l.cpu()) for l in losses]


def main():
    out, GB, T, steps, model = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    from ntmtrack import parallel
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rank, world = parallel.init_from_env(backend=os.environ.get("NTK_DIST_BACKEND", "gloo"))
    lo, hi = parallel.shard_range(GB, rank, world)
    frames, gts0, offs = make_inputs(GB, T)
    abort = model == "dnc_abort"          # rank 1's cluster launch "times out" before the LAST step (fault injection)
    trk = make_tracker("dnc" if abort else model, hi - lo, T, dev)
    parallel.broadcast_parameters(trk._ckpt_params().flat)
    snap = {}

    def before_step(i):
        if abort and i == steps - 1:
            torch.cuda.synchronize()
            snap["flat"] = trk._ckpt_params().flat.cpu()          # parameters after the last good step
            if rank == 1:
                assert trk.core.inject_abort(hi - lo), "no cluster form at this shape: nothing to inject"
    losses = run_steps(trk, frames[lo * T:hi * T].to(dev), gts0[lo:hi].to(dev), offs[lo:hi].to(dev), steps, before_step)
    res = {"flat": trk._ckpt_params().flat.cpu(), "losses": torch.tensor(losses, dtype=torch.float64)}
    if abort:
        from ntmtrack._lib import NtkError
        try:
            trk.check_step()
            raised = 0
        except NtkError:
            raised = 1
        res.update(flat_before_last=snap["flat"], raised=torch.tensor(raised), skipped_after_check=torch.tensor(int(trk.opt.skipped)))
    torch.save(res, out)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

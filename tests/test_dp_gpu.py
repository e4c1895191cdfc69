"""GPU: the data-parallel training path on the HIP kernels.  Two ranks, started as fresh child
processes that share the one visible device and exchange the flat gradient bucket over gloo, each run
NTMOffsetTracker.submit_features / train_on_submitted on their shard; the replicas must stay bit-identical and equal
the single-process run on the global batch (the loss is an un-normalised sum, direct_offset_output.py:606, so a SUM
all-reduce reproduces it)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("model", ["ntm", "dnc"])
def test_two_ranks_hip_training_steps_equal_global_batch(cuda, tmp_path, model):
    sys.path.insert(0, HERE)
    import dp_rank_worker as W
    GB, T, steps, world = 2, 2, 2, 2
    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        out = str(tmp_path / ("rank%d.pt" % rank))
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NTK_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_rank_worker.py"), out, str(GB), str(T),
                                       str(steps), model], env=env))
        outs.append(out)
    # meanwhile: the single-process run on the global batch, in this process
    frames, gts0, offs = W.make_inputs(GB, T)
    trk = W.make_tracker(model, GB, T, cuda)
    losses = W.run_steps(trk, frames.to(cuda), gts0.to(cuda), offs.to(cuda), steps)
    ref = trk._ckpt_params().flat.cpu()
    for p in procs:
        assert p.wait(timeout=600) == 0
    res = [torch.load(o, weights_only=True) for o in outs]
    assert torch.equal(res[0]["flat"], res[1]["flat"]), "replicas diverged"
    # per-rank losses sum to the global loss; parameters equal the single-process ones (different summation order only)
    for s in range(steps):
        np.testing.assert_allclose(float(res[0]["losses"][s] + res[1]["losses"][s]), losses[s], rtol=1e-5)
    assert float((res[0]["flat"] - ref).abs().max()) < 1e-6
    assert float((ref - W.make_tracker(model, GB, T, cuda)._ckpt_params().flat.cpu()).abs().max()) > 1e-4      # it trained


def test_cluster_abort_on_one_rank_stops_the_step_on_every_rank(cuda, tmp_path):
    """Data-parallel failure path: rank 1's DNC cluster launch aborts (its sticky error word is planted, as a timed-out
    hand-off plants it) before the last of three steps.  DNC.guard poisons rank 1's loss and gradient with NaN, the SUM
    all-reduce carries the NaN to rank 0, and the checked optimiser step applies NOTHING on either rank: parameters equal
    those after the last good step, bit for bit and on both ranks; both ranks' loss for that step is NaN; check_step()
    raises on both (rank 1: its error word; rank 0: the skipped step) -- no rank steps on a partial gradient."""
    GB, T, steps, world = 2, 2, 3, 2
    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        out = str(tmp_path / ("rank%d.pt" % rank))
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NTK_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_rank_worker.py"), out, str(GB), str(T),
                                       str(steps), "dnc_abort"], env=env))
        outs.append(out)
    for p in procs:
        assert p.wait(timeout=600) == 0
    res = [torch.load(o, weights_only=True) for o in outs]
    for r in res:
        assert torch.isfinite(r["losses"][:steps - 1]).all() and torch.isnan(r["losses"][steps - 1])
        assert torch.equal(r["flat"], r["flat_before_last"]), "a rank applied an update from a failed step"
        assert torch.isfinite(r["flat"]).all()
        assert int(r["raised"]) == 1 and int(r["skipped_after_check"]) == 0
    assert torch.equal(res[0]["flat"], res[1]["flat"]), "replicas diverged"

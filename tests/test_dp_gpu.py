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

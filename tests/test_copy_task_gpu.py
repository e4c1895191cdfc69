"""GPU: NTM copy task (BASELINE configs[0], main.py:1540-1644): loss/gradient parity with the autograd oracle on
one batch, and an end-to-end check that BPTT + clip + RMSProp actually learn (the loss goes down)."""
import numpy as np
import pytest
import torch

from oracle import ntm_oracle as O
from oracle import ntm_oracle_torch as OT

pytestmark = pytest.mark.gpu


def test_copy_task_loss_and_gradients_match_oracle(cuda):
    from ntmtrack.copy_task import CopyTask, make_batch
    B, L = 3, 6
    task = CopyTask(B, L, hidden_size=100, device=cuda, seed=2, init_scale=0.2)
    sd = {k: v.numpy() for k, v in task.cell.state_dict().items()}
    cfg = O.NTMConfig(4, 4, mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=100, controller_num_layers=1,
                      write_head_size=1, read_head_size=1)
    bits = torch.randint(0, 2, (B, L, 3), generator=torch.Generator().manual_seed(1)).float()
    x, y = make_batch(bits)
    # layout contract of main.py:1546-1559
    assert x.shape == (B, 2 * L + 1, 4) and (x[:, L] == torch.tensor([0., 0., 0., 1.])).all() and not x[:, L + 1:].any()
    assert not y[:, :L + 1].any() and torch.equal(y[:, L + 1:, :3], bits)
    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sd.items()}
    logits, _ = OT.loop(cfg, pt, x.double())
    p = torch.sigmoid(logits)
    yl = y.double()
    loss_ref = (-(yl * torch.log(p + 1e-7) + (1 - yl) * torch.log(1 - p + 1e-7))).mean()     # tf.losses.log_loss
    loss_ref.backward()
    loss, lg = task.loss_and_grads(x.to(cuda), y.to(cuda))
    torch.cuda.synchronize()
    np.testing.assert_allclose(lg.cpu().numpy(), logits.detach().numpy(), atol=2e-5)
    np.testing.assert_allclose(float(loss.cpu()), float(loss_ref.detach()), rtol=1e-5)
    got = task.cell.params.to_tf(grad=True)
    for k in sorted(sd):
        ref = pt[k].grad.numpy()
        err = np.max(np.abs(got[k].numpy() - ref)) / (np.max(np.abs(ref)) + 1e-30)
        assert err < 3e-3, (k, err)


def test_copy_task_learns(cuda):
    from ntmtrack.copy_task import CopyTask
    B, L = 32, 4
    task = CopyTask(B, L, hidden_size=100, learning_rate=3e-3, device=cuda, seed=3)
    g = torch.Generator().manual_seed(7)
    losses = []
    for it in range(400):
        bits = torch.randint(0, 2, (B, L, 3), generator=g).float().to(cuda)
        losses.append(task.train_step(bits))
    torch.cuda.synchronize()
    vals = [float(l.cpu()) for l in losses]
    first, last = np.mean(vals[:10]), np.mean(vals[-10:])
    assert np.isfinite(vals).all()
    assert last < 0.75 * first, (first, last)

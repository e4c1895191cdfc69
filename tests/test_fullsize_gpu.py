"""GPU: BASELINE.json full sizes (config 2: 32 sequences x 20 frames of 224x224, 1300 NTM steps) checked through
size-independent properties -- the oracle cannot run these sizes in seconds:
batch invariance (a frame / sequence computed inside the full batch equals the same one computed alone, bit for
bit: no cross-sample coupling, no tile-boundary effects), run-to-run bitwise determinism of the whole training
step (fixed-order reductions everywhere), and the invariants of the addressing weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _vgg_weights(seed):
    from ntmtrack.vgg import VGG_LAYERS
    rng = np.random.default_rng(seed)
    return {n: ((rng.standard_normal((3, 3, ci, co)) * np.sqrt(2.0 / (9 * ci))).astype(np.float32), np.zeros(co, np.float32))
            for n, ci, co, _ in VGG_LAYERS}


def test_vgg_full_batch_is_frame_invariant(cuda):
    from ntmtrack import vgg
    net = vgg.VGG16Conv43(_vgg_weights(1), device=cuda)
    g = torch.Generator().manual_seed(2)
    F = 640
    frames = torch.empty((F, 224, 224, 3))
    for i in range(0, F, 64):
        frames[i:i + 64] = torch.rand((64, 224, 224, 3), generator=g) * 255 - 117
    frames = frames.to(cuda)
    full = net(frames)
    assert full.shape == (F, 28, 28, 512) and torch.isfinite(full).all() and (full >= 0).all()
    for i in (0, 37, 639):                       # first, middle (odd row-tile offset: 784 rows/frame is not a multiple of 128), last
        alone = net(frames[i:i + 1].contiguous())
        assert torch.equal(alone[0], full[i]), "frame %d differs between batch and single-frame launch" % i
    perm = torch.tensor([5, 3, 11, 7])
    sub = net(frames[perm].contiguous())
    assert torch.equal(sub, full[perm])


def test_ntm_full_length_batch_invariance_and_weight_invariants(cuda):
    from ntmtrack import tracker
    B, T = 32, 20
    trk = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=cuda, seed=5)
    g = torch.Generator().manual_seed(3)
    fmap = torch.relu(torch.randn((B * T, 28, 28, 512), generator=g)).to(cuda)
    gts0 = torch.rand((B, 64), generator=g).to(cuda)
    X, st0, logits, rec = trk.forward_features(fmap, gts0, record=True)
    assert logits.shape == (B, T * 65, 2) and torch.isfinite(logits).all()
    w = rec["w"]                                  # [B,S,H,N] head weights of every step
    assert (w >= 0).all()
    assert (w.sum(-1) < 1.0).all()                # sharpen divides by sum + 1e-3 (quirk Q4): never a full distribution
    wc = rec["wc"]
    assert torch.allclose(wc.sum(-1), torch.ones_like(wc.sum(-1)), atol=1e-4)      # content weights are a softmax
    # sequence 7 alone == sequence 7 inside the batch (one workgroup per sequence, nothing shared but weights)
    trk1 = tracker.NTMOffsetTracker(1, T, vgg_weights=None, device=cuda, seed=5)
    X1, _s, logits1, _r = trk1.forward_features(fmap[7 * T:8 * T].contiguous(), gts0[7:8].contiguous())
    assert torch.equal(logits1[0], logits[7])


def test_training_step_is_bitwise_deterministic(cuda):
    from ntmtrack import tracker
    B, T = 32, 20
    g = torch.Generator().manual_seed(4)
    fmap = torch.relu(torch.randn((B * T, 28, 28, 512), generator=g)).to(cuda)
    gts0 = torch.rand((B, 64), generator=g).to(cuda)
    offs = (torch.rand((B, T, 2), generator=g) - 0.5).to(cuda)
    grads, params = [], []
    for _ in range(2):
        trk = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=cuda, seed=6)
        loss, _ = trk.loss_and_grads(fmap, gts0, offs)
        grads.append(trk.cell.params.grad.clone())
        trk.opt.step()
        params.append(trk.cell.params.flat.clone())
    torch.cuda.synchronize()
    assert torch.equal(grads[0], grads[1]) and torch.equal(params[0], params[1])
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0


def test_config4_shape_training_step_properties(cuda):
    """BASELINE config 4's per-GPU shape (64 sequences x 30 frames -> 1950 strictly sequential steps): the whole
    NTM training step runs, is finite, batch-invariant and deterministic; doubling the batch by repeating the
    sequences doubles the (un-normalised, summed) loss and gradient exactly as direct_offset_output.py:606 implies --
    the property the data-parallel all-reduce relies on."""
    from ntmtrack import tracker
    B, T = 64, 30
    g = torch.Generator().manual_seed(9)
    fmap_h = torch.relu(torch.randn((B // 2 * T, 28, 28, 512), generator=g)).to(cuda)
    gts_h = torch.rand((B // 2, 64), generator=g).to(cuda)
    offs_h = (torch.rand((B // 2, T, 2), generator=g) - 0.5).to(cuda)
    half = tracker.NTMOffsetTracker(B // 2, T, vgg_weights=None, device=cuda, seed=6)
    loss_h, _ = half.loss_and_grads(fmap_h, gts_h, offs_h)
    full = tracker.NTMOffsetTracker(B, T, vgg_weights=None, device=cuda, seed=6)
    loss_f, _ = full.loss_and_grads(torch.cat([fmap_h, fmap_h]), torch.cat([gts_h, gts_h]), torch.cat([offs_h, offs_h]))
    torch.cuda.synchronize()
    assert torch.isfinite(full.cell.params.grad).all()
    np.testing.assert_allclose(float(loss_f.cpu()), 2 * float(loss_h.cpu()), rtol=1e-5)
    gh, gf = half.cell.params.grad, full.cell.params.grad
    rel = float((gf - 2 * gh).abs().max() / gf.abs().max())
    assert rel < 1e-4, rel


def test_config5_shape_dnc_training_step_properties(cuda):
    """BASELINE config 5's cell (DNC 512x128, 4 read heads) at a reduced batch and length, through the SEGMENTED
    BPTT that its full length (3250 steps) requires: finite, the duplicated batch doubles loss and gradient, and
    one optimiser step on the same batch lowers the loss."""
    from ntmtrack import tracker
    B, T = 4, 3
    g = torch.Generator().manual_seed(10)
    fmap_h = torch.relu(torch.randn((B // 2 * T, 28, 28, 512), generator=g)).to(cuda)
    gts_h = torch.rand((B // 2, 64), generator=g).to(cuda)
    offs_h = (torch.rand((B // 2, T, 2), generator=g) - 0.5).to(cuda)
    kw = dict(vgg_weights=None, mem_size=512, mem_dim=128, device=cuda, seed=6, learning_rate=1e-3)
    half = tracker.DNCOffsetTracker(B // 2, T, **kw)
    half.core.bptt_segment = 65
    loss_h, _ = half.loss_and_grads(fmap_h, gts_h, offs_h)
    full = tracker.DNCOffsetTracker(B, T, **kw)
    full.core.bptt_segment = 50                  # segment boundaries need not align with frames
    args = (torch.cat([fmap_h, fmap_h]), torch.cat([gts_h, gts_h]), torch.cat([offs_h, offs_h]))
    loss_f, _ = full.loss_and_grads(*args)
    torch.cuda.synchronize()
    assert full.core.last_segments is not None and len(full.core.last_segments[1]) == 4
    gh, gf = half.core.params.grad, full.core.params.grad
    assert torch.isfinite(gf).all()
    np.testing.assert_allclose(float(loss_f.cpu()), 2 * float(loss_h.cpu()), rtol=1e-5)
    assert float((gf - 2 * gh).abs().max() / gf.abs().max()) < 1e-4
    full.opt.step()
    loss_2, _ = full.loss_and_grads(*args)
    assert float(loss_2.cpu()) < float(loss_f.cpu())


@pytest.mark.parametrize("form,k", [("lds", 8), ("mp", 4)], ids=["lds_k8_whole_chip", "mp_k4_half_chip"])
def test_config3_fullsize_dnc_training_step_properties(cuda, form, k):
    """(Both cluster forms: the LDS-resident one at 8 workgroups per sequence = every CU, and the memory-partitioned one at 4 =
    half the chip, which is what DNCOffsetTracker picks when it has a trunk to run beside the core.)
    BASELINE config 3 at full size (DNC 256x64, 4 read heads; 32 sequences x 20 frames -> 1300 strictly sequential
    steps, every CU of the chip in an 8-workgroup cluster per sequence): the whole training step (recorded forward,
    loss, BPTT, weight-gradient GEMMs) is finite, bitwise reproducible run to run (fixed-order reductions and
    hand-offs: no float atomics on the cluster path), no hand-off times out, and duplicating a half batch doubles the
    un-normalised loss and the gradient (the property the data-parallel SUM all-reduce relies on)."""
    from ntmtrack import tracker
    B, T = 32, 20
    g = torch.Generator().manual_seed(12)
    fmap_h = torch.relu(torch.randn((B // 2 * T, 28, 28, 512), generator=g)).to(cuda)
    gts_h = torch.rand((B // 2, 64), generator=g).to(cuda)
    offs_h = (torch.rand((B // 2, T, 2), generator=g) - 0.5).to(cuda)
    kw = dict(vgg_weights=None, mem_size=256, mem_dim=64, device=cuda, seed=6)
    half = tracker.DNCOffsetTracker(B // 2, T, **kw)
    half.core.cluster_form, half.core.cluster_k = form, k
    loss_h, _ = half.loss_and_grads(fmap_h, gts_h, offs_h)
    half.core.check_cluster()
    assert half.core.last_cluster_k == k and half.core.last_cluster_bwd_k == k and half.core.last_cluster_form == form
    args = (torch.cat([fmap_h, fmap_h]), torch.cat([gts_h, gts_h]), torch.cat([offs_h, offs_h]))
    grads = []
    for _ in range(2):
        full = tracker.DNCOffsetTracker(B, T, **kw)
        full.core.cluster_form, full.core.cluster_k = form, k
        loss_f, _ = full.loss_and_grads(*args)
        full.core.check_cluster()
        assert full.core.last_cluster_k == k and full.core.last_cluster_bwd_k == k and full.core.last_cluster_bwd_form == form
        grads.append(full.core.params.grad.clone())
    torch.cuda.synchronize()
    assert torch.equal(grads[0], grads[1]), "the DNC training step is not bitwise reproducible"
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0
    np.testing.assert_allclose(float(loss_f.cpu()), 2 * float(loss_h.cpu()), rtol=1e-5)
    gh, gf = half.core.params.grad, grads[0]
    assert float((gf - 2 * gh).abs().max() / gf.abs().max()) < 1e-4


def test_config5_fullsize_dnc_training_step_properties(cuda):
    """BASELINE config 5's per-GPU shape AT FULL SIZE: DNC 512 x 128, 4 read heads, 64 sequences x 50 frames -> 3250
    strictly sequential steps, on the memory-partitioned cluster kernels (ntk_dnc_mp_*: 4 workgroups per sequence = every
    CU, the link streamed through HBM) with the segmented BPTT its 283 GB of records require (re-recording serialised:
    two cooperative grids of 256 workgroups cannot share the chip).  One whole training step (recorded forward, loss,
    BPTT, weight-gradient GEMMs): no hand-off times out, everything is finite, two runs are bitwise identical (fixed-order
    reductions and hand-offs), and duplicating a half batch doubles the un-normalised loss and the gradient
    (direct_offset_output_with_dnc.py:606-620: the property the data-parallel SUM all-reduce relies on)."""
    from ntmtrack import tracker
    B, T = 64, 50
    g = torch.Generator().manual_seed(14)
    fmap_h = torch.relu(torch.randn((B // 2 * T, 28, 28, 512), generator=g)).to(cuda)
    gts_h = torch.rand((B // 2, 64), generator=g).to(cuda)
    offs_h = (torch.rand((B // 2, T, 2), generator=g) - 0.5).to(cuda)
    kw = dict(vgg_weights=None, mem_size=512, mem_dim=128, device=cuda, seed=6)
    half = tracker.DNCOffsetTracker(B // 2, T, **kw)
    loss_h, _ = half.loss_and_grads(fmap_h, gts_h, offs_h)
    half.core.check_cluster()
    assert half.core.last_cluster_form == "mp" and half.core.last_cluster_k == 4 and half.core.last_cluster_bwd_k == 4
    loss_h, gh = float(loss_h.cpu()), half.core.params.grad.clone()
    del half
    torch.cuda.empty_cache()
    args = (torch.cat([fmap_h, fmap_h]), torch.cat([gts_h, gts_h]), torch.cat([offs_h, offs_h]))
    grads, losses = [], []
    for _ in range(2):
        full = tracker.DNCOffsetTracker(B, T, **kw)
        assert full.serial_trunk                                  # 64 x 4 workgroups: the trunk pass cannot run beside the core
        loss_f, _ = full.loss_and_grads(*args)
        full.core.check_cluster()
        assert full.core.last_cluster_form == "mp" and full.core.last_cluster_k == 4 and full.core.last_cluster_bwd_k == 4
        assert full.core.last_segments is not None and full.core.last_rerecord_overlapped is False
        grads.append(full.core.params.grad.clone())
        losses.append(float(loss_f.cpu()))
        del full
        torch.cuda.empty_cache()
    assert torch.equal(grads[0], grads[1]) and losses[0] == losses[1], "the config-5 training step is not bitwise reproducible"
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0 and np.isfinite(losses[0])
    np.testing.assert_allclose(losses[0], 2 * loss_h, rtol=1e-5)
    assert float((grads[0] - 2 * gh).abs().max() / grads[0].abs().max()) < 1e-4

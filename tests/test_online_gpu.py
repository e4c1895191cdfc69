"""GPU: online single-frame tracker vs the oracle chain on synthetic frames."""
import numpy as np
import pytest
import torch

from oracle import ntm_oracle as O
from oracle import online_oracle as OO

pytestmark = pytest.mark.gpu


def test_crop_and_resize_matches_oracle(cuda):
    from ntmtrack import online
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 255, size=(37, 53, 3)).astype(np.float32)
    for box in ([0.1, 0.2, 0.8, 0.9], [-0.2, -0.1, 0.6, 1.3], [0.25, 0.25, 0.75, 0.75]):     # second box leaves the image
        ref = OO.crop_and_resize(img.astype(np.float64) - np.array(online.VGG_MEAN), box, 24, 24)
        got = online.crop_and_resize(torch.from_numpy(img).to(cuda), box, crop=24).cpu().numpy()
        np.testing.assert_allclose(got, ref, atol=2e-3)
    # outside the image the crop is 0 in mean-subtracted space (extrapolation_value of the reference graph)
    got = online.crop_and_resize(torch.from_numpy(img).to(cuda), [-0.5, -0.5, -0.1, -0.1], crop=8).cpu().numpy()
    assert not got.any()


def test_online_serialisation_puts_delimiter_first(cuda):
    from ntmtrack import online, _lib
    rng = np.random.default_rng(1)
    fmap = rng.standard_normal((1, 28, 28, 512)).astype(np.float32)
    gt = rng.uniform(size=(64,)).astype(np.float32)
    ref = OO.frame_block(None, fmap, gt)
    X = torch.empty((1, 65, 516), device=cuda)
    P = _lib.ptr
    _lib.check(_lib.lib().ntk_gather_serialize_online(P(torch.from_numpy(fmap).to(cuda)), P(torch.from_numpy(gt.reshape(1, 64)).to(cuda)),
                                                      P(X), 1, 1, 28, 28, 512, 516, 6, 2, 8, _lib.stream()), "ser")
    got = X.cpu().numpy()[0]
    assert np.array_equal(got[:, :514], ref.astype(np.float32))
    assert got[0, 512] == 1 and not got[0, :512].any()


def test_online_tracker_two_frames_match_oracle(cuda):
    """__init__(image, region) + track(image): crop, VGG trunk, Q8 serialisation, 65 steps with the state kept on
    the device, last-step tanh, bbox decode -- against the numpy chain."""
    from ntmtrack import online
    from ntmtrack.ntm import NTMCell
    from ntmtrack.vgg import VGG16Conv43
    rng = np.random.default_rng(5)
    ws = O.init_vgg_weights(rng)
    cfg = O.NTMConfig(514, 2, mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=200, controller_num_layers=1,
                      write_head_size=1, read_head_size=4)
    params = O.init_params(cfg, rng, scale=0.05)
    H, W = 90, 120
    frames = [rng.uniform(0, 255, size=(H, W, 3)).astype(np.float32) for _ in range(2)]
    region = (40.0, 30.0, 36.0, 27.0)                         # x, y, w, h in pixels

    # ---- oracle
    def o_update(size, reg):
        x1, y1, w, h = reg
        nb = online.normalize_bbox(size, (y1, x1, y1 + h, x1 + w))
        cb = online.calculate_cropbox(nb, 8, 6)
        return nb, cb, online.calculate_transformation(cb)
    nb, cb, tr = o_update((W, H), region)
    ws64 = {k: (w.astype(np.float64), b.astype(np.float64)) for k, (w, b) in ws.items()}
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    st = O.zero_state(cfg, p64, 1)

    def o_frame(img, first, st, nb, cb, tr):
        crop = OO.crop_and_resize(img.astype(np.float64) - np.array(online.VGG_MEAN), cb, 224, 224)
        fmap = O.vgg16_conv43(crop[None], ws64)
        gt = online.generate_gt(online.apply_transformation(nb, tr), 8, 6) if first else None
        blk = OO.frame_block(cfg, fmap, gt)
        _, logits, st = O.loop_ntm_tracker(cfg, p64, blk[None], state=st)
        return np.tanh(logits[0, -1]), st
    _, st = o_frame(frames[0], True, st, nb, cb, tr)
    offs, st = o_frame(frames[1], False, st, nb, cb, tr)
    bbox = online.offset_bbox([.5 - .375, .5 - .375, .5 + .375, .5 + .375], offs)
    y1, x1, y2, x2 = online.apply_transformation(bbox, np.linalg.inv(tr))
    ref_region = (x1 * W, y1 * H, (x2 - x1) * W, (y2 - y1) * H)

    # ---- HIP
    cell = NTMCell(2, mem_size=128, mem_dim=20, controller_hidden_size=200, controller_num_layers=1, write_head_size=1,
                   read_head_size=4, device=cuda)
    cell.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, input_dim=514)
    trk = online.NTMTracker(frames[0], region, cell, VGG16Conv43(ws, device=cuda), device=cuda)
    got = trk.track(frames[1])
    torch.cuda.synchronize()
    np.testing.assert_allclose(trk.offsets, offs, atol=1e-4)          # north_star tolerance on the emitted offsets
    np.testing.assert_allclose(np.array(got), np.array(ref_region), rtol=0, atol=1e-2)   # pixels
    assert isinstance(got, online.Rectangle)

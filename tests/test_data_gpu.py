"""GPU: get_input (format (B) -> the four tensors): resize(720x1280) - mean, crop_and_resize(224) vs a numpy chain."""
import os

import numpy as np
import pytest
import torch

from oracle import online_oracle as OO

pytestmark = pytest.mark.gpu


def _resize_ref(img, OH, OW):
    H, W, C = img.shape
    ys = np.arange(OH) * (H / OH); xs = np.arange(OW) * (W / OW)
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, H - 1); x1 = np.minimum(x0 + 1, W - 1)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    top = img[y0][:, x0] + (img[y0][:, x1] - img[y0][:, x0]) * fx
    bot = img[y1][:, x0] + (img[y1][:, x1] - img[y1][:, x0]) * fx
    return top + (bot - top) * fy


def test_get_input_matches_numpy_chain(cuda, tmp_path):
    from PIL import Image
    from ntmtrack import data
    rng = np.random.default_rng(0)
    d = tmp_path / "train_seq_0"
    d.mkdir()
    names = []
    for i in range(2):
        arr = rng.integers(0, 256, size=(60, 80, 3), dtype=np.uint8)
        png = str(d / ("img%d.png" % i))
        Image.fromarray(arr).save(png)
        stem = str(d / ("%06d" % i))
        gt = rng.random((8, 8)); gt.tofile(stem + ".bin")
        with open(stem + ".txt", "w") as f:
            f.write("0.1,0.15,0.9,0.8,0.3,0.3,0.6,0.6,%s,%g,%g" % (png, 0.05 * i, -0.1 * i))
        names.append(stem)
    img, gts, yo, xo = data.get_input(names, device=cuda)
    torch.cuda.synchronize()
    assert img.shape == (2, 224, 224, 3) and gts.shape == (2, 8, 8)
    np.testing.assert_allclose(yo.cpu().numpy(), [0.0, 0.05], atol=1e-7)
    np.testing.assert_allclose(xo.cpu().numpy(), [0.0, -0.1], atol=1e-7)
    arr0 = np.asarray(Image.open(str(d / "img0.png")).convert("RGB"), dtype=np.float64)
    big = _resize_ref(arr0, 720, 1280) - np.array(data.VGG_MEAN)
    ref = OO.crop_and_resize(big, [0.1, 0.15, 0.9, 0.8], 224, 224)
    np.testing.assert_allclose(img[0].cpu().numpy(), ref, atol=5e-3)

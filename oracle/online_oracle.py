"""NumPy restatement of the reference's online tracker (test_tracker.py:104-405).  TEST INFRASTRUCTURE ONLY.
tf.image.crop_and_resize (TF 1.x, un-vendored, documented semantics: bilinear, normalised boxes mapped through
(dim-1), extrapolation_value outside the image) is **parity unpinned**."""
import numpy as np

from . import ntm_oracle as O


def crop_and_resize(image, box, crop_h, crop_w, extrapolation=0.0):
    H, W, C = image.shape
    y1, x1, y2, x2 = box
    out = np.full((crop_h, crop_w, C), extrapolation, dtype=np.float64)
    hs = (y2 - y1) * (H - 1) / (crop_h - 1) if crop_h > 1 else 0.0
    ws = (x2 - x1) * (W - 1) / (crop_w - 1) if crop_w > 1 else 0.0
    for y in range(crop_h):
        in_y = y1 * (H - 1) + y * hs if crop_h > 1 else 0.5 * (y1 + y2) * (H - 1)
        if in_y < 0 or in_y > H - 1:
            continue
        ty, by = int(np.floor(in_y)), int(np.ceil(in_y))
        yl = in_y - ty
        for x in range(crop_w):
            in_x = x1 * (W - 1) + x * ws if crop_w > 1 else 0.5 * (x1 + x2) * (W - 1)
            if in_x < 0 or in_x > W - 1:
                continue
            lx, rx = int(np.floor(in_x)), int(np.ceil(in_x))
            xl = in_x - lx
            top = image[ty, lx] + (image[ty, rx] - image[ty, lx]) * xl
            bot = image[by, lx] + (image[by, rx] - image[by, lx]) * xl
            out[y, x] = top + (bot - top) * yl
    return out


def frame_block(cfg_unused, fmap, gt_or_none):
    """test_tracker.py:383-404: [65, 514] block, delimiter row FIRST, then 64 rows [feat, 0, gt]."""
    feats = O.extract_features(fmap)[0]                         # [64, C]
    C = feats.shape[1]
    tgt = gt_or_none.reshape(-1, 1) if gt_or_none is not None else np.zeros((64, 1))
    rows = np.concatenate([feats, np.zeros((64, 1)), tgt], axis=1)
    delim = np.concatenate([np.zeros((1, C)), np.ones((1, 1)), np.zeros((1, 1))], axis=1)
    return np.concatenate([delim, rows], axis=0)

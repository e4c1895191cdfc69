"""Online single-frame tracker: the contract of test_tracker.NTMTracker
(test_tracker.py:104-405) -- ``NTMTracker(image, region)`` then ``track(image) -> Rectangle`` -- on the HIP
path.  The reference runs 65 separate ``sess.run`` calls per frame and round-trips the whole NTM state
host<->device through feed_dict at every step (:284-299); here one frame is one crop kernel, one VGG trunk
pass, one serialise kernel and ONE 65-step sequence-kernel launch with the state resident on the device.

Quirk Q8 is kept: inference puts the delimiter row FIRST and reads the output of the LAST feature step
(:400-404, :274-282), although training puts it last and reads the delimiter step.
Box geometry (preprocess.py:73-149, :205-240 for behaviour) lives in ntmtrack.geometry.
"""
import collections

import numpy as np
import torch

from . import _lib
from .ntm import NTMCell, _P, _np
from .tracker import GRID_N, GRID_START, GRID_STEP, NUM_FEATURES
from .vgg import VGG16Conv43

Rectangle = collections.namedtuple("Rectangle", ["x", "y", "width", "height"])     # vot.py:23-25

VGG_MEAN = (123.68, 116.78, 103.94)


# ---- box geometry lives in ntmtrack.geometry (one implementation); re-exported here for callers of this module
from .geometry import (normalize_bbox, calculate_cropbox, calculate_transformation, apply_transformation,   # noqa: E402,F401
                       offset_bbox, discrete_gauss, generate_gt)


def crop_and_resize(image, box, crop=224, mean=VGG_MEAN, out=None):
    """(image [H,W,3] fp32 device tensor - mean) cropped to `box` (normalised y1,x1,y2,x2) and resized bilinearly."""
    H, W, C = image.shape
    if out is None:
        out = torch.empty((crop, crop, C), device=image.device, dtype=torch.float32)
    m = torch.tensor(mean, device=image.device, dtype=torch.float32) if mean is not None else None
    y1, x1, y2, x2 = [float(v) for v in box]
    _lib.check(_lib.lib().ntk_crop_and_resize(_P(image.contiguous()), H, W, C, _np(m), y1, x1, y2, x2, _P(out), crop, crop, 0.0,
                                              _lib.stream()), "ntk_crop_and_resize")
    return out


class NTMTracker(object):
    """test_tracker.NTMTracker on the HIP path.  `image`: [H,W,3] RGB array (uint8 or float), `region`: (x, y, w, h)
    in pixels (or normalised if all < 1, :306-309).  `cell`: an ntmtrack.ntm.NTMCell with loaded parameters;
    `vgg`: a VGG16Conv43."""

    def __init__(self, image, region, cell, vgg, cropbox_grid=8, bbox_grid=6, device="cuda"):
        self.cell, self.vgg = cell, vgg
        self.device = torch.device(device)
        self.cropbox_grid, self.bbox_grid = cropbox_grid, bbox_grid
        self.frame = 0
        self.init_region = region
        img = self._to_device(image)
        h, w, _ = img.shape
        self.image_size = (w, h)
        self._update_bbox(self.image_size, region)
        self.state = self.cell.zero_state(1)
        feats = self._preprocess_image(img, True)
        self._run_tracker(feats)                    # this output is discarded (test_tracker.py:146-148)

    def _to_device(self, image):
        t = torch.as_tensor(np.asarray(image), dtype=torch.float32) if not torch.is_tensor(image) else image.float()
        return t.to(self.device).contiguous()

    def _update_bbox(self, image_size, region):     # test_tracker.py:300-329
        x1, y1, w, h = region
        normalized = x1 < 1 and y1 < 1 and w < 1 and h < 1
        bbox = (y1, x1, y1 + h, x1 + w)
        self.normalized_bbox = list(bbox) if normalized else normalize_bbox(image_size, bbox)
        self.cropbox = calculate_cropbox(self.normalized_bbox, self.cropbox_grid, self.bbox_grid)
        self.transformation = calculate_transformation(self.cropbox)

    def _preprocess_image(self, img, is_first_frame):
        """-> serialised block [1, 65, ldx] on the device: delimiter row first, then the 64 feature rows (:370-405)."""
        crop = crop_and_resize(img, self.cropbox)
        self.cropped_input_image = crop
        try:
            fmap = self.vgg(crop.unsqueeze(0), latency=True)             # one frame per call: the trunk form with the shorter critical path
        except TypeError:                                                # (a caller's own trunk object without the flag)
            fmap = self.vgg(crop.unsqueeze(0))
        gts0 = None
        if is_first_frame:
            gt = generate_gt(apply_transformation(self.normalized_bbox, self.transformation), self.cropbox_grid, self.bbox_grid)
            gts0 = torch.as_tensor(gt.reshape(1, -1), dtype=torch.float32).to(self.device).contiguous()
        ldx = self.cell.dims.ldx
        X = torch.empty((1, NUM_FEATURES + 1, ldx), device=self.device)
        _lib.check(_lib.lib().ntk_gather_serialize_online(_P(fmap), _np(gts0), _P(X), 1, 1, fmap.shape[1], fmap.shape[2],
                                                          fmap.shape[3], ldx, GRID_START, GRID_STEP, GRID_N, _lib.stream()),
                   "ntk_gather_serialize_online")
        return X

    def _run_tracker(self, X):
        logits, _o, self.state, _rec = self.cell.run_sequence(X, self.state, record=False, want_outputs=False)
        return logits

    def _initial_normal_bbox(self):
        width = self.bbox_grid / float(self.cropbox_grid)
        return [.5 - width / 2, .5 - width / 2, .5 + width / 2, .5 + width / 2]

    def _decode_bbox(self, normalized_bbox):
        y1, x1, y2, x2 = apply_transformation(normalized_bbox, np.linalg.inv(self.transformation))
        w, h = self.image_size
        y1, x1, y2, x2 = y1 * h, x1 * w, y2 * h, x2 * w
        return Rectangle(x1, y1, x2 - x1, y2 - y1)

    def track(self, image):
        """One frame: returns the new region as Rectangle(x, y, width, height) in image coordinates."""
        self.frame += 1
        img = self._to_device(image)
        logits = self._run_tracker(self._preprocess_image(img, False))
        offsets = torch.tanh(logits[0, -1]).cpu().numpy()          # output of the LAST step (:274-282); 2 values -> host
        self.offsets = offsets
        self.output_bbox = offset_bbox(self._initial_normal_bbox(), offsets)
        region = self._decode_bbox(self.output_bbox)
        self._update_bbox(self.image_size, region)
        return region

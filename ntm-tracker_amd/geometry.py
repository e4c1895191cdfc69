"""Host-side box geometry of the tracking contract: the ONE implementation used by the online tracker, the
synthetic-input generator and bench.py.

Behaviour follows the reference's offline preparation (preprocess.py:73-149 box helpers, :205-240 heat-map);
the bodies are written from the definitions:

  * a box is (y1, x1, y2, x2); "normalised" means divided by (height-1, width-1);
  * the crop box is the object box scaled about its centre by cropbox_grid / bbox_grid;
  * the crop transformation is the affine map, in homogeneous (x, y, 1) coordinates, that sends the crop box
    to the unit square;
  * the ground-truth heat-map is an isotropic Gaussian sampled at the cell centres of a w x h grid,
    thresholded at eps * max and normalised to sum 1.
"""
import numpy as np


def normalize_bbox(size, bbox):
    """Pixel box -> fractions of the last valid pixel index (size = (width, height))."""
    width, height = size
    span = np.array([height - 1, width - 1, height - 1, width - 1], dtype=np.float64)
    return (np.asarray(bbox, dtype=np.float64) / span).tolist()


def calculate_cropbox(normalbbox, cropbox_grid, bbox_grid):
    """Object box scaled about its centre by cropbox_grid / bbox_grid."""
    box = np.asarray(normalbbox, dtype=np.float64)
    centre = (box[:2] + box[2:]) / 2
    half = (box[2:] - box[:2]) * (cropbox_grid / float(bbox_grid)) / 2
    return np.concatenate([centre - half, centre + half]).tolist()


def calculate_transformation(cropbox):
    """3x3 affine matrix over (x, y, 1) that maps the crop box onto [0,1] x [0,1]."""
    y1, x1, y2, x2 = cropbox
    sx, sy = 1.0 / (x2 - x1), 1.0 / (y2 - y1)
    T = np.diag([sx, sy, 1.0])
    T[0, 2] = -x1 * sx
    T[1, 2] = -y1 * sy
    return T


def apply_transformation(normalbbox, transformation):
    """Both corners of a box through a 3x3 homogeneous transformation."""
    y1, x1, y2, x2 = normalbbox
    corners = np.array([[x1, x2], [y1, y2], [1.0, 1.0]])
    out = np.asarray(transformation) @ corners
    return [out[1, 0], out[0, 0], out[1, 1], out[0, 1]]


def offset_bbox(init_bbox, offsets):
    """Translate a box by (dy, dx)."""
    dy, dx = offsets
    shift = (dy, dx, dy, dx)
    return tuple(c + s for c, s in zip(init_bbox, shift))


def discrete_gauss(center=(.5, .5), shape=(8, 8), sigma=1.0):
    """exp(-r^2 / 2 sigma^2) at the cell centres of a (w, h) grid whose origin is moved to `center`
    (normalised x, y); values below eps * max are dropped; normalised to sum 1 (left as is if all zero)."""
    w, h = shape
    xs = (0.5 - center[0] * w) + np.arange(w, dtype=np.float64)
    ys = (0.5 - center[1] * h) + np.arange(h, dtype=np.float64)
    g = np.exp(-(ys[:, None] ** 2 + xs[None, :] ** 2) / (2.0 * sigma * sigma))
    g[g < np.finfo(g.dtype).eps * g.max()] = 0
    total = g.sum()
    return g / total if total != 0 else g


def generate_gt(normalbbox, cropbox_grid, bbox_grid, focus=3):
    """Heat-map of a (transformed) box on the cropbox_grid x cropbox_grid grid.  The reference computes
    sigma = bbox_grid / focus under Python-2 integer division when both are ints (6 / 4 -> 1)."""
    y1, x1, y2, x2 = normalbbox
    both_int = isinstance(bbox_grid, int) and isinstance(focus, int)
    sigma = bbox_grid // focus if both_int else bbox_grid / focus
    return discrete_gauss(((x1 + x2) / 2., (y1 + y2) / 2.), (cropbox_grid, cropbox_grid), sigma)

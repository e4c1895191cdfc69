"""Input contract of the tracking path ("sequence_generator input contract").

Two generations exist in the reference:
 (A) legacy pickle from sequence_generator.py:76-154: a list of
     ``(seq_dir, obj_name, subseq_id, seq_len, [(frame.JPEG, (w,h), bbox, [gt maps])])`` consumed by
     ``default_get_batch`` (direct_offset_output.py:122-142);
 (B) current on-disk format from preprocess.py:321-334: ``<out>/<seq>_<track>/NNNNNN.txt`` (one CSV line
     ``crop_y1,crop_x1,crop_y2,crop_x2,bbox_y1,bbox_x1,bbox_y2,bbox_x2,image_path,y_offset,x_offset``) and
     ``NNNNNN.bin`` (8x8 float64 heat-map, 512 bytes), enumerated by ``get_valid_sequences`` (:94-120), batched
     by ``sevenbyseven_get_batch`` (:144-157) and decoded by ``get_input`` (:159-224) into the four tensors the
     hot path consumes: batch_img [B*T,224,224,3] (mean-subtracted), batch_gt [B*T,8,8], y_offsets, x_offsets.
Host-side file parsing is plain Python; the image math (resize to 720x1280, mean subtraction, crop_and_resize to
224x224) runs in HIP kernels.  ``SyntheticSequences`` produces the same four tensors without a dataset.
"""
import os

import numpy as np
import torch

from . import _lib
from .geometry import discrete_gauss

VGG_MEAN = (123.68, 116.78, 103.94)


SPLITS = ("train", "val")


def _stems(seqdir, suffix=".txt"):
    """Frame stems ('000000', ...) of one <seq>_<track> folder, in name order."""
    return sorted(name[:-len(suffix)] for name in os.listdir(seqdir) if name.endswith(suffix))


def get_valid_sequences(sequences_dir, min_length):
    """Enumerate the format-(B) folders under `sequences_dir` that hold at least `min_length` frames.

    Contract (behaviour of direct_offset_output.py:94-120): a folder with n >= min_length frame records is
    subsampled with the integer stride n // min_length starting at its first frame, keeping exactly min_length
    stems; shorter folders are dropped.  Returns (all, train, val), each a list of (folder, stems); the split is
    decided by 'train' / 'val' appearing in the folder path ('train' is tested first), and a folder that names
    neither is an error."""
    everything = []
    by_split = {name: [] for name in SPLITS}
    for entry in sorted(os.listdir(sequences_dir)):
        folder = os.path.join(sequences_dir, entry)
        stems = _stems(folder)
        stride = len(stems) // min_length
        if stride < 1:
            continue
        item = (folder, stems[0:(min_length - 1) * stride + 1:stride])
        split = next((name for name in SPLITS if name in folder), None)
        everything.append(item)
        if split is None:
            raise ValueError("sequence folder %r names neither of the splits %s" % (folder, "/".join(SPLITS)))
        by_split[split].append(item)
    return everything, by_split["train"], by_split["val"]


def default_get_batch(index, batch_size, seq_length, seqs):
    """Legacy pickle contract (A) (behaviour of direct_offset_output.py:122-142): `seqs` holds records
    (seq_dir, obj_name, subseq_id, seq_len, frames) with frames = [(path, (w, h), bbox, [gt maps per layer])].
    Takes records [index, index + batch_size), the first `seq_length` frames of each, and returns
    (frame paths flattened [B*T], first-layer gt maps flattened to [B, T, F], index + batch_size)."""
    stop = index + batch_size
    names, gts = [], []
    for record in seqs[index:stop]:
        frames = record[4][:seq_length]
        names.extend(frame[0] for frame in frames)
        gts.append(np.array([np.asarray(frame[-1][0]).reshape(-1) for frame in frames]))
    return names, np.array(gts), stop


def sevenbyseven_get_batch(index, batch_size, seqs):
    """Format (B) batching (behaviour of direct_offset_output.py:144-157): `seqs` as returned by
    get_valid_sequences; -> (frame paths without suffix [B*T], index + batch_size)."""
    stop = index + batch_size
    names = [os.path.join(folder, stem) for folder, stems in seqs[index:stop] for stem in stems]
    return names, stop


def load_frame_record(path_nosuffix, gt_width=8):
    """One frame of format (B): the CSV line and the float64 heat-map."""
    with open(path_nosuffix + '.txt') as f:
        fields = f.readline().strip().split(',')
    if len(fields) != 11:
        raise ValueError("%s.txt: expected 11 comma-separated fields, got %d" % (path_nosuffix, len(fields)))
    vals = [float(v) for v in fields[:8]]
    raw = np.fromfile(path_nosuffix + '.bin', dtype=np.float64)
    if raw.size != gt_width * gt_width:
        raise ValueError("%s.bin: expected %d float64 values, got %d" % (path_nosuffix, gt_width * gt_width, raw.size))
    return {"cropbox": vals[:4], "bbox": vals[4:8], "image_path": fields[8], "y_offset": float(fields[9]),
            "x_offset": float(fields[10]), "gt": raw.astype(np.float32).reshape(gt_width, gt_width)}


def _decode_image(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.float32)


def preprocess_frame(image, cropbox, device, resize_to=(720, 1280), crop=224, out=None):
    """image [H,W,3] float array -> mean-subtracted 224x224 crop on the device
    (direct_offset_output.py:193-211: resize_images(720x1280) - VGG_MEAN, then crop_and_resize)."""
    L, P = _lib.lib(), _lib.ptr
    img = torch.as_tensor(image, dtype=torch.float32).to(device).contiguous()
    H, W, C = img.shape
    big = torch.empty((resize_to[0], resize_to[1], C), device=device)
    _lib.check(L.ntk_resize_bilinear(P(img), H, W, C, P(big), resize_to[0], resize_to[1], _lib.stream()), "ntk_resize_bilinear")
    if out is None:
        out = torch.empty((crop, crop, C), device=device)
    mean = torch.tensor(VGG_MEAN, device=device)
    y1, x1, y2, x2 = [float(v) for v in cropbox]
    _lib.check(L.ntk_crop_and_resize(P(big), resize_to[0], resize_to[1], C, P(mean), y1, x1, y2, x2, P(out), crop, crop, 0.0,
                                     _lib.stream()), "ntk_crop_and_resize")
    return out


def get_input(frame_names_nosuffix, device="cuda", reverse_image=False, image_root=None):
    """direct_offset_output.py:159-224 -> (batch_img [n,224,224,3], batch_gt [n,8,8], y_offsets [n], x_offsets [n])."""
    n = len(frame_names_nosuffix)
    dev = torch.device(device)
    batch_img = torch.empty((n, 224, 224, 3), device=dev)
    gts, ys, xs = [], [], []
    for i, name in enumerate(frame_names_nosuffix):
        rec = load_frame_record(name)
        path = rec["image_path"] if image_root is None else os.path.join(image_root, rec["image_path"])
        preprocess_frame(_decode_image(path), rec["cropbox"], dev, out=batch_img[i])
        gts.append(rec["gt"]); ys.append(rec["y_offset"]); xs.append(rec["x_offset"])
    batch_gt = torch.from_numpy(np.stack(gts)).to(dev)
    y_off = torch.tensor(ys, dtype=torch.float32, device=dev)
    x_off = torch.tensor(xs, dtype=torch.float32, device=dev)
    if reverse_image:                                   # :187-188, :203-204
        x_off = -x_off
        batch_img = torch.flip(batch_img, dims=[2])
    return batch_img, batch_gt, y_off, x_off


class SyntheticSequences(object):
    """The four tensors of the contract without a dataset: frames U[0,255) - VGG_MEAN, frame-0 heat-map
    = discrete_gauss((.5,.5),(8,8),1), offsets U(-.5,.5) with frame 0 = 0."""

    def __init__(self, batch_size, sequence_length, seed=42, device="cuda"):
        self.B, self.T, self.seed, self.device = batch_size, sequence_length, seed, torch.device(device)

    def batch(self, step=0):
        B, T = self.B, self.T
        g = torch.Generator(device="cpu").manual_seed(self.seed + step)
        mean = torch.tensor(VGG_MEAN)
        frames = torch.empty((B * T, 224, 224, 3), dtype=torch.float32)
        for i in range(0, B * T, 64):
            n = min(64, B * T - i)
            frames[i:i + n] = torch.rand((n, 224, 224, 3), generator=g) * 255.0 - mean
        hm = discrete_gauss((.5, .5), (8, 8), 1.0)
        gts = torch.from_numpy(np.tile(hm.astype(np.float32)[None], (B * T, 1, 1)))
        offs = torch.rand((B, T, 2), generator=g) - 0.5
        offs[:, 0, :] = 0
        d = self.device
        return frames.to(d), gts.to(d), offs[:, :, 0].reshape(-1).to(d), offs[:, :, 1].reshape(-1).to(d)

"""CPU: the input contract (format (B) files, sequence enumeration, batching) on hand-written fixtures."""
import os

import numpy as np
import pytest


def _write_seq(root, name, nframes, gt_seed=0):
    d = os.path.join(root, name)
    os.makedirs(d)
    rng = np.random.default_rng(gt_seed)
    for i in range(nframes):
        stem = os.path.join(d, "%06d" % i)
        gt = rng.random((8, 8))
        gt.tofile(stem + ".bin")                                     # 8x8 float64 = 512 bytes (preprocess.py:322-324)
        with open(stem + ".txt", "w") as f:                          # preprocess.py:329-334
            f.write("0.1,0.2,0.8,0.9,0.3,0.35,0.6,0.7,/data/img_%06d.JPEG,%g,%g" % (i, 0.01 * i, -0.02 * i))
    return d


def test_load_frame_record(tmp_path):
    from ntmtrack import data
    d = _write_seq(str(tmp_path), "train_seqA_0", 2, gt_seed=3)
    rec = data.load_frame_record(os.path.join(d, "000001"))
    assert rec["cropbox"] == [0.1, 0.2, 0.8, 0.9] and rec["bbox"] == [0.3, 0.35, 0.6, 0.7]
    assert rec["image_path"] == "/data/img_000001.JPEG"
    assert rec["y_offset"] == pytest.approx(0.01) and rec["x_offset"] == pytest.approx(-0.02)
    assert os.path.getsize(os.path.join(d, "000001.bin")) == 512
    rng = np.random.default_rng(3); rng.random((8, 8))
    np.testing.assert_allclose(rec["gt"], rng.random((8, 8)).astype(np.float32))
    with open(os.path.join(d, "bad.txt"), "w") as f:
        f.write("1,2,3")
    np.zeros(64).tofile(os.path.join(d, "bad.bin"))
    with pytest.raises(ValueError):
        data.load_frame_record(os.path.join(d, "bad"))


def test_get_valid_sequences_and_batching(tmp_path):
    from ntmtrack import data
    root = str(tmp_path)
    _write_seq(root, "train_seqA_0", 45)
    _write_seq(root, "train_seqB_1", 19)          # shorter than min_length 20 -> dropped (skip == 0)
    _write_seq(root, "val_seqC_0", 20)
    result, train, val = data.get_valid_sequences(root, 20)
    assert [os.path.basename(s) for s, _ in result] == ["train_seqA_0", "val_seqC_0"]
    assert len(train) == 1 and len(val) == 1
    seq, frames = train[0]
    assert len(frames) == 20 and frames[0] == "000000" and frames[1] == "000002"     # stride 45 // 20 = 2
    names, idx = data.sevenbyseven_get_batch(0, 2, result)
    assert idx == 2 and len(names) == 40 and names[0].endswith(os.path.join("train_seqA_0", "000000"))
    import tempfile
    other = tempfile.mkdtemp(prefix="seqs_")        # pytest's tmp path contains "valid": the reference tests the whole path string
    _write_seq(other, "other_seq", 25)
    with pytest.raises(Exception):
        data.get_valid_sequences(other, 20)                         # neither 'train' nor 'val' in the path (:118-119)


def test_default_get_batch_legacy_contract():
    from ntmtrack import data
    gt = lambda v: [np.full((8, 8), v)]
    seqs = [("d0", "obj", 0, 3, [("f%d.JPEG" % i, (640, 480), [(1, 2), (3, 4)], gt(i)) for i in range(3)]),
            ("d1", "obj", 0, 3, [("g%d.JPEG" % i, (640, 480), [(1, 2), (3, 4)], gt(10 + i)) for i in range(3)])]
    names, gts, idx = data.default_get_batch(0, 2, 2, seqs)
    assert names == ["f0.JPEG", "f1.JPEG", "g0.JPEG", "g1.JPEG"] and idx == 2
    assert gts.shape == (2, 2, 64) and gts[1, 1, 0] == 11

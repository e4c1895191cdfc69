"""CPU: host-side logic of the product package that needs no GPU -- packed parameter layout,
dimension bookkeeping, flop accounting, data contract helpers."""
import numpy as np
import torch

from oracle import ntm_oracle as O


def test_packed_params_roundtrip_and_layout():
    from ntmtrack.ntm import NTMDims, PackedParams
    d = NTMDims(514, 2, 128, 20, 1, 200, 4, 1)
    assert (d.P, d.PP, d.K, d.ldx, d.ldz, d.ldh) == (170, 172, 280, 516, 284, 204)
    cfg = O.NTMConfig(514, 2, mem_size=128, mem_dim=20, shift_range=1, controller_hidden_size=200,
                      controller_num_layers=1, write_head_size=1, read_head_size=4)
    assert cfg.control_dim == d.P
    rng = np.random.default_rng(0)
    p = O.init_params(cfg, rng)
    p["lstm/cell_0/biases"] = rng.standard_normal(800).astype(np.float32)
    p["addressing/biases"] = rng.standard_normal(170).astype(np.float32)
    p["output/biases"] = rng.standard_normal(2).astype(np.float32)
    pk = PackedParams(d, "cpu")
    pk.load_tf({k: torch.from_numpy(v) for k, v in p.items()})
    back = pk.to_tf()
    for k in p:
        assert np.array_equal(back[k].numpy(), p[k]), k
    # trainable element count and the structural zero padding
    assert sum(v.size for v in p.values()) == 673852
    assert int((pk.flat != 0).sum()) <= 673852
    # the packed gate product equals the TF-layout one: [x;read;h] @ W + b with columns regrouped per unit
    x = rng.standard_normal(514).astype(np.float32)
    z = rng.standard_normal(280).astype(np.float32)
    g_tf = np.concatenate([x, z]) @ p["lstm/cell_0/weights"] + p["lstm/cell_0/biases"]
    WxT, Wr = pk.view("WxT").numpy(), pk.view("Wr").numpy()
    g_pk = WxT[:, :514] @ x + z @ Wr[:280] + Wr[280]
    np.testing.assert_allclose(g_pk.reshape(200, 4).T.reshape(-1), g_tf, rtol=1e-5, atol=1e-5)
    # unpack/output linear
    h = rng.standard_normal(200).astype(np.float32)
    Wa = pk.view("Wa").numpy()
    u = h @ Wa[:200] + Wa[200]
    np.testing.assert_allclose(u[:170], h @ p["addressing/weights"] + p["addressing/biases"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(u[170:172], h @ p["output/weights"] + p["output/biases"], rtol=1e-5, atol=1e-5)


def test_conv_flop_accounting_matches_survey():
    from ntmtrack.vgg import conv_flops_per_frame, VGG_LAYERS
    assert conv_flops_per_frame() == 2 * 13959364608            # 13.959 GMAC / frame
    assert [l[0] for l in VGG_LAYERS] == [l[0] for l in O.VGG_LAYERS]


def test_shard_range_partitions_sequences():
    from ntmtrack import parallel
    assert parallel.shard_range(64, 0, 1) == (0, 64)
    spans = [parallel.shard_range(512, r, 8) for r in range(8)]
    assert spans[0] == (0, 64) and spans[7] == (448, 512)
    assert all(spans[i][1] == spans[i + 1][0] for i in range(7))
    try:
        parallel.shard_range(10, 0, 4)
        assert False
    except ValueError:
        pass


def test_geometry_known_answers_of_the_reference():
    """The reference's own self-tests for the box helpers (preprocess.py:152-157 calculate_transformation_test,
    :223-226 discrete_gauss_test against the MATLAB-style fspecial of :195-204), restated on ntmtrack.geometry."""
    from ntmtrack import geometry as G
    box = [.3, .4, .5, .6]
    np.testing.assert_almost_equal(G.apply_transformation(box, G.calculate_transformation(box)), [0, 0, 1, 1])
    # fspecial('gaussian', (7,7), 0.75): centred grid -3..3
    y, x = np.ogrid[-3:4, -3:4]
    h = np.exp(-(x * x + y * y) / (2. * 0.75 * 0.75))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    h /= h.sum()
    np.testing.assert_almost_equal(G.discrete_gauss((.5, .5), (7, 7), 0.75), h)
    # the oracle's independent restatement agrees, also off-centre
    np.testing.assert_allclose(G.discrete_gauss((.3, .6), (8, 8), 1.5), O.discrete_gauss((.3, .6), (8, 8), 1.5), rtol=1e-12)
    assert G.normalize_bbox((641, 481), (48, 64, 96, 128)) == [0.1, 0.1, 0.2, 0.2]
    np.testing.assert_allclose(G.calculate_cropbox([.4, .4, .6, .6], 8, 6), [.5 - .4 / 3, .5 - .4 / 3, .5 + .4 / 3, .5 + .4 / 3])
    assert G.offset_bbox((.1, .2, .3, .4), (.5, -.1)) == (.6, .1, .8, .30000000000000004)
    assert G.generate_gt([.25, .25, .75, .75], 8, 6, 4).shape == (8, 8)      # sigma = 6 // 4 = 1 (Python-2 division)
    np.testing.assert_allclose(G.generate_gt([.25, .25, .75, .75], 8, 6, 4), G.discrete_gauss((.5, .5), (8, 8), 1))


def test_bench_synthetic_inputs_follow_the_shard_ranges():
    """Rank r of a data-parallel bench run generates exactly the sequences parallel.shard_range gives it: the union over
    ranks is what a single process holds (bench.py)."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    from ntmtrack import parallel
    GB, T, world = 4, 2, 2
    whole = bench.synth_inputs(GB, T, "cpu", 42)
    for r in range(world):
        lo, hi = parallel.shard_range(GB, r, world)
        part = bench.synth_inputs(hi - lo, T, "cpu", 42, first_sequence=lo)
        assert torch.equal(part[0], whole[0][lo * T:hi * T]) and torch.equal(part[2], whole[2][lo:hi])
        assert torch.equal(part[1], whole[1][lo:hi])
    assert float(whole[2][:, 0].abs().max()) == 0.0               # frame 0 carries no offset (direct_offset_output.py:581-606)
    q = bench._cgroup_cpu_quota()
    assert q is None or q > 0


def test_dnc_segment_plan_is_even_and_records_two_tail_segments():
    """Segmented BPTT (config 5): the segment length is the even split of the smallest segment count the record budget
    allows, so a record set is no larger than it has to be (the allocator's reserved peak fell from 300 to 205 GB)."""
    from ntmtrack.dnc import DNC
    core = object.__new__(DNC)                                    # host arithmetic only: no device, no library
    core.N, core.W, core.R, core.Wn, core.hid, core.O = 512, 128, 4, 1, 200, 2
    core.ldz, core.ldh, core.ldy, core.IP = 716, 204, 716, 920
    core.bptt_segment = None
    per_step = 4 * 64 * core._record_floats_per_step()
    assert 86e6 < per_step < 88e6                                 # 1.36 MB per sequence-step x 64 sequences
    seg = core._segment_len(64, 3250)
    assert seg == 1084 and -(-3250 // seg) == 3 and seg * per_step <= DNC.record_budget_bytes
    assert core._segment_len(32, 3250) == 1625                    # half the batch: two segments
    assert core._segment_len(32, 1300) == 1300                    # config 3's shape fits whole
    core.bptt_segment = 50
    assert core._segment_len(4, 195) == 50
    assert DNC.recorded_tail_segments == 2


def test_bench_self_launch_starts_n_ranks_as_a_child_and_returns_its_exit_code():
    """`python bench.py --gpus 2` without WORLD_SIZE (how the driver calls N = 1) must not die of a usage error: it starts
    `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` as a child process and returns the child's
    exit code.  No GPU here, so every rank refuses to run (the HIP path has no CPU fallback) and the code is non-zero --
    what this checks is the mechanism: two ranks were started with RANK / WORLD_SIZE set, and their failure came back."""
    import os
    import subprocess
    import sys
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: tests/test_bench_gpu.py runs the real thing")
    env = dict(os.environ, NTK_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "launching 2 ranks as a child process" in res.stderr
    # both ranks refuse; the elastic launcher may tear the second one down before it has printed its own refusal
    assert 1 <= res.stderr.count("no GPU visible") <= 2, res.stderr[-2000:]
    assert "local_rank" in res.stderr or "RANK" in res.stderr, res.stderr[-2000:]
    assert not [l for l in res.stdout.splitlines() if l.strip().startswith("{")]

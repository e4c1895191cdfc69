"""GPU: bench.py contract -- one JSON line with the driver's keys plus roofline / memory_step (tiny workload)."""
import importlib
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("model", ["ntm", "dnc"])
def test_bench_emits_contract_json(cuda, capsys, monkeypatch, model):
    # in-process (this process already owns the GPU: no exec of a second program from here)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2",
                                      "--seq-len", "2", "--no-cpu-baseline", "--model", model])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "memory_step", "memory_step_bptt"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["scaling"] == "weak" and out["vs_baseline"] is None and out["data"] == "synthetic" and out["dtype"] == "f32"
    assert "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    if model == "ntm":          # the NTM tracker's trunk is the split form: three fp16 MFMA products per fp32 product, priced against the fp16 pipe
        assert r["peak"] == 2500.0 and "bf16p_kernel<X3>" in r["kernel"] and "dtype_note" in out
        assert 2.9 < r["achieved"] / r["algorithmic_tflops"] <= 3.0
    else:                       # the DNC tracker keeps the F(4x4) Winograd trunk on the fp32 pipe (a quarter of the direct form's multiplies)
        assert r["peak"] == 157.3 and "wino43" in r["kernel"] and "dtype_note" not in out
        assert r["algorithmic_tflops"] >= r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0.0 < r["frac"] <= 1.0, "roofline.frac is executed MFMA flops / peak: a utilisation"
    for ms in (out["memory_step"], out["memory_step_bptt"]):
        for key in ("kernel", "achieved", "peak", "unit", "frac", "traffic", "us_per_step"):
            assert key in ms, key
        assert ms["unit"] == "GB/s" and 0.0 < ms["frac"] <= 1.0
    if model == "dnc":          # a multi-CU form ran (LDS-resident or memory-partitioned cluster kernels), not the one-workgroup fallback
        for ms in (out["memory_step"], out["memory_step_bptt"]):
            assert "dnc_cluster_" in ms["kernel"] or "dnc_mp_" in ms["kernel"], ms["kernel"]
    # frames/s = frames per step / seconds per step
    assert abs(out["value"] - 2 * 2 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-2


def test_bench_two_ranks_print_one_json_line(cuda, tmp_path):
    """The N > 1 contract, rehearsed on the one-GPU box: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`
    with NTK_DIST_BACKEND=gloo (both ranks share the device; the driver's run uses nccl = RCCL, one GPU per rank) prints
    EXACTLY ONE JSON line (rank 0's), with n_gpus 2, the global batch = 2 x the per-GPU batch (weak scaling), and a
    frames/s value that is the whole job's: both ranks' frames over the max-over-ranks time."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, NTK_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--seq-len", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert abs(out["value"] - 2 * 2 * 2 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-2
    assert "cpu_baseline" not in out                                  # rank 0 at N = 1 only


def test_bench_gpus_2_without_torchrun_launches_its_own_ranks(cuda):
    """`python bench.py --gpus 2` the way the driver calls N = 1 (no torch.distributed.run, no WORLD_SIZE): bench.py starts
    the two ranks itself as a child process, forwards rank 0's ONE JSON line and returns the child's exit code."""
    import subprocess
    env = dict(os.environ, NTK_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--seq-len", "2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert "launching 2 ranks as a child process" in res.stderr


def test_rccl_world_size_1_training_step_equals_the_undistributed_step(cuda):
    """RCCL smoke on the one-GPU box: init_process_group("nccl", world_size=1, device_id=...) and one pipelined training
    step whose SUM all-reduce is issued inside the high-priority core stream -- the first place RCCL
    initialisation and that stream ordering run at all.  A world of one must change nothing: parameters after the step are
    bit-equal to the step of a tracker that never saw torch.distributed.  Runs in a child process (a process group is
    process-global state; the pytest process keeps none)."""
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import torch, torch.distributed as dist
        import dp_rank_worker as W
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        frames, gts0, offs = W.make_inputs(2, 2)
        frames, gts0, offs = frames.to(dev), gts0.to(dev), offs.to(dev)
        res = {}
        for model in ("ntm", "dnc"):
            ref = W.make_tracker(model, 2, 2, dev)
            W.run_steps(ref, frames, gts0, offs, 2)
            ref.check_step()
            res[model] = ref._ckpt_params().flat.clone()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from ntmtrack import parallel
        assert parallel.world() == (0, 1) and dist.get_backend() == "nccl"
        calls = []
        real = dist.all_reduce
        def spy(t, *a, **k):
            calls.append((torch.cuda.current_stream(dev).priority, t.numel()))
            return real(t, *a, **k)
        dist.all_reduce = spy
        parallel.allreduce_gradients.__globals__["world"] = lambda: (0, 2)     # force the collective at world size 1
        for model in ("ntm", "dnc"):
            trk = W.make_tracker(model, 2, 2, dev)
            n0 = len(calls)
            W.run_steps(trk, frames, gts0, offs, 2)
            trk.check_step()
            assert len(calls) - n0 == 2, calls
            assert all(p < 0 for p, _ in calls[n0:]), "the all-reduce must be issued inside the high-priority core stream"
            assert calls[-1][1] == trk._flat_grad().numel()
            assert torch.equal(trk._ckpt_params().flat, res[model]), model
        dist.barrier()
        dist.destroy_process_group()
        print("RCCL_WORLD1_OK")
    """) % (ROOT, ROOT, str(_free_port()))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NTK_DIST_BACKEND"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "RCCL_WORLD1_OK" in res.stdout, (res.stdout[-1000:], res.stderr[-3000:])


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p

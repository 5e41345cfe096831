"""bench.py's N > 1 path: `python bench.py --gpus N` with no launcher must start N ranks itself (or fail
loudly), headline the node-range partition BASELINE.json's north_star names, and never print n_gpus: 1
for --gpus 2.  On CPU the kernels cannot run (no CPU fallback by design): `--rehearse` runs ranks,
rendezvous, partition and exchange over gloo and says so in the line; the GPU form runs the real kernels
with two gloo ranks sharing one card."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout)


def _line(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[0])


def test_gpus2_self_launch_rehearsal_on_cpu():
    r = _run(["--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r)
    assert line["n_gpus"] == 2 and line["rehearsal"] is True and line["value"] is None
    assert line["config"]["partition"] == "nodes" and "node-range partition x2" in line["config"]["parallelism"]
    assert line["scaling"] == "strong" and line["cpu_baseline"] is None
    ex = line["exchange"]
    # two ranks, random sources: each needs nearly all of the other's rows (D = 128 floats each)
    assert ex["kind"] == "halo" and 0.8 * 169343 / 2 * 512 < ex["bytes_received_max_rank"] <= 169343 * 512
    assert abs(ex["edges_max_rank"] - 1166243 / 2) < 0.02 * 1166243       # edge-balanced cut


def test_gpus2_without_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has two GPUs: the launch would succeed")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], env={"STAG_BENCH_BACKEND": "nccl"})
    assert r.returncode != 0 and "Nothing was measured" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--rehearse", "--steps", "1", "--warmup", "0"],
             env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout


@pytest.mark.gpu
def test_gpus2_real_kernels_two_gloo_ranks_on_one_card():
    """The N = 2 step with real kernels (exchange + overlapped local rows + remote rows), gloo standing in
    for RCCL because two RCCL ranks cannot share a device; timings are meaningless, the plumbing is not."""
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
             env={"STAG_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = _line(r)
    assert line["n_gpus"] == 2 and line["config"]["partition"] == "nodes" and line["value"] > 0
    assert line["alt_partition"]["partition"] == "channels" and "error" not in line["alt_partition"]
    ex = line["exchange"]
    assert ex["exchange_only_us"] > 0 and ex["kernels_only_us"] > 0 and ex["local_units"] + ex["remote_units"] > 0


def _torchrun(args, env=None, timeout=600):
    """The driver's own launch line: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", *args]
    return subprocess.run(cmd, env=e, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_gpus2_under_torchrun_rehearsal_on_cpu():
    r = _torchrun(["--rehearse", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = _line(r)
    assert line["n_gpus"] == 2 and line["rehearsal"] is True and line["config"]["partition"] == "nodes"


@pytest.mark.gpu
def test_gpus2_under_torchrun_real_kernels():
    r = _torchrun(["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-alt"], env={"STAG_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = _line(r)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["exchange"]["kernels_only_us"] > 0


@pytest.mark.gpu
def test_single_gpu_line_contract():
    r = _run(["--steps", "20", "--warmup", "5", "--cpu-budget-s", "4"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = _line(r)
    assert line["n_gpus"] == 1 and line["config"]["partition"] == "none" and line["value"] > 1e9
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None or "NOT collected in this run" in rf["traffic_source"]
    assert rf["ceilings"]["gather"]["us"] > 0 and rf["ceilings"]["valu_rng"]["us"] > 0
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1

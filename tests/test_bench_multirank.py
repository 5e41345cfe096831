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


def test_launcher_counts_gpus_from_the_kfd_topology_without_hip(tmp_path, monkeypatch):
    """The launcher parent must not initialise a GPU before it starts its ranks: it reads the KFD
    topology — nodes with simd_count > 0 are GPUs — and honours the *_VISIBLE_DEVICES lists."""
    sys.path.insert(0, ROOT)
    import bench
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):         # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.node_gpu_count(str(tmp_path)) == (3, "kfd topology (sysfs)")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.node_gpu_count(str(tmp_path))[0] == 2
    src = open(os.path.join(ROOT, "bench.py")).read()
    launcher = src[src.index("def self_launch"):src.index("# ---", src.index("def self_launch"))]
    assert "torch.cuda" not in launcher, "the launcher parent must never ask the HIP runtime for anything"


def test_a_failing_rank_is_named_with_its_stderr():
    r = _run(["--gpus", "2", "--rehearse", "--steps", "1", "--warmup", "0", "--exchange", "halo"],
             env={"STAG_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0 and "rank 1 exited with code" in r.stderr and "STAG_BENCH_FAIL_RANK" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_a_rank_dying_behind_the_headline_does_not_lose_the_line():
    """An extra that takes a rank down hard (os._exit in rank 1 after the headline loop): the launcher stops the job,
    rank 0 dies in its collective — and its guardian process prints the line as far as it got, marked."""
    r = _run(["--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "1"], env={"STAG_BENCH_CRASH_IN_EXTRAS": "1"})
    assert r.returncode != 0 and "rank 1 exited with code 13" in r.stderr
    lines = [json.loads(t) for t in r.stdout.splitlines() if t.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and "ms_per_step" in lines[0]
    # (rank 0 either died in the collective — the guardian's line — or saw its peer vanish as an exception inside the extra)
    assert "extras_crashed" in lines[0] or "error" in lines[0]["exchange"]
    # ... and a clean run prints exactly one line, without the mark
    r = _run(["--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "1"])
    lines = [json.loads(t) for t in r.stdout.splitlines() if t.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1 and "extras_crashed" not in lines[0]


def test_a_rank_dying_behind_the_headline_under_torchrun():
    """The same under the driver's launcher: torchrun SIGTERMs rank 0 when rank 1 is gone; the guardian survives it."""
    r = _torchrun(["--rehearse", "--steps", "2", "--warmup", "1"], env={"STAG_BENCH_CRASH_IN_EXTRAS": "1"})
    lines = [json.loads(t) for t in r.stdout.splitlines() if t.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1 and lines[0]["n_gpus"] == 2
    assert "extras_crashed" in lines[0] or "error" in lines[0]["exchange"]


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


@pytest.mark.gpu
def test_gpus4_full_size_through_the_real_collective_is_bit_identical():
    """`bench.py --gpus 4` as the driver's multi-GPU run executes it, FOUR gloo ranks sharing the one card (the box
    admits six GPU processes: this test's own, and four ranks; never eight), the cfg2 graph and the cfg5 GAT at full
    size: the line's own `partition_check` — every rank launches the whole graph once and compares its rows with what
    the partitioned step produced through the REAL collective — must read bit-identical, forward with and without
    the overlap, for the aggregation (`_ShardAggregate`) and for BASELINE configs[4]'s GAT step (`_ShardGat`), and the
    backward through the transposed exchange within 1e-5 of the whole graph's."""
    r = _run(["--gpus", "4", "--steps", "5", "--warmup", "2", "--no-alt"], env={"STAG_BENCH_BACKEND": "gloo"}, timeout=1100)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = _line(r)
    assert line["n_gpus"] == 4 and len(line["device_ms_per_step_per_rank"]) == 4 and line["comm"]["backend"] == "gloo"
    for pc in (line["partition_check"], line["gat_partition"]["partition_check"]):
        assert "error" not in pc, pc
        assert pc["partition_bit_identical"] is True and pc["rows_compared"] == 169343 and pc["rows_differing"] == 0
        assert pc["backward_within_1e-5"] is True, pc
        assert len(set(pc["row_checksums_per_rank"])) == 4
    gp = line["gat_partition"]
    assert gp["value"] > 0 and gp["exchange"]["bytes_received_max_rank"] > 0 and gp["exchange"]["kernels_only_us"] > 0
    assert len(line["exchange"]["exchange_GBs_per_rank"]) == 4
    out = os.environ.get("STAG_REHEARSAL_OUT")
    if out:
        with open(out, "w") as f:
            f.write(json.dumps(line, indent=1) + "\n")


@pytest.mark.gpu
def test_the_multi_gpu_code_path_over_rccl_itself_with_one_rank():
    """RCCL refuses two ranks on one card, so the rehearsals above run over gloo.  What they cannot show is whether the
    N > 1 code path is well-formed ON THE nccl BACKEND — `init_process_group("nccl", device_id=...)`, barriers, all_reduce /
    all_gather of the line's float64 and int64 records, the partition checks, the library's own communicator as the second
    transport.  STAG_BENCH_FORCE_DIST=1 runs exactly that path with ONE rank over RCCL: every extra of the line must be
    there without an `error`."""
    r = _run(["--gpus", "1", "--steps", "5", "--warmup", "2"], env={"STAG_BENCH_FORCE_DIST": "1", "STAG_BENCH_BACKEND": "nccl"},
             timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = _line(r)
    assert line["n_gpus"] == 1 and line["comm"]["backend"] == "nccl" and line["comm"]["rccl_version"][0].isdigit()
    assert "extras_timed_out_after_s" not in line
    for key in ("exchange", "partition_check", "gat_partition", "alt_partition", "comm_variants"):
        assert key in line and "error" not in line[key], (key, line.get(key))
    assert line["partition_check"]["partition_bit_identical"] and line["partition_check"]["rows_compared"] == 169343
    gp = line["gat_partition"]
    assert "error" not in gp["partition_check"] and gp["partition_check"]["partition_bit_identical"] and "error" not in gp["exchange"]
    cv = line["comm_variants"]
    assert cv["torch_distributed"]["headline"] is True and cv["native_rccl_group"]["device_ms_per_step"] > 0
    assert cv["native_rccl_group"]["partition_check"]["partition_bit_identical"] is True


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

#!/usr/bin/env python
"""bench.py — aggregated edges/s of ONE stochastic-aggregation layer-forward on the
ogbn-arxiv-shaped synthetic CSR (BASELINE.json configs[1]: N=169,343, E=1,166,243,
D=128, fp32, Normal(1, 0.5) per-edge per-channel noise, one Monte-Carlo sample).

A "step" = one pass of the hot path: `ops.aggregate(graph, x, EdgeNoise)` =
noise draw + gather + weighted segmented sum, fresh Philox offset per step.  Inputs
are resident in HBM before the timed region.  `--workload gat` makes BASELINE configs[4]'s
layer-forward the step instead (arxiv GAT, 8 heads x 32, noise [E, 8]: `ops.gat_aggregate`).

`--gpus N` with N > 1: the SAME graph is node-range partitioned over N ranks
(stag_amd.partition.GraphShard: contiguous destination-row ranges cut at equal edge counts) and
a step is the RCCL exchange of the referenced source rows over xGMI + the local kernel, the rows
with only local sources overlapping the collective.  Total work is fixed => "strong" scaling.
Started without a launcher (`python bench.py --gpus N`), the script spawns its N ranks itself —
the parent counts the node's GPUs from the KFD topology in sysfs and never loads the HIP runtime —;
under `python -m torch.distributed.run` it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.

The N > 1 line carries its own correctness evidence (`partition_check`: every rank also runs the
WHOLE-graph launch once and compares its rows bit for bit with what the partitioned step produced
through the real collective; the backward against the whole graph's at 1e-5), the GAT partition of
BASELINE configs[4] in a second, shorter loop (`gat_partition`, with the same check), the
exchange-free channel partition (`alt_partition`), per-rank device times, the exchange alone and the
kernels alone (`exchange`), the RCCL version, and the two transports of the exchange as timed
variants (`comm_variants`: torch.distributed's all_to_all_single | the library's own grouped
ncclSend/ncclRecv, `stag_halo_exchange_multi`).  Everything after the headline loop is best-effort:
an exception becomes an `error` string in its object, a watchdog prints the line as far as it
got if an extra hangs, and if a rank dies hard behind the headline loop (the launcher then tears the
job down) rank 0's guardian process — forked before rank 0 touches the GPU — prints the last snapshot
of the line with `extras_crashed`.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects: `roofline`
(algorithmic bytes / device time of the op, measured with HIP events on the launch stream, against
the 8 TB/s HBM peak; `ceilings` = the two limits that bind before HBM does), `cpu_baseline` (the
oracle's reference-dataflow twin on the host cores; N=1 only; a baseline, not a target), `variants`
(N=1: the other single-GPU BASELINE configs and forms, each with its own device time and fraction).
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GAT_H, GAT_F = 8, 32    # BASELINE configs[4]: 8 heads, hidden 256
CHECK_OFFSET = 7777     # Philox offset of the launches the partition check compares


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 2000 x ~100 us: a fifth of a second.  The first ~15 launches after any host sync run at
    # ~140 us and the clocks keep rising for ~25 ms of sustained load (rocprofv3 kernel trace:
    # 121 -> 112.5 us over 200 launches), so a 200-step run reads higher than 1000+ steps.
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="agg", choices=["agg", "gat"],
                    help="agg (default): BASELINE configs[1], the metric's configuration; gat: configs[4]'s layer-forward "
                         "(arxiv GAT 8 x 32, noise [E, 8]) as the step — at N > 1 the node-range partitioned GAT")
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--noise", default="normal", choices=["normal", "uniform", "bernoulli", "none"])
    ap.add_argument("--graph", default="arxiv", choices=["arxiv", "arxiv_sym"],
                    help="arxiv: the 1,166,243-edge directed CSR; arxiv_sym: the script's "
                         "self-loop + reverse-edge variant (scripts/arxiv_mle/gcn/run.py:53-55)")
    ap.add_argument("--seg-len", type=int, default=64)
    ap.add_argument("--partition", default="nodes", choices=["nodes", "channels"],
                    help="N>1 only. nodes (default; BASELINE north_star, SURVEY.md 8e): dst-range shards + "
                         "RCCL exchange of the referenced source rows per step. channels: every rank keeps "
                         "the whole CSR and D/N channels; the step has no exchange (partition.ChannelShard)")
    ap.add_argument("--no-alt", action="store_true",
                    help="N>1: skip the second, shorter timed loop over the partition NOT chosen "
                         "(reported as `alt_partition` in the same JSON line)")
    ap.add_argument("--no-gat", action="store_true",
                    help="N>1, --workload agg: skip the shorter loop over the partitioned GAT step (`gat_partition`)")
    ap.add_argument("--no-check", action="store_true",
                    help="N>1: skip `partition_check` (whole-graph launch on every rank, bitwise comparison)")
    ap.add_argument("--no-comm-variants", action="store_true",
                    help="N>1 over RCCL: skip the second timed loop over the other transport of the exchange")
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"])
    ap.add_argument("--no-overlap", action="store_true",
                    help="N>1, nodes: launch all rows behind the collective instead of overlapping the "
                         "local-source rows with it")
    ap.add_argument("--native-comm", action="store_true",
                    help="N>1, nodes: the HEADLINE exchange through the library's own RCCL communicator "
                         "(stag_halo_exchange_multi, include/stag_hip.h) instead of torch.distributed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="N=1: skip the short loops over the other workload variants (`variants` in the line) and the "
                         "cold reading")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed launches of the same step before the W warm-up steps, until this much wall time "
                         "has passed: the card raises its clocks over the first ~100 ms of load, and a 20-step run "
                         "(2 ms of work) would otherwise time the ramp (DESIGN.md section 5); 0 = none")
    ap.add_argument("--extras-timeout-s", type=float, default=float(os.environ.get("STAG_BENCH_EXTRAS_TIMEOUT_S", "420")),
                    help="N>1: wall-clock budget of everything after the headline loop; when it runs out rank 0 prints "
                         "the line as far as it got and every rank exits")
    ap.add_argument("--rehearse", action="store_true",
                    help="plumbing check without a GPU: ranks, rendezvous (gloo), partition and exchange run on "
                         "CPU tensors, NO kernel is launched; the line carries rehearsal=true and value=null")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------- launcher
def node_gpu_count(topology="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs of this node as the KFD driver lists them, WITHOUT loading the HIP runtime (the launcher parent must
    not initialise a GPU before it starts its ranks): topology nodes with a non-zero simd_count are GPUs, the
    CPUs have simd_count 0.  Narrowed by ROCR_/HIP_/CUDA_VISIBLE_DEVICES.  Falls back to a throw-away child
    interpreter when sysfs is not readable.  -> (count, how)"""
    n, how = 0, "kfd topology (sysfs)"
    files = sorted(glob.glob(os.path.join(topology, "*", "properties")))
    readable = False
    for f in files:
        try:
            with open(f) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            readable = True
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            pass
    if not readable:
        how = "child interpreter (torch.cuda.device_count)"
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                               capture_output=True, text=True, timeout=300)
            n = int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
        except (subprocess.SubprocessError, ValueError):
            n = 0
    else:
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            v = os.environ.get(var)
            if v is not None:
                n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n, how


def self_launch(args):
    """`--gpus N` without a launcher: start the N ranks as child processes (fresh interpreters — this
    process never loads the HIP runtime) and relay rank 0's line.  A rank that fails is named with its
    exit code and the tail of its stderr.  -> exit code."""
    import socket
    backend = "gloo" if args.rehearse else os.environ.get("STAG_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        ndev, how = node_gpu_count()
        if ndev < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node exposes {ndev} GPU(s) [{how}]; RCCL needs one "
                             f"device per rank. Nothing was measured.\n")
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    tmp = tempfile.mkdtemp(prefix="stag_bench_")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        log = open(os.path.join(tmp, f"rank{r}.stderr"), "w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=log))
    rank_of = {p.pid: r for r, p in enumerate(procs)}
    rc, failed = 0, None
    deadline = time.time() + float(os.environ.get("STAG_BENCH_TIMEOUT_S", "1500"))
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc, failed = code, rank_of[p.pid]
        if rc != 0 or time.time() > deadline:      # one rank failed (or hung): stop the others, exactly these PIDs
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    for r, log in enumerate(logs):
        log.flush()
        log.seek(0)
        text = log.read()
        log.close()
        if rc != 0 and (r == failed or (failed is None and text.strip())):
            sys.stderr.write(f"bench.py: rank {r} " + (f"exited with code {rc}" if r == failed else "(stopped)") +
                             f"; last lines of its stderr:\n" + "\n".join(text.strip().splitlines()[-25:]) + "\n")
        elif rc == 0 and r == 0 and text.strip():
            sys.stderr.write(text)          # rank 0's warnings, as if it had written them itself
    if rc == 124 and failed is None:
        sys.stderr.write("bench.py: the ranks did not finish within STAG_BENCH_TIMEOUT_S; stopped.\n")
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return rc


# ------------------------------------------------------------------------------------------------- helpers
def make_noise(stag_amd, graph, D, kind, offset):
    from stag_amd import _lib
    if kind == "none":
        return None
    k, p0, p1 = {"normal": (_lib.NOISE_NORMAL, 1.0, 0.5),
                 "uniform": (_lib.NOISE_UNIFORM, 1.0 - 0.5 * 3 ** 0.5, 1.0 + 0.5 * 3 ** 0.5),
                 "bernoulli": (_lib.NOISE_BERNOULLI, 0.5, None)}[kind]
    return stag_amd.EdgeNoise(graph, D, k, p0, p1, seed=0x5747A6, offset=offset,
                              in_norm=(kind == "bernoulli"))


def cpu_baseline(src, dst, n, x, kind, budget_s):
    """Reference dataflow on the host: materialise w[E,D], x[src]*w, dst-segmented sum
    (oracle/stag_oracle.c: stag_agg_ref_dataflow_cpu), OpenMP over all host cores."""
    from oracle import oracle as O
    O.build()
    # the GPU box shares its host: 16 cores are this job's share (task statement)
    threads = min(len(os.sched_getaffinity(0)), int(os.environ.get("STAG_CPU_THREADS", "16")))
    O.set_threads(threads)
    indptr, indices, eid, _, _ = O.csr_build(src, dst, n, n)
    g = O.CsrGraph(indptr, indices, eid, n_src=n)
    E, D = len(src), x.shape[1]
    if kind == "none":
        spec = O.make_spec("none")
    elif kind == "normal":
        spec = O.make_spec("normal", 1.0, 0.5, seed=0x5747A6, Dn=D, n_edges=E)
    elif kind == "uniform":
        spec = O.make_spec("uniform", 1.0 - 0.5 * 3 ** 0.5, 1.0 + 0.5 * 3 ** 0.5, seed=0x5747A6, Dn=D, n_edges=E)
    else:
        spec = O.make_spec("bernoulli", 0.5, in_norm=True, seed=0x5747A6, Dn=D, n_edges=E)
    bufs = (np.empty((E, D), np.float32), np.empty((E, D), np.float32))
    times, fused = [], []
    t_all = time.perf_counter()
    while len(times) < 3 and (time.perf_counter() - t_all) < budget_s * 0.6:
        t0 = time.perf_counter()
        O.agg_ref_dataflow(g, src, dst, x, spec, bufs)
        times.append(time.perf_counter() - t0)
    while len(fused) < 2 and (time.perf_counter() - t_all) < budget_s:
        t0 = time.perf_counter()
        O.agg_fwd(g, x, spec)
        fused.append(time.perf_counter() - t0)
    t = float(np.median(times))
    out = {"value": E / t, "unit": "edges/s", "cores": threads, "kind": "port",
           "sample": f"full workload ({E} edges x {D} channels), {len(times)} passes of the "
                     f"reference dataflow (materialise w, gather*mul, segmented sum), median {t:.3f} s"}
    if fused:
        out["fused_twin_value"] = E / float(np.median(fused))
    return out


def committed_profile(args, world):
    """-> (HBM-side bytes per launch, file it came from) from the committed rocprofv3 --pmc passes of this
    same workload (profiles/<round>/bench_*_pmc_summary.json: FETCH_SIZE x2 per the gfx950 correction,
    plus WRITE_SIZE); (None, None) when no profile of this exact workload is committed.  NOT measured in
    this run: PMC collection needs its own rocprofv3 passes (tools/profile_bench.py)."""
    if world != 1:
        return None, None
    key = f"{args.graph}/{args.noise}/D{args.feat}/seg{args.seg_len}" if args.workload == "agg" else \
        f"{args.graph}/gat{GAT_H}x{GAT_F}/{args.noise}/seg{args.seg_len}"
    best = (None, None)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "bench_*_pmc_summary.json"))):
        try:
            t = json.load(open(f)).get("traffic", {})
            if t.get("workload") == key:
                best = (float(t["traffic_bytes_per_launch"]), os.path.relpath(f, ROOT))   # the latest round
        except (ValueError, KeyError):
            pass
    return best


class Watchdog:
    """Everything after the headline loop of an N > 1 run is best-effort.  If it has not finished after `seconds`
    (a collective that never completes), rank 0 prints the line as far as it got and every rank leaves."""

    def __init__(self, seconds, rank, line_fn, guardian=None):
        self.rank, self.line_fn, self.done, self.guardian = rank, line_fn, threading.Event(), guardian
        self.t = threading.Thread(target=self._run, args=(seconds,), daemon=True)
        self.t.start()

    def _run(self, seconds):
        if self.done.wait(seconds):
            return
        if self.rank == 0:
            line = self.line_fn()
            line["extras_timed_out_after_s"] = seconds
            if self.guardian is not None:
                self.guardian.final(line)
            else:
                print(json.dumps(line), flush=True)
        sys.stderr.write(f"bench.py: rank {self.rank}: the extra loops did not finish in {seconds:.0f} s; "
                         f"the headline measurement stands, leaving.\n")
        sys.stderr.flush()
        os._exit(0)

    def cancel(self):
        self.done.set()


class Guardian:
    """Rank 0 of an N > 1 run keeps a forked child — forked BEFORE this process touches the GPU; the child never
    does — that holds the line as far as rank 0 got.  If rank 0 dies before it has printed the line itself (an
    extra crashed a rank and the launcher tore the job down, a fault inside a collective), the child prints the
    last snapshot with `extras_crashed`: the headline measurement of the first multi-GPU run is not lost with
    the extras behind it.  Nothing is printed when rank 0 never got as far as the headline.  With a guardian, rank 0's own
    finished line goes out through it as well (`final`): whatever kills rank 0, and whenever, stdout carries ONE line."""

    def __init__(self):
        import signal
        r, w = os.pipe()
        sys.stdout.flush()
        sys.stderr.flush()
        pid = os.fork()
        if pid == 0:                    # the child: pipe -> memory; os-level calls only
            try:
                os.close(w)
                try:
                    os.setsid()         # a launcher that signals rank 0's process group does not take the child along
                except OSError:
                    pass
                for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
                    signal.signal(sig, signal.SIG_IGN)
                data = b""
                while True:
                    chunk = os.read(r, 1 << 16)
                    if not chunk:
                        break
                    data += chunk
                msgs = [m for m in data.decode("utf-8", "replace").split("\n") if m]
                if msgs and msgs[-1].startswith("FINAL "):         # rank 0 finished: its line, as it is
                    os.write(1, (msgs[-1][6:] + "\n").encode())
                elif msgs:
                    snap = json.loads(msgs[-1])
                    snap["extras_crashed"] = ("rank 0 ended before finishing this line (an extra after the headline loop "
                                              "failed hard on some rank); printed by its guardian process from the last "
                                              "snapshot: everything present was complete when it was taken")
                    os.write(1, (json.dumps(snap) + "\n").encode())
            finally:
                os._exit(0)
        os.close(r)
        self.w, self.pid = w, pid

    def update(self, line):
        if self.w is not None and line is not None:
            try:
                os.write(self.w, (json.dumps(line) + "\n").encode())
            except OSError:               # the guardian is gone: rank 0 carries on without it
                pass

    def final(self, line):
        """The finished line goes out THROUGH the guardian (exactly one line whatever happens to this process between
        here and its exit); returns when the guardian has written it."""
        if self.w is None:                # (second call: the line is out already)
            return
        try:
            os.write(self.w, ("FINAL " + json.dumps(line) + "\n").encode())
            os.close(self.w)
        except OSError:                   # the guardian is gone: print it ourselves
            print(json.dumps(line), flush=True)
        finally:
            self.w = None
        try:
            os.waitpid(self.pid, 0)
        except OSError:
            pass


# ------------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # STAG_BENCH_FORCE_DIST=1 (tests): the N > 1 code path — process group, partition, exchange, every extra of the line — with
    # ONE rank: on a one-GPU box it is the only way to run that path over RCCL itself ("nccl"), which refuses two ranks on a card
    multi = world > 1 or os.environ.get("STAG_BENCH_FORCE_DIST") == "1"
    rehearse = args.rehearse
    if os.environ.get("STAG_BENCH_FAIL_RANK") == str(rank) and world > 1:      # test hook: how the launcher reports a dead rank
        raise SystemExit(f"bench.py: rank {rank} asked to fail (STAG_BENCH_FAIL_RANK)")
    guardian = None
    if world > 1 and rank == 0:           # forked before anything below initialises the GPU
        try:
            guardian = Guardian()
        except OSError as exc:            # no fork: the line is printed the plain way
            sys.stderr.write(f"bench.py: no guardian process ({exc}); continuing without one\n")
    if not rehearse and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the stochastic-aggregation path has no CPU fallback "
                         "(--rehearse checks the multi-rank plumbing without one)")
    if rehearse:
        dev = torch.device("cpu")
    else:
        local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    backend = None
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:        # (the forced one-rank form, started without a launcher)
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", str(world))
        # RCCL ("nccl") always in a real run; STAG_BENCH_BACKEND=gloo only rehearses the N>1 code
        # path with several ranks sharing one card (RCCL refuses two ranks on one device)
        backend = "gloo" if rehearse else os.environ.get("STAG_BENCH_BACKEND", "nccl")
        import datetime
        if backend == "nccl":
            # a collective that never completes should end the run in minutes, not in half an hour
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=5))

    import stag_amd
    from stag_amd import _lib, ops, synthetic
    from stag_amd.partition import GraphShard

    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    if args.graph == "arxiv_sym":
        src, dst = synthetic.with_self_loops_and_reverse(src, dst, n)
    E, D = len(src), args.feat
    H, F = GAT_H, GAT_F
    gen0 = torch.Generator().manual_seed(0)
    x_host = torch.randn(n, D, generator=gen0)
    gat_host = None

    def gat_inputs():
        """el, er [N, H], ft [N, H, F] of the GAT step (seeded, the same on every rank)."""
        nonlocal gat_host
        if gat_host is None:
            g = torch.Generator().manual_seed(5)
            gat_host = (torch.randn(n, H, generator=g), torch.randn(n, H, generator=g), torch.randn(n, H, F, generator=g))
        return gat_host

    def sync():
        if not rehearse:
            torch.cuda.synchronize()

    def fence():
        sync()
        if multi:
            dist.barrier()
            sync()

    _whole = {}

    def whole_graph():
        if "g" not in _whole:
            _whole["g"] = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
            _whole["g"].csr.plan(args.seg_len)
        return _whole["g"]

    _shards = {}

    def node_shard():
        if "s" not in _shards:
            _shards["s"] = GraphShard(src, dst, n, rank, world, device=dev, exchange=args.exchange)
        return _shards["s"]

    def use_native(shard, on):
        """Switch the shard's exchange between torch.distributed (None) and the library's own communicator."""
        if not on:
            shard.native_comm = None
            return
        if "native" not in _shards:
            from stag_amd.partition import NativeComm
            _shards["native"] = NativeComm(rank, world, dev)
        shard.native_comm = _shards["native"]

    def gat_noise(graph, i, pos_base=None):
        nz = make_noise(stag_amd, graph, H, args.noise, i)
        if nz is not None and pos_base is not None:
            nz.pos_base = pos_base
        return nz

    def make_step(workload, partition):
        """-> (step(i), description, parts).  All inputs end up resident in HBM here.  parts: for the node
        partition, the exchange alone and the kernels alone (timed separately for the `exchange` object)."""
        if not multi:
            graph = whole_graph()
            if workload == "gat":
                el, er, ft = (t.to(dev) for t in gat_inputs())
                graph.csr.plan(args.seg_len, need=True)
                return (lambda i: ops.gat_aggregate(graph, el, er, ft, 0.2, gat_noise(graph, i),
                                                    seg_len=args.seg_len)), "single GPU", None
            x = x_host.to(dev)
            return (lambda i: ops.aggregate(graph, x, make_noise(stag_amd, graph, D, args.noise, i),
                                            seg_len=args.seg_len)), "single GPU", None
        if partition == "channels":
            if workload == "gat":
                raise ValueError("the channel partition is for the aggregation workload")
            from stag_amd.partition import ChannelShard
            whole = whole_graph()
            shard = ChannelShard(whole, D, rank, world)
            x = shard.scatter_cols(x_host).to(dev)
            return (lambda i: shard.aggregate(x, make_noise(stag_amd, whole, shard.dn, args.noise, i),
                                              seg_len=args.seg_len)), (
                f"channel shards x{world}: whole CSR per rank, D/{world} channels each, "
                f"no exchange in the step"), None
        shard = node_shard()
        overlap = not args.no_overlap
        coll = "RCCL" if dist.get_backend() == "nccl" else f"{dist.get_backend()} (rehearsal backend, not RCCL)"
        desc = (f"node-range partition x{world}: dst-row ranges cut at equal edge counts, {coll} "
                f"{'all-to-all of the referenced source rows (halo)' if args.exchange == 'halo' else 'all-gather of padded row shards'}"
                f" per step over xGMI" + (", local-source rows overlap the collective" if overlap else ""))
        if workload == "gat":
            el_h, er_h, ft_h = gat_inputs()
            lo, hi = shard.row_lo, shard.row_hi
            el, er, ft = el_h[lo:hi].to(dev), er_h[lo:hi].to(dev), ft_h[lo:hi].to(dev)

            def exchange_only(i):
                bufs, work = shard.halo_start_multi([ft, el], persistent=True)
                if work is not None:
                    work.wait()
                return bufs[0]
            if rehearse:
                return (lambda i: exchange_only(i)), desc, {"shard": shard}
            shard.csr.plan(args.seg_len, need=True)
            p_loc, p_rem = shard.plan_split(args.seg_len)
            ft0, el0 = (b.clone() for b in shard.halo_gather_multi([ft, el]))

            def kernels_only(i):
                return ops.gat_aggregate(shard, el0, er, ft0, 0.2, gat_noise(shard, i, shard.pos_base),
                                         seg_len=args.seg_len, _gathered=True)
            return (lambda i: shard.gat_aggregate(el, er, ft, 0.2, gat_noise(shard, i), seg_len=args.seg_len,
                                                  overlap=overlap)), desc + "; two tables per step: ft [n, 8, 32], el [n, 8]", {
                "shard": shard, "exchange_only": exchange_only, "kernels_only": kernels_only,
                "local_units": p_loc["n_units"], "remote_units": p_rem["n_units"], "inputs": (el, er, ft)}
        # this rank's rows live INSIDE the shard's persistent exchange buffer (where a previous layer would have
        # written them): the step copies nothing into it, the send rows go into a persistent send buffer
        x = shard.local_rows(D, device=dev)
        x.copy_(x_host[shard.row_lo:shard.row_hi])

        def exchange_only(i):
            buf, work = shard.halo_start(x, persistent=True)
            if work is not None:
                work.wait()
            return buf

        if rehearse:          # no kernel exists on the CPU: the step is the exchange alone
            return (lambda i: exchange_only(i)), desc, {"shard": shard}
        shard.csr.plan(args.seg_len)
        p_loc, p_rem = shard.plan_split(args.seg_len)
        buf0 = shard.halo_gather(x).clone()         # a filled buffer for the kernels-only loop

        def make_noise_on_shard(i):
            nz = make_noise(stag_amd, shard, D, args.noise, i)
            if nz is not None:
                nz.pos_base = shard.pos_base
            return nz

        def kernels_only(i):
            return ops.aggregate(shard, buf0, make_noise_on_shard(i), seg_len=args.seg_len, _gathered=True)
        return (lambda i: shard.aggregate(x, make_noise(stag_amd, shard, D, args.noise, i),
                                          seg_len=args.seg_len, overlap=overlap)), desc, {
            "shard": shard, "exchange_only": exchange_only, "kernels_only": kernels_only,
            "local_units": p_loc["n_units"], "remote_units": p_rem["n_units"], "inputs": (x,)}

    def timed(step, steps, warmup, per_rank=False):
        """-> (wall seconds, device ms per step), MAX over ranks; barrier + sync both sides.
        per_rank: also every rank's own device ms per step."""
        with torch.no_grad():
            for i in range(warmup):
                step(i)
            # one HIP event pair brackets the K steps on the launch stream (an event pair per step
            # adds ~14 us of queue bubbles per step and would be charged to the kernel)
            if not rehearse:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fence()
            t0 = time.perf_counter()
            if not rehearse:
                ev0.record()
            for i in range(steps):
                out = step(warmup + i)
            if not rehearse:
                ev1.record()
            fence()
            t1 = time.perf_counter()
        wall = t1 - t0
        dev_ms = ev0.elapsed_time(ev1) / steps if not rehearse else wall / steps * 1e3
        ranks_ms = None
        if multi:
            if per_rank:
                own = torch.tensor([dev_ms], dtype=torch.float64, device=dev)
                allr = [torch.zeros_like(own) for _ in range(world)]
                dist.all_gather(allr, own)
                ranks_ms = [float(t[0]) for t in allr]
            t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(t[0]), float(t[1])
        assert torch.isfinite(out).all()
        return (wall, dev_ms, ranks_ms) if per_rank else (wall, dev_ms)

    def settle(step):
        """Untimed launches of the step until --settle-ms of wall time has gone by (every rank, same count)."""
        if rehearse or args.settle_ms <= 0:
            return 0
        done, t_end = 0, time.perf_counter() + args.settle_ms * 1e-3
        with torch.no_grad():
            while True:
                for i in range(50):
                    step(i)
                sync()
                done += 50
                more = torch.tensor([1.0 if time.perf_counter() < t_end else 0.0], device=dev)
                if multi:
                    dist.all_reduce(more, op=dist.ReduceOp.MIN)   # ranks must agree: the step is a collective
                if float(more) == 0.0:
                    return done

    def b_alg_of(workload, n_, E_, D_=None):
        """indptr + indices + the source table once + the output once (+ el / er for GAT)."""
        if workload == "gat":
            return 4 * (n_ + 1) + 4 * E_ + 8 * n_ * H + 2 * 4 * n_ * H * F
        return 4 * (n_ + 1) + 4 * E_ + 8 * n_ * D_

    # ---------------------------------------------------------------------------- partition check (N > 1)
    def scaled_err(a, b):
        return float(((a - b).abs() / (1.0 + b.abs())).max()) if a.numel() else 0.0

    def partition_check(workload, parts):
        """Every rank runs the WHOLE-graph launch once and compares ITS rows with what the partitioned step —
        through the real collective — produced: forward bit for bit, with and without the overlap; backward
        (`_ShardAggregate` / `_ShardGat`: the transposed exchange, fixed-order adds) at 1e-5 against the
        whole graph's gradient rows (one GPU adds a source row's out-edges in one sum, P ranks add P partial
        sums: fp32 does not regroup)."""
        shard = parts["shard"]
        lo, hi = shard.row_lo, shard.row_hi
        whole = whole_graph()
        res = {}
        if workload == "gat":
            el_h, er_h, ft_h = gat_inputs()
            el_w, er_w, ft_w = (t.to(dev).requires_grad_(True) for t in (el_h, er_h, ft_h))
            gout = torch.randn(n, H, F, generator=torch.Generator().manual_seed(11)).to(dev)
            whole.csr.plan(args.seg_len, need=True)
            ref = ops.gat_aggregate(whole, el_w, er_w, ft_w, 0.2, gat_noise(whole, CHECK_OFFSET), seg_len=args.seg_len)
            ref.backward(gout)
            el, er, ft = (t.detach().clone().requires_grad_(True) for t in parts["inputs"])
            got = shard.gat_aggregate(el, er, ft, 0.2, gat_noise(shard, CHECK_OFFSET), seg_len=args.seg_len, overlap=True)
            got.backward(gout[lo:hi])
            with torch.no_grad():
                got2 = shard.gat_aggregate(el, er, ft, 0.2, gat_noise(shard, CHECK_OFFSET), seg_len=args.seg_len,
                                           overlap=False)
            berr = max(scaled_err(ft.grad, ft_w.grad[lo:hi]), scaled_err(el.grad, el_w.grad[lo:hi]),
                       scaled_err(er.grad, er_w.grad[lo:hi]))
        else:
            x_w = x_host.to(dev).requires_grad_(True)
            gout = torch.randn(n, D, generator=torch.Generator().manual_seed(11)).to(dev)
            ref = ops.aggregate(whole, x_w, make_noise(stag_amd, whole, D, args.noise, CHECK_OFFSET), seg_len=args.seg_len)
            ref.backward(gout)
            x = parts["inputs"][0].detach().clone().requires_grad_(True)
            got = shard.aggregate(x, make_noise(stag_amd, shard, D, args.noise, CHECK_OFFSET), seg_len=args.seg_len,
                                  overlap=True)
            got.backward(gout[lo:hi])
            with torch.no_grad():
                got2 = shard.aggregate(x, make_noise(stag_amd, shard, D, args.noise, CHECK_OFFSET), seg_len=args.seg_len,
                                       overlap=False)
            berr = scaled_err(x.grad, x_w.grad[lo:hi])
        ref_rows = ref.detach()[lo:hi].reshape(hi - lo, -1)
        got_rows, got2_rows = got.detach().reshape(hi - lo, -1), got2.reshape(hi - lo, -1)
        bad = int((got_rows.view(torch.int32) != ref_rows.view(torch.int32)).any(1).sum())
        bad2 = int((got2_rows.view(torch.int32) != ref_rows.view(torch.int32)).any(1).sum())
        # a 64-bit checksum of this rank's output bits, position-weighted: listed per rank in the line
        bits = got_rows.view(torch.int32).to(torch.int64)
        wts = (torch.arange(bits.shape[1], device=dev, dtype=torch.int64) * 2 + 1).unsqueeze(0)
        cks = int(((bits * wts).sum(1) * (torch.arange(lo, hi, device=dev, dtype=torch.int64) * 2 + 1)).sum())
        t = torch.tensor([bad, bad2, hi - lo], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        e = torch.tensor([berr], dtype=torch.float64, device=dev)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        own = torch.tensor([cks], dtype=torch.int64, device=dev)
        allc = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(allc, own)
        res.update({"partition_bit_identical": bool(int(t[0]) == 0 and int(t[1]) == 0),
                    "rows_compared": int(t[2]), "rows_differing": int(t[0]), "rows_differing_without_overlap": int(t[1]),
                    "backward_max_scaled_err_vs_whole_graph": float(e[0]), "backward_within_1e-5": bool(float(e[0]) <= 1e-5),
                    "row_checksums_per_rank": [f"{int(c[0]) & ((1 << 64) - 1):016x}" for c in allc],
                    "what": "every rank launched the whole graph once (Philox offset %d) and compared its own rows with the "
                            "partitioned step's through the real collective: forward bitwise (overlap on and off), "
                            "backward (transposed exchange) at 1e-5" % CHECK_OFFSET})
        return res

    def exchange_report(parts, width_floats, steps_cap):
        """bytes of the exchange and, from separate short loops, the collective alone / the kernels alone."""
        shard = parts["shard"]
        rb, sb = shard.exchange_bytes(width_floats)
        own = torch.tensor([rb, sb, shard.number_of_edges(), shard.n_rows], dtype=torch.float64, device=dev)
        cnt, mx = own.clone(), own.clone()
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        allr = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(allr, own)
        ex = {"kind": args.exchange, "bytes_received_all_ranks": float(cnt[0]),
              "bytes_received_max_rank": float(mx[0]), "bytes_sent_max_rank": float(mx[1]),
              "edges_max_rank": float(mx[2]), "rows_max_rank": float(mx[3]),
              "bytes_received_per_rank": [float(t[0]) for t in allr], "edges_per_rank": [float(t[2]) for t in allr]}
        if not rehearse:
            k2 = max(1, min(args.steps, steps_cap))
            _, ex_ms, ex_ranks = timed(parts["exchange_only"], k2, min(args.warmup, 10), per_rank=True)
            _, kr_ms, kr_ranks = timed(parts["kernels_only"], k2, min(args.warmup, 10), per_rank=True)
            ex.update({"exchange_only_us": ex_ms * 1e3, "kernels_only_us": kr_ms * 1e3, "steps": k2,
                       "exchange_only_us_per_rank": [v * 1e3 for v in ex_ranks],
                       "kernels_only_us_per_rank": [v * 1e3 for v in kr_ranks],
                       "exchange_GBs_max_rank": float(mx[0]) / (ex_ms * 1e-3) / 1e9 if ex_ms > 0 else None,
                       "exchange_GBs_per_rank": [float(t[0]) / (v * 1e-3) / 1e9 if v > 0 else None
                                                 for t, v in zip(allr, ex_ranks)],
                       "local_units": parts["local_units"], "remote_units": parts["remote_units"],
                       "note": "separate short loops after the headline loop: the collective alone "
                               "(the send rows gathered into the persistent send buffer + all-to-all) and the local kernels alone "
                               "on an already exchanged buffer; device time, max over ranks and per rank"})
        return ex

    # ---------------------------------------------------------------------------- the headline loop
    partition = args.partition
    workload = args.workload
    step, parallelism, parts = make_step(workload, partition)
    if multi and args.native_comm and not rehearse and parts is not None:
        use_native(parts["shard"], True)
    cold = None
    if not multi and not rehearse and not args.no_variants:
        # the same K steps BEFORE the settle phase: what a short run reads while the card is still raising its
        # clocks — reported beside the headline number, never as it
        cw, cd = timed(step, args.steps, args.warmup)
        cold = {"ms_per_step": cw / args.steps * 1e3, "device_ms_per_step": cd, "steps": args.steps,
                "warmup": args.warmup, "note": "timed before the settle phase (clocks still rising); the headline "
                                               "loop below runs after it"}
    settle_steps = settle(step)
    if multi:
        wall, dev_ms, dev_ms_ranks = timed(step, args.steps, args.warmup, per_rank=True)
    else:
        (wall, dev_ms), dev_ms_ranks = timed(step, args.steps, args.warmup), None

    if guardian is not None:             # the bare headline, the moment it exists (the full line replaces it a few ms later)
        guardian.update({"metric": "aggregated edges/sec, stochastic-aggregation layer-forward, ogbn-arxiv-shaped CSR",
                         "value": None if rehearse else E / (wall / args.steps), "unit": "edges/s", "n_gpus": world,
                         "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
                         "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                         "config": {"workload": f"{workload} on the arxiv-shaped CSR, N={n}, E={E}", "parallelism": parallelism,
                                    "partition": partition},
                         "device_ms_per_step": dev_ms, "snapshot": "headline only: rank 0 ended before it had formed the full line"})

    def variant_loops():
        """Short timed loops over the other single-GPU BASELINE configs and forms, each with its own
        device time and fraction of the HBM roofline on its own algorithmic bytes: configs[1] after the script's own
        preprocessing (scripts/arxiv_mle/gcn/run.py:53-55: E = 2,671,154), with the script's Bernoulli + in-norm
        noise (:70-74) and without noise; configs[2] (PPI batch, GraphSAGE mean, D = 256: XCD-aware walk and plan
        order); configs[3] (4096 molecules, GIN sum, D = 128); configs[4]'s layer on one GPU (arxiv GAT 8 x 32)."""
        out = {}
        ks, kw = 200, 20
        graphs = {"arxiv": (src, dst)}

        def entry(Ev, w_, d_ms, balg, **kv):
            return dict(kv, E=Ev, steps=ks, ms_per_step=w_ / ks * 1e3, device_ms_per_step=d_ms, edges_per_s=Ev / (w_ / ks),
                        algorithmic_bytes_per_step=balg, frac=balg / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
        if workload == "agg":
            for gname, noise in (("arxiv", "bernoulli"), ("arxiv", "none"), ("arxiv_sym", "normal"), ("arxiv_sym", "bernoulli")):
                if gname == args.graph and noise == args.noise:
                    continue
                if gname not in graphs:
                    graphs[gname] = synthetic.with_self_loops_and_reverse(*synthetic.arxiv_like(seed=1), n)
                s_, d_ = graphs[gname]
                gkey = "_g_" + gname
                if gkey not in graphs:
                    graphs[gkey] = stag_amd.Graph(torch.from_numpy(s_), torch.from_numpy(d_), n, device=dev)
                    graphs[gkey].csr.plan(args.seg_len)
                gr, xv, Ev = graphs[gkey], x_host.to(dev), len(s_)
                st = lambda i, gr=gr, xv=xv, noise=noise: ops.aggregate(gr, xv, make_noise(stag_amd, gr, D, noise, i), seg_len=args.seg_len)
                w_, d_ms = timed(st, ks, kw)
                out[f"{gname}/{noise}"] = entry(Ev, w_, d_ms, b_alg_of("agg", n, Ev, D), graph=gname,
                                                noise=noise + ("+in_norm" if noise == "bernoulli" else ""))
        # BASELINE configs[4]'s layer-forward on ONE GPU (the partitioned form is the N > 1 line's gat_partition)
        if workload == "agg":
            gw = whole_graph()
            gw.csr.plan(args.seg_len, need=True)
            el, er, ft = (t.to(dev) for t in gat_inputs())
            for noise in ("normal", "none"):
                st = lambda i, noise=noise: ops.gat_aggregate(gw, el, er, ft, 0.2, make_noise(stag_amd, gw, H, noise, i),
                                                              seg_len=args.seg_len)
                settle(st)
                w_, d_ms = timed(st, ks, kw)
                out[f"arxiv_gat/{noise}"] = entry(E, w_, d_ms, b_alg_of("gat", n, E), noise=noise,
                                                  graph="BASELINE configs[4] on one GPU: arxiv GAT forward, H=8, F=32, noise [E, 8]")
            del el, er, ft
        # BASELINE configs[2]: the PPI-sized block-diagonal batch (scripts/ppi_mle/run.py:12-14; GraphSAGE mean, hidden
        # 256) — the workload the XCD-aware walk is for: the same launches with it and in plan order
        s3, d3, sizes3 = synthetic.ppi_like()
        n3, E3, D3 = int(sizes3.sum()), len(s3), 256
        x3 = torch.randn(n3, D3, device=dev)
        balg3 = b_alg_of("agg", n3, E3, D3)
        for order in ("xcd", "plan_order"):
            g3 = stag_amd.Graph(torch.from_numpy(s3), torch.from_numpy(d3), n3,
                                batch_num_nodes=torch.from_numpy(sizes3).to(dev), device=dev)
            p3 = g3.csr.plan(args.seg_len)
            if order == "xcd":
                g3.csr._add_xcd_order(p3)                 # what "auto" does for this graph after 16 launches
            else:
                p3["xcd_decided"] = True                  # never
            for noise in ("normal", "none"):
                st = lambda i, noise=noise: ops.aggregate(g3, x3, make_noise(stag_amd, g3, D3, noise, i), reduce="mean",
                                                          seg_len=args.seg_len)
                settle(st)                     # a VALU-bound launch reads its clocks: the headline's settle phase again
                w_, d_ms = timed(st, ks, kw)
                out[f"ppi_batch/{noise}/{order}"] = entry(
                    E3, w_, d_ms, balg3, graph="BASELINE configs[2]: 24 PPI-sized graphs batched (block-diagonal), SAGE mean, D=256",
                    noise=noise, unit_order=order, stripe_locality=g3.csr.stripe_locality())
        del x3, g3
        # BASELINE configs[3]: 4096 molecules batched, GIN sum aggregation at the hidden width
        s4, d4, sizes4 = synthetic.molecules_like(4096)
        n4, E4, D4 = int(sizes4.sum()), len(s4), 128
        g4 = stag_amd.Graph(torch.from_numpy(s4), torch.from_numpy(d4), n4,
                            batch_num_nodes=torch.from_numpy(sizes4).to(dev), device=dev)
        x4 = torch.randn(n4, D4, device=dev)
        for noise in ("normal", "none"):
            st = lambda i, noise=noise: ops.aggregate(g4, x4, make_noise(stag_amd, g4, D4, noise, i), seg_len=args.seg_len)
            settle(st)                         # (also takes the view past the launches after which it gets its plan and order)
            w_, d_ms = timed(st, ks, kw)
            out[f"molecule_batch/{noise}"] = entry(E4, w_, d_ms, b_alg_of("agg", n4, E4, D4), noise=noise,
                                                   graph="BASELINE configs[3]: 4096 molecules batched (block-diagonal), GIN sum, D=128",
                                                   stripe_locality=g4.csr.stripe_locality())
        return out

    variants = variant_loops() if (not multi and not rehearse and not args.no_variants) else None

    # ---------------------------------------------------------------------------- the line, as far as the headline
    line = None
    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        b_alg = b_alg_of(workload, n, E, D)
        achieved = b_alg / (dev_ms * 1e-3) / 1e9
        traffic, traffic_source = committed_profile(args, 2 if multi else 1)
        row_bytes = 4 * (H * F if workload == "gat" else D)
        b_gather = b_alg - 4 * n * (row_bytes // 4) + E * row_bytes   # SURVEY §8d B_gather: every edge pulls its row
        n_blocks = E * (((H if workload == "gat" else D) + 3) // 4)
        if workload == "gat":
            wl = (f"BASELINE configs[4]'s layer-forward: ogbn-arxiv-shaped synthetic CSR ({args.graph}), N={n}, E={E}, "
                  f"GAT {H} heads x {F}, fp32, int32 CSR, noise={args.noise}(per edge, per head, Philox4x32-10) on the "
                  f"logits, edge softmax + weighted sum, 1 MC sample")
            note = ("one step = one stag_gat_fwd call = ONE kernel launch (gat_fwd_block_kernel); device time = HIP event "
                    "pair around the K launches on the launch stream / K; gather-bound (1-KB rows), see DESIGN.md 4.2")
        else:
            wl = (f"BASELINE configs[1]: ogbn-arxiv-shaped synthetic CSR ({args.graph}), N={n}, E={E}, D={D}, fp32, "
                  f"int32 CSR, noise={args.noise}(per edge, per channel, Philox4x32-10), 1 layer-forward, 1 MC sample")
            note = ("one step = one stag_agg_fwd call = ONE kernel launch (agg_kernel); device time = "
                    "HIP event pair around the K launches on the launch stream / K; D=128 per-channel "
                    "Normal noise is RNG(VALU)- and gather-bound, see DESIGN.md")
        line = {
            "metric": "aggregated edges/sec, stochastic-aggregation layer-forward, ogbn-arxiv-shaped CSR",
            "value": None if rehearse else E / (wall / args.steps), "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "settle_ms": args.settle_ms, "settle_steps": settle_steps,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl, "parallelism": parallelism, "partition": "none" if not multi else partition,
                       "seg_len": args.seg_len},
            # N > 1: whole-job algorithmic bytes over the slowest rank's device time, against N x 8 TB/s
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBS * world), "traffic": traffic,
                         "traffic_source": (f"{traffic_source} (committed rocprofv3 --pmc passes of this workload; "
                                            f"NOT collected in this run)") if traffic_source else None,
                         "algorithmic_bytes_per_step": b_alg, "bytes_per_edge": b_alg / E,
                         "device_ms_per_step": dev_ms,
                         # the two limits that bind before HBM streaming does on this workload:
                         # numbers are per launch on ONE GPU, measured with tools/ubench_valu.hip and PMC
                         "ceilings": {
                             "gather": {"bytes": b_gather, "rate_TBs": 7.4,
                                        "us": b_gather / 7.4e12 * 1e6,
                                        "why": "uniform-random sources: every edge pulls its whole source row, which misses "
                                               "the 4 MB per-XCD L2 (PMC: FETCH x2 = E * row bytes); rows come from the "
                                               "Infinity Cache at the guide's measured random-row rate, 7.4-7.9 TB/s"},
                             "valu_rng": {"philox_blocks": n_blocks, "cycles_per_block_per_wave": 315,
                                          "us": n_blocks / 64 * 315 / 1024 / 2.3e9 * 1e6 if args.noise == "normal" else None,
                                          "why": "one Philox4x32-10 block (40 VALU) + 4 Box-Muller normals (24 VALU, 8 of them "
                                                 "transcendental) + the multiply-adds = ~315 issue cycles per wave at the 4.4 cycles "
                                                 "per VALU instruction PMC reads, 1024 SIMDs at 2.3 GHz; the launch without its row "
                                                 "gathers (ids + draw + adds: -DSTAG_EXP_NO_ROWS) reads 95 us, DESIGN.md section 5"},
                             "frac_if_at_max_of_ceilings": None},
                         "note": note},
        }
        c = line["roofline"]["ceilings"]
        lim = max(c["gather"]["us"], c["valu_rng"]["us"] or 0.0)
        c["frac_if_at_max_of_ceilings"] = b_alg / (lim * 1e-6) / 1e9 / HBM_PEAK_GBS
        if rehearse:
            line["rehearsal"] = True
            line["roofline"] = None
            line["note"] = ("plumbing rehearsal on CPU tensors over gloo: ranks, rendezvous, partition and "
                            "exchange ran, NO kernel was launched, nothing here is a measurement")
        line["cpu_baseline"] = None
        if cold is not None:
            line["cold"] = cold
        if variants is not None:
            line["variants"] = variants
        if multi:
            line["device_ms_per_step_per_rank"] = dev_ms_ranks
            line["comm"] = {"backend": dist.get_backend(),
                            "headline_transport": "stag_halo_exchange_multi (library-owned RCCL communicator)"
                            if (args.native_comm and not rehearse) else "torch.distributed all_to_all_single / all_gather_into_tensor"}
            if not rehearse:
                try:
                    line["comm"]["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception as exc:        # noqa: BLE001 — diagnostics only
                    line["comm"]["rccl_version"] = f"unavailable ({type(exc).__name__})"
                line["comm"]["hip"], line["comm"]["torch"] = torch.version.hip, torch.__version__

    # ---------------------------------------------------------------------------- N > 1: everything after the headline
    def extra(name, fn):
        """Run one best-effort extra on EVERY rank; rank 0 records its result, or the error."""
        try:
            val = fn()
        except Exception as exc:        # noqa: BLE001 — the headline stands on its own
            import traceback
            val = {"error": f"{type(exc).__name__}: {exc}", "where": traceback.format_exc().strip().splitlines()[-3:]}
        if line is not None and val is not None:
            line[name] = val
        if guardian is not None:
            guardian.update(line)

    if guardian is not None:
        guardian.update(line)            # the headline is safe from here on
    if multi and parts is not None:
        if os.environ.get("STAG_BENCH_CRASH_IN_EXTRAS") == str(rank):      # test hook: a rank that dies hard behind the headline
            time.sleep(2.0)              # (somewhere inside the extras, not in the instant rank 0 is still forming its line)
            os._exit(13)
        dog = Watchdog(args.extras_timeout_s, rank, lambda: dict(line or {}), guardian) if not rehearse else None
        width = (H * F + H) if workload == "gat" else D
        extra("exchange", lambda: exchange_report(parts, width, 200))
        if not rehearse and not args.no_check:
            extra("partition_check", lambda: partition_check(workload, parts))

        if workload == "agg" and not rehearse and not args.no_gat and partition == "nodes":
            def gat_partition():
                """BASELINE configs[4] as written: the node-range partitioned GAT layer-forward, timed in its own
                shorter loop, with its own exchange report and partition check."""
                step_g, par_g, parts_g = make_step("gat", "nodes")
                kg = max(1, min(args.steps, 200))
                settle(step_g)
                w_, d_ms, ranks_ms = timed(step_g, kg, min(args.warmup, 10), per_rank=True)
                balg = b_alg_of("gat", n, E)
                res = {"workload": f"BASELINE configs[4]: arxiv GAT {H} heads x {F}, noise [E, {H}] = {args.noise}, "
                                   f"node-range partitioned x{world}", "parallelism": par_g, "steps": kg,
                       "ms_per_step": w_ / kg * 1e3, "value": E / (w_ / kg), "unit": "edges/s",
                       "device_ms_per_step": d_ms, "device_ms_per_step_per_rank": ranks_ms,
                       "algorithmic_bytes_per_step": balg,
                       "frac": balg / (d_ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)}
                try:
                    res["exchange"] = exchange_report(parts_g, H * F + H, 100)
                except Exception as exc:        # noqa: BLE001
                    res["exchange"] = {"error": f"{type(exc).__name__}: {exc}"}
                if not args.no_check:
                    try:
                        res["partition_check"] = partition_check("gat", parts_g)
                    except Exception as exc:        # noqa: BLE001
                        res["partition_check"] = {"error": f"{type(exc).__name__}: {exc}"}
                return res
            extra("gat_partition", gat_partition)

        if not args.no_alt and not rehearse and workload == "agg":
            def alt():
                other = "nodes" if partition == "channels" else "channels"
                step2, par2, _ = make_step("agg", other)
                k2 = max(1, min(args.steps, 50))
                w2, d2 = timed(step2, k2, min(args.warmup, 5))
                return {"partition": other, "parallelism": par2, "steps": k2, "ms_per_step": w2 / k2 * 1e3,
                        "value": E / (w2 / k2), "unit": "edges/s", "device_ms_per_step": d2}
            extra("alt_partition", alt)

        if not rehearse and not args.no_comm_variants and partition == "nodes" and backend == "nccl":
            def comm_variants():
                """The same headline step over the OTHER transport of the exchange (last: the library's own
                communicator has only ever run with one rank before the first multi-GPU run)."""
                shard = parts["shard"]
                k2 = max(1, min(args.steps, 200))
                res = {}
                head = "native_rccl_group" if args.native_comm else "torch_distributed"
                res[head] = {"ms_per_step": wall / args.steps * 1e3, "device_ms_per_step": dev_ms, "steps": args.steps,
                             "headline": True}
                other = "torch_distributed" if args.native_comm else "native_rccl_group"
                use_native(shard, not args.native_comm)
                try:
                    w2, d2 = timed(step, k2, min(args.warmup, 10))
                    res[other] = {"ms_per_step": w2 / k2 * 1e3, "device_ms_per_step": d2, "steps": k2}
                    _, ex2 = timed(parts["exchange_only"], k2, min(args.warmup, 10))
                    res[other]["exchange_only_us"] = ex2 * 1e3
                    if not args.no_check:
                        res[other]["partition_check"] = partition_check(workload, parts)
                finally:
                    use_native(shard, args.native_comm)
                res["what"] = ("torch_distributed: dist.all_to_all_single on RCCL's stream; native_rccl_group: "
                               "stag_halo_exchange_multi — grouped ncclSend/ncclRecv on a library-owned communicator "
                               "and side stream, no torch.distributed on the data path")
                return res
            extra("comm_variants", comm_variants)
        if dog is not None:
            dog.cancel()

    if rank == 0:
        if not multi and not args.no_cpu_baseline and not rehearse and workload == "agg":
            line["cpu_baseline"] = cpu_baseline(src, dst, n, x_host.numpy(), args.noise, args.cpu_budget_s)
        if guardian is not None:
            guardian.final(line)
        else:
            print(json.dumps(line), flush=True)
    if multi:
        try:
            dist.destroy_process_group()
        except Exception:        # noqa: BLE001 — the line is out
            pass


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py — aggregated edges/s of ONE stochastic-aggregation layer-forward on the
ogbn-arxiv-shaped synthetic CSR (BASELINE.json configs[1]: N=169,343, E=1,166,243,
D=128, fp32, Normal(1, 0.5) per-edge per-channel noise, one Monte-Carlo sample).

A "step" = one pass of the hot path: `ops.aggregate(graph, x, EdgeNoise)` =
noise draw + gather + weighted segmented sum, fresh Philox offset per step.  Inputs
are resident in HBM before the timed region.

`--gpus N` with N > 1: the SAME graph is node-range partitioned over N ranks
(stag_amd.partition.GraphShard: contiguous destination-row ranges cut at equal edge counts) and
a step is the RCCL exchange of the referenced source rows over xGMI + the local kernel, the rows
with only local sources overlapping the collective.  Total work is fixed => "strong" scaling.
Started without a launcher (`python bench.py --gpus N`), the script spawns its N ranks itself
before anything touches a GPU; under `python -m torch.distributed.run` it reads RANK / LOCAL_RANK /
WORLD_SIZE from the environment.  `--partition channels` (the exchange-free alternative for graphs
that fit one GPU) is timed in a second, shorter loop and reported as `alt_partition`.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects: `roofline`
(algorithmic bytes / device time of the op, measured with HIP events on the launch stream, against
the 8 TB/s HBM peak; `ceilings` = the two limits that bind before HBM does), `cpu_baseline` (the
oracle's reference-dataflow twin on the host cores; N=1 only; a baseline, not a target) and, for
N > 1, `exchange` (bytes and device time of the collective alone and of the kernels alone).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 2000 x ~115 us: a quarter of a second.  The first ~15 launches after any host sync run at
    # ~140 us and the clocks keep rising for ~25 ms of sustained load (rocprofv3 kernel trace:
    # 121 -> 112.5 us over 200 launches), so a 200-step run reads 121 us, 1000+ steps 113 us.
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
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
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"])
    ap.add_argument("--no-overlap", action="store_true",
                    help="N>1, nodes: launch all rows behind the collective instead of overlapping the "
                         "local-source rows with it")
    ap.add_argument("--native-comm", action="store_true",
                    help="N>1, nodes: the exchange through the library's own RCCL communicator "
                         "(stag_halo_exchange, include/stag_hip.h) instead of torch.distributed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="N=1: skip the short loops over the other workload variants (`variants` in the line: the "
                         "script's preprocessed graph arxiv_sym, Bernoulli + in-norm, no noise) and the cold reading")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="untimed launches of the same step before the W warm-up steps, until this much wall time "
                         "has passed: the card raises its clocks over the first ~100 ms of load, and a 20-step run "
                         "(2 ms of work) would otherwise time the ramp (DESIGN.md section 5); 0 = none")
    ap.add_argument("--rehearse", action="store_true",
                    help="plumbing check without a GPU: ranks, rendezvous (gloo), partition and exchange run on "
                         "CPU tensors, NO kernel is launched; the line carries rehearsal=true and value=null")
    return ap.parse_args(argv)


def self_launch(args):
    """`--gpus N` without a launcher: start the N ranks as child processes (fresh interpreters — this
    process has not touched a GPU and never will) and relay rank 0's line.  -> exit code."""
    import socket
    backend = "gloo" if args.rehearse else os.environ.get("STAG_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        ndev = torch.cuda.device_count()          # counts devices without initialising the runtime
        if ndev < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node exposes {ndev} GPU(s); RCCL needs one "
                             f"device per rank. Nothing was measured.\n")
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + float(os.environ.get("STAG_BENCH_TIMEOUT_S", "1500"))
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if rc != 0 or time.time() > deadline:      # one rank failed (or hung): stop the others, exactly these PIDs
            for p in procs:
                p.terminate()
            for p in procs:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            return rc or 124
        time.sleep(0.05)
    return rc


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
    key = f"{args.graph}/{args.noise}/D{args.feat}/seg{args.seg_len}"
    best = (None, None)
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "bench_*_pmc_summary.json"))):
        try:
            t = json.load(open(f)).get("traffic", {})
            if t.get("workload") == key:
                best = (float(t["traffic_bytes_per_launch"]), os.path.relpath(f, ROOT))   # the latest round
        except (ValueError, KeyError):
            pass
    return best


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
    rehearse = args.rehearse
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    from stag_amd import ops, synthetic
    from stag_amd.partition import GraphShard

    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    if args.graph == "arxiv_sym":
        src, dst = synthetic.with_self_loops_and_reverse(src, dst, n)
    E, D = len(src), args.feat
    x_host = torch.randn(n, D, generator=torch.Generator().manual_seed(0))

    def sync():
        if not rehearse:
            torch.cuda.synchronize()

    def fence():
        sync()
        if world > 1:
            dist.barrier()
            sync()

    def make_step(partition):
        """-> (step(i), description, parts).  All inputs end up resident in HBM here.  parts: for the node
        partition, the exchange alone and the kernels alone (timed separately for the `exchange` object)."""
        if world == 1:
            graph = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
            graph.csr.plan(args.seg_len)
            x = x_host.to(dev)
            return (lambda i: ops.aggregate(graph, x, make_noise(stag_amd, graph, D, args.noise, i),
                                            seg_len=args.seg_len)), "single GPU", None
        if partition == "channels":
            from stag_amd.partition import ChannelShard
            whole = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
            whole.csr.plan(args.seg_len)
            shard = ChannelShard(whole, D, rank, world)
            x = shard.scatter_cols(x_host).to(dev)
            return (lambda i: shard.aggregate(x, make_noise(stag_amd, whole, shard.dn, args.noise, i),
                                              seg_len=args.seg_len)), (
                f"channel shards x{world}: whole CSR per rank, D/{world} channels each, "
                f"no exchange in the step"), None
        shard = GraphShard(src, dst, n, rank, world, device=dev, exchange=args.exchange)
        # this rank's rows live INSIDE the shard's persistent exchange buffer (where a previous layer would have
        # written them): the step copies nothing into it, the send rows go into a persistent send buffer
        x = shard.local_rows(D, device=dev)
        x.copy_(x_host[shard.row_lo:shard.row_hi])
        if args.native_comm and not rehearse:
            from stag_amd.partition import NativeComm
            shard.native_comm = NativeComm(rank, world, dev)
        overlap = not args.no_overlap
        coll = "RCCL" if dist.get_backend() == "nccl" else f"{dist.get_backend()} (rehearsal backend, not RCCL)"
        desc = (f"node-range partition x{world}: dst-row ranges cut at equal edge counts, {coll} "
                f"{'all-to-all of the referenced source rows (halo)' if args.exchange == 'halo' else 'all-gather of padded row shards'}"
                f" per step over xGMI" + (", local-source rows overlap the collective" if overlap else ""))
        def exchange_only(i):
            buf, work = shard.halo_start(x, persistent=True)
            if work is not None:
                work.wait()
            return buf

        if rehearse:          # no kernel exists on the CPU: the step is the exchange alone
            return (lambda i: exchange_only(i)), desc, {"shard": shard}
        shard.csr.plan(args.seg_len)
        p_loc, p_rem = shard.plan_split(args.seg_len)
        buf0 = shard.halo_gather(x)         # a filled buffer for the kernels-only loop

        def kernels_only(i):
            return ops.aggregate(shard, buf0, make_noise_on_shard(i), seg_len=args.seg_len, _gathered=True)

        def make_noise_on_shard(i):
            nz = make_noise(stag_amd, shard, D, args.noise, i)
            if nz is not None:
                nz.pos_base = shard.pos_base
            return nz
        return (lambda i: shard.aggregate(x, make_noise(stag_amd, shard, D, args.noise, i),
                                          seg_len=args.seg_len, overlap=overlap)), desc, {
            "shard": shard, "exchange_only": exchange_only, "kernels_only": kernels_only,
            "local_units": p_loc["n_units"], "remote_units": p_rem["n_units"]}

    def timed(step, steps, warmup):
        """-> (wall seconds, device ms per step), MAX over ranks; barrier + sync both sides."""
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
        if world > 1:
            t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(t[0]), float(t[1])
        assert torch.isfinite(out).all()
        return wall, dev_ms

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
                if world > 1:
                    dist.all_reduce(more, op=dist.ReduceOp.MIN)   # ranks must agree: the step is a collective
                if float(more) == 0.0:
                    return done

    partition = args.partition
    step, parallelism, parts = make_step(partition)
    cold = None
    if world == 1 and not rehearse and not args.no_variants:
        # the same K steps BEFORE the settle phase: what a short run reads while the card is still raising its
        # clocks — reported beside the headline number, never as it
        cw, cd = timed(step, args.steps, args.warmup)
        cold = {"ms_per_step": cw / args.steps * 1e3, "device_ms_per_step": cd, "steps": args.steps,
                "warmup": args.warmup, "note": "timed before the settle phase (clocks still rising); the headline "
                                               "loop below runs after it"}
    settle_steps = settle(step)
    wall, dev_ms = timed(step, args.steps, args.warmup)

    def variant_loops():
        """Short timed loops over the other workload variants of BASELINE configs[1]: the graph after
        the script's own preprocessing (scripts/arxiv_mle/gcn/run.py:53-55: E = 2,671,154), the script's
        Bernoulli + in-norm noise (:70-74), and no noise (the plain gather); each with its own device time and
        fraction of the HBM roofline on its own algorithmic bytes."""
        out = {}
        ks, kw = 200, 20
        graphs = {"arxiv": (src, dst)}
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
            balg = 4 * (n + 1) + 4 * Ev + 8 * n * D
            out[f"{gname}/{noise}"] = {"graph": gname, "noise": noise + ("+in_norm" if noise == "bernoulli" else ""),
                                       "E": Ev, "steps": ks, "ms_per_step": w_ / ks * 1e3, "device_ms_per_step": d_ms,
                                       "edges_per_s": Ev / (w_ / ks), "algorithmic_bytes_per_step": balg,
                                       "frac": balg / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # BASELINE configs[2]: the PPI-sized block-diagonal batch (scripts/ppi_mle/run.py:12-14; GraphSAGE mean, hidden
        # 256) — the workload the XCD-aware walk is for: the same launches with it and in plan order
        s3, d3, sizes3 = synthetic.ppi_like()
        n3, E3, D3 = int(sizes3.sum()), len(s3), 256
        x3 = torch.randn(n3, D3, device=dev)
        balg3 = 4 * (n3 + 1) + 4 * E3 + 8 * n3 * D3
        for order in ("xcd", "plan_order"):
            g3 = stag_amd.Graph(torch.from_numpy(s3), torch.from_numpy(d3), n3, device=dev)
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
                out[f"ppi_batch/{noise}/{order}"] = {
                    "graph": "24 PPI-sized graphs batched (block-diagonal), SAGE mean, D=256", "noise": noise, "unit_order": order,
                    "stripe_locality": g3.csr.stripe_locality(), "E": E3, "steps": ks, "ms_per_step": w_ / ks * 1e3,
                    "device_ms_per_step": d_ms, "edges_per_s": E3 / (w_ / ks), "algorithmic_bytes_per_step": balg3,
                    "frac": balg3 / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        return out

    variants = variant_loops() if (world == 1 and not rehearse and not args.no_variants) else None

    exchange = None
    if world > 1 and parts is not None:
        shard = parts["shard"]
        rb, sb = shard.exchange_bytes(D)
        cnt = torch.tensor([rb, sb, shard.number_of_edges(), shard.n_rows], dtype=torch.float64, device=dev)
        mx = cnt.clone()
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        exchange = {"kind": args.exchange, "bytes_received_all_ranks": float(cnt[0]),
                    "bytes_received_max_rank": float(mx[0]), "bytes_sent_max_rank": float(mx[1]),
                    "edges_max_rank": float(mx[2]), "rows_max_rank": float(mx[3])}
        if not rehearse:
            k2 = max(1, min(args.steps, 200))
            _, ex_ms = timed(parts["exchange_only"], k2, min(args.warmup, 10))
            _, kr_ms = timed(parts["kernels_only"], k2, min(args.warmup, 10))
            exchange.update({"exchange_only_us": ex_ms * 1e3, "kernels_only_us": kr_ms * 1e3, "steps": k2,
                             "exchange_GBs_max_rank": float(mx[0]) / (ex_ms * 1e-3) / 1e9 if ex_ms > 0 else None,
                             "local_units": parts["local_units"], "remote_units": parts["remote_units"],
                             "note": "separate short loops after the headline loop: the collective alone "
                                     "(the send rows gathered into the persistent send buffer + all-to-all) and the local kernels alone "
                                     "on an already exchanged buffer; device time, max over ranks"})

    alt = None
    if world > 1 and not args.no_alt and not rehearse:
        other = "nodes" if partition == "channels" else "channels"
        try:
            step2, par2, _ = make_step(other)
            k2 = max(1, min(args.steps, 50))
            w2, d2 = timed(step2, k2, min(args.warmup, 5))
            alt = {"partition": other, "parallelism": par2, "steps": k2, "ms_per_step": w2 / k2 * 1e3,
                   "value": E / (w2 / k2), "unit": "edges/s", "device_ms_per_step": d2}
        except Exception as exc:   # the headline partition's numbers stand on their own
            alt = {"partition": other, "error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        b_alg = 4 * (n + 1) + 4 * E + 4 * n * D + 4 * n * D   # indptr + indices + x once + out once
        achieved = b_alg / (dev_ms * 1e-3) / 1e9
        traffic, traffic_source = committed_profile(args, world)
        b_gather = 4 * (n + 1) + 4 * E + 4 * E * D + 4 * n * D   # SURVEY §8d B_gather: every edge pulls its row
        n_blocks = E * ((D + 3) // 4)
        line = {
            "metric": "aggregated edges/sec, stochastic-aggregation layer-forward, ogbn-arxiv-shaped CSR",
            "value": None if rehearse else E / (wall / args.steps), "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "settle_ms": args.settle_ms, "settle_steps": settle_steps,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: ogbn-arxiv-shaped synthetic CSR "
                                   f"({args.graph}), N={n}, E={E}, D={D}, fp32, int32 CSR, "
                                   f"noise={args.noise}(per edge, per channel, Philox4x32-10), "
                                   f"1 layer-forward, 1 MC sample",
                       "parallelism": parallelism, "partition": "none" if world == 1 else partition,
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
                                        "why": "uniform-random sources: every edge pulls a D*4-byte row that misses the "
                                               "4 MB per-XCD L2 (PMC: FETCH x2 = E*D*4); rows come from the Infinity "
                                               "Cache at the guide's measured random-row rate, 7.4-7.9 TB/s"},
                             "valu_rng": {"philox_blocks": n_blocks, "cycles_per_block_per_wave": 315,
                                          "us": n_blocks / 64 * 315 / 1024 / 2.3e9 * 1e6 if args.noise == "normal" else None,
                                          "why": "one Philox4x32-10 block + 4 Box-Muller normals = 315 issue cycles per "
                                                 "wave (20 v_mad_u64_u32 at 7.6 + transcendentals), 1024 SIMDs at 2.3 GHz"},
                             "frac_if_at_max_of_ceilings": None},
                         "note": "one step = one stag_agg_fwd call = ONE kernel launch (agg_kernel); device time = "
                                 "HIP event pair around the K launches on the launch stream / K; D=128 per-channel "
                                 "Normal noise is RNG(VALU)- and gather-bound, see DESIGN.md"},
        }
        c = line["roofline"]["ceilings"]
        lim = max(c["gather"]["us"], c["valu_rng"]["us"] or 0.0)
        c["frac_if_at_max_of_ceilings"] = b_alg / (lim * 1e-6) / 1e9 / HBM_PEAK_GBS
        if rehearse:
            line["rehearsal"] = True
            line["roofline"] = None
            line["note"] = ("plumbing rehearsal on CPU tensors over gloo: ranks, rendezvous, partition and "
                            "exchange ran, NO kernel was launched, nothing here is a measurement")
        if world == 1 and not args.no_cpu_baseline and not rehearse:
            line["cpu_baseline"] = cpu_baseline(src, dst, n, x_host.numpy(), args.noise, args.cpu_budget_s)
        else:
            line["cpu_baseline"] = None
        if cold is not None:
            line["cold"] = cold
        if variants is not None:
            line["variants"] = variants
        if exchange is not None:
            line["exchange"] = exchange
        if alt is not None:
            line["alt_partition"] = alt
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

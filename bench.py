#!/usr/bin/env python
"""bench.py — aggregated edges/s of ONE stochastic-aggregation layer-forward on the
ogbn-arxiv-shaped synthetic CSR (BASELINE.json configs[1]: N=169,343, E=1,166,243,
D=128, fp32, Normal(1, 0.5) per-edge per-channel noise, one Monte-Carlo sample).

A "step" = one pass of the hot path: `ops.aggregate(graph, x, EdgeNoise)` =
noise draw + gather + weighted segmented sum, fresh Philox offset per step.  Inputs
are resident in HBM before the timed region.  With --gpus N > 1 the same graph is
node-range partitioned over N ranks (stag_amd.partition) and a step is the halo
exchange (RCCL) + the local kernel: total work fixed => "strong" scaling.
(--partition channels: the exchange-free alternative for graphs that fit one GPU.)

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: `roofline` (algorithmic bytes / device time of the op, measured with HIP
events on the launch stream, against the 8 TB/s HBM peak) and `cpu_baseline` (the
oracle's reference-dataflow twin on the host cores; N=1 only; a baseline, not a target).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
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
    ap.add_argument("--partition", default="auto", choices=["auto", "nodes", "channels"],
                    help="N>1 only. nodes: dst-range shards + RCCL exchange of the referenced source "
                         "rows per step (BASELINE north_star: graphs larger than one GPU). channels: "
                         "every rank keeps the whole CSR and D/N channels; the step has no exchange "
                         "(partition.ChannelShard). auto: channels when the whole graph fits one GPU "
                         "(it does for the arxiv CSR), else nodes")
    ap.add_argument("--no-alt", action="store_true",
                    help="N>1: skip the second, shorter timed loop over the partition NOT chosen "
                         "(reported as `alt_partition` in the same JSON line)")
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    return ap.parse_args()


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


def measured_traffic(args, world):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this same workload
    (profiles/<round>/bench_*_pmc_summary.json: FETCH_SIZE x2 per the gfx950 correction, plus
    WRITE_SIZE); None when no profile of this exact workload is committed."""
    if world != 1:
        return None
    key = f"{args.graph}/{args.noise}/D{args.feat}/seg{args.seg_len}"
    best = None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "bench_*_pmc_summary.json"))):
        try:
            t = json.load(open(f)).get("traffic", {})
            if t.get("workload") == key:
                best = float(t["traffic_bytes_per_launch"])     # the latest round that measured it
        except (ValueError, KeyError):
            pass
    return best


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the stochastic-aggregation path has no CPU fallback")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") always in a real run; STAG_BENCH_BACKEND=gloo only rehearses the N>1 code
        # path with several ranks sharing one card (RCCL refuses two ranks on one device)
        backend = os.environ.get("STAG_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            import datetime
            # a collective that never completes should end the run in minutes, not in half an hour
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))
        else:
            dist.init_process_group(backend)

    import stag_amd
    from stag_amd import ops, synthetic
    from stag_amd.partition import GraphShard

    src, dst = synthetic.arxiv_like(seed=1)
    n = synthetic.ARXIV_NODES
    if args.graph == "arxiv_sym":
        src, dst = synthetic.with_self_loops_and_reverse(src, dst, n)
    E, D = len(src), args.feat
    x_host = torch.randn(n, D, generator=torch.Generator().manual_seed(0))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def make_step(partition):
        """-> (step(i), description).  All inputs end up resident in HBM here."""
        if world == 1:
            graph = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
            graph.csr.plan(args.seg_len)
            x = x_host.to(dev)
            return (lambda i: ops.aggregate(graph, x, make_noise(stag_amd, graph, D, args.noise, i),
                                            seg_len=args.seg_len)), "single GPU"
        if partition == "channels":
            from stag_amd.partition import ChannelShard
            whole = stag_amd.Graph(torch.from_numpy(src), torch.from_numpy(dst), n, device=dev)
            whole.csr.plan(args.seg_len)
            shard = ChannelShard(whole, D, rank, world)
            x = shard.scatter_cols(x_host).to(dev)
            return (lambda i: shard.aggregate(x, make_noise(stag_amd, whole, shard.dn, args.noise, i),
                                              seg_len=args.seg_len)), (
                f"channel shards x{world}: whole CSR per rank, D/{world} channels each, "
                f"no exchange in the step")
        shard = GraphShard(src, dst, n, rank, world, device=dev, exchange=args.exchange)
        shard.csr.plan(args.seg_len)
        x = x_host[shard.row_lo:shard.row_hi].to(dev)
        return (lambda i: shard.aggregate(x, make_noise(stag_amd, shard, D, args.noise, i),
                                          seg_len=args.seg_len)), (
            f"dst-range partition x{world} + RCCL {args.exchange} exchange of the referenced "
            f"source rows per step")

    def timed(step, steps, warmup):
        """-> (wall seconds, device ms per step), MAX over ranks; barrier + sync both sides."""
        with torch.no_grad():
            for i in range(warmup):
                step(i)
            # one HIP event pair brackets the K steps on the launch stream (an event pair per step
            # adds ~14 us of queue bubbles per step and would be charged to the kernel)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fence()
            t0 = time.perf_counter()
            ev0.record()
            for i in range(steps):
                out = step(warmup + i)
            ev1.record()
            fence()
            t1 = time.perf_counter()
        wall, dev_ms = t1 - t0, ev0.elapsed_time(ev1) / steps
        if world > 1:
            t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall, dev_ms = float(t[0]), float(t[1])
        assert torch.isfinite(out).all()
        return wall, dev_ms

    # the arxiv CSR (5 MB) and x (87 MB) fit any GPU: auto => channel shards
    fits = (4 * (n + 1) + 8 * E + 8 * n * D) < 0.5 * torch.cuda.get_device_properties(dev).total_memory
    partition = args.partition if args.partition != "auto" else ("channels" if fits else "nodes")
    step, parallelism = make_step(partition)
    wall, dev_ms = timed(step, args.steps, args.warmup)

    alt = None
    if world > 1 and not args.no_alt:
        other = "nodes" if partition == "channels" else "channels"
        try:
            step2, par2 = make_step(other)
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
        line = {
            "metric": "aggregated edges/sec, stochastic-aggregation layer-forward, ogbn-arxiv-shaped CSR",
            "value": E / (wall / args.steps), "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: ogbn-arxiv-shaped synthetic CSR "
                                   f"({args.graph}), N={n}, E={E}, D={D}, fp32, int32 CSR, "
                                   f"noise={args.noise}(per edge, per channel, Philox4x32-10), "
                                   f"1 layer-forward, 1 MC sample",
                       "parallelism": parallelism,
                       "seg_len": args.seg_len},
            # N > 1: whole-job algorithmic bytes over the slowest rank's device time, against N x 8 TB/s
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK_GBS * world), "traffic": measured_traffic(args, world),
                         "algorithmic_bytes_per_step": b_alg, "bytes_per_edge": b_alg / E,
                         "device_ms_per_step": dev_ms,
                         "note": "one step = one stag_agg_fwd call = ONE kernel launch (agg_kernel); "
                                 "device time = HIP event pair around the K launches on the launch "
                                 "stream / K; D=128 per-channel Normal noise is RNG(VALU)- and "
                                 "gather-bound, see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(src, dst, n, x_host.numpy(), args.noise, args.cpu_budget_s)
        else:
            line["cpu_baseline"] = None
        if alt is not None:
            line["alt_partition"] = alt
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

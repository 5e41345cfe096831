#!/usr/bin/env python
"""rocprofv3 evidence for BASELINE configs 3, 4 and 5 (tools/bench_configs.py), written to
gpurun_out/profiles/<round>/ (run on the GPU box from the repo root; copy into profiles/<round>/):

    python tools/profile_configs.py --round r02 [--only cfg5,cfg5_train,cfg3,cfg3_l1,cfg4,cfg4_l1]

Per config, four processes (counters never share a run with a trace domain):
  1. rocprofv3 --kernel-trace --stats -> <cfg>_kernel_stats.csv, <cfg>.json (the script's own line)
  2. --pmc FETCH_SIZE   3. --pmc WRITE_SIZE   4. --pmc <SQ counters>  -> <cfg>_pmc_summary.json:
     per-dispatch averages for every stag kernel of the config, gfx950 FETCH_SIZE correction applied.
This script never touches the GPU itself; the profiled program is `python3 tools/bench_configs.py`
directly after `--`.
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Two SQ passes, each with its OWN cycle counter: a ratio is formed from counters of one pass only (round 3 divided
# counters of one replay by the cycles of another, at another clock: "VALU busy" 1.049 — VERDICT r03)
SQ_A = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"]
SQ_B = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"]


def derived(a, b):
    """Ratios from the two SQ passes (a: SQ_A's per-dispatch averages, b: SQ_B's), each normalised by the cycles counted
    in ITS pass.  GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_ACTIVE_INST_VALU counts quad-cycles (x4 = SIMD cycles
    issuing VALU; 1024 SIMDs).  A busy fraction cannot exceed 1: a raw value above it (counter granularity, the replay's
    clock) is reported as 1.0 with `valu_saturated`."""
    out = {}
    if b.get("GRBM_GUI_ACTIVE"):
        cyc = b["GRBM_GUI_ACTIVE"] / 8.0
        raw = b["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
        out.update({"valu_busy_frac": min(raw, 1.0), "valu_busy_frac_raw": raw, "valu_saturated": raw >= 0.995,
                    "cycles_per_valu_inst": b["SQ_ACTIVE_INST_VALU"] * 4.0 / max(b["SQ_INSTS_VALU"], 1.0),
                    "xcd_cycles_valu_pass": cyc})
    if a.get("GRBM_GUI_ACTIVE"):
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        out.update({"avg_waves_per_simd": a["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * cyc),
                    "wait_frac_of_wave_life": a["SQ_WAIT_ANY"] / max(a["SQ_WAVE_CYCLES"], 1.0),
                    "xcd_cycles_wave_pass": cyc})
    if a.get("SQ_WAVES") and b.get("SQ_INSTS_VALU"):
        out["valu_insts_per_wave"] = b["SQ_INSTS_VALU"] / max(a["SQ_WAVES"], 1.0)    # (both are counts: no clock in it)
    return out


KERNELS = re.compile(r"agg_kernel|gat_\w+_kernel|segment_reduce_kernel|agg_bwd_w_kernel|noise_materialize")


def run_prof(tag, prof_args, prog_args, scratch):
    out = os.path.join(scratch, tag)
    cmd = ["rocprofv3", *prof_args, "-d", out, "-o", tag, "--output-format", "csv", "--",
           "python3", "tools/bench_configs.py", *prog_args]
    print("+", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-4000:])
        raise SystemExit(f"{tag}: rocprofv3 exited {r.returncode}")
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and '"config"' in l]
    return out, lines


def find(out, suffix):
    hits = glob.glob(os.path.join(out, "**", f"*{suffix}"), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {out}")
    return hits[0]


def counters(out):
    """{kernel name: (dispatches, {counter: average per dispatch})} for the stag kernels."""
    per, disp = {}, {}
    for row in csv.DictReader(open(find(out, "counter_collection.csv"))):
        name = row["Kernel_Name"]
        if not KERNELS.search(name):
            continue
        disp.setdefault(name, set()).add(row["Dispatch_Id"])
        per.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: (len(disp[k]), {c: sum(v) / max(len(disp[k]), 1) for c, v in per[k].items()}) for k in per}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r02")
    ap.add_argument("--only", default="cfg5,cfg5_train,cfg3,cfg3_l1,cfg4,cfg4_l1")
    ap.add_argument("--noise", default="normal")
    ap.add_argument("--suffix", default="", help="appended to the file names (e.g. _none)")
    args = ap.parse_args()
    dst = os.path.join(ROOT, "gpurun_out", "profiles", args.round)
    os.makedirs(dst, exist_ok=True)
    for cfg in [c for c in args.only.split(",") if c]:
        tag = cfg + args.suffix
        scratch = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
        prog = ["--only", cfg, "--noise", args.noise]
        out, lines = run_prof("stats", ["--kernel-trace", "--stats"], [*prog, "--steps", "100", "--warmup", "10"], scratch)
        stats = find(out, "kernel_stats.csv")
        rows = [r for r in csv.DictReader(open(stats))]
        with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
            f.write(open(stats).read())
        with open(os.path.join(dst, f"{tag}.json"), "w") as f:
            json.dump(lines, f, indent=1)
            f.write("\n")
        avg_ns = {r["Name"]: (float(r["AverageNs"]), int(r["Calls"])) for r in rows if KERNELS.search(r["Name"])}
        print("kernel stats:", {k[:70]: v for k, v in avg_ns.items()}, flush=True)
        summary = {"config": cfg, "noise": args.noise, "script_lines": lines, "kernels": {}}
        passes = {}
        for ptag, ctrs in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]), ("pmc_sq_a", SQ_A), ("pmc_sq_b", SQ_B)):
            out, _ = run_prof(ptag, ["--pmc", *ctrs], [*prog, "--steps", "10", "--warmup", "2", "--settle-ms", "0"], scratch)
            passes[ptag] = counters(out)
        for name in sorted(set().union(*[set(p) for p in passes.values()])):
            k = {"avg_us_stats_pass": avg_ns.get(name, (None, 0))[0] and avg_ns[name][0] / 1e3,
                 "calls_stats_pass": avg_ns.get(name, (None, 0))[1]}
            c, per_pass = {}, {}
            for ptag in passes:
                if name in passes[ptag]:
                    k[f"dispatches_{ptag}"] = passes[ptag][name][0]
                    per_pass[ptag] = passes[ptag][name][1]
                    c.update({(f"{cn}@{ptag}" if cn == "GRBM_GUI_ACTIVE" else cn): v for cn, v in passes[ptag][name][1].items()})
            k["counters_avg_per_dispatch"] = c
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                # gfx950: FETCH_SIZE tallies 128-B fabric read requests at 64 B => x2 (MI355X_MICROARCH.md, HBM);
                # WRITE_SIZE is exact for 16-B-per-lane stores; both in KB
                k["traffic_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
                if k["avg_us_stats_pass"]:
                    k["traffic_TBs"] = k["traffic_bytes_per_launch"] / (k["avg_us_stats_pass"] * 1e-6) / 1e12
            d = derived(per_pass.get("pmc_sq_a", {}), per_pass.get("pmc_sq_b", {}))
            if d:
                k["derived"] = d
            summary["kernels"][name] = k
        with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
            f.write("\n")
        print(json.dumps({n[:60]: {"us": k["avg_us_stats_pass"], "traffic_MB": k.get("traffic_bytes_per_launch", 0) / 1e6}
                          for n, k in summary["kernels"].items()}, indent=1), flush=True)


if __name__ == "__main__":
    main()

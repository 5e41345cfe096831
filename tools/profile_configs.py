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
SQ = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU",
      "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"]
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
        for ptag, ctrs in (("pmc_fetch", ["FETCH_SIZE"]), ("pmc_write", ["WRITE_SIZE"]), ("pmc_sq", SQ)):
            out, _ = run_prof(ptag, ["--pmc", *ctrs], [*prog, "--steps", "10", "--warmup", "2", "--settle-ms", "0"], scratch)
            passes[ptag] = counters(out)
        for name in sorted(set().union(*[set(p) for p in passes.values()])):
            k = {"avg_us_stats_pass": avg_ns.get(name, (None, 0))[0] and avg_ns[name][0] / 1e3,
                 "calls_stats_pass": avg_ns.get(name, (None, 0))[1]}
            c = {}
            for ptag in passes:
                if name in passes[ptag]:
                    k[f"dispatches_{ptag}"] = passes[ptag][name][0]
                    c.update(passes[ptag][name][1])
            k["counters_avg_per_dispatch"] = c
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                # gfx950: FETCH_SIZE tallies 128-B fabric read requests at 64 B => x2 (MI355X_MICROARCH.md, HBM);
                # WRITE_SIZE is exact for 16-B-per-lane stores; both in KB
                k["traffic_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
                if k["avg_us_stats_pass"]:
                    k["traffic_TBs"] = k["traffic_bytes_per_launch"] / (k["avg_us_stats_pass"] * 1e-6) / 1e12
            if c.get("GRBM_GUI_ACTIVE") and k["avg_us_stats_pass"]:
                xcd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
                k["derived"] = {
                    "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * xcd_cycles),
                    "valu_insts_per_wave": c["SQ_INSTS_VALU"] / max(c["SQ_WAVES"], 1.0),
                    "avg_waves_per_simd": c["SQ_WAVE_CYCLES"] * 4.0 / (1024.0 * xcd_cycles),
                    "wait_frac_of_wave_life": c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
                }
            summary["kernels"][name] = k
        with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
            json.dump(summary, f, indent=1)
            f.write("\n")
        print(json.dumps({n[:60]: {"us": k["avg_us_stats_pass"], "traffic_MB": k.get("traffic_bytes_per_launch", 0) / 1e6}
                          for n, k in summary["kernels"].items()}, indent=1), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""rocprofv3 evidence for one command (runs on the GPU box): a --kernel-trace --stats run plus separate
--pmc passes, reduced to one summary.json per tag.

usage: tools/profile_cmd.py <tag> [--kernel SUBSTR] [--passes valu,misc,lds,fetch,write] [--env K=V]... -- <program> <args>...

The program itself must follow `--` (python3 bench.py ..., not env/bash wrappers): the profiler's
preloaded library has initialised the GPU before the program starts.  Counter values are summed over
every dispatch whose kernel name contains SUBSTR and reported per dispatch.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = {
    # at most 8 SQ_, 4 TCC_ (FETCH_SIZE takes 3, WRITE_SIZE 2) and 2 GRBM_ counters fit one pass
    "valu": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU",
             "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"],
    "misc": ["SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_ANY",
             "SQ_WAVES", "SQ_INSTS_BRANCH", "SQ_ACTIVE_INST_SCA", "GRBM_GUI_ACTIVE"],
    "lds": ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
            "SQ_LDS_ADDR_CONFLICT", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"],
    "mfma": ["SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES",
             "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"],
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
}


def main():
    argv = sys.argv[1:]
    if "--" not in argv or not argv:
        print(__doc__)
        return 2
    cut = argv.index("--")
    opts, cmd = argv[:cut], argv[cut + 1:]
    tag = opts[0]
    kernel, passes, env = "render_kernel", ["valu", "misc", "fetch", "write"], dict(os.environ)
    i = 1
    while i < len(opts):
        if opts[i] == "--kernel":
            kernel = opts[i + 1]
        elif opts[i] == "--passes":
            passes = opts[i + 1].split(",")
        elif opts[i] == "--env":
            k, v = opts[i + 1].split("=", 1)
            env[k] = v
        else:
            raise SystemExit(f"unknown option {opts[i]}")
        i += 2
    env["TMPDIR"] = "/tmp"
    out = os.path.join(ROOT, "gpurun_out", tag)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    summary = {"command": " ".join(cmd), "kernel_filter": kernel,
               "env": {k: v for k, v in env.items() if k.startswith("RT_AMD")}}

    def run(args, log, limit, must=False):
        with open(os.path.join(out, log), "w") as f:
            rc = subprocess.call(args, stdout=f, stderr=subprocess.STDOUT, env=env, cwd=ROOT, timeout=limit + 30)
        if rc != 0:
            print(f"{tag}: '{' '.join(args[:6])} ...' exited with {rc}; see {out}/{log}")
            # a time limit, an abort or a fault ends the whole call; a refused counter list only skips the pass
            if rc in (124, 137, 134, 139) or rc < 0 or must:
                sys.exit(1)
        return rc

    run(["timeout", "-k", "10", "400", "rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv",
         "-d", os.path.join(out, "stats"), "--"] + cmd, "stats.log", 400, must=True)
    rows = []
    for f in glob.glob(os.path.join(out, "stats", "*", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(out, "kernel_stats.csv"))
        for r in csv.DictReader(open(f)):
            if kernel in r["Name"]:
                rows.append({"name": r["Name"].split("(")[0][-60:], "calls": int(r["Calls"]),
                             "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6,
                             "max_ms": float(r["MaxNs"]) / 1e6})
    summary["kernel_stats"] = rows
    # the line the program printed (bench.py: its JSON)
    for line in open(os.path.join(out, "stats.log")):
        if line.startswith('{"metric"'):
            summary["bench_line"] = json.loads(line)
        elif "Msamples/s" in line:
            summary["program_line"] = line.strip()

    for name in passes:
        if run(["timeout", "-s", "KILL", "300", "rocprofv3", "--pmc"] + PASSES[name] +
               ["--output-format", "csv", "-d", os.path.join(out, "pmc_" + name), "--"] + cmd, f"pmc_{name}.log", 300):
            summary[name] = {"error": "pass failed"}
            continue
        agg, disp = collections.defaultdict(float), set()
        for f in glob.glob(os.path.join(out, "pmc_" + name, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if kernel in r["Kernel_Name"]:
                    agg[r["Counter_Name"]] += float(r["Counter_Value"])
                    disp.add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
        n = max(len(disp), 1)
        summary[name] = {"dispatches": len(disp), **{k: v / n for k, v in sorted(agg.items())}}
        shutil.rmtree(os.path.join(out, "pmc_" + name), ignore_errors=True)  # raw per-dispatch CSVs are large

    v, m = summary.get("valu", {}), summary.get("misc", {})
    d = {}
    if v.get("SQ_INSTS_VALU"):
        d["lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_INSTS_VALU"])
        d["cycles_per_valu_per_simd"] = v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0 / v["SQ_INSTS_VALU"]
        d["shader_cycles"] = v["GRBM_GUI_ACTIVE"] / 8.0
        d["wave_cycle_split"] = {k: v[k] / v["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")
                                 if k in v}
        # share of the shader cycles in which a SIMD's vector issue port was taken (an instruction holds it 4 cycles)
        d["issue_busy"] = 4.0 * v["SQ_ACTIVE_INST_VALU"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    f = summary.get("mfma", {})
    if f.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        d["mfma_busy"] = f["SQ_VALU_MFMA_BUSY_CYCLES"] / (f["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        d["mfma_insts"] = f["SQ_INSTS_MFMA"]
    if m.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
        d["lds_busy_share_of_cu_cycles"] = m["SQ_LDS_IDX_ACTIVE"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 256.0)
    if "fetch" in summary and "write" in summary:
        # both counters report KiB; FETCH_SIZE tallies 128-byte requests at 64 on gfx950
        d["hbm_bytes_per_launch"] = (2 * summary["fetch"].get("FETCH_SIZE", 0) + summary["write"].get("WRITE_SIZE", 0)) * 1024
    summary["derived"] = d
    shutil.rmtree(os.path.join(out, "stats"), ignore_errors=True)
    json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print(tag, json.dumps({"kernel_stats": rows, "derived": d}))
    return 0


if __name__ == "__main__":
    sys.exit(main())

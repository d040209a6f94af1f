#!/usr/bin/env python3
"""Produce the files under profiles/ for one round tag, on a GPU box (run through gpurun from the repo root):

    python3 tools/collect_profiles.py r02_a            # -> gpurun_out/profiles/r02_a_*  (copy the ones to keep into profiles/)

It runs, each as its own rocprofv3 invocation (counters never share a run with --stats; the program after `--` is python3
itself, no wrapper):
  1. rocprofv3 --kernel-trace --stats  -- python3 bench.py --sustained-budget 0        kernel_stats.csv + the bench JSON line
  2. rocprofv3 --kernel-trace --stats  -- python3 bench.py --workload hicedrn64 --steps 5 --warmup 1
  3. rocprofv3 --kernel-trace --pmc FETCH_SIZE  -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --chains 1
  4. rocprofv3 --kernel-trace --pmc WRITE_SIZE  -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --chains 1
  5./6. rocprofv3 --kernel-trace --pmc <SQ counters, two passes>  -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --chains 1
        -> <tag>_unet64_b256_sq.json: per kernel and launch MFMA-busy, wave wait / issue-stall shares, LDS busy and bank-conflict share
and reduces 3 + 4 to <tag>_unet64_b256_hbm_traffic.json: per kernel, bytes per launch = 1024 * counter / launches, FETCH_SIZE
doubled (gfx950 tallies 128-byte read requests at 64 bytes: /opt/skills/guides/MI355X_MICROARCH.md, HBM).
"""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rocprof(outdir, extra, bench_args, log):
    cmd = ["rocprofv3", "--kernel-trace"] + extra + ["--output-format", "csv", "-d", outdir, "-o", "p", "--", "python3", os.path.join(ROOT, "bench.py")] + bench_args
    env = dict(os.environ, TMPDIR="/tmp")
    with open(log, "w") as f:
        subprocess.run(cmd, check=True, cwd="/tmp", env=env, stdout=f, stderr=subprocess.STDOUT)


def per_kernel(path, counter):
    agg, n = collections.defaultdict(float), collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]] += float(r["Counter_Value"])
                n[r["Kernel_Name"]] += 1
    return agg, n


SQ_A = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_LDS",
        "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"]
SQ_B = ["SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"]


def sq_reduce(paths, out_path, command):
    """Per kernel: counters summed over the chip, averaged per launch, plus the shares DESIGN.md quotes.  Units (MI355X_MICROARCH.md,
    cycle constants): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave, SQ_VALU_MFMA_BUSY_CYCLES cycles per SIMD
    (32 per 32x32x16 bf16 MFMA), GRBM_GUI_ACTIVE the sum over the 8 XCDs of busy cycles."""
    tot, cnt, dur = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(collections.Counter), collections.defaultdict(list)
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if "at::native" in k or k.startswith("__amd_rocclr") or "Cijk_" in k:
                    continue
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
                if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_LDS_IDX_ACTIVE"):
                    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rec = {"command": command, "kernels": {}}
    order = sorted(tot, key=lambda k: -sum(dur[k]))
    for k in order:
        per = {c: tot[k][c] / max(cnt[k][c], 1) for c in tot[k]}
        d = {"launches": max(cnt[k].values()), "avg_us_under_profiler": round(sum(dur[k]) / max(len(dur[k]), 1) / 1e3, 1)}
        d.update({c: round(v) for c, v in per.items()})
        cyc = per.get("GRBM_GUI_ACTIVE", 0) / 8
        if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in per:
            d["mfma_busy_frac"] = round(per["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
        if per.get("SQ_WAVE_CYCLES"):
            for c, nm in (("SQ_WAIT_ANY", "wave_parked_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_WAIT_INST_LDS", "lds_issue_stall_frac"),
                          ("SQ_ACTIVE_INST_ANY", "issuing_frac")):
                if c in per:
                    d[nm] = round(per[c] / per["SQ_WAVE_CYCLES"], 4)
        if per.get("SQ_LDS_IDX_ACTIVE"):
            if "SQ_LDS_BANK_CONFLICT" in per:
                d["lds_bank_conflict_frac"] = round(per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"], 4)
            if cyc:
                d["lds_busy_frac"] = round(per["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 4)
        rec["kernels"][k] = d
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    work = os.path.join(ROOT, "gpurun_out", "profiles_work")
    out = os.path.join(ROOT, "gpurun_out", "profiles")
    os.makedirs(work, exist_ok=True)
    os.makedirs(out, exist_ok=True)
    # --sustained-budget 0: the post-timing whole-chain run (1000 more steps) would only bloat the traces
    rocprof(os.path.join(work, "unet64"), ["--stats"], ["--sustained-budget", "0"], os.path.join(work, "unet64.log"))
    # the same with ONE whole-batch chain: the launches bench.py's `roofline` describes (its per-launch events are recorded on eager whole-batch
    # steps; the default run replays the kernels on two 128-tile halves whose durations overlap in the trace)
    rocprof(os.path.join(work, "unet64_c1"), ["--stats"], ["--sustained-budget", "0", "--chains", "1", "--no-cpu-baseline"], os.path.join(work, "unet64_c1.log"))
    rocprof(os.path.join(work, "hicedrn64"), ["--stats"], ["--workload", "hicedrn64", "--steps", "5", "--warmup", "1", "--sustained-budget", "0"],
            os.path.join(work, "hicedrn64.log"))
    # counter passes: one whole-batch chain, so that a launch is the 256-tile launch the bench line's algorithmic bytes describe
    short = ["--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--sustained-budget", "0", "--chains", "1"]
    rocprof(os.path.join(work, "fetch"), ["--pmc", "FETCH_SIZE"], short, os.path.join(work, "fetch.log"))
    rocprof(os.path.join(work, "write"), ["--pmc", "WRITE_SIZE"], short, os.path.join(work, "write.log"))
    if "--no-sq" not in sys.argv:
        rocprof(os.path.join(work, "sqa"), ["--pmc"] + SQ_A, short, os.path.join(work, "sqa.log"))
        rocprof(os.path.join(work, "sqb"), ["--pmc"] + SQ_B, short, os.path.join(work, "sqb.log"))
        sq_reduce([os.path.join(work, "sqa", "p_counter_collection.csv"), os.path.join(work, "sqb", "p_counter_collection.csv")],
                  os.path.join(out, f"{tag}_unet64_b256_sq.json"),
                  "rocprofv3 --kernel-trace --pmc " + " ".join(SQ_A) + " | " + " ".join(SQ_B) + " (one pass each) -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --chains 1")
    shutil.copy(os.path.join(work, "unet64_c1", "p_kernel_stats.csv"), os.path.join(out, f"{tag}_unet64_b256_chains1_kernel_stats.csv"))
    for wl in ("unet64", "hicedrn64"):
        shutil.copy(os.path.join(work, wl, "p_kernel_stats.csv"), os.path.join(out, f"{tag}_{wl}_b256_kernel_stats.csv"))
        with open(os.path.join(work, wl + ".log")) as f:
            lines = [ln for ln in f if ln.startswith("{")]
        if lines:
            with open(os.path.join(out, f"{tag}_{wl}_b256_bench.json"), "w") as f:
                f.write(lines[-1])
    fe, nf = per_kernel(os.path.join(work, "fetch", "p_counter_collection.csv"), "FETCH_SIZE")
    wr, nw = per_kernel(os.path.join(work, "write", "p_counter_collection.csv"), "WRITE_SIZE")
    steps = 9 + 1 + 4 + 4        # graph set-up (3 per arithmetic) + warm-up + timed + the profiled eager steps bench.py adds
    rows, tf, tw = [], 0.0, 0.0
    for k in fe:
        fb, wb = 2 * fe[k] * 1024, wr.get(k, 0.0) * 1024
        tf += fb
        tw += wb
        rows.append((k, nf[k], fb / nf[k], wb / max(nw.get(k, 1), 1)))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    rec = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (one pass each) -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --chains 1",
           "correction": "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): gfx950 FETCH_SIZE counts 128-byte requests as 64 (MI355X_MICROARCH.md, HBM)",
           "steps_in_run": steps, "per_step_fetch_GB": round(tf / steps / 1e9, 3), "per_step_write_GB": round(tw / steps / 1e9, 3),
           "kernels": {k: {"launches": n, "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb)} for k, n, fb, wb in rows}}
    with open(os.path.join(out, f"{tag}_unet64_b256_hbm_traffic.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(f"wrote {out}/{tag}_*: {rec['per_step_fetch_GB']} GB fetched + {rec['per_step_write_GB']} GB written per step")


if __name__ == "__main__":
    main()

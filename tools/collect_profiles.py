#!/usr/bin/env python3
"""Produce the files under profiles/ for one round tag, on a GPU box (run through gpurun from the repo root):

    python3 tools/collect_profiles.py r02_a            # -> gpurun_out/profiles/r02_a_*  (copy the ones to keep into profiles/)

It runs, each as its own rocprofv3 invocation (counters never share a run with --stats; the program after `--` is python3
itself, no wrapper):
  1. rocprofv3 --kernel-trace --stats  -- python3 bench.py                      kernel_stats.csv + the bench JSON line
  2. rocprofv3 --kernel-trace --stats  -- python3 bench.py --workload hicedrn64 --steps 5 --warmup 1
  3. rocprofv3 --kernel-trace --pmc FETCH_SIZE  -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
  4. rocprofv3 --kernel-trace --pmc WRITE_SIZE  -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
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


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    work = os.path.join(ROOT, "gpurun_out", "profiles_work")
    out = os.path.join(ROOT, "gpurun_out", "profiles")
    os.makedirs(work, exist_ok=True)
    os.makedirs(out, exist_ok=True)
    rocprof(os.path.join(work, "unet64"), ["--stats"], [], os.path.join(work, "unet64.log"))
    rocprof(os.path.join(work, "hicedrn64"), ["--stats"], ["--workload", "hicedrn64", "--steps", "5", "--warmup", "1"], os.path.join(work, "hicedrn64.log"))
    short = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    rocprof(os.path.join(work, "fetch"), ["--pmc", "FETCH_SIZE"], short, os.path.join(work, "fetch.log"))
    rocprof(os.path.join(work, "write"), ["--pmc", "WRITE_SIZE"], short, os.path.join(work, "write.log"))
    for wl in ("unet64", "hicedrn64"):
        shutil.copy(os.path.join(work, wl, "p_kernel_stats.csv"), os.path.join(out, f"{tag}_{wl}_b256_kernel_stats.csv"))
        with open(os.path.join(work, wl + ".log")) as f:
            lines = [ln for ln in f if ln.startswith("{")]
        if lines:
            with open(os.path.join(out, f"{tag}_{wl}_b256_bench.json"), "w") as f:
                f.write(lines[-1])
    fe, nf = per_kernel(os.path.join(work, "fetch", "p_counter_collection.csv"), "FETCH_SIZE")
    wr, nw = per_kernel(os.path.join(work, "write", "p_counter_collection.csv"), "WRITE_SIZE")
    steps = 3 + 1 + 3            # timed + warm-up + the profiled eager steps bench.py adds
    rows, tf, tw = [], 0.0, 0.0
    for k in fe:
        fb, wb = 2 * fe[k] * 1024, wr.get(k, 0.0) * 1024
        tf += fb
        tw += wb
        rows.append((k, nf[k], fb / nf[k], wb / max(nw.get(k, 1), 1)))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    rec = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (one pass each) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
           "correction": "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): gfx950 FETCH_SIZE counts 128-byte requests as 64 (MI355X_MICROARCH.md, HBM)",
           "steps_in_run": steps, "per_step_fetch_GB": round(tf / steps / 1e9, 3), "per_step_write_GB": round(tw / steps / 1e9, 3),
           "kernels": {k: {"launches": n, "fetch_bytes_per_launch": round(fb), "write_bytes_per_launch": round(wb)} for k, n, fb, wb in rows}}
    with open(os.path.join(out, f"{tag}_unet64_b256_hbm_traffic.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(f"wrote {out}/{tag}_*: {rec['per_step_fetch_GB']} GB fetched + {rec['per_step_write_GB']} GB written per step")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Do the two half-batch chains of a step overlap on the device?  Reads a `rocprofv3 --kernel-trace --output-format csv` directory of
`bench.py --chains 2` and reports, over the traced run:

    python3 tools/lane_overlap.py <dir> [--tail-steps N]

  * per queue: launches, summed kernel time;
  * time with >= 1 / >= 2 kernels of DIFFERENT queues in flight;
  * of the two-queue time, the share in which an MFMA-bound 3x3 convolution of one chain runs next to an HBM/latency-bound launch of
    the other (everything that is not a 3x3 `conv_igemm_bf16x3_kernel<..., 9>`), and the share with 3x3 next to 3x3.
Timestamps are the profiler's (ns); a kernel's interval is [start, end)."""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    paths = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not paths:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    rows = []
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                if "at::native" in name or name.startswith("__amd_rocclr"):
                    continue
                q = r.get("Queue_Id") or r.get("Stream_Id") or "0"
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), q, name))
    rows.sort()
    # keep the part of the run where two queues are active (the timed region and after): from the first launch of the second-busiest queue
    by_q = {}
    for s, e, q, n in rows:
        by_q.setdefault(q, []).append((s, e, n))
    busy = sorted(by_q, key=lambda q: -sum(e - s for s, e, _ in by_q[q]))
    print(f"{len(rows)} kernel launches on {len(by_q)} queues")
    for q in busy[:4]:
        t = sum(e - s for s, e, _ in by_q[q])
        print(f"  queue {q}: {len(by_q[q])} launches, {t / 1e6:.2f} ms of kernel time, first at {by_q[q][0][0]}")
    if len(busy) < 2:
        print("only one queue carried kernels: nothing to overlap")
        return
    a, b = busy[0], busy[1]
    t0 = max(by_q[a][0][0], by_q[b][0][0])
    t1 = min(by_q[a][-1][1], by_q[b][-1][1])
    is33 = lambda n: "conv_igemm_bf16x3_kernel" in n and ", 9>(" in n
    ev = []
    for q in (a, b):
        for s, e, n in by_q[q]:
            if e <= t0 or s >= t1:
                continue
            ev.append((max(s, t0), 1, q, is33(n)))
            ev.append((min(e, t1), -1, q, is33(n)))
    ev.sort(key=lambda x: (x[0], x[1]))
    live = {a: [0, 0], b: [0, 0]}          # [3x3 in flight, other in flight] per queue
    last = t0
    any1 = both = mix = conv_conv = other_other = 0
    for t, dlt, q, c in ev:
        dt = t - last
        if dt > 0:
            na, nb = sum(live[a]), sum(live[b])
            if na or nb:
                any1 += dt
            if na and nb:
                both += dt
                ca, cb = live[a][0] > 0, live[b][0] > 0
                if ca and cb:
                    conv_conv += dt
                elif ca or cb:
                    mix += dt
                else:
                    other_other += dt
        live[q][0 if c else 1] += dlt
        last = t
    span = t1 - t0
    print(f"window where both chains run: {span / 1e6:.2f} ms")
    print(f"  >= 1 kernel in flight: {any1 / 1e6:.2f} ms ({any1 / span:.3f} of the window)")
    print(f"  kernels of BOTH chains in flight: {both / 1e6:.2f} ms ({both / span:.3f} of the window)")
    if both:
        print(f"    3x3 conv next to a non-3x3 launch of the other chain: {mix / 1e6:.2f} ms ({mix / both:.3f})")
        print(f"    3x3 next to 3x3: {conv_conv / 1e6:.2f} ms ({conv_conv / both:.3f});  non-3x3 next to non-3x3: {other_other / 1e6:.2f} ms ({other_other / both:.3f})")
    ka = sum(e - s for s, e, _ in by_q[a] if s >= t0 and e <= t1) + sum(e - s for s, e, _ in by_q[b] if s >= t0 and e <= t1)
    print(f"  summed kernel time of the two chains in the window: {ka / 1e6:.2f} ms = {ka / span:.3f} x the window (1.0 = no overlap at all)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per training iteration.

Iterations are delimited by launches of the forward gather kernel (one per iteration in the engine); the first
`--skip` iterations (warm-up) are dropped.  Prints per-kernel time per iteration, sorted, plus the total.
usage: python tools/prof_summary.py <kernel_trace.csv> [--skip 3 | --last 9] [--top 40]
"""
import argparse
import csv
import collections
import re


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"at::native::", "", name)
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    return name[:120]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--skip", type=int, default=3)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--marker", default="gather_aggregate_fwd_kernel")
    ap.add_argument("--last", type=int, default=0, help="use only the last N iterations (the bench's history pre-roll runs up to T evaluation "
                    "iterations before the training steps)")
    ap.add_argument("--count", type=int, default=0, help="use only this many iterations after --skip")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
    if len(marks) <= a.skip + 1:
        raise SystemExit("not enough iterations in trace")
    # iteration i spans [first kernel after previous iteration's end ... ]: use marker-to-marker windows
    if a.last:
        a.skip = max(a.skip, len(marks) - 1 - a.last)
    lo, hi = marks[a.skip], marks[-1]
    n_iter = len(marks) - 1 - a.skip
    if a.count and a.count < n_iter:
        hi, n_iter = marks[a.skip + a.count], a.count
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in rows[lo:hi]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = short(r["Kernel_Name"])
        agg[k][0] += d
        agg[k][1] += 1
    span = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
    tot = sum(v[0] for v in agg.values())
    print(f"# {n_iter} iterations, wall/iter {span / n_iter / 1e6:.3f} ms, kernel-busy/iter {tot / n_iter / 1e6:.3f} ms, "
          f"{sum(v[1] for v in agg.values()) / n_iter:.0f} launches/iter")
    print(f"# {'us/iter':>10} {'calls/iter':>10} {'avg us':>9}  kernel")
    for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[: a.top]:
        print(f"{d / n_iter / 1e3:12.1f} {c / n_iter:10.1f} {d / c / 1e3:9.1f}  {k}")
    cats = collections.defaultdict(lambda: [0.0, 0])
    for k, (d, c) in agg.items():
        if k.startswith("Cijk"):
            g = "GEMM (hipBLASLt via torch)"
        elif "lstep::" in k:
            g = "lstep:" + k.split("lstep::")[1].split("(")[0]
        elif "rocprim" in k or "sort" in k.lower():
            g = "torch sort/scan (rocprim)"
        elif "copyBuffer" in k or "fillBuffer" in k:
            g = "memcpy/memset"
        elif "multi_tensor" in k:
            g = "Adam"
        else:
            g = "torch elementwise/index/reduce"
        cats[g][0] += d
        cats[g][1] += c
    print("# by category")
    for g, (d, c) in sorted(cats.items(), key=lambda kv: -kv[1][0]):
        print(f"{d / n_iter / 1e3:12.1f} {c / n_iter:10.1f}             {g}")


if __name__ == "__main__":
    main()

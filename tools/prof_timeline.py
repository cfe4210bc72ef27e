#!/usr/bin/env python3
"""Timeline of ONE steady-state training iteration from a rocprofv3 --kernel-trace CSV: per stream (queue), every launch with
its start offset, duration and the idle gap in front of it; plus per-stream busy/idle totals.  Shows where the critical
(main-stream) path waits for the host or for the other stream.
usage: python tools/prof_timeline.py <kernel_trace.csv> [--iter 5] [--min-gap 15]
"""
import argparse
import csv
import collections
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|at::native::|void ", "", name)
    return name[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--iter", type=int, default=5)
    ap.add_argument("--min-gap", type=float, default=15.0, help="list launches preceded by at least this idle time (us)")
    ap.add_argument("--marker", default="gather_aggregate_fwd_kernel")
    ap.add_argument("--all", action="store_true", help="list every launch")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
    lo, hi = marks[a.iter], marks[a.iter + 1]
    t0 = int(rows[lo]["Start_Timestamp"])
    span = (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    per = collections.defaultdict(list)
    for r in rows[lo:hi]:
        per[r[qkey]].append(r)
    print(f"# iteration {a.iter}: {span:.1f} us marker-to-marker, {hi - lo} launches, streams: { {k: len(v) for k, v in per.items()} }")
    for q, rs in per.items():
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e3
        print(f"## stream {q}: {len(rs)} launches, busy {busy:.1f} us, first at +{(int(rs[0]['Start_Timestamp']) - t0) / 1e3:.1f}, "
              f"last ends +{(int(rs[-1]['End_Timestamp']) - t0) / 1e3:.1f}")
        prev_end = None
        gaps = 0.0
        for r in rs:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = 0.0 if prev_end is None else max(0.0, (s - prev_end) / 1e3)
            gaps += gap
            if a.all or gap >= a.min_gap:
                print(f"   +{(s - t0) / 1e3:8.1f} us  gap {gap:7.1f}  dur {(e - s) / 1e3:7.1f}  {short(r['Kernel_Name'])}")
            prev_end = e if prev_end is None else max(prev_end, e)
        print(f"   idle inside the stream's span: {gaps:.1f} us")


if __name__ == "__main__":
    main()

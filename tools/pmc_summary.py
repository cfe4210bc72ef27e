#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into per-kernel HBM traffic per launch.

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (16 B/lane), so
read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact for 16-B/lane stores and float atomics.
usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [--skip 2 | --last 6] [--out json]
"""
import argparse
import collections
import csv
import json


def load(path, counter, skip, last):
    per = collections.defaultdict(list)
    rows = list(csv.DictReader(open(path)))
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        if r["Counter_Name"] == counter and "lstep::" in r["Kernel_Name"]:
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    pick = (lambda v: v[-last:]) if last else (lambda v: v[skip:])
    return {k: sum(pick(v)) / max(1, len(pick(v))) for k, v in per.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("--skip", type=int, default=2, help="launches to drop per kernel (warm-up)")
    ap.add_argument("--last", type=int, default=0, help="use only the last N launches of every kernel (the timed training steps: the bench "
                    "first evolves the history with up to T evaluation iterations, whose launches have other shapes)")
    ap.add_argument("--out")
    a = ap.parse_args()
    f, w = load(a.fetch, "FETCH_SIZE", a.skip, a.last), load(a.write, "WRITE_SIZE", a.skip, a.last)
    out = {}
    for k in sorted(f):
        rd, wr = 2.0 * f[k] * 1024.0, w.get(k, 0.0) * 1024.0
        out[k] = {"read_bytes": rd, "write_bytes": wr, "traffic_bytes": rd + wr, "FETCH_SIZE_KB_raw": f[k], "WRITE_SIZE_KB_raw": w.get(k, 0.0)}
        print(f"{k:60s} read {rd / 1e9:7.3f} GB  write {wr / 1e9:7.3f} GB  traffic {(rd + wr) / 1e9:7.3f} GB per launch")
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-step, per-rank budget of lstep_amd.parallel.DistributedLstep for the three forms of update_pe
(LSTEP_PHASE2 = replicate | allgather | pull): bytes every rank receives per global batch, the link time they cost, and the rows every
rank pushes through update_pe's MLP (the matrix-core work of the update), for the workloads of BASELINE.json under weak scaling
(B = 16384 edges per GPU).

Pure arithmetic on the workload's shape (uniform endpoints: expected distinct counts from occupancy, N (1 - exp(-draws / N))); no GPU.
usage: python tools/comm_budget.py
"""
import math

P_ROW = 172 * 4            # one PE / feature row, bytes
ROW_ID = 176 * 4           # a PE row travelling with its id packed into padding columns (parallel._rows_with_ids)
XGMI_LINK = 153e9 / 2      # bytes/s one direction of one xGMI link (7 links x ~153 GB/s bidirectional per GPU)
MLP_ROWS_PER_MS = 290e3 / 0.46      # lstep_update_rows_pre on one MI355X: 290 k touched rows in 0.46 ms (profiles/r02_f_*)


def distinct(draws: float, n: float) -> float:
    return n * (1.0 - math.exp(-draws / n))


def budget(name, N, B, K, W):
    """B = per-GPU batch (weak scaling): global batch W * B edges on a graph of N nodes.  Returns {form: (bytes in, link ms, MLP rows)}."""
    gb = W * B
    U = distinct(2 * gb, N)                      # distinct batch nodes of the global batch
    U2 = distinct(U * K, N)                      # distinct neighbours phase 2 touches (upper bound: every slot a real neighbour)
    need = distinct(3 * B * (K + 1) + 1, N)      # rows ONE rank's gather stage reads (its 3 B rows, their K neighbours, row 0)
    frac = (W - 1) / W if W > 1 else 0.0         # share of a gathered buffer that arrives over the links
    links = max(1, min(W - 1, 7))
    bw = links * XGMI_LINK                       # direct exchange: every peer link carries its own block
    common = {
        "FFT rows all-gather (critical path, under the edge/node gather)": U * P_ROW * frac,
        "spliced-row gradient reduce-scatter (under the weight-gradient stream)": U * P_ROW * frac,
        "parameter gradients all-reduce": 2 * 2.3e6 * frac,
    }
    forms = {
        "replicate": ({}, U + U2),
        "allgather": ({"phase-1 rows all-gather (side stream)": U * ROW_ID * frac,
                       "phase-2 rows all-gather (side stream, under the backward pass)": U2 * ROW_ID * frac}, (U + U2) / W),
        "pull": ({"phase-1 rows all-gather (side stream)": U * ROW_ID * frac,
                  "request ids all-to-all (4 B per id, blocks of 1.5 x an even split)": 1.5 * (3 * B * (K + 1) + 1) / W * 4 * (W - 1),
                  "pulled rows all-to-all-v (second communicator, under the backward pass)": need * P_ROW * frac}, (U + U2) / W),
    }
    print(f"\n{name}: N = {N:,}, global batch {gb:,} (W = {W} x {B:,}), K = {K}:  U = {U:,.0f} batch nodes, U2 = {U2:,.0f} rows touched by phase 2 "
          f"({U2 / N:.0%} of the table), one rank's gather reads {need:,.0f} distinct PE rows ({need / N:.0%})")
    for what, nbytes in common.items():
        print(f"    every form: {what:78s} {nbytes / 1e6:8.1f} MB in per rank  {nbytes / bw * 1e3:5.2f} ms at {links} link(s)")
    out = {}
    for form, (extra, mlp_rows) in forms.items():
        total = sum(common.values()) + sum(extra.values())
        upd = sum(extra.values())
        print(f"  {form}:")
        for what, nbytes in extra.items():
            print(f"    {what:90s} {nbytes / 1e6:8.1f} MB in per rank  {nbytes / bw * 1e3:5.2f} ms")
        print(f"    -> update traffic {upd / 1e6:8.1f} MB ({upd / bw * 1e3:5.2f} ms of link time), all collectives {total / 1e6:8.1f} MB ({total / bw * 1e3:5.2f} ms); "
              f"update_pe MLP rows per rank {mlp_rows:10,.0f}  (~{mlp_rows / MLP_ROWS_PER_MS:4.2f} ms of matrix-core time; single GPU, c4: ~0.5 ms)")
        out[form] = (upd, upd / bw * 1e3, mlp_rows)
    return out


if __name__ == "__main__":
    table = []
    for label, N, W in (("c4", 1_000_000, 1), ("c4", 1_000_000, 2), ("c4", 1_000_000, 4), ("c4-sized graph", 1_000_000, 8), ("c5", 4_000_000, 8)):
        res = budget(f"{label}  synthetic {N // 1_000_000} M nodes", N, 16384, 20, W)
        table.append((label, W, res))
    print("\nSummary -- update_pe per rank and step: MB received for the update / ms of link time / rows through the MLP")
    print(f"{'workload':16s} {'W':>2s} | " + " | ".join(f"{f:^34s}" for f in ("replicate", "allgather", "pull")))
    for label, W, res in table:
        print(f"{label:16s} {W:2d} | " + " | ".join(f"{res[f][0] / 1e6:8.1f} MB {res[f][1]:5.2f} ms {res[f][2] / 1e3:8.0f} k rows" for f in ("replicate", "allgather", "pull")))
    print("\nReading: 'replicate' moves nothing but its MLP rows grow with the global batch (x 6 from W = 1 to c5 at W = 8); 'allgather' keeps the rows\n"
          "flat but receives every touched row (1.8 GB at c5); 'pull' keeps the rows flat AND receives only what the next gather reads: its link\n"
          "time falls with W (more links; the pulled rows alone: 2.9 -> 1.5 -> 1.0 ms) and it runs on a second communicator underneath the backward pass.  O(U) work\n"
          "that stays replicated in 'pull' (sampling the batch nodes' neighbourhoods, sorting U x K keys, the [U, 172] x [172, 176] product of the\n"
          "pre-multiplied messages): ~0.3 ms at c5 / W = 8.")
    print("\nmemory per rank at c5 / W = 8: edge_raw 68.8 GB + node_raw 2.75 GB + PE table (owned rows + cache) 2.75 GB + CSR 3.2 GB + "
          "history shard (T + 2) x 0.5 M x 688 B = 35.1 GB  =  112.6 GB of 288 GB")

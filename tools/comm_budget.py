#!/usr/bin/env python3
"""Per-step collective byte budget of lstep_amd.parallel.DistributedLstep: what every rank sends / receives per
global batch for the workloads of BASELINE.json at W = 2, 4, 8, and what the alternatives the design rejects would move instead.

Pure arithmetic on the workload's shape (uniform endpoints: expected distinct counts from occupancy, 1 - exp(-draws / N)); no GPU.
usage: python tools/comm_budget.py
"""
import math

P_ROW = 172 * 4            # one PE / feature row, bytes
ROW_ID = 176 * 4           # a PE row travelling with its id packed into padding columns (parallel._rows_with_ids)
EMB = 176 * 4              # one padded embedding row
XGMI_LINK = 153e9 / 2      # bytes/s one direction of one xGMI link (7 links x ~153 GB/s bidirectional per GPU)


def distinct(draws: float, n: float) -> float:
    return n * (1.0 - math.exp(-draws / n))


def budget(name, N, B, K, W):
    """B = per-GPU batch (weak scaling): global batch W * B edges on a graph of N nodes."""
    gb = W * B
    U = distinct(2 * gb, N)                      # distinct batch nodes
    U2 = distinct(U * K, N)                      # distinct neighbours phase 2 touches (upper bound: every slot a real neighbour)
    need = distinct(3 * B * (K + 1), N)          # rows ONE rank's gather stage reads (its 3 B rows and their K neighbours)
    frac = (W - 1) / W                           # share of an all-gathered buffer that arrives over the links
    links = min(W - 1, 7)
    bw = links * XGMI_LINK                       # direct all-gather: every peer link carries its own block
    rows = {
        "FFT rows all-gather (critical path, under the edge/node gather)": U * P_ROW,
        "spliced-row gradient reduce-scatter (under the weight-gradient stream)": U * P_ROW,
        "phase-1 rows all-gather (side stream)": U * ROW_ID,
        "phase-2 rows all-gather (side stream, under the backward pass)": U2 * ROW_ID,
        "parameter gradients all-reduce": 2 * 2.3e6,
    }
    print(f"\n{name}: N = {N:,}, global batch {gb:,} (W = {W} x {B:,}), K = {K}:  U = {U:,.0f} batch nodes, U2 = {U2:,.0f} touched rows "
          f"({U2 / N:.0%} of the table)")
    total = 0.0
    for what, nbytes in rows.items():
        inbound = nbytes * frac
        total += inbound
        print(f"    {what:75s} {nbytes / 1e6:9.1f} MB total  {inbound / 1e6:9.1f} MB in per rank  {inbound / bw * 1e3:6.2f} ms at {links} links")
    print(f"    {'sum per rank':75s} {'':9s}           {total / 1e6:9.1f} MB in per rank  {total / bw * 1e3:6.2f} ms")
    # the alternatives
    emb = 3 * gb * EMB
    print(f"    -- destination-owner sharding of the gather stage instead of batch slices: + all-gather of the embeddings {emb * frac / 1e6:7.1f} MB in "
          f"per rank forward and the same again for their gradient, both on the critical path ({2 * emb * frac / bw * 1e3:.2f} ms), to save "
          f"{(1 - 2 / W) * 100:.0f} % of the edge table per rank")
    # the replicated-update form (DistributedLstep's default): no update collective, but every rank runs update_pe for the whole global batch.
    # Matrix-core time from the one-GPU measurement (lstep_update_rows_pre: 0.46 ms per 290 k touched rows, 0.09 ms per 32 k phase-1 rows),
    # segment sums 0.1 ms and sorting 0.08 ms per 0.65 M messages
    msgs = U * K
    rep_ms = U2 / 290e3 * 0.46 + U / 32e3 * 0.09 + msgs / 0.65e6 * (0.10 + 0.08)
    print(f"    -- replicated update_pe instead of the two update all-gathers: 0 MB, ~{rep_ms:.1f} ms of kernels on the side stream (single GPU: ~0.9 ms), "
          f"against ~2 ms of backward pass to hide under")
    pull = need * P_ROW * frac
    print(f"    -- owner-sharded PE table with a pull of the rows the next gather reads: {pull / 1e6:7.1f} MB in per rank "
          f"({pull / bw * 1e3:.2f} ms) instead of the phase-2 all-gather's {rows['phase-2 rows all-gather (side stream, under the backward pass)'] * frac / 1e6:.1f} MB, "
          f"but as a request / response pair that needs the NEXT batch's ids and negatives")


if __name__ == "__main__":
    for W in (2, 4):
        budget("c4  synthetic 1 M nodes / 20 M edges", 1_000_000, 16384, 20, W)
    budget("c5  synthetic 4 M nodes / 100 M edges", 4_000_000, 16384, 20, 8)
    print("\nmemory per rank at c5 / W = 8 with the replicated tables: edge_raw 68.8 GB + node_raw 2.75 GB + PE table 2.75 GB + CSR 3.2 GB + "
          "history shard (T + 2) x 0.5 M x 688 B = 35.1 GB  =  112.6 GB of 288 GB")

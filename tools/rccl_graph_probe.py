#!/usr/bin/env python3
"""Does this torch / RCCL replay collectives from a HIP graph ACROSS RANKS?  ``bench.py --gpus N`` (N > 1) runs this as a child process of
every rank BEFORE the rank touches its GPU: the multi-GPU iteration is one captured graph with its RCCL collectives inside
(``lstep_amd.parallel.GraphedDistStep``), which this pool could only ever rehearse with ONE rank (tools/rccl_capture_probe.py), where RCCL
short-cuts every collective into a copy.  The probe captures the same kinds of calls the step makes -- all_gather_into_tensor,
reduce_scatter_tensor, all_reduce and a synchronous all_to_all_single on the capturing stream, a synchronous all_gather on a forked side
stream, ``capture_error_mode="thread_local"`` -- replays the graph three times on changing inputs and checks every result.  Exit code 0 =
safe to capture the step; anything else (a wrong value, an exception, a crash, the parent's timeout) makes the bench fall back to the
launch-by-launch device-driven iteration instead of failing the run.

Round 5: a SECOND graph rehearses the owner-sharded ("pull") form's exchange pattern exactly as ``parallel.RowPullDev`` issues it inside a
captured step -- two equal-split ``all_to_all_single`` calls issued synchronously from the capturing stream, last: int32 id blocks [W, Cp]
one way, the float32 rows [W, Cp, P] they name the other way, every value checked on three replays with changing ids -- on the main
communicator and on a second one (``dist.new_group``: what the launch-by-launch pull uses).  The last stdout line of rank 0's child AND of
every other rank's is a JSON verdict {"captured": true, "pull": true|false}: ``bench.py`` lets ``DistributedLstep`` choose "pull" beyond four
ranks only when every rank says pull = true, and "replicate" otherwise (exit code 0 either way: the basic capture is what decides the graph).

env: RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT (the parent passes its own port + an offset)."""
import os
import sys

import torch
import torch.distributed as dist


def quiesce():
    """Before a capture: the device idle and one sweep of the NCCL watchdog over its list of launch-by-launch collectives (it polls their end
    events; an event on a stream that has meanwhile started capturing makes hipEventQuery fail and the watchdog abort the process --
    lstep_amd.parallel.quiesce_collectives, tools/nccl_capture_after_eager_probe.py)."""
    import time
    torch.cuda.synchronize()
    time.sleep(0.25)


def main() -> int:
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("LSTEP_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    os.environ.setdefault("RANK", "0")                # (run by hand on a one-GPU box: one rank)
    os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    n, m = 1 << 16, 1 << 10
    base = torch.zeros(1, dtype=torch.float32, device=dev)
    x = torch.empty(n, dtype=torch.float32, device=dev)
    out = torch.zeros(world * n, dtype=torch.float32, device=dev)
    rs = torch.zeros(n, dtype=torch.float32, device=dev)
    ar = torch.zeros(n, dtype=torch.float32, device=dev)
    a_in = torch.empty(world * m, dtype=torch.float32, device=dev)
    a_out = torch.zeros(world * m, dtype=torch.float32, device=dev)
    side_out = torch.zeros(world * n, dtype=torch.float32, device=dev)
    blocks = torch.arange(world, dtype=torch.float32, device=dev).repeat_interleave(m)
    side = torch.cuda.Stream(device=dev)

    def body():
        x.copy_((base + rank).expand(n))
        dist.all_gather_into_tensor(out, x)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                     # (a synchronous all_gather on a side stream that joined the capture)
            dist.all_gather_into_tensor(side_out, x * 3.0)
        z = out * 2.0
        dist.reduce_scatter_tensor(rs, z)
        ar.copy_(rs)
        dist.all_reduce(ar)
        a_in.copy_(blocks + (rank * world) + base)
        dist.all_to_all_single(a_out, a_in)
        torch.cuda.current_stream().wait_stream(side)

    # the collectives once launch by launch (communicator set-up), then captured
    body()
    quiesce()        # (`side` carried a collective and joins the capture: the watchdog's list must be empty by then)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        body()
    ranks = torch.arange(world, dtype=torch.float32, device=dev)
    for it in range(3):
        b = float(10 * it + 1)
        base.fill_(b)
        for t in (out, rs, ar, a_out, side_out):
            t.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        ok = bool((out.view(world, n) == (ranks + b)[:, None]).all())
        ok &= bool((side_out.view(world, n) == 3.0 * (ranks + b)[:, None]).all())
        ok &= bool((rs == 2.0 * world * (b + rank)).all())
        ok &= bool((ar == 2.0 * world * (world * b + world * (world - 1) / 2.0)).all())
        ok &= bool((a_out.view(world, m) == (ranks * world + rank + b)[:, None]).all())
        if not ok:
            print(f"rccl_graph_probe: rank {rank}: wrong values in replay {it}", file=sys.stderr)
            return 3
    dist.barrier()
    if rank == 0:
        print(f"rccl_graph_probe: {world} rank(s): captured collectives replay correctly")
    pull_ok, pull_note = True, "ok"
    try:
        pull_ok, pull_note = pull_pattern(rank, world, dev)
    except Exception as e:  # noqa: BLE001  (the basic capture above is what decides the graph; the pull pattern only decides the form)
        pull_ok, pull_note = False, f"{type(e).__name__}: {e}"
    if not pull_ok:
        print(f"rccl_graph_probe: rank {rank}: pull pattern: {pull_note}", file=sys.stderr)
    import json
    print(json.dumps({"captured": True, "pull": bool(pull_ok), "pull_note": pull_note}))
    sys.stdout.flush()
    os._exit(0)        # (no communicator teardown: graphs that hold the communicator's kernels are still alive; the process has nothing left to do)


def pull_pattern(rank: int, world: int, dev):
    """The owner-sharded form's row pull as a captured step issues it (``parallel.RowPullDev`` with ``defer_exchange``): requester r asks owner
    p for the rows ``req[p, :]`` (-1 = unused slot), the owner gathers them from its table, the rows come back in the same block layout."""
    cp, width, rows = 512, 172, 4096
    table = (torch.arange(rows, dtype=torch.float32, device=dev)[:, None] * 8.0 + float(rank)
             + torch.arange(width, dtype=torch.float32, device=dev)[None, :] / 256.0)        # table[i, c] encodes (row, owner, column) exactly (24 bits)
    seed = torch.zeros(1, dtype=torch.int64, device=dev)
    req = torch.empty((world, cp), dtype=torch.int32, device=dev)
    asked = torch.empty((world, cp), dtype=torch.int32, device=dev)
    rows_out = torch.empty((world * cp, width), dtype=torch.float32, device=dev)
    rows_in = torch.zeros((world * cp, width), dtype=torch.float32, device=dev)
    slots = torch.arange(world * cp, dtype=torch.int64, device=dev).reshape(world, cp)

    def body(group):
        ids = (slots * 7 + seed + rank * 13) % rows
        ids = torch.where(slots % 5 == 4, torch.full_like(ids, -1), ids)                              # holes, as in the fixed-capacity blocks
        req.copy_(ids.to(torch.int32))
        dist.all_to_all_single(asked.reshape(-1), req.reshape(-1), group=group)                       # synchronous, capturing stream
        a = asked.reshape(-1).long()
        rows_out.copy_(torch.where((a >= 0)[:, None], table[a.clamp(min=0)], torch.zeros((), device=dev)))
        dist.all_to_all_single(rows_in, rows_out, group=group)

    def check():
        ids = ((slots * 7 + seed + rank * 13) % rows)
        hole = slots % 5 == 4
        owner = torch.arange(world, dtype=torch.float32, device=dev)[:, None].expand(world, cp)
        want = ids.to(torch.float32)[:, :, None] * 8.0 + owner[:, :, None] + torch.arange(width, dtype=torch.float32, device=dev) / 256.0
        want = torch.where(hole[:, :, None], torch.zeros((), device=dev), want)
        return bool((rows_in.reshape(world, cp, width) == want).all())

    second = dist.new_group(backend="nccl")
    for name, group in (("main communicator", None), ("second communicator", second)):
        body(group)                                   # once launch by launch (communicator set-up)
        quiesce()
        if not check():
            return False, f"wrong values launch by launch on the {name}"
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            body(group)
        for it in range(3):
            seed.fill_(101 * it + 17)
            rows_in.fill_(-1.0)
            graph.replay()
            torch.cuda.synchronize()
            if not check():
                return False, f"wrong values in replay {it} on the {name}"
    dist.barrier()
    return True, "ok"


if __name__ == "__main__":
    try:
        code = main()
    except Exception as e:  # noqa: BLE001
        print(f"rccl_graph_probe: {type(e).__name__}: {e}", file=sys.stderr)
        code = 2
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)

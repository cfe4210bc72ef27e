#!/usr/bin/env python3
"""Can this torch / RCCL capture collectives into a HIP graph?  One rank (world size 1 over RCCL, the only multi-process set-up a
one-GPU box allows); every case prints one line.  Decides how ``lstep_amd.parallel`` replays its iteration: whole-step graph with the
collectives inside, or graph segments between eagerly issued collectives.
usage: python tools/rccl_capture_probe.py  (through gpurun)"""
import os
import time
import traceback

import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    g2 = dist.new_group(backend="nccl")
    n = 1 << 20
    a = torch.arange(n, dtype=torch.float32, device=dev)
    out = torch.zeros(n, dtype=torch.float32, device=dev)
    rs = torch.zeros(n, dtype=torch.float32, device=dev)
    a2a = torch.zeros(n, dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(device=dev)

    def body(async_ops: bool, second_comm: bool):
        x = a * 2.0
        if async_ops:
            w = dist.all_gather_into_tensor(out, x, async_op=True)
            y = x + 1.0            # compute while the collective is "in flight"
            w.wait()
        else:
            dist.all_gather_into_tensor(out, x)
            y = x + 1.0
        z = out + y
        dist.reduce_scatter_tensor(rs, z)
        dist.all_reduce(rs)
        variant = os.environ.get("PROBE_VARIANT", "")
        if second_comm and variant == "comm2_main_stream":           # second communicator, but issued from the capturing stream itself
            dist.all_to_all_single(a2a, rs, group=g2)
        elif second_comm and variant == "comm1_side_stream":         # main communicator, issued from a side stream that joined the capture
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dist.all_to_all_single(a2a, rs)
            torch.cuda.current_stream().wait_stream(side)
        elif second_comm and variant == "a2a_async_main":            # asynchronous all-to-all issued from the capturing stream, waited for later
            w = dist.all_to_all_single(a2a, rs, async_op=True)
            y2 = rs + 1.0
            w.wait()
            a2a.add_(y2 * 0.0)
        elif second_comm and variant == "a2a_twice_main":            # two exchanges in a row (the pull: requests, then rows), a kernel in between
            w = dist.all_to_all_single(a2a, rs, async_op=True)
            mid = w.wait() if False else None
            w.wait()
            tmp = a2a * 1.0
            w2 = dist.all_to_all_single(a2a, tmp, async_op=True)
            w2.wait()
        elif second_comm and variant == "allgather_async_side":      # asynchronous all-gather from a side stream (update_pe's phase-1 rows)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                w = dist.all_gather_into_tensor(a2a, rs, async_op=True)
                w.wait()
            torch.cuda.current_stream().wait_stream(side)
        elif second_comm and variant == "comm1_side_stream_allgather":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dist.all_gather_into_tensor(a2a, rs)
            torch.cuda.current_stream().wait_stream(side)
        elif second_comm:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dist.all_to_all_single(a2a, rs, group=g2)
            torch.cuda.current_stream().wait_stream(side)
        else:
            dist.all_to_all_single(a2a, rs)
        return a2a * 1.0

    only = os.environ.get("PROBE_CASE")       # "a,s" -- one case per process (a faulting capture must not hide the other cases)
    for async_ops in (False, True):
        for second_comm in (False, True):
            if only is not None and only != f"{int(async_ops)},{int(second_comm)}":
                continue
            tag = f"async={int(async_ops)} second_comm={int(second_comm)}"
            try:
                for _ in range(3):
                    ref = body(async_ops, second_comm).clone()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    res = body(async_ops, second_comm)
                a.add_(1.0)
                want = body(async_ops, second_comm).clone()
                res.zero_()
                g.replay()
                torch.cuda.synchronize()
                ok = bool(torch.equal(res, want))
                t0 = time.perf_counter()
                for _ in range(200):
                    g.replay()
                torch.cuda.synchronize()
                t_rep = (time.perf_counter() - t0) / 200
                t0 = time.perf_counter()
                for _ in range(200):
                    body(async_ops, second_comm)
                torch.cuda.synchronize()
                t_eag = (time.perf_counter() - t0) / 200
                print(f"[probe] capture {tag}: ok={ok} replay {t_rep * 1e6:.1f} us eager {t_eag * 1e6:.1f} us", flush=True)
                del g
            except Exception as e:  # noqa: BLE001
                print(f"[probe] capture {tag}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)
                traceback.print_exc()
                torch.cuda.synchronize()
    # per-collective eager cost at the sizes of a c4 step (host time per call, one rank)
    for name, fn in (("all_gather 22MB", lambda: dist.all_gather_into_tensor(torch.empty(32768 * 172, device=dev), torch.empty(32768 * 172, device=dev))),
                     ("all_reduce 2.3MB", lambda: dist.all_reduce(torch.empty(600000, device=dev)))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        t_host = (time.perf_counter() - t0) / 100
        torch.cuda.synchronize()
        print(f"[probe] eager {name}: host {t_host * 1e6:.1f} us per call", flush=True)
    print("[probe] torch", torch.__version__, "nccl", torch.cuda.nccl.version(), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if os.environ.get("PROBE_CASE") is None and os.environ.get("PROBE_INLINE") != "1":
        # parent: never touches the GPU; one child per case
        import subprocess
        import sys
        cases = [("0,0", ""), ("0,1", ""), ("1,0", ""), ("0,1", "comm2_main_stream"), ("0,1", "comm1_side_stream"), ("0,1", "comm1_side_stream_allgather"),
                 ("0,1", "a2a_async_main"), ("0,1", "a2a_twice_main"), ("0,1", "allgather_async_side")]
        if os.environ.get("PROBE_ONLY"):
            cases = [c for c in cases if c[1] in os.environ["PROBE_ONLY"].split(",")]
        for i, (case, variant) in enumerate(cases):
            env = dict(os.environ, PROBE_CASE=case, PROBE_VARIANT=variant, MASTER_PORT=str(29577 + i))
            print(f"[probe] ---- variant '{variant}'")
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=240)
            lines = [ln for ln in (r.stdout + r.stderr).splitlines() if "[probe]" in ln or "Error" in ln or "error" in ln]
            print(f"[probe] case async,second_comm={case}: exit code {r.returncode}")
            print("\n".join(lines[-12:]), flush=True)
    else:
        main()

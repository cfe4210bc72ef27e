#!/usr/bin/env python3
"""Does an NCCL-work event that was recorded INSIDE a graph capture poison the process group's event cache?  torch caches the HIP events
of its collective Work objects (TORCH_NCCL_CUDA_EVENT_CACHE, on by default); a Work created during a capture records its end event into
the capturing stream and hands the event back to the cache when it dies; an eager collective that picks the same event up later is tracked
by the watchdog thread, whose hipEventQuery then fails with hipErrorCapturedEvent and terminates the process (seen once in round 4:
`bench.py` in the pull form, in the launch-by-launch iterations behind the timed graph replays).
usage (through gpurun): python tools/nccl_event_cache_probe.py            # runs both settings in child processes
"""
import os
import subprocess
import sys
import time


def child():
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("PROBE_PORT", "29641"), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    n = 1 << 16
    x = torch.ones(n, device=dev)
    out = torch.zeros(n, device=dev)
    for _ in range(4):        # communicator set-up, launch by launch
        dist.all_gather_into_tensor(out, x, async_op=True).wait()
    torch.cuda.synchronize()
    a_in, a_out = torch.ones(n, device=dev), torch.zeros(n, device=dev)
    side = torch.cuda.Stream(device=dev)
    graphs = []
    for rounds in range(2):   # (the pull form captures again when its request blocks grow)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            works = [dist.all_gather_into_tensor(out, x, async_op=True) for _ in range(8)]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                dist.all_gather_into_tensor(out.clone(), x)          # synchronous, on a side stream that joined the capture
            for w in works:
                w.wait()
            dist.all_reduce(out)
            dist.all_to_all_single(a_out, a_in)                      # synchronous, capturing stream
            dist.all_to_all_single(a_in, a_out)
            torch.cuda.current_stream().wait_stream(side)
        del works, w          # the captured Work objects die: their events go back to the cache
        graphs.append(g)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        for i in range(120):  # launch-by-launch collectives behind the replays, as bench.py's event-timed iterations issued them
            w = dist.all_gather_into_tensor(out, x, async_op=True)
            w2 = dist.all_to_all_single(a_out, a_in, async_op=True)
            y = out * 2.0
            w.wait()
            w2.wait()
            dist.all_reduce(y)
            if i % 15 == 0:
                time.sleep(0.15)  # (the watchdog polls its list of outstanding works every 100 ms)
    torch.cuda.synchronize()
    time.sleep(1.0)
    print("ok", flush=True)
    os._exit(0)


if __name__ == "__main__":
    if os.environ.get("PROBE_CHILD") == "1":
        child()
    for i, setting in enumerate(("1", "0")):
        env = dict(os.environ, PROBE_CHILD="1", TORCH_NCCL_CUDA_EVENT_CACHE=setting, PROBE_PORT=str(29641 + i))
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
        tail = [ln for ln in (r.stderr or "").splitlines() if "HIP error" in ln or "terminate" in ln][:2]
        print(f"TORCH_NCCL_CUDA_EVENT_CACHE={setting}: exit code {r.returncode} {r.stdout.strip()} {' | '.join(tail)[:300]}")

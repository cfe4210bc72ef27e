#!/usr/bin/env python3
"""Hypothesis for the sporadic "Process group watchdog thread terminated with exception: HIP error: operation not permitted on an event last
recorded in a capturing stream" (round 4 once in bench.py, round 5 once in the RCCL world-1 golden-trace test): torch's NCCL watchdog
polls the END EVENT of every collective issued launch by launch until it has completed (a sweep every ~100 ms).  A synchronous collective
runs on the CURRENT stream, so its end event is recorded on that stream.  If the same stream starts (or joins) a graph capture before the
watchdog's next sweep, HIP's hipEventQuery on that event fails with hipErrorCapturedEvent -- although the record itself was not captured --
and the watchdog takes the process down.

    python tools/nccl_capture_after_eager_probe.py          # runs the scenarios in child processes, prints which ones die

scenarios: same   eager collective on stream S, capture begins on S at once, held open 0.4 s
           other  eager collective on stream S, capture on another stream T (S never captures)
           drain  as `same`, but the device is synchronised and the host sleeps 0.25 s (one watchdog sweep) before the capture begins
           lib    as `same`, with lstep_amd.parallel.quiesce_collectives() in front of the capture (what the package does)
"""
import os
import subprocess
import sys
import time


def child(mode: str):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("PROBE_PORT", "29651"), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    n = 1 << 20
    x = torch.ones(n, device=dev)
    out = torch.zeros(n, device=dev)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    time.sleep(0.3)
    s, t = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for rep in range(int(os.environ.get("PROBE_REPS", "20"))):
        with torch.cuda.stream(s):
            dist.all_gather_into_tensor(out, x)               # synchronous: on the current stream s; its Work goes to the watchdog's list
            dist.all_reduce(out)
        if mode == "drain":
            torch.cuda.synchronize()
            time.sleep(0.25)
        if mode == "lib":
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
            from lstep_amd.parallel import quiesce_collectives
            quiesce_collectives(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=(t if mode == "other" else s), capture_error_mode="thread_local"):
            y = x * 2.0
            time.sleep(0.4)                                   # the capture stays open across several watchdog sweeps
            y = y + 1.0
        g.replay()
        torch.cuda.synchronize()
    print("ok", flush=True)
    os._exit(0)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        return
    modes = sys.argv[1:] or ["other", "drain", "lib", "same"]
    for i, mode in enumerate(modes):
        env = dict(os.environ, PROBE_PORT=str(29651 + i))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode], env=env, capture_output=True, text=True, timeout=600)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if "watchdog" in l or "capturing" in l or l.strip() == "ok"]
        print(f"{mode:6s} exit code {r.returncode}: " + (" | ".join(tail[-2:]) if tail else "(no output)"), flush=True)


if __name__ == "__main__":
    main()

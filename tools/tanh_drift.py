#!/usr/bin/env python3
"""Long-trace drift of the hardware-exp `tanh_fast` (csrc/lstep_mma.h) against libm's tanhf (round-3 ADVICE).

Builds a second library with -DLSTEP_EXACT_TANH=1, runs the SAME long training trace (launch by launch, deterministic kernels) once per
library in a child process each, and compares the PE table, the newest history snapshot, the loss and the weights at checkpoints.
The golden traces cover 16 batches; this covers thousands, on a graph small enough that every row is rewritten hundreds of times
(`pe += tanh(...)` in update_pe and in the tail: the error of one evaluation, <= 1.5e-7, enters the table every step).
usage: python tools/tanh_drift.py [steps=2000] [workload=tiny]     (through gpurun; prints one line per checkpoint)"""
import json
import os
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)


def child(steps: int, workload: str, out: str):
    import torch
    from lstep_amd.optim import FusedAdam
    from lstep_amd.workload import build_workload, evolve_history
    dev = torch.device("cuda", 0)
    wl = build_workload(workload, dev, seed=0, num_fft_batches=16)
    eng, model = wl.engine, wl.model
    model.train()
    opt = FusedAdam(model.parameters(), lr=1e-4)
    B = wl.batch
    start = 20 * B
    evolve_history(eng, wl.stream, start, B, wl.num_nodes)
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    marks = sorted({m for m in (16, 64, 256, 1024, 4096, steps) if m <= steps})
    rec = {}
    for i in range(steps):
        lo = start + (i * B) % (wl.num_edges - start - B)
        src, dst, ts, eid = wl.stream.batch(lo, lo + B)
        neg = torch.randint(1, wl.num_nodes + 1, (B,), generator=gen, device=dev)
        res = eng.train_iteration(opt, 1000 + i, src, dst, ts, eid, neg)
        if i + 1 in marks:
            rec[i + 1] = {"table": eng.ring.last().detach().cpu(), "loss": float(res["loss"]),
                          "weights": torch.cat([(torch.view_as_real(p) if p.is_complex() else p).detach().reshape(-1).cpu() for p in model.parameters()])}
    torch.save(rec, out)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    workload = sys.argv[2] if len(sys.argv) > 2 else "tiny"
    if os.environ.get("TANH_DRIFT_CHILD"):
        return child(steps, workload, os.environ["TANH_DRIFT_CHILD"])
    from lstep_amd import _native as nat
    exact = os.path.join(nat.CSRC, "liblstep_hip_exact_tanh.so")
    nat.build_library(defines=["LSTEP_EXACT_TANH=1"], lib_path=exact)
    nat.build_library()
    outs = {}
    # controls: the SAME library run twice (run-to-run noise: the few float atomics left in the step), and the exact-tanh library with another
    # -- equally valid -- summation order of the segment sums (LSTEP_SEGMENT_ATOMICS=1): how far do two trajectories drift apart under ANY
    # rounding-level perturbation?  (Adam's updates are +-lr per step almost regardless of the gradient's size: a perturbed sign is a
    # 1e-4 step, and training is a chaotic map of its rounding errors.)
    runs = (("fast", nat.LIB_PATH, {}), ("exact", exact, {}), ("fast_again", nat.LIB_PATH, {}), ("exact_other_sum_order", exact, {"LSTEP_SEGMENT_ATOMICS": "1"}))
    for name, lib, extra in runs:
        outs[name] = f"/tmp/tanh_drift_{name}.pt"
        env = dict(os.environ, TANH_DRIFT_CHILD=outs[name], LSTEP_LIB=lib, **extra)
        subprocess.run([sys.executable, os.path.abspath(__file__), str(steps), workload], env=env, check=True)
    import torch
    res = {k: torch.load(v) for k, v in outs.items()}
    for what, (x, y) in (("tanh_fast vs tanhf", ("fast", "exact")), ("control: tanh_fast vs tanh_fast (second run)", ("fast", "fast_again")),
                         ("control: tanhf vs tanhf with float-atomic segment joins", ("exact", "exact_other_sum_order"))):
        a, b = res[x], res[y]
        for m in sorted(a):
            dt = float((a[m]["table"] - b[m]["table"]).abs().max())
            dw = float((a[m]["weights"] - b[m]["weights"]).abs().max())
            print(json.dumps({"pair": what, "steps": m, "workload": workload, "max_abs_table_diff": dt, "max_abs_weight_diff": dw,
                              "loss_a": a[m]["loss"], "loss_b": b[m]["loss"], "table_abs_max": float(a[m]["table"].abs().max())}))


if __name__ == "__main__":
    main()

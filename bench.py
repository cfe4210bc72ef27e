#!/usr/bin/env python3
"""bench.py -- processed edges/s of the L-STEP training iteration (fwd + bwd + update_pe + Adam) on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line (rank 0).  A step is one training
iteration of the reference protocol (``train_LSTEP_link_prediction.py:204-311``: FFT splice, 3x combining_pe_raw_feat,
link predictor, BCE + PE loss, update_pe, history append, backward, Adam) on one batch of synthetic edges, history
window full (t = T).  Inputs are resident in HBM before the timed region.

Extra objects on the line:
  roofline      the neighbour-gather kernel (lstep_gather_aggregate_fwd): algorithmic bytes of SURVEY.md 8(d)
                (sum over rows of 688*(2k + v + 3) + 24k + 8v, k/v counted exactly from the kernel's own per-row
                counts) / the launch's duration from HIP events recorded on the launch stream, vs 8 TB/s HBM.
  cpu_baseline  the CPU oracle (oracle/lstep_oracle.py, the parity-checked restatement that executes the reference's op
                sequence) timed on this box's host cores on a bounded, scaled-down sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
# edge + node + PE channels, CSR search (not the explicit-list variant); last argument: the instantiation with the workgroup-shared long-row path
# (graphs whose longest adjacency row exceeds 256 entries: power-law graphs, the reference's small dense datasets)
GATHER_KERNEL = "lstep::gather_aggregate_fwd_kernel<true, true, false, false>"
GATHER_KERNEL_LONG_ROWS = "lstep::gather_aggregate_fwd_kernel<true, true, false, true>"


def gather_algorithmic_bytes(count: torch.Tensor, K: int, G: int, row_bytes: int = 688) -> float:
    """per node-call 688*(k + k + v + 3) + 24k + 8v with k = min(c, K), v = min(c, G)."""
    c = count.to(torch.int64)
    k = c.clamp(max=K)
    v = c.clamp(max=G)
    return float((row_bytes * (2 * k + v + 3) + 24 * k + 8 * v).sum().item())


def pair_gather_launches(sink, K: int, G: int):
    """(ms per gather STAGE, algorithmic bytes per stage) from the (start event, end event, per-row counts, branches) records of the timed
    launches.  The single-GPU engine computes all channels in one launch (branches = 3).  The multi-GPU engine launches the stage as TWO
    kernels per step -- edge + node channels (1), then the PE channel (2) once the all-gathered spliced rows are in: a (1, 2) pair over
    the same rows is one stage, its durations add up and its bytes are counted once.  Anything else in the sink is an error, not a
    silently mis-priced launch."""
    ms, nbytes, i = [], [], 0
    while i < len(sink):
        e0, e1, count, br = sink[i]
        if br == 3:
            ms.append(e0.elapsed_time(e1))
            nbytes.append(gather_algorithmic_bytes(count, K, G))
            i += 1
            continue
        if br != 1 or i + 1 >= len(sink) or sink[i + 1][3] != 2 or sink[i + 1][2].numel() != count.numel():
            raise RuntimeError(f"gather event sink: launch {i} (branches {br}) is not the first half of an (edge+node, PE) pair")
        f0, f1 = sink[i + 1][0], sink[i + 1][1]
        ms.append(e0.elapsed_time(e1) + f0.elapsed_time(f1))
        nbytes.append(gather_algorithmic_bytes(count, K, G))
        i += 2
    return ms, nbytes


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(workload: str = "synth-1M-20M", time_gap: int = 2000, batch: int = None, seconds_budget: float = 25.0):
    """Oracle train iterations (the reference's op sequence) on this box's host cores.

    The reference's own batch sizes fit the CPU oracle whole, so the enron / wikipedia / reddit lines are timed on THE SAME configuration the
    GPU ran (nodes, edges, batch, K, time_gap, T; same synthetic generator) -- a like-for-like ratio.  The 1 M- and 4 M-node workloads do
    not (the oracle's history tensor alone is 69 / 275 GB and one iteration takes minutes): they keep the bounded sample of earlier
    rounds -- same degree structure (E / N = 20), K, time_gap, T; 50 k nodes, batch 512 -- and the line says so (smaller N favours the CPU:
    its per-batch O(N T P) history cat shrinks, so the GPU / CPU ratio read from it is conservative)."""
    from lstep_amd import protocol, synth
    from lstep_amd.workload import WORKLOADS
    from oracle.lstep_oracle import OracleNeighborSampler, build_oracle_model  # checker / baseline only

    n_w, e_w, b_w, k_w = WORKLOADS[workload]
    same = n_w <= 20_000
    if same:
        N, E, B, K = n_w, e_w, (batch or b_w), k_w
    else:
        N, E, B, K = 50_000, 1_000_000, 512, k_w
    G, T = time_gap, 100
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    # the GPU box gives one GPU a share of 16 host cores; more torch threads than that only oversubscribes
    # (256 threads ran this sample 20x slower than 8).  `cores` reports the threads actually used.
    cores = max(1, min(avail, int(os.environ.get("LSTEP_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    g = synth.make_temporal_graph(N, E, seed=0)
    node_raw, edge_raw = synth.make_features(N, E, seed=1)
    model = build_oracle_model(node_raw, edge_raw, OracleNeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=N), K, T,
                               synth.make_state_dict(K, T))
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    hist = 0.1 * torch.randn(N + 1, T, synth.PE_DIM, generator=torch.Generator().manual_seed(0))
    state = protocol.ProtocolState(history=hist)
    start = E // 2
    times, first = [], None
    it = 0
    t_begin = time.perf_counter()
    while True:
        sl = slice(start + it * B, start + (it + 1) * B)
        neg = synth.make_negatives(N, B, seed=it)
        t0 = time.perf_counter()
        protocol.train_iteration(model[0], model[1], opt, state, 1000 + it, g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl], neg, K, G, T)
        dt = time.perf_counter() - t0
        if it > 0:  # first iteration = warm-up
            times.append(dt)
        else:
            first = dt
        it += 1
        spent = time.perf_counter() - t_begin
        if (len(times) >= 2 and spent > seconds_budget) or len(times) >= 8:
            break
        if len(times) >= 1 and spent + dt > 2 * seconds_budget:      # (Reddit shape: ~10 s per iteration -- one timed iteration after the warm-up)
            break
    per_iter = float(np.mean(times))
    what = (f"the SAME configuration as the GPU line ({N} nodes / {E} edges, batch {B}, K={K}, time_gap={G}, T={T} history full)" if same else
            f"a bounded SAMPLE of the workload: synthetic {N} nodes / {E} edges (same E/N as the GPU workload), batch {B}, K={K}, time_gap={G}, T={T} history full")
    return {"value": B / per_iter, "unit": "edges/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(), "host_cores_visible": avail,
            "same_config_as_gpu_line": bool(same),
            "sample": f"oracle (reference op sequence) train iteration on {what}; {len(times)} timed iteration(s) after 1 warm-up ({first * 1e3:.0f} ms), "
                      f"{per_iter * 1e3:.0f} ms each"}


# Where the gather kernel's rows come from, by table size (MI355X_MICROARCH.md: L2 4 MiB per XCD / 32 MiB aggregate at ~34.5 TB/s; 256 MiB Infinity
# Cache, random-row gathers from a 38 MB table at 8.6 TB/s chip-wide; HBM 8 TB/s spec).  A table of at most 4 MiB is L2-resident on every
# XCD; one that, together with the launch's other tables, fits ~200 MiB stays in the Infinity Cache between launches; anything else is HBM.
LEVEL_PEAK_GBS = {"l2": 34500.0, "infinity_cache": 8600.0, "hbm": HBM_PEAK_GBS}


def table_level(nbytes: float) -> str:
    return "l2" if nbytes <= 4 * 2 ** 20 else ("infinity_cache" if nbytes <= 200 * 2 ** 20 else "hbm")


def gather_roofline_bound(count: torch.Tensor, K: int, G: int, num_nodes: int, num_edges: int, row_bytes: int = 688):
    """(bound, blended peak GB/s, per-level byte split) of one gather launch: SURVEY.md 8(d)'s algorithmic bytes attributed to the table they
    are read from -- node rows (v + 1 per row) from node_raw, edge rows (k) from edge_raw, PE rows (k + 1) from the PE table, the output
    row and the CSR slices streamed once (HBM) -- each priced at the peak of the level that table lives in.  The blended peak is total
    bytes / sum(bytes_level / peak_level): the rate at which a launch that ran every level at its peak would move the algorithmic bytes;
    `bound` names the level with the largest share of that time.  For the 1 M-node workloads every table is HBM-sized: bound = hbm, peak = 8 TB/s."""
    c = count.to(torch.int64)
    k = c.clamp(max=K).sum().item()
    v = c.clamp(max=G).sum().item()
    rows = c.numel()
    node_tab, edge_tab = (num_nodes + 1) * row_bytes, (num_edges + 1) * row_bytes
    split = {"l2": 0.0, "infinity_cache": 0.0, "hbm": 0.0}
    split[table_level(node_tab)] += row_bytes * (v + rows)             # neighbours' node rows + the row's own
    split[table_level(node_tab)] += row_bytes * (k + rows)             # PE table: same shape as node_raw
    split[table_level(edge_tab)] += row_bytes * k
    split["hbm"] += row_bytes * rows + 24 * k + 8 * v                  # output row, CSR slices
    total = sum(split.values())
    t = {lv: split[lv] / LEVEL_PEAK_GBS[lv] for lv in split}
    bound = max(t, key=t.get)
    return bound, total / sum(t.values()), {lv: split[lv] for lv in split}


def measure_gather_traffic(args, kernel_names):
    """HBM bytes per launch of the gather kernel from the PMC counters of a short CHILD run of this very command under rocprofv3 -- a process
    cannot read its own launches' counters.  Collected and corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
    `--pmc` passes (with --kernel-trace only), read bytes = 2 x FETCH_SIZE x 1024 on gfx950 (wide coalesced reads are reported at half),
    write bytes = WRITE_SIZE x 1024; averaged over the child's timed training iterations.  Returns (bytes, description) or None (no
    rocprofv3, a pass failed or timed out: the caller falls back to the committed file)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None
    steps = 4
    child = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", "1", "--prime", "2", "--history", "random", "--graph", "off",
             "--no-cpu-baseline", "--traffic", "off", "--workload", args.workload, "--time-gap", str(args.time_gap)]
    if args.batch is not None:
        child += ["--batch", str(args.batch)]
    if args.zipf:
        child += ["--zipf", str(args.zipf)]
    tmp = tempfile.mkdtemp(prefix="lstep_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "--kernel-trace", "-d", out, "-o", "c", "--output-format", "csv", "--"] + child
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=float(os.environ.get("LSTEP_PMC_TIMEOUT", "240")))
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            rows = list(csv.DictReader(open(files[0])))
            if rows and "Dispatch_Id" in rows[0]:
                rows.sort(key=lambda q: int(q["Dispatch_Id"]))
            vals = [float(q["Counter_Value"]) for q in rows
                    if q["Counter_Name"] == counter and any(q["Kernel_Name"].replace("void ", "").startswith(k) for k in kernel_names)]
            if len(vals) < steps:
                return None
            got[counter] = float(np.mean(vals[-steps:]))
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    traffic = 2.0 * got["FETCH_SIZE"] * 1024.0 + got["WRITE_SIZE"] * 1024.0
    return traffic, (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over a child run of this command ({steps} training iterations, launch by "
                     f"launch): 2 x {got['FETCH_SIZE']:.0f} KiB read (gfx950 correction) + {got['WRITE_SIZE']:.0f} KiB written per launch")


def default_workload(gpus: int) -> str:
    """BASELINE.json: configs[3] (1 M nodes / 20 M edges, B = 16384) is the single-GPU workload and the 4-GPU one; configs[4]
    (4 M / 100 M) the 8-GPU one.  Weak scaling: every rank contributes B = 16384 edges to the global batch."""
    return "synth-4M-100M" if gpus >= 8 else "synth-1M-20M"


def visible_gpus() -> int:
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (the parent of the ranks must stay GPU-free: a process that
    has initialised HIP must not fork / exec workers): the *_VISIBLE_DEVICES lists if set, otherwise the KFD topology (GPU nodes have a
    non-zero simd_count)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return -1        # unknown: let the ranks fail fast on a missing device
    return n


def launch_ranks(gpus: int, argv) -> int:
    import subprocess
    have = visible_gpus()
    if 0 <= have < gpus and os.environ.get("LSTEP_SINGLE_DEVICE") != "1":
        print(f"bench.py: --gpus {gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    # --standalone: torchrun picks (and keeps) a free rendezvous port itself -- no bind-then-release race
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={gpus}",
           os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def rccl_probe_verdict(world: int, rank: int, local_rank: int) -> dict:
    """Multi-rank runs replay ONE captured graph per step with the RCCL collectives inside (``parallel.GraphedDistStep``) -- something a
    one-GPU box can only rehearse with a single rank.  Before this rank touches its GPU, a child process per rank captures and replays
    the same kinds of collectives across the N ranks (tools/rccl_graph_probe.py), and then the owner-sharded form's exchange pattern (two
    synchronous equal-split all-to-alls from the capturing stream, on the main and on a second communicator).  Returns
    {"captured": bool, "pull": bool, "note": str}: on anything but a clean exit the bench keeps the device-driven iteration but issues it
    launch by launch instead of failing the run on a capture it could never try; "pull" (the form ``DistributedLstep`` takes beyond four
    ranks) needs the child's own verdict on the exchange pattern as well, else the run takes "replicate"."""
    import subprocess
    if world <= 1 and os.environ.get("LSTEP_FORCE_GRAPH_PROBE") != "1":
        return {"captured": True, "pull": True, "note": "skipped (one rank)"}
    if os.environ.get("LSTEP_DIST_GRAPH", "1") == "0":
        return {"captured": False, "pull": False, "note": "skipped (LSTEP_DIST_GRAPH=0)"}
    if os.environ.get("LSTEP_SKIP_GRAPH_PROBE") == "1" or os.environ.get("LSTEP_DIST_BACKEND", "nccl") != "nccl":
        return {"captured": True, "pull": True, "note": "skipped"}
    env = dict(os.environ)
    env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank), MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
               MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29533")) + 23))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE"):
        env.pop(k, None)       # (the child is a plain env:// rendezvous on its own port, not a member of the parent's elastic job)
    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "rccl_graph_probe.py")]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=float(os.environ.get("LSTEP_GRAPH_PROBE_TIMEOUT", "240")))
    except subprocess.TimeoutExpired:
        return {"captured": False, "pull": False, "note": "timeout"}
    if r.returncode == 0:
        pull, note = False, "ok (no verdict on the pull pattern)"
        for ln in reversed((r.stdout or "").strip().splitlines()):
            if ln.startswith("{"):
                try:
                    v = json.loads(ln)
                    pull = bool(v.get("pull"))
                    note = "ok" if pull else f"ok; pull pattern: {v.get('pull_note')}"
                except ValueError:
                    pass
                break
        return {"captured": True, "pull": pull, "note": note}
    tail = (r.stderr or r.stdout or "").strip().splitlines()
    return {"captured": False, "pull": False, "note": f"exit code {r.returncode}" + (f": {tail[-1][:160]}" if tail else "")}


def rccl_graph_probe(world: int, rank: int, local_rank: int):
    """(captured ok, note) of ``rccl_probe_verdict``."""
    v = rccl_probe_verdict(world, rank, local_rank)
    return v["captured"], v["note"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, help="default by --gpus: 1, 2, 4 -> synth-1M-20M (BASELINE.json configs[3]), 8 -> synth-4M-100M "
                                                     "(configs[4]); the per-GPU batch is fixed (weak scaling), the global batch is gpus x batch")
    ap.add_argument("--time-gap", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prime", type=int, default=12, help="training iterations run as part of the set-up pre-roll, before the warm-up steps")
    ap.add_argument("--graph", choices=["on", "off"], default="on",
                    help="on: steady-state training iterations are replayed from ONE captured HIP graph (engine.GraphedTrainStep, single GPU); "
                         "off: every launch issued from Python")
    ap.add_argument("--mode", choices=["train", "eval"], default="train", help="eval = evaluate_model_utils.py:38-142 iteration (4x combine, no backward)")
    ap.add_argument("--zipf", type=float, default=None, help="power-law endpoint popularity exponent (hub-skew variant)")
    ap.add_argument("--sampler", choices=["recent", "uniform", "time_interval_aware"], default="recent",
                    help="neighbour sampling strategy (utils/utils.py:175-208).  The RNG-defined ones are drawn on the host in the reference's order "
                         "(native replay of numpy's generator) and fed to the explicit-neighbourhood kernels: time_gap draws per row and call")
    ap.add_argument("--traffic", choices=["auto", "file", "off"], default="auto",
                    help="roofline.traffic (HBM bytes per launch of the gather kernel, PMC counters): auto = two short rocprofv3 --pmc child runs of "
                         "this very command when rocprofv3 is on the box (FETCH_SIZE and WRITE_SIZE in separate passes), else the committed "
                         "profiles/*pmc_traffic.json of the same workload; file = the committed file only; off = null")
    ap.add_argument("--history", choices=["evolved", "random"], default="evolved",
                    help="state of the T-snapshot PE history when the run starts: evolved = built by the algorithm itself over the T batches "
                         "before the first one (snapshots are clones of their predecessor plus the rows the batch wrote, as in the reference); "
                         "random = T independent random tables (every row differs between snapshots: the worst case for the filter)")
    args = ap.parse_args()

    if args.workload is None:
        args.workload = default_workload(args.gpus)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU under torch.distributed.run, BEFORE anything
        # in this process touches the GPU (a process that has initialised HIP must not be re-exec'ed or forked), and hand its exit
        # code on.  Rank 0 of the child job prints the JSON line.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus {args.gpus}` (it starts its own ranks) "
                         f"or under torch.distributed.run with --nproc-per-node {args.gpus}")
    verdict = rccl_probe_verdict(world, rank, local_rank)      # (a child process: before THIS process initialises HIP)
    probe_ok, probe_note, pull_ok = verdict["captured"], verdict["note"], verdict["pull"]
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    if os.environ.get("LSTEP_SINGLE_DEVICE") == "1":   # rehearsal of the N > 1 plumbing on a one-GPU box (with LSTEP_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from lstep_amd.workload import build_workload

    # LSTEP_FORCE_DIST=1 runs the distributed engine (RCCL collectives, owner-sharded ring) even on one rank (rehearsal)
    use_dist = world > 1 or os.environ.get("LSTEP_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        from lstep_amd.parallel import DistributedLstep  # owner-sharded engine (RCCL)
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        backend = os.environ.get("LSTEP_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    wl = build_workload(args.workload, dev, time_gap=args.time_gap, batch=args.batch, seed=0, sharded=use_dist, zipf=args.zipf, sampler=args.sampler)
    eng, model = wl.engine, wl.model
    model.train()
    # Adam, lr 1e-4 (reference defaults, utils/load_configs.py:45,48) through the single-kernel implementation
    from lstep_amd.optim import FusedAdam
    opt = FusedAdam(model.parameters(), lr=1e-4)
    if not use_dist:
        runner = eng
        eng.use_step_graph = args.graph == "on" and args.mode == "train"
    else:
        from lstep_amd.workload import prefill_distributed
        if world > 1:       # every rank takes the same path: the graph / the owner-sharded form only if every rank's probe came back clean
            flag = torch.tensor([1 if probe_ok else 0, 1 if pull_ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            got = flag.tolist()
            if got[0] == 0 and probe_ok:
                probe_ok, probe_note = False, "another rank's probe failed"
            if got[1] == 0 and pull_ok:
                pull_ok, probe_note = False, probe_note + "; another rank's pull-pattern probe failed"
        # the probe's verdict decides what DistributedLstep does by default: the whole-step graph (W > 1) only with a clean capture probe, the
        # owner-sharded "pull" form beyond four ranks only with a clean pull-pattern probe ("replicate" otherwise)
        runner = DistributedLstep(eng, opt, probe={"captured": probe_ok, "pull": probe_ok and pull_ok})
        if not probe_ok:
            runner.use_step_graph = False
            if rank == 0:
                print(f"bench.py: captured-collective probe: {probe_note}: the multi-GPU iteration is issued launch by launch", file=sys.stderr)
        prefill_distributed(runner, seed=0)
    B = wl.batch
    need = (args.warmup + args.steps + 17) * B * world      # (+10 + 5: the launch-by-launch iterations that time the gather kernel in-step and with idle neighbours)
    if need > wl.num_edges:
        raise SystemExit(f"{args.warmup + args.steps} batches of {B * world} edges do not fit the {wl.num_edges}-edge stream of workload {args.workload}")
    start = min(wl.num_edges // 2, wl.num_edges - need)   # from the middle of the stream when it fits
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    prerolled = 0
    prime = args.prime if args.mode == "train" else 0
    prime = max(0, min(prime, start // (B * world) - 1))
    if args.history == "evolved":
        from lstep_amd.workload import evolve_history
        prerolled = evolve_history(runner, wl.stream, start - prime * B * world, B * world, wl.num_nodes)

    negatives = {}

    def draw(i):
        """Negative destinations (and sources, evaluation) of step i, drawn in step order from ONE generator -- one step ahead of their
        use, so that the multi-GPU engine can request the rows of the next gather while the current step runs (``parallel.RowPull``)."""
        if i not in negatives:
            assert not negatives or i == max(negatives) + 1
            neg = torch.randint(1, wl.num_nodes + 1, (B * world,), generator=gen, device=dev)
            neg_src = torch.randint(1, wl.num_nodes + 1, (B * world,), generator=gen, device=dev) if args.mode == "eval" else None
            negatives[i] = (neg, neg_src)
        return negatives[i]

    def step(i):
        lo = start + i * B * world
        src, dst, ts, eid = wl.stream.batch(lo, lo + B * world)
        neg, neg_src = draw(i)
        negatives.pop(i - 1, None)
        nxt = None
        if lo + 2 * B * world <= wl.num_edges:     # the edge stream is known ahead: let the engine group the next batch's endpoints early
            s2, d2, t2, _ = wl.stream.batch(lo + B * world, lo + 2 * B * world)
            n2, ns2 = draw(i + 1)
            nxt = (s2, d2, t2, ns2, n2) if args.mode == "eval" else (s2, d2, t2, n2)
        if args.mode == "eval":
            with torch.no_grad():
                return runner.eval_iteration(1000 + i, src, dst, ts, eid, neg_src, neg, lookahead=nxt)
        return runner.train_iteration(opt, 1000 + i, src, dst, ts, eid, neg, lookahead=nxt)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Setup, not measurement: `prime` TRAINING iterations on the batches right before the first warm-up batch finish the pre-roll.  The
    # first training iterations of a process are 1.5-2x slower than the steady state (HIP-graph capture of the weight composition, the
    # first synchronous gradient sort of every call site, and above all the caching allocator, which needs about ten iterations of the
    # three-stream schedule before it stops calling hipMalloc); with the driver's `--warmup 5` they sat inside the timed region
    # (round-1 record: 4.35 ms/step over 20 steps whose last twelve ran at 3.6).
    pre_sink, pre_log, pre_steps = None, None, 0
    if use_dist and args.mode == "train" and prime >= 6:
        # Multi-GPU, graph mode: the launch-by-launch iterations that carry the timed events (gather kernel, every collective: the `comm`
        # object) are taken from the pre-roll, BEFORE the step is captured.  Behind the replays they ended the process once (round 4, one
        # rank over RCCL, pull form): the process group's watchdog thread queried an event "last recorded in a capturing stream" of a
        # collective issued launch by launch after a capture -- not a risk to take on the line a multi-GPU run is measured by.
        pre_steps = min(6, prime - 4)
        pre_sink, runner.comm_log = [], []
        model[0].gather_event_sink = pre_sink
        graph_flag = getattr(runner, "use_step_graph", False)      # (False: the captured-collective probe failed or LSTEP_DIST_GRAPH=0)
        runner.use_step_graph = False
        for i in range(-prime, -prime + pre_steps):
            step(i)
            if i == -prime:        # (the very first training iteration sets communicators and lazy state up: not logged)
                barrier()
                pre_sink.clear()
                runner.comm_log = []
        barrier()
        runner.use_step_graph = graph_flag
        model[0].gather_event_sink = None
        pre_log, runner.comm_log = runner.comm_log, None
    for i in range(-prime + pre_steps, 0):
        step(i)
    for i in range(args.warmup):
        step(i)
    # Host hygiene before the timed region: move everything allocated so far (torch, the workload, the captured graphs: a few million
    # tracked objects) out of the cyclic collector's reach.  A full collection otherwise lands inside the run about once per 60 steps
    # and costs 35-50 ms (measured: tools/variance.py -- one 20-step block at 7-8 ms/step among blocks at 5.2).
    import gc
    gc.collect()
    gc.freeze()
    if os.environ.get("LSTEP_SYNC_DEBUG") == "1":      # diagnostic: print every call that makes the host wait for the GPU inside the timed steps
        torch.cuda.set_sync_debug_mode("warn")
    sink = []
    model[0].gather_event_sink = sink
    steps_issued = args.warmup + args.steps       # (batches consumed so far; the event-timed launch-by-launch iterations below go on from here)
    trace = os.environ.get("LSTEP_BENCH_TRACE") == "1"      # per-step host / GPU times on stderr (does not drain the GPU between steps)
    marks = []
    barrier()
    t0 = time.perf_counter()
    host_s = 0.0
    for i in range(args.steps):
        h0 = time.perf_counter()
        step(args.warmup + i)
        host_s += time.perf_counter() - h0
        if trace:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((time.perf_counter() - t0, ev))
    barrier()
    elapsed = time.perf_counter() - t0
    if trace and rank == 0:
        prev_h, prev_g = 0.0, None
        rows = []
        for h, ev in marks:
            g = marks[0][1].elapsed_time(ev)
            rows.append(f"{(h - prev_h) * 1e3:.2f}/{(g - prev_g) if prev_g is not None else float('nan'):.2f}")
            prev_h, prev_g = h, g
        print("[trace] per step host-enqueue ms / GPU ms since previous step end: " + " ".join(rows), file=sys.stderr)
    timing_note = "HIP events on the launch stream around every gather launch of the timed steps"
    if pre_log is not None:
        runner.comm_log = pre_log          # (the collectives' timed events always come from the pre-roll)
    if not sink and pre_sink:
        sink = pre_sink
        timing_note = (f"the timed steps are graph replays, which cannot carry timed events: HIP events around the gather launch (and around every "
                       f"collective) of {pre_steps - 1} iterations issued launch by launch in the pre-roll, before the step was captured")
    elif not sink:
        # the timed steps were graph replays (engine.GraphedTrainStep), and a graph cannot carry timed events: the gather kernel's launch
        # duration is measured the same way on the next batches of the same stream, issued launch by launch right after the timed region
        eng_g = getattr(runner, "use_step_graph", None)
        runner.use_step_graph = False
        if use_dist:
            runner.comm_log = []        # the same launch-by-launch iterations time every collective between HIP events (the `comm` object)
        extra = max(3, min(10, args.steps))
        for i in range(extra):
            step(args.warmup + args.steps + i)
        steps_issued = args.warmup + args.steps + extra
        barrier()
        runner.use_step_graph = eng_g
        timing_note = (f"the timed steps are graph replays, which cannot carry timed events: HIP events around the gather launch of the {extra} "
                       "iterations issued launch by launch right after the timed region (same stream of batches, same state)")
    model[0].gather_event_sink = None
    # The same launch once more with every other stream IDLE (a device synchronisation right in front of it): inside a step the gather kernel
    # runs beside the window slide (history_advance_oldest, 0.4 GB on the auxiliary stream) and whatever the previous step left in flight;
    # `launch_ms` / `achieved` / `frac` stay the in-step figures, `launch_ms_idle` / `frac_idle` say what the same ISA does alone.
    idle_sink = []
    if world == 1 and not use_dist and args.mode == "train" and sink:
        eng_g = getattr(runner, "use_step_graph", None)
        runner.use_step_graph = False
        model[0].gather_event_sink, model[0].gather_event_idle = idle_sink, True
        for i in range(5):
            if (start + (steps_issued + i + 2) * B * world) <= wl.num_edges:
                step(steps_issued + i)
        barrier()
        model[0].gather_event_sink, model[0].gather_event_idle = None, False
        runner.use_step_graph = eng_g
    comm = None
    if use_dist:
        # per collective: bytes moved through this rank's buffers per call and the event-timed duration (launch-by-launch iterations);
        # host_ms_per_step: time this rank's Python thread spent issuing one TIMED step (a graph replay: input copies + one launch)
        log, runner.comm_log = (runner.comm_log or []), None
        agg = {}
        for name, nbytes, e0, e1 in log:
            a = agg.setdefault(name, [0, 0, 0.0])
            a[0] += 1
            a[1] += nbytes
            a[2] += e0.elapsed_time(e1)
        steps_logged = max(1, len([1 for n_, *_ in log if n_ == "all_reduce parameter gradients"]))
        comm = {"host_ms_per_step": host_s / args.steps * 1e3, "device_driven": bool(getattr(runner, "device_driven", False)),
                "collectives_per_step": {k: {"calls": v[0] / steps_logged, "bytes_per_call": v[1] // max(v[0], 1), "ms_per_call": v[2] / max(v[0], 1)}
                                         for k, v in agg.items()},
                "timing": "HIP events around every collective of the launch-by-launch iterations " + ("of the pre-roll, before the step was captured" if pre_sink else "run after the timed region") + " (rank 0)"}
        comm["timed_steps_are_graph_replays"] = bool(getattr(runner, "use_step_graph", False)) and args.mode == "train"
        comm["captured_collective_probe"] = probe_note
        comm["probe_verdict"] = {"captured": bool(probe_ok), "pull_pattern": bool(pull_ok)}
        comm["update_form"] = getattr(runner, "form", None)
        comm["update_form_chosen_by"] = ("LSTEP_PHASE2=" + os.environ["LSTEP_PHASE2"]) if os.environ.get("LSTEP_PHASE2", "auto") != "auto" else \
            "auto: pull beyond four ranks with a clean pull-pattern probe on every rank, replicate otherwise"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        # HBM traffic per launch of the gather kernel: not measurable from inside the process; taken from the committed PMC
        # passes of this same command (profiles/*pmc_traffic.json, made by tools/pmc_summary.py), default workload only
        gather_kernel = GATHER_KERNEL_LONG_ROWS if (getattr(wl.sampler, "max_degree", 0) > 256 and wl.G > 256) else GATHER_KERNEL
        traffic, traffic_src = None, None
        if args.traffic == "auto" and world == 1 and not use_dist and args.mode == "train" and args.sampler == "recent":
            measured = measure_gather_traffic(args, (gather_kernel,))
            if measured is not None:
                traffic, traffic_src = measured
        if (traffic is None and args.traffic != "off" and args.batch is None and args.time_gap == 2000 and world == 1 and not args.zipf
                and args.mode == "train" and args.sampler == "recent"):
            import glob
            others = [w for w in ("enron", "wikipedia", "reddit", "tiny", "synth-4M-100M") if w != args.workload]
            pat = "*pmc_traffic.json" if args.workload == "synth-1M-20M" else f"*{args.workload}*pmc_traffic.json"
            cands = sorted(c for c in glob.glob(os.path.join(ROOT, "profiles", pat)) if not any(o in os.path.basename(c) for o in others))
            if cands:
                table = json.load(open(cands[-1]))
                rec = None
                for name in (gather_kernel, GATHER_KERNEL, GATHER_KERNEL_LONG_ROWS, "lstep::gather_aggregate_fwd_kernel<true, true, false>",
                             "lstep::gather_aggregate_fwd_kernel<true, true>"):       # (this round's names, then the names of earlier rounds' files)
                    rec = rec or table.get(name)
                if rec:
                    traffic, traffic_src = rec["traffic_bytes"], os.path.relpath(cands[-1], ROOT)
        ms, bytes_per_launch = pair_gather_launches(sink, wl.K, wl.G)
        if args.sampler != "recent":
            # explicit neighbourhoods (sampling with replacement): every row with history fills all K / time_gap slots; the per-row count
            # the kernel reports is K.  Upper bound of the stage's algorithmic bytes: every slot of every row read.
            rows = int(sink[0][2].numel())
            bytes_per_launch = [float(rows * (688 * (2 * wl.K + wl.G + 3) + 24 * wl.K + 8 * wl.G))] * len(ms)
        avg_ms = float(np.mean(ms))
        achieved = float(np.mean(bytes_per_launch)) / (avg_ms * 1e-3) / 1e9
        bound, peak, split = gather_roofline_bound(sink[0][2] if args.sampler == "recent" else torch.full_like(sink[0][2], wl.G), wl.K, wl.G,
                                                   wl.num_nodes, wl.num_edges)
        scale = float(np.mean(bytes_per_launch)) / max(sum(split.values()), 1.0)      # (the split of the first timed launch, scaled to the mean launch)
        split = {lv: b * scale for lv, b in split.items()}
        bound_note, frac_vs_l2 = None, None
        if achieved > peak:
            # more algorithmic bytes per second than the level the table-size model assigns can deliver: rows are re-read from a cache above
            # it (hub rows of a power-law graph, the slot lists of a sampled-with-replacement neighbourhood).  The model's bound, peak and
            # frac (> 1) stay on the line -- they say that the model mis-prices re-read rows --, `frac_vs_l2` prices the launch against L2.
            frac_vs_l2 = achieved / LEVEL_PEAK_GBS["l2"]
            bound_note = (f"the table-size model says {bound} ({peak:.0f} GB/s) but the launch moved {achieved:.0f} GB/s of algorithmic bytes: "
                          f"re-read rows are served from L2 (frac_vs_l2 = {frac_vs_l2:.3f} of {LEVEL_PEAK_GBS['l2']:.0f} GB/s)")
        launch_ms_idle, frac_idle = None, None
        if idle_sink:
            ms_i, bytes_i = pair_gather_launches(idle_sink, wl.K, wl.G)
            launch_ms_idle = float(np.mean(ms_i))
            frac_idle = float(np.mean(bytes_i)) / (launch_ms_idle * 1e-3) / 1e9 / peak
        line = {
            "metric": "processed edges/sec (L-STEP fwd+bwd)" if args.mode == "train" else "processed edges/sec (L-STEP eval iteration, no bwd)",
            "value": B * world * args.steps / elapsed,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "host_ms_per_step": host_s / args.steps * 1e3,      # this rank's Python thread issuing one timed step (graph replay: input copies + one launch)
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl.describe() + (f", zipf {args.zipf} endpoints" if args.zipf else ""), "global_batch": B * world,
                       "per_gpu_batch": B, "workload_name": args.workload,
                       "history": (f"evolved: {prerolled} pre-roll batches through the engine's own eval iteration + {prime} through its training iteration"
                                   if args.history == "evolved" else f"random: T independent snapshots (+ {prime} training iterations of set-up)"),
                       "step_graph": bool(getattr(runner, "use_step_graph", False)),
                       "parallelism": ((f"x{world}: PE history ring and FFT filter sharded by node owner (id % {world}), RCCL all-gather of the "
                                        f"filtered rows / reduce-scatter of their gradient; gather, dense tail and loss on each rank's {B}-edge "
                                        f"slice of the global batch; update_pe " +
                                        {"replicate": "replicated on the global batch (no update collective)",
                                         "allgather": "sharded by owner with RCCL all-gather of every updated PE row into replicated tables",
                                         "pull": "and the PE table sharded by owner: RCCL all-gather of the batch nodes' updated rows, all-to-all pull of the "
                                                 "rows the next gather reads (requested one step ahead)"}[getattr(runner, "form", "replicate")])
                                       if use_dist else "single GPU"),
                       "update_form": getattr(runner, "form", None)},
            "roofline": {"bound": bound, "kernel": ("lstep::gather_aggregate_fwd_kernel<.., true> x 2 (explicit neighbour lists: edge + node channels, PE channel)" if args.sampler != "recent" else gather_kernel) if not use_dist else "lstep::gather_aggregate_fwd_kernel<true, false, false> + <false, true, false> (two launches per step)", "achieved": achieved, "peak": peak,
                         "unit": "GB/s", "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_by_level": split, "bound_note": bound_note, "frac_vs_l2": frac_vs_l2,
                         "launch_ms": avg_ms, "launch_ms_idle": launch_ms_idle, "frac_idle": frac_idle,
                         "launch_timing": timing_note + ("; launch_ms_idle / frac_idle: the same launch of 5 more such iterations with a device "
                                                         "synchronisation right in front of it (no neighbour kernel on any stream)" if idle_sink else ""), "algorithmic_bytes_per_launch": float(np.mean(bytes_per_launch)),
                         "rows_per_launch": int(sink[0][2].numel()) if sink else 0},
        }
        hub = model[0].__dict__.get("_last_hub")
        if hub is not None and args.sampler == "recent":
            # (csrc/hub.hip) rows whose node channel came from prefix differences over the union of their node's windows: the launch then
            # reads far fewer node rows than SURVEY 8(d)'s per-row count prices (`achieved` / `frac` keep that algorithmic count)
            served_rows = int(hub[0].sum().item())
            if served_rows and line["roofline"]["bound_note"]:
                line["roofline"]["bound_note"] += (f"; AND {served_rows} of the launch's rows take their node channel from the hub kernels' pass over the union "
                                                   "of their windows, so most of the priced bytes are not moved at all (hub_rows)")
            line["roofline"]["hub_rows"] = {"rows_served": served_rows, "work_items": int(hub[1].item()),
                                            "rows_per_launch": int(hub[0].numel()),
                                            "note": "node channel of batch rows whose node occurs >= 4 times in the batch: one pass over the union of "
                                                    "their windows (lstep_hub_node_sums) instead of time_gap row reads each; launch_ms covers the "
                                                    "work-list kernels, the gather kernel and the hub kernel"}
        if comm is not None:
            line["comm"] = comm
        if world == 1 and not args.no_cpu_baseline and args.mode == "train" and not args.zipf and args.sampler == "recent":
            line["cpu_baseline"] = cpu_baseline(args.workload, args.time_gap, args.batch)
        print(json.dumps(line))
    if use_dist:
        # Orderly teardown, bounded.  Round 4 left through os._exit(0) because one `pull` rehearsal (world size 1 over RCCL) sat in
        # destroy_process_group until the caller's timeout: the captured step's graph -- which holds kernels of BOTH communicators -- was still
        # alive, and a communicator cannot be destroyed before the graphs that captured its collectives.  Now: drop the captured steps
        # (explicit graph lifetime, model.drain_dead_graphs), drain the GPU, destroy the process group from a helper thread and give it
        # 30 s; only if THAT does not return is the process ended with os._exit, and the line says so on stderr (nothing is hidden: the
        # JSON line is out, every rank has passed the barrier).
        sys.stdout.flush()
        sys.stderr.flush()
        barrier()
        try:
            runner.close()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: rank {rank}: closing the captured steps failed: {type(e).__name__}: {e}", file=sys.stderr)
        import threading
        done = threading.Event()

        def teardown():
            try:
                dist.destroy_process_group()
            finally:
                done.set()
        th = threading.Thread(target=teardown, name="lstep-bench-teardown", daemon=True)      # (no GPU launches: communicator teardown only)
        th.start()
        if not done.wait(timeout=float(os.environ.get("LSTEP_BENCH_TEARDOWN_TIMEOUT", "30"))):
            print(f"bench.py: rank {rank}: destroy_process_group did not return within its timeout; leaving through os._exit(0)", file=sys.stderr)
            sys.stderr.flush()
            os._exit(0)

if __name__ == "__main__":
    main()

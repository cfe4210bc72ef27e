"""Multi-process tests of lstep_amd.parallel.

* CPU (gloo, world_size 2): the collective helpers the distributed engine is built from.
* GPU (gloo over CUDA tensors staged through the host, 2 ranks on the one GPU of the test box): the owner-sharded
  engine must reproduce the SAME golden training / evaluation trace as the reference (tests/golden/traces.npz).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import TRACE_B, TRACE_BATCHES, TRACE_G, TRACE_K, TRACE_START, TRACE_T, eval_batches, trace_batches, trace_inputs
from lstep_amd import synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if backend == "nccl":       # = RCCL on ROCm
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)


def _cpu_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        from lstep_amd.parallel import PendingGather, all_gather_var, all_reduce_gradients, all_reduce_sum, owned_rows, reduce_scatter_var
        # uneven row blocks, incl. an empty one
        t = torch.arange(rank * 3 * 4, dtype=torch.float32).reshape(rank * 3, 4) + 100 * rank
        cat, counts = all_gather_var(t)
        assert counts == [0, 3] and cat.shape == (3, 4) and torch.equal(cat, torch.arange(12, dtype=torch.float32).reshape(3, 4) + 100)
        ids = torch.tensor([5, 7, 9][: rank + 1], dtype=torch.int64)
        cat, counts = all_gather_var(ids)
        assert cat.tolist() == [5, 5, 7] and counts == [1, 2]
        blk = reduce_scatter_var(torch.arange(10, dtype=torch.float32).reshape(5, 2) * (rank + 1), [2, 3])
        assert torch.equal(blk, (torch.arange(10, dtype=torch.float32).reshape(5, 2) * 3)[:2] if rank == 0 else (torch.arange(10, dtype=torch.float32).reshape(5, 2) * 3)[2:])
        # equal counts (no un-padding), rows with trailing dimensions, counts known locally (no size exchange), and the asynchronous form
        t3 = torch.arange(2 * 3 * 2, dtype=torch.float32).reshape(2, 3, 2) + 1000 * rank
        cat, counts = all_gather_var(t3, counts=[2, 2])
        assert counts == [2, 2] and cat.shape == (4, 3, 2) and torch.equal(cat[2:], torch.arange(12, dtype=torch.float32).reshape(2, 3, 2) + 1000)
        pend = PendingGather(torch.full((rank + 2, 5), float(rank)))
        got = pend.wait()
        assert pend.counts == [2, 3] and got.shape == (5, 5) and got[:2].eq(0).all() and got[2:].eq(1).all() and pend.wait() is got
        # padded reduce-scatter against all_reduce + slice, uneven incl. an empty block, with trailing dimensions
        for cnt in ([2, 3], [5, 0], [0, 5], [4, 4]):
            full = torch.randn(sum(cnt), 3, 2, generator=torch.Generator().manual_seed(7 + rank))
            want = all_reduce_sum(full.clone())
            off = [0, cnt[0], sum(cnt)]
            blk = reduce_scatter_var(full, cnt)
            assert blk.shape[0] == cnt[rank] and torch.allclose(blk, want[off[rank]:off[rank + 1]], atol=1e-6)
        v = torch.full((3,), float(rank + 1))
        assert all_reduce_sum(v).tolist() == [3.0, 3.0, 3.0]
        lin = torch.nn.Linear(3, 2)
        cplx = torch.nn.Parameter(torch.zeros(2, 2, dtype=torch.complex64))
        lin.weight.grad = torch.full_like(lin.weight, rank + 1.0)      # bias.grad stays None -> treated as zero
        cplx.grad = torch.full((2, 2), complex(rank + 1.0, -rank), dtype=torch.complex64)
        all_reduce_gradients(list(lin.parameters()) + [cplx])
        assert torch.all(lin.weight.grad == 3.0) and torch.all(lin.bias.grad == 0.0) and torch.all(cplx.grad == complex(3.0, -1.0))
        assert [owned_rows(65, 2, r) for r in (0, 1)] == [33, 32] and sum(owned_rows(1000001, 8, r) for r in range(8)) == 1000001
        assert owned_rows(1, 4, 3) == 0
        # all-to-all-v of row blocks (the pull of the owner-sharded PE table): uneven blocks incl. empty ones, rows with a trailing dimension
        from lstep_amd.parallel import TensorsKey, exchange_rows
        send_counts = [[1, 2], [3, 0]][rank]                       # rank 0 keeps 1 row, sends 2; rank 1 sends 3, keeps none
        recv_counts = [[1, 3], [2, 0]][rank]
        send = (torch.arange(sum(send_counts) * 2, dtype=torch.float32).reshape(-1, 2) + 100 * rank)
        got = exchange_rows(send, send_counts, recv_counts)
        if rank == 0:
            assert torch.equal(got, torch.cat([send[:1], torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100]))
        else:
            assert torch.equal(got, torch.arange(6, dtype=torch.float32).reshape(3, 2)[1:])
        pend = exchange_rows(torch.full((4,), rank, dtype=torch.int32), [2, 2], [2, 2], async_op=True)
        assert pend.wait().tolist() == [0, 0, 1, 1]
        a, b = torch.zeros(3), torch.ones(3)
        key = TensorsKey(a, b)
        assert key.matches(a, b) and not key.matches(b, a)
        a.add_(1)
        assert not key.matches(a, b)
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def _run(worker, world, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    res, deadline = [], time.time() + 600
    while len(res) < len(procs):
        try:
            res.append(q.get(timeout=5))
        except queue.Empty:
            # a rank that died without reporting (a fault inside a native call) must fail the test at once, not after the queue's timeout
            dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
            if dead or time.time() > deadline:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise AssertionError(f"ranks died without a report (rank, exit code): {dead}" if dead else "ranks timed out")
    for p in procs:
        p.join(timeout=60)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_collective_helpers_gloo_world2():
    _run(_cpu_worker, 2)


def _probe_and_form(world):
    """(probe handed to DistributedLstep, the form LSTEP_PHASE2 must resolve to).  LSTEP_TEST_PROBE=clean stands for bench.py's cross-rank
    rehearsal of captured collectives and of the pull exchange pattern having come back clean on every rank."""
    probe = {"captured": True, "pull": True} if os.environ.get("LSTEP_TEST_PROBE") == "clean" else None
    policy = os.environ.get("LSTEP_PHASE2", "auto")
    if policy == "auto":
        policy = "pull" if (probe is not None and world >= int(os.environ.get("LSTEP_PULL_MIN_WORLD", "5"))) else "replicate"
    return probe, policy


def _gpu_worker(rank, world, port, q, backend="gloo", ahead=False):
    try:
        torch.cuda.set_device(0)
        _init(rank, world, port, backend)
        dev = "cuda:0"
        from lstep_amd.engine import EdgeStream, LstepEngine
        from lstep_amd.parallel import DistributedLstep, all_gather_var
        from lstep_amd.sampler import NeighborSampler
        from lstep_amd.workload import build_hip_model
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traces.npz"))
        g, node_raw, edge_raw, pe0 = trace_inputs()
        sampler = NeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=g["num_nodes"], device=dev)
        model = build_hip_model(node_raw, edge_raw, sampler, TRACE_K, TRACE_T, synth.make_state_dict(TRACE_K, TRACE_T), dev)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        probe, want_form = _probe_and_form(world)
        dl = DistributedLstep(LstepEngine(model[0], model[1], TRACE_K, TRACE_G, make_ring=False), opt, probe=probe)
        assert dl.ring.rows == (65 - rank + world - 1) // world
        stream = EdgeStream.from_numpy(g["src"], g["dst"], g["ts"], g["eid"], dev)
        init = torch.from_numpy(pe0.copy()).to(dev)
        tol = dict(rtol=0, atol=5e-5)
        half = TRACE_B // world

        def global_predicts(local):  # rank-local [pos(b), neg(b)] -> global [pos(B), neg(B)]
            cat, _ = all_gather_var(local.reshape(2, half))   # rows: r0 pos, r0 neg, r1 pos, r1 neg
            cat = cat.reshape(world, 2, half)
            return torch.cat([cat[:, 0, :].reshape(-1), cat[:, 1, :].reshape(-1)]).cpu().numpy()

        pull = dl.form == "pull"
        assert dl.form == want_form, (dl.form, want_form)
        assert dl.device_driven == (dl.form != "allgather" and os.environ.get("LSTEP_DIST_HOST_COUNTS") != "1")

        poisoned = [0]

        def check_table(want):
            """Owned rows always; every row of a replicated table; an owner-sharded table as assembled from its owners."""
            poisoned[0] += int(torch.isnan(dl.table).any(dim=1).sum())
            np.testing.assert_allclose(dl.table[rank::world].cpu().numpy(), want[rank::world], **tol)
            np.testing.assert_allclose((dl.full_table() if pull else dl.table).cpu().numpy(), want, **tol)

        batches = trace_batches(g)
        negs = [torch.from_numpy(b_[4]).to(dev) for b_ in batches]
        for b, (src, dst, t, eid, neg) in enumerate(batches):
            lo = TRACE_START + b * TRACE_B
            nxt = None
            if ahead and b + 1 < len(batches) and b % 3 != 2:        # look-ahead on two batches of three: requested-ahead and on-the-spot pulls
                s2, d2, t2, _ = stream.batch(lo + TRACE_B, lo + 2 * TRACE_B)
                nxt = (s2, d2, t2, negs[b + 1])
            res = dl.train_iteration(opt, b, *stream.batch(lo, lo + TRACE_B), negs[b], initial_pe=init, lookahead=nxt)
            if pull and ahead and b > 0:
                assert (dl._pending_pull is not None) == (nxt is not None)
            check_table(z[f"train/b{b}/snapshot"])
            np.testing.assert_allclose(dl.ring.last().cpu().numpy(), z[f"train/b{b}/snapshot"][rank::world], **tol)
            if res is not None:
                got = [float(res["lp_loss"]), float(res["pe_loss"]), float(res["loss"])]
                np.testing.assert_allclose(got, z[f"train/b{b}/losses"], rtol=0, atol=2e-5)
                np.testing.assert_allclose(global_predicts(res["predicts"]), z[f"train/b{b}/predicts"], **tol)
        np.testing.assert_allclose(dl.ring.as_reference_tensor().cpu().numpy(), z["train/final_history"][rank::world, -TRACE_T:, :], **tol)
        model.eval()
        with torch.no_grad():
            ebatches = eval_batches(g)
            eneg = [(torch.from_numpy(b_[4]).to(dev), torch.from_numpy(b_[5]).to(dev)) for b_ in ebatches]
            for b, (src, dst, t, eid, neg_src, neg_dst) in enumerate(ebatches):
                lo = TRACE_START + (TRACE_BATCHES + b) * TRACE_B
                nxt = None
                if ahead and b + 1 < len(ebatches):            # the evaluation loop's look-ahead names both negative sets
                    s2, d2, t2, _ = stream.batch(lo + TRACE_B, lo + 2 * TRACE_B)
                    nxt = (s2, d2, t2, eneg[b + 1][0], eneg[b + 1][1])
                res = dl.eval_iteration(b, *stream.batch(lo, lo + TRACE_B), eneg[b][0], eneg[b][1], lookahead=nxt)
                if pull and ahead:
                    assert (dl._pending_pull is not None) == (nxt is not None)
                np.testing.assert_allclose(global_predicts(res["predicts"]), z[f"eval/b{b}/predicts"], **tol)
                check_table(z[f"eval/b{b}/snapshot"])
        # LSTEP_PULL_POISON=1 did poison: rows nobody delivered were NaN in the cache (and no result above ever saw one)
        assert (poisoned[0] > 0) == (pull and world > 1), poisoned
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.gpu
@pytest.mark.parametrize("world,phase2,ahead,host_counts", [(2, "allgather", False, False), (2, "replicate", False, False), (4, "auto", False, False),
                                                            (2, "pull", False, False), (2, "pull", True, False), (4, "pull", True, False),
                                                            (2, "replicate", False, True), (2, "pull", True, True),
                                                            (2, "auto+clean-probe", True, False), (4, "auto+clean-probe", True, False)])
def test_distributed_engine_reproduces_golden_trace_ranks_on_one_gpu(monkeypatch, world, phase2, ahead, host_counts):
    """W ranks share the one GPU of the test box (gloo staging the device tensors through the host) and must reproduce the REFERENCE's
    golden training + evaluation trace: owner-sharded history ring and FFT filter, batch slices through the gather stage, and the three
    forms of update_pe (LSTEP_PHASE2) -- incl. the owner-sharded PE table ("pull": owner-computes update, all-gather of the batch nodes'
    rows, all-to-all pull of the rows the next gather reads, requested on the spot and one step ahead), two ranks and four (17 / 16 /
    16 / 16 owned rows, 4 edges of the batch per rank)."""
    assert torch.cuda.is_available()
    if phase2 == "auto+clean-probe":
        # the default (LSTEP_PHASE2 unset = auto) with a clean cross-rank probe: the owner-sharded form beyond LSTEP_PULL_MIN_WORLD - 1 ranks
        # (five in production: what `python bench.py --gpus 8` runs; lowered here to the ranks one test box can hold)
        monkeypatch.delenv("LSTEP_PHASE2", raising=False)
        monkeypatch.setenv("LSTEP_TEST_PROBE", "clean")
        monkeypatch.setenv("LSTEP_PULL_MIN_WORLD", "2")
    else:
        monkeypatch.setenv("LSTEP_PHASE2", phase2)      # (inherited by the spawned ranks)
    monkeypatch.setenv("LSTEP_PULL_POISON", "1")    # "pull": rows a rank does not own are NaN unless a collective delivered them
    if host_counts:                                 # the host-sized iterations of rounds 2-3 (A/B switch; what "allgather" always takes)
        monkeypatch.setenv("LSTEP_DIST_HOST_COUNTS", "1")
    _run(_gpu_worker, world, "gloo", ahead)


@pytest.mark.gpu
@pytest.mark.parametrize("phase2", ["auto", "pull"])
def test_distributed_engine_on_rccl_world_size_1_reproduces_golden_trace(monkeypatch, phase2):
    """The RCCL calls themselves (``backend="nccl"``: all_gather_into_tensor of padded blocks, the asynchronous gather left in flight on
    RCCL's stream, reduce_scatter_tensor, the flat gradient all-reduce; "pull": all_to_all_single with split sizes on a second
    communicator) on the one GPU of the test box: world size 1 with LSTEP_FORCE_COLLECTIVES=1, so no collective is short-circuited.
    Same golden trace as the gloo runs above."""
    assert torch.cuda.is_available()
    monkeypatch.setenv("LSTEP_FORCE_COLLECTIVES", "1")      # (inherited by the spawned rank)
    monkeypatch.setenv("LSTEP_PHASE2", phase2)
    monkeypatch.setenv("LSTEP_PULL_POISON", "1")
    _run(_gpu_worker, 1, "nccl", phase2 == "pull")


def _long_worker(rank, world, port, q, backend, expect_replays):
    """tests/golden/traces_long.npz (16 consecutive training batches the REFERENCE ran) through ``DistributedLstep`` with the one-launch
    Adam: the device-driven iteration (fixed-capacity collectives, counts on the device) and -- over RCCL -- its whole-step HIP graph
    with the collectives captured inside."""
    try:
        torch.cuda.set_device(0)
        _init(rank, world, port, backend)
        dev = "cuda:0"
        from helpers import LONG_BATCHES, check_long_trace_gradients, long_trace_batches
        from lstep_amd.engine import EdgeStream, LstepEngine
        from lstep_amd.optim import FusedAdam
        from lstep_amd.parallel import DistributedLstep, all_gather_var
        from lstep_amd.sampler import NeighborSampler
        from lstep_amd.workload import build_hip_model
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "traces_long.npz"))
        g, node_raw, edge_raw, pe0 = trace_inputs()
        sampler = NeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=g["num_nodes"], device=dev)
        model = build_hip_model(node_raw, edge_raw, sampler, TRACE_K, TRACE_T, synth.make_state_dict(TRACE_K, TRACE_T), dev)
        model.train()
        opt = FusedAdam(model.parameters(), lr=1e-4)
        probe, want_form = _probe_and_form(world)
        dl = DistributedLstep(LstepEngine(model[0], model[1], TRACE_K, TRACE_G, make_ring=False), opt, probe=probe)
        assert dl.form == want_form, (dl.form, want_form)
        assert dl.device_driven and dl.use_step_graph == (backend == "nccl")
        stream = EdgeStream.from_numpy(g["src"], g["dst"], g["ts"], g["eid"], dev)
        init = torch.from_numpy(pe0.copy()).to(dev)
        tol = dict(rtol=0, atol=5e-5)
        half = TRACE_B // world
        pull = dl.form == "pull"

        def global_predicts(local):
            cat, _ = all_gather_var(local.reshape(2, half))
            cat = cat.reshape(world, 2, half)
            return torch.cat([cat[:, 0, :].reshape(-1), cat[:, 1, :].reshape(-1)]).cpu().numpy()

        batches = long_trace_batches(g)
        negs = [torch.from_numpy(b_[4]).to(dev) for b_ in batches]
        worst = {"snapshot": 0.0, "loss": 0.0, "predicts": 0.0, "gradient": 0.0}
        for b in range(LONG_BATCHES):
            lo = TRACE_START + b * TRACE_B
            nxt = None
            if b + 1 < LONG_BATCHES and b != 9:           # (one step without a look-ahead: the next replay must fetch its rows itself)
                s2, d2, t2, _ = stream.batch(lo + TRACE_B, lo + 2 * TRACE_B)
                nxt = (s2, d2, t2, negs[b + 1])
            res = dl.train_iteration(opt, b, *stream.batch(lo, lo + TRACE_B), negs[b], initial_pe=init, lookahead=nxt)
            want = z[f"b{b}/snapshot"]
            table = (dl.full_table() if pull else dl.table).cpu().numpy()
            worst["snapshot"] = max(worst["snapshot"], float(np.abs(table - want).max()))
            np.testing.assert_allclose(table, want, err_msg=f"b{b} table", **tol)
            np.testing.assert_allclose(dl.ring.last().cpu().numpy(), want[rank::world], **tol)
            if res is None:
                continue
            got = [float(res["lp_loss"]), float(res["pe_loss"]), float(res["loss"])]
            np.testing.assert_allclose(got, z[f"b{b}/losses"], rtol=0, atol=2e-5, err_msg=f"b{b} losses")
            pr = global_predicts(res["predicts"])
            worst["loss"] = max(worst["loss"], float(np.abs(np.asarray(got) - z[f"b{b}/losses"]).max()))
            worst["predicts"] = max(worst["predicts"], float(np.abs(pr - z[f"b{b}/predicts"]).max()))
            np.testing.assert_allclose(pr, z[f"b{b}/predicts"], err_msg=f"b{b} predicts", **tol)
            d, _ = check_long_trace_gradients(model, z, b, atol=2e-6, digest_atol=2e-4)
            worst["gradient"] = max(worst["gradient"], d)
        np.testing.assert_allclose(dl.ring.as_reference_tensor().cpu().numpy(), z["final_history"][rank::world], **tol)
        dl.check_capacity(wait=True)
        gs = dl._graphed.get(TRACE_B)
        if expect_replays:
            assert gs is not None and gs.graph is not None and gs.replays == LONG_BATCHES - 7, (gs and gs.replays)
            assert int(dl.ring.dev_start.item()) == dl.ring.start
            # the step behind the one without a look-ahead fetched its rows on the spot -- through the small captured graph of its own,
            # not through collectives issued launch by launch between two replays
            assert gs.pull_now_replays == (1 if pull else 0), gs.pull_now_replays
        else:
            assert gs is None
        if rank == 0:
            print(f"[distributed long trace, {backend} x{world}, {dl.form}, {'graph replay' if expect_replays else 'launch by launch'}] worst: "
                  + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
        dl.close()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.gpu
def test_capture_right_behind_launch_by_launch_collectives_is_quiesced():
    """The sporadic "watchdog thread terminated ... event last recorded in a capturing stream" abort (round 4 in bench.py, round 5 in the RCCL
    golden-trace test): torch's NCCL watchdog polls the end event of a launch-by-launch collective until its next sweep (~100 ms); if the
    stream the collective ran on starts or joins a capture before that, hipEventQuery fails and the process aborts.
    tools/nccl_capture_after_eager_probe.py re-enacts it (scenario `same`); with ``parallel.quiesce_collectives`` in front of the capture
    (scenario `lib`: what ``GraphedDistStep`` does before both of its captures) the same sequence must run through."""
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = dict(os.environ, PROBE_REPS="5", PROBE_PORT=str(_free_port()), PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_capture_after_eager_probe.py"), "--child", "lib"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("phase2", ["replicate", "pull", "auto+clean-probe"])
def test_distributed_graph_replay_reproduces_long_golden_trace_rccl(monkeypatch, phase2):
    """the multi-GPU iteration is device-driven and replayed as ONE HIP graph with its RCCL collectives captured inside
    (world size 1 with LSTEP_FORCE_COLLECTIVES=1, the only RCCL set-up a one-GPU box allows): batches 0-3 fill the T = 4 window, 4-5
    prime, 6 is captured, 7-15 are replays -- every step's table, losses, probabilities and parameter gradients against what the
    REFERENCE produced (tests/golden/traces_long.npz)."""
    assert torch.cuda.is_available()
    monkeypatch.setenv("LSTEP_FORCE_COLLECTIVES", "1")
    if phase2 == "auto+clean-probe":       # the auto-selected form of a job whose probe is clean ("pull"), through the captured step
        monkeypatch.delenv("LSTEP_PHASE2", raising=False)
        monkeypatch.setenv("LSTEP_TEST_PROBE", "clean")
        monkeypatch.setenv("LSTEP_PULL_MIN_WORLD", "1")
    else:
        monkeypatch.setenv("LSTEP_PHASE2", phase2)
    monkeypatch.setenv("LSTEP_PULL_POISON", "1")
    _run(_long_worker, 1, "nccl", True)


@pytest.mark.gpu
@pytest.mark.parametrize("world,phase2", [(2, "replicate"), (2, "pull"), (4, "pull")])
def test_distributed_device_driven_long_golden_trace_ranks_on_one_gpu(monkeypatch, world, phase2):
    """The same 16-batch reference trace on 2 / 4 ranks sharing the test box's GPU (gloo): fixed-capacity blocks with per-owner counts on the
    device, the gradient buffer laid out by (owner, slot), poisoned caches in the owner-sharded form."""
    assert torch.cuda.is_available()
    monkeypatch.setenv("LSTEP_PHASE2", phase2)
    monkeypatch.setenv("LSTEP_PULL_POISON", "1")
    _run(_long_worker, world, "gloo", False)


def _c4_worker(rank, world, port, q, phase2):
    """The multi-GPU iteration AT THE BENCH SHAPE (1 M nodes / 20 M edges, B = 16384, K = 20, time_gap 2000, T = 100) on one rank over RCCL with
    every collective forced, against the single-GPU engine run from the SAME state on the same batches: two models that share the feature
    tables and the CSR, the engine's evolved history copied into the rank's ring shard; five consecutive training iterations (launch by
    launch, captured with the collectives inside, replayed) -- per step the link probabilities, the three losses, every parameter gradient
    and the whole 1 000 001-row table each side leaves behind."""
    try:
        torch.cuda.set_device(0)
        _init(rank, world, port, "nccl")
        dev = torch.device("cuda", 0)
        from lstep_amd.engine import LstepEngine
        from lstep_amd.model import LSTEP, MergeLayer
        from lstep_amd.optim import FusedAdam
        from lstep_amd.parallel import DistributedLstep
        from lstep_amd.workload import build_workload, evolve_history
        wl = build_workload("synth-1M-20M", dev, seed=0)
        N, E, B, K, G, T = wl.num_nodes, wl.num_edges, wl.batch, wl.K, wl.G, wl.T
        eng, hm = wl.engine, wl.model
        eng.use_step_graph = True
        start = E // 2
        hm.eval()
        with torch.no_grad():
            assert evolve_history(eng, wl.stream, start, B, N) == T
        hm.train()
        # the second model: same tables (no copy: they are device-resident float32 already), same sampler, same weights
        bb2 = LSTEP(hm[0].node_raw_features, hm[0].edge_raw_features, wl.sampler, wl.sampler, pe_dim=synth.PE_DIM, num_neighbors=K,
                    time_feat_dim=synth.TIME_DIM, num_fft_batches=T, device=dev)
        assert bb2.edge_raw_features.data_ptr() == hm[0].edge_raw_features.data_ptr()
        pred2 = MergeLayer(synth.FEAT_DIM, synth.FEAT_DIM, synth.FEAT_DIM, 1).to(dev)
        m2 = torch.nn.Sequential(bb2, pred2)
        m2.load_state_dict(hm.state_dict())
        m2.train()
        opt1, opt2 = FusedAdam(hm.parameters(), lr=1e-4), FusedAdam(m2.parameters(), lr=1e-4)
        dl = DistributedLstep(LstepEngine(bb2, pred2, K, G, make_ring=False), opt2)
        assert dl.form == phase2 and dl.device_driven and dl.use_step_graph
        r1, r2 = eng.ring, dl._ring
        assert (r1.rows, r1.S) == (r2.rows, r2.S)
        r2.start, r2.len = r1.start, r1.len
        for name in ("buf", "mask", "oldest"):
            getattr(r2, name).copy_(getattr(r1, name))
        dl.table.copy_(r1.table)
        r2.generation += 1
        r2.begin_slot()
        torch.cuda.synchronize()
        gen = torch.Generator(device=dev).manual_seed(77)
        worst = {"predicts": 0.0, "loss": 0.0, "grad": 0.0, "table": 0.0}
        steps = 5
        negs = [torch.randint(1, N + 1, (B,), generator=gen, device=dev) for _ in range(steps + 1)]
        for i in range(steps):
            lo = start + i * B
            src, dst, ts, eid = wl.stream.batch(lo, lo + B)
            s2, d2, t2, _ = wl.stream.batch(lo + B, lo + 2 * B)
            nxt = (s2, d2, t2, negs[i + 1])
            o1 = eng.train_iteration(opt1, 1000 + i, src, dst, ts, eid, negs[i], lookahead=nxt)
            o2 = dl.train_iteration(opt2, 1000 + i, src, dst, ts, eid, negs[i], lookahead=nxt)
            d = float((o1["predicts"] - o2["predicts"]).abs().max())
            worst["predicts"] = max(worst["predicts"], d)
            assert d <= 5e-5, f"step {i}: probabilities differ by {d:.3e}"
            for k in ("lp_loss", "pe_loss", "loss"):
                d = abs(float(o1[k]) - float(o2[k]))
                worst["loss"] = max(worst["loss"], d)
                assert d <= 2e-5, f"step {i} {k}: {d:.3e}"
            for (k, pa), (_, pb) in zip(hm.named_parameters(), m2.named_parameters()):
                if pa.grad is None:
                    assert pb.grad is None or float(pb.grad.abs().max()) == 0.0, k
                    continue
                d = float((torch.view_as_real(pa.grad - pb.grad) if pa.grad.is_complex() else (pa.grad - pb.grad)).abs().max())
                worst["grad"] = max(worst["grad"], d)
                assert d <= 5e-6, f"step {i} d({k}): {d:.3e}"
            d = float((r1.table - dl.table).abs().max())
            worst["table"] = max(worst["table"], d)
            assert d <= 5e-5, f"step {i}: tables differ by {d:.3e}"
            # both sides go on from IDENTICAL state (Adam turns rounding-level gradient differences into +-lr weight steps)
            with torch.no_grad():
                for pb, pa in zip(m2.parameters(), hm.parameters()):
                    pb.copy_(pa)
            opt2.load_state_dict(opt1.state_dict())
            torch.cuda.synchronize()
        gs = dl._graphed.get(B)
        assert gs is not None and gs.replays >= 2, (gs and gs.replays)
        dl.check_capacity(wait=True)
        print(f"[c4 on one rank over RCCL, {phase2}, vs the single-GPU engine, {steps} steps] worst: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
        dl.close()
        eng.close()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.gpu
@pytest.mark.parametrize("phase2", ["replicate", "pull"])
def test_c4_distributed_iteration_on_rccl_matches_the_single_gpu_engine(monkeypatch, phase2):
    """VERDICT r3 "what's weak" 1 (iii): ``DistributedLstep.train_iteration`` itself at the bench shape -- not a stubbed share."""
    assert torch.cuda.is_available()
    import gc
    gc.collect()
    torch.cuda.empty_cache()          # (earlier tests of this process may have left ~100 GB in the caching allocator: the rank is another process)
    import time
    for _ in range(30):               # (the rank process of the test before this one may still be giving its memory back)
        free, _ = torch.cuda.mem_get_info()
        if free >= 200 * 2 ** 30:
            break
        time.sleep(1.0)
    if free < 200 * 2 ** 30:
        pytest.skip(f"needs ~170 GB of HBM (two 70 GB history rings + the 13.8 GB edge table); {free / 2 ** 30:.0f} GB are free")
    monkeypatch.setenv("LSTEP_FORCE_COLLECTIVES", "1")
    monkeypatch.setenv("LSTEP_PHASE2", phase2)
    _run(_c4_worker, 1, phase2)


def _hub_worker(rank, world, port, q):
    """Power-law graph whose hub rows collect hundreds of update_pe messages per global batch (segments of many 64-entry chunks): the
    single-GPU engine, the replicated form and the owner-sharded form on the same global batches."""
    try:
        torch.cuda.set_device(0)
        _init(rank, world, port, "gloo")
        dev = "cuda:0"
        from lstep_amd.engine import EdgeStream, LstepEngine
        from lstep_amd.optim import FusedAdam
        from lstep_amd.parallel import DistributedLstep, all_gather_var
        from lstep_amd.sampler import NeighborSampler
        from lstep_amd.workload import build_hip_model
        N, E, K, T, B, G, steps, first = 300, 40000, 16, 5, 256, 2000, 9, 30000
        g = synth.make_temporal_graph(num_nodes=N, num_edges=E, seed=131, zipf=1.3)
        node_raw, edge_raw = synth.make_features(N, E, seed=132)
        sd = synth.make_state_dict(K, T, seed=134)
        stream = EdgeStream.from_numpy(g["src"], g["dst"], g["ts"], g["eid"], dev)
        # the premise: some row receives far more than two chunks of messages in one batch (phase 1: its incident batch edges)
        busiest = max(int(np.bincount(np.concatenate([g["src"][first + b * B:first + (b + 1) * B], g["dst"][first + b * B:first + (b + 1) * B]])).max())
                      for b in range(steps))
        assert busiest > 128, busiest        # more than two 64-entry chunks: three or more partial sums per row
        results = {}
        for form in ("single", "replicate", "pull"):
            os.environ["LSTEP_PHASE2"] = form if form != "single" else "auto"
            sampler = NeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=N, device=dev)
            model = build_hip_model(node_raw, edge_raw, sampler, K, T, sd, dev)
            model.train()
            opt = FusedAdam(model.parameters(), lr=1e-4)
            init = torch.from_numpy(synth.make_initial_pe(N, seed=133)).to(dev)
            if form == "single":
                run = LstepEngine(model[0], model[1], K, G)
            else:
                run = DistributedLstep(LstepEngine(model[0], model[1], K, G, make_ring=False), opt)
                assert run.form == form
            tables, losses = [], []
            negs = [torch.from_numpy(synth.make_negatives(N, B, seed=140 + b)).to(dev) for b in range(steps)]
            for b in range(steps):
                lo = first + b * B
                nxt = None
                if b + 1 < steps:
                    s2, d2, t2, _ = stream.batch(lo + B, lo + 2 * B)
                    nxt = (s2, d2, t2, negs[b + 1])
                out = run.train_iteration(opt, b, *stream.batch(lo, lo + B), negs[b], initial_pe=init, lookahead=nxt)
                if form == "single":
                    tables.append(run.ring.last().clone())
                elif form == "replicate":
                    tables.append(run.table.clone())
                    # replicas: every rank ran the whole update on its own copy -- they must be BIT-identical (deterministic segment sums)
                    cat, _ = all_gather_var(run.table.reshape(1, -1))
                    assert torch.equal(cat[0], cat[-1]), f"replicas differ after batch {b}: {float((cat[0] - cat[-1]).abs().max()):.3e}"
                else:
                    tables.append(run.full_table())
                    assert torch.equal(run.table[rank::world], tables[-1][rank::world])
                if out is not None:
                    losses.append(float(out["loss"]))
            results[form] = (torch.stack(tables), np.asarray(losses))
        (t1, l1), (t2, l2), (t3, l3) = results["single"], results["replicate"], results["pull"]
        for name, (tt, ll) in (("replicate", (t2, l2)), ("pull", (t3, l3))):
            d = float((tt - t1).abs().max())
            assert d <= 5e-5, f"{name} vs single GPU: tables differ by {d:.3e}"
            np.testing.assert_allclose(ll, l1, rtol=0, atol=2e-5, err_msg=name)
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.gpu
def test_hub_graph_two_ranks_replicas_and_shards_agree():
    """Zipf-1.3 graph, global batch 256 over two ranks: hub rows receive > 128 messages per batch, i.e. segment sums of three or more
    chunk partials -- the case in which float atomics made the replicas of the "replicate" form drift apart by ulps per step.
    Nine training iterations each: the two replicas stay BIT-identical; the owner-sharded table ("pull", rows requested one step ahead),
    assembled from its owners, and the replicated one both match the single-GPU engine on the same global batches (tables 5e-5,
    losses 2e-5)."""
    assert torch.cuda.is_available()
    _run(_hub_worker, 2)


@pytest.mark.gpu
def test_bench_multi_rank_plumbing_gloo_two_ranks_one_gpu():
    """`bench.py --gpus 2` under torch.distributed.run (the driver's launch line), rehearsed on one GPU with gloo: the JSON
    line must come out with n_gpus = 2 and a global batch of 2 x B."""
    import json
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = dict(os.environ, LSTEP_SINGLE_DEVICE="1", LSTEP_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "tiny", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 512 and line["value"] > 0
    assert line["roofline"]["rows_per_launch"] == 3 * 256


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no WORLD_SIZE in the environment): bench.py must start the two ranks itself, before touching the
    GPU, and rank 0's JSON line must come out of the parent's stdout.  Rehearsed on one GPU with gloo."""
    import json
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LSTEP_SINGLE_DEVICE="1", LSTEP_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "tiny",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 512 and line["config"]["per_gpu_batch"] == 256 and line["value"] > 0


@pytest.mark.gpu
def test_bench_single_gpu_line_keeps_the_contract():
    """`python bench.py` on one GPU (tiny workload, the graphed step like the default run): ONE JSON line with the driver's keys, BASELINE.json's
    metric, the roofline object of the gather kernel and the workload named in `config` -- and the timed steps must really have been graph replays."""
    import json
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LSTEP_FORCE_DIST")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--workload", "tiny", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):      # (cpu_baseline is left out by --no-cpu-baseline: it takes 10-30 s of host time)
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["warmup"] == 2 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["unit"] == "edges/s" and line["dtype"] == "f32" and line["data"] == "synthetic" and line["scaling"] == "weak"
    assert abs(line["value"] - line["config"]["global_batch"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert line["config"]["workload_name"] == "tiny" and "model" not in line["config"] and line["config"]["step_graph"] is True
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launch_ms", "algorithmic_bytes_per_launch"):
        assert key in roof, key
    # (the 2000-node "tiny" tables are cache-resident: the bound is the cache level its rows come from, bench.gather_roofline_bound)
    assert roof["bound"] in ("l2", "infinity_cache") and roof["unit"] == "GB/s" and 8000.0 <= roof["peak"] <= 34500.0 and roof["launch_ms"] > 0
    assert abs(sum(roof["bytes_by_level"].values()) - roof["algorithmic_bytes_per_launch"]) <= 1e-6 * roof["algorithmic_bytes_per_launch"]
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["launch_ms"] * 1e-3) / 1e9) <= 1e-6 * roof["achieved"]


def test_bench_default_workload_by_gpu_count():
    import importlib.util
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    spec = importlib.util.spec_from_file_location("lstep_bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert [mod.default_workload(n) for n in (1, 2, 4, 8)] == ["synth-1M-20M"] * 3 + ["synth-4M-100M"]
    # the roofline bound follows the level the gathered tables live in: HBM (8 TB/s) at the 1 M-node shape, the caches at the reference's own
    count = torch.tensor([0, 3, 20, 40, 2500], dtype=torch.int32)
    bound, peak, split = mod.gather_roofline_bound(count, 20, 2000, 1_000_000, 20_000_000)
    assert bound == "hbm" and abs(peak - 8000.0) < 1e-6 and split["l2"] == split["infinity_cache"] == 0.0
    assert abs(sum(split.values()) - mod.gather_algorithmic_bytes(count, 20, 2000)) < 1e-6
    bound, peak, split = mod.gather_roofline_bound(count, 20, 2000, 184, 125_235)         # Enron shape: node / PE tables in L2, edge rows in the Infinity Cache
    assert bound in ("l2", "infinity_cache") and 8000.0 < peak < 34500.0 and split["l2"] > split["infinity_cache"] > 0
    assert abs(sum(split.values()) - mod.gather_algorithmic_bytes(count, 20, 2000)) < 1e-6


def test_bench_captured_collective_probe_decisions(monkeypatch, tmp_path):
    """``bench.py --gpus N`` decides between the whole-step graph and the launch-by-launch iteration from a child-process probe
    (tools/rccl_graph_probe.py); the decision logic without a GPU: one rank skips it, an opt-out skips it, a child that fails or never
    answers turns the graph off instead of failing the run."""
    import importlib.util
    import subprocess
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    spec = importlib.util.spec_from_file_location("lstep_bench_probe", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for k in ("LSTEP_FORCE_GRAPH_PROBE", "LSTEP_DIST_GRAPH", "LSTEP_SKIP_GRAPH_PROBE", "LSTEP_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    assert mod.rccl_graph_probe(1, 0, 0) == (True, "skipped (one rank)")
    monkeypatch.setenv("LSTEP_DIST_GRAPH", "0")
    assert mod.rccl_graph_probe(2, 0, 0)[0] is False
    monkeypatch.delenv("LSTEP_DIST_GRAPH")
    monkeypatch.setenv("LSTEP_SKIP_GRAPH_PROBE", "1")
    assert mod.rccl_graph_probe(2, 0, 0) == (True, "skipped")
    monkeypatch.delenv("LSTEP_SKIP_GRAPH_PROBE")
    seen = {}

    class Done:
        def __init__(self, rc, err=""):
            self.returncode, self.stdout, self.stderr = rc, "", err

    def fake_run(cmd, env=None, **kw):
        seen["env"] = env
        seen["cmd"] = cmd
        return Done(seen["rc"], seen.get("err", ""))
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("MASTER_PORT", "29700")
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")
    seen["rc"] = 0
    ok, note = mod.rccl_graph_probe(4, 3, 3)
    assert ok is True and note.startswith("ok")
    assert seen["env"]["RANK"] == "3" and seen["env"]["WORLD_SIZE"] == "4" and seen["env"]["MASTER_PORT"] == "29723"
    assert "TORCHELASTIC_USE_AGENT_STORE" not in seen["env"] and seen["cmd"][-1].endswith(os.path.join("tools", "rccl_graph_probe.py"))
    seen["rc"], seen["err"] = 3, "rccl_graph_probe: rank 3: wrong values in replay 1\n"
    ok, note = mod.rccl_graph_probe(4, 3, 3)
    assert ok is False and "exit code 3" in note and "wrong values" in note

    def hang(cmd, env=None, timeout=None, **kw):
        raise subprocess.TimeoutExpired(cmd, timeout)
    monkeypatch.setattr(subprocess, "run", hang)
    assert mod.rccl_graph_probe(4, 0, 0) == (False, "timeout")
    assert mod.rccl_probe_verdict(8, 0, 0) == {"captured": False, "pull": False, "note": "timeout"}

    # round 5: the child's last stdout line is its verdict on the owner-sharded form's exchange pattern
    class Out(Done):
        def __init__(self, rc, out):
            super().__init__(rc)
            self.stdout = out
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None, **kw: Out(0, 'rccl_graph_probe: 8 rank(s): ok\n{"captured": true, "pull": true, "pull_note": "ok"}\n'))
    assert mod.rccl_probe_verdict(8, 0, 0) == {"captured": True, "pull": True, "note": "ok"}
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None, **kw: Out(0, '{"captured": true, "pull": false, "pull_note": "wrong values in replay 1 on the second communicator"}\n'))
    v = mod.rccl_probe_verdict(8, 0, 0)
    assert v["captured"] is True and v["pull"] is False and "second communicator" in v["note"]
    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None, **kw: Out(0, "no verdict line\n"))
    assert mod.rccl_probe_verdict(8, 0, 0)["pull"] is False
    assert mod.rccl_probe_verdict(1, 0, 0) == {"captured": True, "pull": True, "note": "skipped (one rank)"}


def test_distributed_defaults_follow_the_cross_rank_probe(monkeypatch):
    """What a multi-rank job runs BY DEFAULT (round-4 VERDICT item 2, ADVICE): the owner-sharded "pull" form beyond four ranks and the
    whole-step graph with the collectives inside only behind a clean cross-rank probe; "replicate", launch by launch, without one.  The
    decision logic alone, without a GPU or a process group."""
    import types

    import torch.distributed as dist
    from lstep_amd.parallel import DistributedLstep
    for k in ("LSTEP_PHASE2", "LSTEP_PULL_MIN_WORLD", "LSTEP_DIST_GRAPH"):
        monkeypatch.delenv(k, raising=False)

    def stub(world, probe, ok=True):
        return types.SimpleNamespace(W=world, probe=probe, device="cuda:0", _device_update_ok=lambda: ok)
    clean, no_pull = {"captured": True, "pull": True}, {"captured": True, "pull": False}
    form = DistributedLstep._choose_form
    assert form(stub(8, clean)) == "pull" and form(stub(5, clean)) == "pull"
    assert form(stub(8, None)) == "replicate" and form(stub(8, no_pull)) == "replicate"
    assert form(stub(4, clean)) == "replicate" and form(stub(2, clean)) == "replicate" and form(stub(1, clean)) == "replicate"
    assert form(stub(8, clean, ok=False)) == "allgather"             # (configurations the device-count update does not cover)
    monkeypatch.setenv("LSTEP_PHASE2", "pull")
    assert form(stub(2, None)) == "pull"                              # (explicit choice: no probe needed)
    monkeypatch.setenv("LSTEP_PHASE2", "auto")
    monkeypatch.setenv("LSTEP_PULL_MIN_WORLD", "2")
    assert form(stub(2, clean)) == "pull" and form(stub(2, None)) == "replicate"
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    allowed = DistributedLstep._graph_allowed
    assert allowed(stub(1, None), None) is True                       # one rank: rehearsed on hardware
    assert allowed(stub(8, None), None) is False and allowed(stub(8, {"captured": False, "pull": False}), None) is False
    assert allowed(stub(8, clean), None) is True
    monkeypatch.setenv("LSTEP_DIST_GRAPH", "1")
    assert allowed(stub(8, None), None) is True                       # (the caller insists)
    monkeypatch.setenv("LSTEP_DIST_GRAPH", "0")
    assert allowed(stub(1, clean), None) is False
    monkeypatch.delenv("LSTEP_DIST_GRAPH")
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "gloo")
    assert allowed(stub(1, clean), None) is False

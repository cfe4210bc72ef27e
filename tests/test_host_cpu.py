"""CPU-only checks of the host layer: the C-ABI library loads and exports every symbol of include/lstep_hip.h
(no compute call: there is no GPU here), the host CSR builder reproduces the reference adjacency order, and the
FFT-filter coefficient table is the reference's fft -> filter -> ifft -> agg pipeline."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import METHOD_K, METHOD_T, SAMPLER_GRAPHS, method_inputs
from lstep_amd import _native as nat
from lstep_amd import synth
from lstep_amd.sampler import build_csr_arrays
from oracle.lstep_oracle import OracleNeighborSampler, build_oracle_model

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


@pytest.fixture(scope="module")
def lib():
    nat.build_library()
    return nat.load_library()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lstep_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lstep_[a-z0-9_]+)\s*\(", text)))


def test_abi_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert len(names) >= 14
    raw = ctypes.CDLL(nat.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/lstep_hip.h but not exported"
        assert n in nat.SIGNATURES, f"{n} has no ctypes prototype"
    assert sorted(nat.SIGNATURES) == names
    assert lib.lstep_abi_version() == nat.ABI_VERSION == 40


def test_abi_argument_validation_without_gpu(lib):
    # validation happens before any HIP call, so these are safe on a CPU-only host
    rc = lib.lstep_sample_recent(None, None, 4, None, 4, 0, None, None, None, None, None)
    assert rc == nat.LSTEP_EINVAL and b"greater than 0" in lib.lstep_last_error()
    with pytest.raises(AssertionError):
        nat.check(rc)
    csr = nat.CsrStruct(0, 0, 0, 0, 0, 0, 0)
    assert lib.lstep_sample_recent(ctypes.byref(csr), None, 4, None, 4, 5, None, None, None, None, None) == nat.LSTEP_EINVAL
    assert lib.lstep_gather_aggregate_fwd(ctypes.byref(csr), None, None, None, 170, 172, None, None, 100, None, None, None, 4, 5, 8, 3,
                                          None, None, None, None, 0, 0, 0, 0, None, None) == nat.LSTEP_EINVAL
    assert b"unsupported widths" in lib.lstep_last_error()
    assert lib.lstep_history_filter_bwd_chunks(0) == 0 and lib.lstep_history_filter_bwd_chunks(129) == 2
    assert lib.lstep_sample_recent(ctypes.byref(csr), None, 0, None, 0, 5, None, None, None, None, None) == nat.LSTEP_OK  # empty batch


@pytest.mark.parametrize("name", list(SAMPLER_GRAPHS))
def test_csr_builder_matches_reference_order(name):
    g = synth.make_temporal_graph(**SAMPLER_GRAPHS[name])
    indptr, nbr, eid, ts, rows = build_csr_arrays(g["src"], g["dst"], g["eid"], g["ts"], g["num_nodes"])
    o = OracleNeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=g["num_nodes"])
    assert rows == o.num_rows
    np.testing.assert_array_equal(indptr, o.indptr)
    np.testing.assert_array_equal(nbr, o.nbr)
    np.testing.assert_array_equal(eid, o.eid)
    np.testing.assert_array_equal(ts, o.ts)
    assert nbr.dtype == np.int32 and eid.dtype == np.int32 and ts.dtype == np.float64


def test_csr_builder_edge_cases():
    indptr, nbr, eid, ts, rows = build_csr_arrays([], [], [], [], num_nodes=3)
    assert rows == 4 and indptr.tolist() == [0, 0, 0, 0, 0] and len(nbr) == 0
    # self loop appears twice in its node's list; node 3 is isolated
    indptr, nbr, eid, ts, rows = build_csr_arrays([1, 2], [1, 1], [1, 2], [5.0, 5.0], num_nodes=3)
    assert indptr.tolist() == [0, 0, 3, 4, 4] and nbr[:3].tolist() == [1, 1, 2] and eid[:3].tolist() == [1, 1, 2]
    with pytest.raises(ValueError):
        build_csr_arrays([-1], [1], [1], [0.0])


def test_fft_coefficient_table_is_the_reference_pipeline():
    from lstep_amd.model import LSTEP
    g, node_raw, edge_raw, pe0 = method_inputs()
    sd = synth.make_state_dict(METHOD_K, METHOD_T)
    oracle = build_oracle_model(node_raw, edge_raw, None, METHOD_K, METHOD_T, sd)[0]
    bb = LSTEP(node_raw, edge_raw, None, None, num_neighbors=METHOD_K, num_fft_batches=METHOD_T, device="cpu")
    bb.load_state_dict({k[2:]: torch.as_tensor(v) for k, v in sd.items() if k.startswith("0.")})
    rng = np.random.RandomState(4)
    hist = torch.from_numpy((0.1 * rng.standard_normal((g["num_nodes"] + 1, METHOD_T, synth.PE_DIM))).astype(np.float32))
    ids = np.arange(0, 65, 7)
    for stored, bidx in ((3, 3), (METHOD_T, 9), (4, 2), (2, 0), (1, 1), (5, 40)):
        x = hist[:, :stored, :]
        ref = oracle.fourier_transform_pe(ids, x, bidx)
        coef = bb.fft_coefficients(stored, bidx)
        got = torch.einsum("usp,sp->up", x[torch.from_numpy(ids)], coef[:stored])
        np.testing.assert_allclose(got.detach().numpy(), ref.detach().numpy(), rtol=0, atol=2e-6)
    # gradients reach fft_filter (complex) and fft_agg through the table exactly as through the FFT pipeline
    x = hist[:, :METHOD_T, :]
    w = torch.from_numpy(rng.standard_normal((len(ids), synth.PE_DIM)).astype(np.float32))
    (oracle.fourier_transform_pe(ids, x, 7) * w).sum().backward()
    (torch.einsum("usp,sp->up", x[torch.from_numpy(ids)], bb.fft_coefficients(METHOD_T, 7)) * w).sum().backward()
    np.testing.assert_allclose(torch.view_as_real(bb.fft_filter.weight.grad).numpy(), torch.view_as_real(oracle.fft_filter.weight.grad).numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(bb.fft_agg.weight.grad.numpy(), oracle.fft_agg.weight.grad.numpy(), rtol=0, atol=2e-6)


def test_product_path_has_no_oracle_import():
    """The oracle is a checker: nothing under l-step_amd/ may import it (the smoke check lives in __graft_entry__.py)."""
    src_dir = os.path.join(ROOT, "l-step_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for fn in os.listdir(src_dir):
        if fn.endswith(".py"):
            hit = pat.search(open(os.path.join(src_dir, fn)).read())
            assert hit is None, fn


@pytest.mark.parametrize("strategy,tsf", [("uniform", 0.0), ("time_interval_aware", 1e-3)])
def test_rng_defined_sampling_replays_the_reference(golden, strategy, tsf):
    """uniform / time_interval_aware sampling are defined by numpy's RandomState.choice call order;
    the host replay must match the reference draw for draw (incl. the RNG state carried between calls and reset)."""
    from lstep_amd.sampler import NeighborSampler
    z = golden("random_sampling")
    g = synth.make_temporal_graph(num_nodes=40, num_edges=900, seed=50, tie_quantum=5.0)
    s = NeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], sample_neighbor_strategy=strategy, time_scaling_factor=tsf, seed=3, device="cpu")
    for call in range(2):
        for k in (3, 10):
            nbr, eid, nt = s.get_historical_neighbors(z["ids"], z["ts"], k)
            np.testing.assert_array_equal(nbr, z[f"{strategy}/call{call}/k{k}/nbr"])
            np.testing.assert_array_equal(eid, z[f"{strategy}/call{call}/k{k}/eid"])
            np.testing.assert_array_equal(nt.view(np.uint32), z[f"{strategy}/call{call}/k{k}/nt"].view(np.uint32))
    s.reset_random_state()
    np.testing.assert_array_equal(s.get_historical_neighbors(z["ids"], z["ts"], 3)[0], z[f"{strategy}/reset/k3/nbr"])


@pytest.mark.parametrize("strategy,tsf", [("uniform", 0.0), ("time_interval_aware", 1e-3), ("time_interval_aware", 0.0)])
@pytest.mark.parametrize("seed", [7, None])
def test_native_rng_replay_equals_the_numpy_loop(monkeypatch, strategy, tsf, seed):
    """``lstep_sample_random_host`` (C++ replay of numpy's legacy ``RandomState.choice``: MT19937, randint's masked rejection, the cdf path)
    against the interpreter loop around numpy's own ``choice`` (LSTEP_PY_RNG_SAMPLER=1, rounds 1-3) -- every sampled id / edge / float32 time
    identical and the generator left in the SAME state (the next numpy draw agrees) -- for K = 1 (a one-entry history consumes nothing) to
    2000 (time_gap), tied timestamps, rows without history, M != M' and the module-level generator (seed None, utils/utils.py:184)."""
    from lstep_amd.sampler import NeighborSampler
    for gkw in (dict(num_nodes=50, num_edges=3000, seed=3), dict(num_nodes=30, num_edges=20000, seed=4, time_span=500.0, tie_quantum=5.0)):
        g = synth.make_temporal_graph(**gkw)
        mk = lambda: NeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], sample_neighbor_strategy=strategy, time_scaling_factor=tsf, seed=seed,  # noqa: E731
                                     device="cpu")
        a, b = mk(), mk()
        rng = np.random.RandomState(1)
        for K in (1, 5, 20, 64, 2000):
            n = 200 if K < 2000 else 24
            ids = rng.randint(0, g["num_nodes"] + 1, n)
            ts = rng.uniform(g["ts"].min() - 1, g["ts"].max() + 1, n if K != 5 else n - 40)
            res, after = [], []
            for smp, py in ((a, "1"), (b, "0")):
                monkeypatch.setenv("LSTEP_PY_RNG_SAMPLER", py)
                if seed is None:
                    np.random.seed(11)
                res.append(smp.get_historical_neighbors(ids, ts, K))
                after.append((np.random if seed is None else smp.random_state).randint(0, 1 << 30))
            for x, y in zip(*res):
                np.testing.assert_array_equal(x, y)
            assert after[0] == after[1], "the generator state diverged"


@pytest.mark.parametrize("strategy,tsf", [("uniform", 0.0), ("time_interval_aware", 1e-3)])
def test_native_row_sort_equals_numpy_argsort_per_row(monkeypatch, strategy, tsf):
    """``lstep_sample_random_sorted_host`` (round 5: draws on one thread, positions sorted and triples gathered on worker threads; only rows
    holding a float32-time tie among DISTINCT interactions are left to numpy) against round 4's path (native draws, numpy's 1-D argsort on
    every row: LSTEP_RNG_NUMPY_SORT=1): identical arrays and generator state -- short histories (counting sort), histories far longer than K
    (std::sort), quantised timestamps (ties: the flagged rows), one worker and several, and the fill-in-place form the model uses
    (``sample_random_into`` on a dirty buffer: rows without history come back zeroed)."""
    from lstep_amd.sampler import NeighborSampler
    graphs = (dict(num_nodes=50, num_edges=3000, seed=3), dict(num_nodes=3, num_edges=30000, seed=5),
              dict(num_nodes=30, num_edges=20000, seed=4, time_span=500.0, tie_quantum=5.0))
    for gi, gkw in enumerate(graphs):
        g = synth.make_temporal_graph(**gkw)
        mk = lambda: NeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], sample_neighbor_strategy=strategy, time_scaling_factor=tsf, seed=9,  # noqa: E731
                                     device="cpu")
        smp = {"numpy": mk(), "one": mk(), "many": mk(), "into": mk()}
        rng = np.random.RandomState(2)
        flagged = 0
        for K in (1, 7, 64, 700):
            n = 150 if K < 700 else 40
            ids = rng.randint(0, g["num_nodes"] + 1, n)
            ts = rng.uniform(g["ts"].min() - 1, g["ts"].max() + 1, n)
            res = {}
            for name, s in smp.items():
                monkeypatch.setenv("LSTEP_RNG_NUMPY_SORT", "1" if name == "numpy" else "0")
                monkeypatch.setenv("LSTEP_HOST_THREADS", "1" if name == "one" else "5")
                if name == "into":
                    out = (np.full((n, K), -7, np.int64), np.full((n, K), -7, np.int64), np.full((n, K), np.nan, np.float32))
                    assert all(a is b for a, b in zip(s.sample_random_into(ids, ts, K, out), out))
                    res[name] = out
                else:
                    res[name] = s.get_historical_neighbors(ids, ts, K)
            flagged += smp["many"].last_numpy_sorted_rows
            for name in ("one", "many", "into"):
                for x, y in zip(res["numpy"], res[name]):
                    np.testing.assert_array_equal(x, y, err_msg=f"{name} K={K} graph {gi}")
                np.testing.assert_array_equal(res["numpy"][2].view(np.uint32), res[name][2].view(np.uint32))
        states = {name: s.random_state.randint(0, 1 << 30) for name, s in smp.items()}
        assert len(set(states.values())) == 1, states
        if "tie_quantum" in gkw:
            assert flagged > 0, "the quantised timestamps must have produced rows with ties (left to numpy)"
        elif gi == 0:
            assert flagged < 100, "only the rows that drew both ends of a self-loop (one interaction listed twice) hold a tie"


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback: with the shared library absent, loading (and therefore every op and both model classes) raises."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from lstep_amd import _native as nat\n"
            "try:\n    nat.load_library()\nexcept nat.LstepNativeError as e:\n    print('RAISED', 'no CPU fallback' in str(e))\n"
            "try:\n    from lstep_amd.sampler import NeighborSampler\n    NeighborSampler([1], [2], [1], [0.5], device='cpu')\n"
            "except nat.LstepNativeError:\n    print('SAMPLER RAISED')\n") % ROOT
    env = dict(os.environ, LSTEP_LIB=os.path.join(tmp_path, "does_not_exist.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300).stdout
    assert "RAISED True" in out and "SAMPLER RAISED" in out, out


def test_tail_weight_composition_matches_autograd():
    """model._TailWeights (hand-derived backward of the padded / pre-multiplied dense-tail weights) against the same
    composition written with F.pad / cat / matmul and differentiated by autograd."""
    from lstep_amd.model import _TailWeights, _pad1, _pad2
    torch.manual_seed(0)
    Fd, D, P, K = 172, 100, 172, 7
    C, CP, Ce, Fn, Cp, Pp = D + Fd, P + D, 288, 176, 288, 176
    mk = lambda *s: (0.1 * torch.randn(*s)).requires_grad_(True)  # noqa: E731
    W1, b1, aw, ab, W2, b2 = mk(C, C), mk(C), mk(1, K), mk(1), mk(C, C), mk(C)
    Wn, bn, Wo, bo = mk(Fd, C + Fd), mk(Fd), mk(Fd, P + Fd), mk(Fd)
    Ws, bs, Wn1, bn1, Wn2, bn2 = mk(P, P), mk(P), mk(P, CP), mk(P), mk(P, P), mk(P)
    params = [W1, b1, aw, ab, W2, b2, Wn, bn, Wo, bo, Ws, bs, Wn1, bn1, Wn2, bn2]
    a = aw.reshape(-1)
    wo_a, wo_b, wn_a, wn_b = Wo[:, :Fd], Wo[:, Fd:], Wn[:, :Fd], Wn[:, Fd:]
    M = wo_a @ wn_b
    ref = (_pad2(W1, Ce, Ce), _pad1(a.sum() * b1 + ab, Ce), _pad2(Wn1, Pp, Cp), _pad1(bn1, Pp),
           torch.cat([_pad2(Ws, Pp, Pp), _pad2(Wn2, Pp, Pp)], 1), _pad1(bs + bn2, Pp),
           torch.cat([_pad2(wo_a @ wn_a, Fn, Fn), _pad2(M @ W2, Fn, Ce), _pad2(wo_b, Fn, Pp)], 1), _pad1(M @ b2 + wo_a @ bn + bo, Fn))
    got = _TailWeights.apply((Fd, C, P, CP, Ce, Fn, Cp, Pp), *params)
    assert len(got) == 12                       # 8 operands + the 4 transposed matrices the backward kernel reads
    for x, y in zip(got, ref):
        np.testing.assert_allclose(x.detach().numpy(), y.detach().numpy(), rtol=0, atol=1e-6)
    for t, src in zip(got[8:], (got[0], got[2], got[4], got[6])):
        assert t.is_contiguous() and not t.requires_grad and torch.equal(t, src.detach().t())
    ws = [torch.randn_like(o) for o in ref]
    g_ref = torch.autograd.grad(sum((o * w).sum() for o, w in zip(ref, ws)), params)
    g_got = torch.autograd.grad(sum((o * w).sum() for o, w in zip(got, ws)), params)
    for i, (x, y) in enumerate(zip(g_got, g_ref)):
        assert x.shape == y.shape, i
        np.testing.assert_allclose(x.numpy(), y.numpy(), rtol=1e-4, atol=2e-5, err_msg=str(i))


def test_graft_entry_build_runs():
    """The driver's build hook: compiles (or finds up to date) the library, checks the ABI version, imports the package."""
    import importlib
    entry = importlib.import_module("__graft_entry__")
    entry.build()


def test_lookahead_batch_key_survives_allocator_reuse():
    """The look-ahead grouping is only picked up for the SAME endpoint buffers, unmodified.  The key keeps the tensors alive, so a later
    batch can never sit at a freed look-ahead batch's address (the collision the old (data_ptr, numel, version) key allowed)."""
    import gc

    import torch

    from lstep_amd.engine import BatchKey
    base_s, base_d = torch.arange(100), torch.arange(100, 200)
    src, dst = base_s[10:30], base_d[10:30]
    key = BatchKey(src, dst)
    assert key.matches(base_s[10:30], base_d[10:30])            # another view of the same memory (what EdgeStream.batch returns)
    assert not key.matches(base_s[11:31], base_d[10:30])
    assert not key.matches(base_s[10:29], base_d[10:29])
    base_s[12] = 7                                              # written since: version counters are shared by all views
    assert not key.matches(base_s[10:30], base_d[10:30])
    # freed look-ahead tensors: the key's references keep the block, a new batch of the same size gets other memory
    a, b = torch.arange(64) + 1, torch.arange(64) + 2
    key = BatchKey(a, b)
    ptrs = (a.data_ptr(), b.data_ptr())
    del a, b
    gc.collect()
    c, d = torch.arange(64) + 3, torch.arange(64) + 4
    assert (c.data_ptr(), d.data_ptr()) != ptrs and not key.matches(c, d)


def test_initial_positional_encodings_properties():
    """LapPE / RWPE initial encodings (utils/PositionalEncoding.py:42-62,69-91; parity UNPINNED: torch_geometric is absent and the
    reference's ARPACK output is an arbitrary basis of hugely degenerate eigenspaces).  What is well defined is checked: the
    eigen-equation, orthonormal columns, ascending eigenvalues in [0, 2], +-1 column signs from the seeded generator, and the
    return-probability definition of RWPE against dense matrix powers."""
    import torch

    from lstep_amd import init_pe
    g = synth.make_temporal_graph(num_nodes=60, num_edges=300, seed=9)
    src, dst = g["src"][:80], g["dst"][:80]          # the first batch only (train_LSTEP_link_prediction.py:168-189)
    n, k = 61, 12
    ei = init_pe.first_batch_edge_index(src, dst)
    assert tuple(ei.shape) == (2, 160) and torch.equal(ei[0, :80], torch.from_numpy(src)) and torch.equal(ei[1, :80], torch.from_numpy(dst))
    lap, ew = init_pe.sym_normalised_laplacian(ei, n)
    dense = lap.toarray()
    assert np.allclose(dense, dense.T) and ew.numel() == 160 + n
    deg = np.bincount(ei[0].numpy(), minlength=n)
    iso = deg == 0
    assert iso.sum() > 0 and np.all(np.diag(dense)[iso] == 1.0) and np.all(np.abs(dense[iso]).sum(1) == 1.0)   # isolated nodes: a lone 1 on the diagonal
    gen = torch.Generator().manual_seed(3)
    pe, _ = init_pe.laplacian_pe(ei, n, k, generator=gen)
    assert pe.dtype == torch.float64 and tuple(pe.shape) == (n, k)
    v = pe.numpy()
    lam = np.einsum("ij,ij->j", v, dense @ v)
    assert np.allclose(dense @ v, v * lam, atol=1e-8), "columns must be eigenvectors"
    assert np.allclose(v.T @ v, np.eye(k), atol=1e-8)
    assert np.all(np.diff(lam) >= -1e-9) and lam.min() >= -1e-9 and lam.max() <= 2 + 1e-9
    full = np.linalg.eigvalsh(dense)
    assert np.allclose(lam, full[1:k + 1], atol=1e-8), "eigenvalues 1..k of the spectrum (the smallest one is dropped)"
    rw = init_pe.random_walk_pe(ei, n, 5)
    p = np.zeros((n, n))
    np.add.at(p, (ei[0].numpy(), ei[1].numpy()), 1.0 / np.maximum(deg, 1)[ei[0].numpy()])
    q = np.eye(n)
    for i in range(5):
        q = q @ p
        assert np.allclose(rw[:, i].numpy(), np.diag(q), atol=1e-6)
    assert rw.dtype == torch.float32 and float(rw[deg == 0].abs().max()) == 0.0


def test_chunked_oracle_training_step_equals_the_protocol_iteration():
    """tests/helpers.py::oracle_train_step_chunked (the form in which the -m gpu tests afford the oracle's full-batch losses and
    gradients at B = 4096 / 16384) against ``protocol.train_iteration`` (the reference's loop body) on the same state: losses,
    probabilities, every parameter gradient and the table update_pe leaves behind, for a chunk size that does not divide the batch."""
    from helpers import oracle_train_step_chunked
    from lstep_amd import protocol
    N, E, B, K, T, G = 50, 1500, 40, 5, 4, 2000
    g = synth.make_temporal_graph(num_nodes=N, num_edges=E, seed=61)
    node_raw, edge_raw = synth.make_features(N, E, seed=62)
    sd = synth.make_state_dict(K, T, seed=63)
    hist = torch.from_numpy((0.1 * np.random.RandomState(64).standard_normal((N + 1, T, synth.PE_DIM))).astype(np.float32))
    hist[0, -1] = 0.01                  # a live padding row
    sl = slice(300, 300 + B)            # early: many padded neighbour slots
    src, dst, ts, eid = g["src"][sl], g["dst"][sl], g["ts"][sl], g["eid"][sl]
    neg = synth.make_negatives(N, B, seed=65)
    out = []
    for chunked in (False, True):
        om = build_oracle_model(node_raw, edge_raw, OracleNeighborSampler(g["src"], g["dst"], g["eid"], g["ts"], num_nodes=N), K, T, sd)
        om.train()
        if not chunked:
            opt = torch.optim.SGD(om.parameters(), lr=0.0)       # the iteration steps its optimiser: keep the weights where they are
            st = protocol.ProtocolState(history=hist.clone())
            res = protocol.train_iteration(om[0], om[1], opt, st, 9, src, dst, ts, eid, neg, K, G, T)
            res["table"] = st.history[:, -1, :]
        else:
            bn = protocol.unique_batch_nodes(src, dst)
            res = oracle_train_step_chunked(om, hist[torch.from_numpy(bn)], hist[:, -1, :].clone(), bn, 9, src, dst, ts, eid, neg, K, G, chunk=16,
                                            want_emb=[0, B + 3, 2 * B + 39])
            assert sorted(res["emb"]) == [0, B + 3, 2 * B + 39]
            with torch.no_grad():
                e = om[0].combining_pe_raw_feat(res["spliced"], dst[3:5], ts[3:5], K, G)[0].numpy()     # (two rows: the reference's squeeze() breaks B == 1)
            np.testing.assert_allclose(res["emb"][B + 3], e, rtol=0, atol=1e-6)
        out.append((res, {k: (None if p.grad is None else p.grad.clone()) for k, p in om.named_parameters()}))
    (ra, ga), (rb, gb) = out
    for k in ("lp_loss", "pe_loss", "loss"):
        assert abs(ra[k] - rb[k]) < 2e-6, k
    np.testing.assert_allclose(rb["predicts"], ra["predicts"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(rb["table"].numpy(), ra["table"].numpy(), rtol=0, atol=1e-6)
    for k in ga:
        if ga[k] is None:
            assert gb[k] is None or float(gb[k].abs().max()) == 0.0, k
            continue
        a, b = ga[k], gb[k]
        d = float((torch.view_as_real(a - b) if a.is_complex() else (a - b)).abs().max())
        assert d <= 2e-6 * max(1.0, float(a.abs().max())), (k, d)


def test_initial_positional_encodings_match_the_reference(golden):
    """``lstep_amd.init_pe`` (scipy) against ``utils/PositionalEncoding.py`` itself, run by tests/golden/make_golden.py on shims of the six
    ``torch_geometric.utils`` functions it calls (their documented semantics; the wheel is absent and unpinned): RWPE entry for entry; the
    normalised Laplacian and the ``edge_weight`` LaplacianPE returns; its eigenvector columns up to the sign the reference randomises
    (``:57-59``) on a graph whose small eigenvalues are simple."""
    from lstep_amd import init_pe
    z = golden("init_pe")
    ei, n = z["edge_index"], 40
    k, walk = z["lappe_abs"].shape[1], z["rwpe"].shape[1]
    rw = init_pe.random_walk_pe(ei, n, walk)
    assert rw.dtype == torch.float32 and tuple(rw.shape) == (n, walk)
    np.testing.assert_allclose(rw.numpy(), z["rwpe"], rtol=0, atol=1e-6)
    assert float(z["rwpe"][-3:].max()) == 0.0 and float(z["rwpe"][:, 1].min()) >= 0.0      # isolated nodes never return; the fixture has them
    lap, edge_weight = init_pe.sym_normalised_laplacian(ei, n)
    np.testing.assert_allclose(lap.toarray(), z["laplacian"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(edge_weight.numpy(), z["lappe_edge_weight"], rtol=0, atol=1e-6)
    pe, ew = init_pe.laplacian_pe(ei, n, k, generator=torch.Generator().manual_seed(1))
    assert pe.dtype == torch.float64 and tuple(pe.shape) == (n, k)
    np.testing.assert_allclose(np.abs(pe.numpy()), z["lappe_abs"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ew.numpy(), z["lappe_edge_weight"], rtol=0, atol=1e-6)
    # A SECOND, independent statement of what the fixture's Laplacian must be (half of LapPE's arithmetic sat in the builder-written
    # get_laplacian shim): the textbook I - D^-1/2 A D^-1/2 assembled densely with numpy from the raw edge list -- A counts parallel edges,
    # self-loops dropped, isolated nodes keep a lone 1 on the diagonal -- and its eigenvectors from a dense symmetric solver instead of ARPACK.
    a = np.zeros((n, n))
    np.add.at(a, (ei[0][ei[0] != ei[1]], ei[1][ei[0] != ei[1]]), 1.0)
    d = a.sum(1)
    dis = np.where(d > 0, 1.0 / np.sqrt(np.maximum(d, 1e-300)), 0.0)
    textbook = np.eye(n) - dis[:, None] * a * dis[None, :]
    np.testing.assert_allclose(z["laplacian"], textbook, rtol=0, atol=1e-6)          # what the reference file produced through the shim
    evals, evecs = np.linalg.eigh(textbook)
    np.testing.assert_allclose(np.abs(evecs[:, 1:k + 1]), z["lappe_abs"], rtol=0, atol=1e-6)
    assert np.all(np.diff(evals[:k + 2]) > 1e-3)

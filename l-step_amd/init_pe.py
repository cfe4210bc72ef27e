"""Initial positional encodings: the one-off host-side step in front of the hot path.

The reference computes them once per run from the FIRST training batch's edges only
(``train_LSTEP_link_prediction.py:168-189``) with ``utils/PositionalEncoding.py``:

* ``LaplacianPE(edge_index, num_nodes, k)`` (``:42-62``): the ``k + 1`` smallest eigenpairs (ARPACK ``eigsh(which='SA')``) of the
  symmetrically normalised Laplacian ``I - D^-1/2 A D^-1/2`` of the undirected multigraph (duplicate edges add up), eigenvectors
  ``1 .. k`` with a random sign per column (``torch.randint``); float64; returns ``(pe [N, k], edge_weight)``;
* ``RandomWalkPE(edge_index, num_nodes, walk_length)`` (``:69-91``): ``pe[n, i] = (P^(i+1))[n, n]`` for the random-walk matrix
  ``P = D_out^-1 A`` (rows without an edge keep degree 1).

**Parity pin.**  The reference builds both on ``torch_geometric.utils`` (``get_laplacian``, ``to_scipy_sparse_matrix``,
``to_torch_csr_tensor``, ``to_edge_index``, ``get_self_loop_attr``, ``scatter``), which is not installed here and pinned nowhere (the
reference has no requirements file).  ``tests/golden/make_golden.py::gen_init_pe`` runs the reference FILE on shims of those six functions
written from their documented semantics (as it does for ``torch_scatter``) and ``tests/test_host_cpu.py`` holds this module to the result:
RWPE entry for entry, the normalised Laplacian and ``edge_weight`` entry for entry, the LapPE columns up to their (random) sign on a graph
whose small eigenvalues are simple.  What no fixture can pin is LapPE on the reference's ACTUAL input: a graph of <= 2 B edges on N nodes
leaves N - O(B) isolated nodes, i.e. one eigenvalue of multiplicity ~N, and ARPACK returns an arbitrary basis of that eigenspace that
depends on its random start vector -- there the output is not a function of the input; the property tests (eigen-equation, orthonormal
columns, eigenvalue order) cover that case.

Host code (scipy), like the reference: this runs once before the first batch, on <= 2 B edges.  ``random_walk_pe_device`` /
``laplacian_pe_device`` compute the same encodings on the GPU (sparse products / a dense float64 ``eigh``) for callers that keep the
initial PE on the device; both are held to the same fixtures (``tests/test_hip_parity.py``).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch
from scipy.sparse.linalg import eigsh


def _undirected(edge_index) -> np.ndarray:
    ei = edge_index.cpu().numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index)
    if ei.ndim != 2 or ei.shape[0] != 2:
        raise ValueError("edge_index must be [2, num_edges]")
    return ei.astype(np.int64)


def sym_normalised_laplacian(edge_index, num_nodes: int):
    """``I - D^-1/2 A D^-1/2`` as scipy CSR plus the COO values in torch_geometric's order (off-diagonal entries first, then the
    diagonal), i.e. what ``get_laplacian(normalization='sym')`` returns as ``edge_weight``.  The reference passes both directions of
    every batch edge (``train:181-182``); parallel edges add up when the COO matrix is assembled."""
    ei = _undirected(edge_index)
    row, col = ei
    w = np.ones(ei.shape[1], dtype=np.float64)
    deg = np.bincount(row, weights=w, minlength=num_nodes)
    with np.errstate(divide="ignore"):
        dis = 1.0 / np.sqrt(deg)
    dis[~np.isfinite(dis)] = 0.0
    off = -dis[row] * w * dis[col]
    rows = np.concatenate([row, np.arange(num_nodes)])
    cols = np.concatenate([col, np.arange(num_nodes)])
    vals = np.concatenate([off, np.ones(num_nodes)])
    lap = sp.coo_matrix((vals, (rows, cols)), shape=(num_nodes, num_nodes)).tocsr()
    return lap, torch.from_numpy(vals.astype(np.float32))


def laplacian_pe(edge_index, num_nodes: int, k: int, generator: torch.Generator = None):
    """Mirror of reference ``LaplacianPE`` (``utils/PositionalEncoding.py:42-62``): ``(pe float64 [num_nodes, k], edge_weight)``."""
    if not 0 < k < num_nodes - 1:
        raise ValueError("ARPACK needs 0 < k + 1 < num_nodes")
    lap, edge_weight = sym_normalised_laplacian(edge_index, num_nodes)
    vals, vecs = eigsh(lap, k=k + 1, which="SA", return_eigenvectors=True)
    vecs = np.real(vecs[:, vals.argsort()])
    pe = torch.from_numpy(np.ascontiguousarray(vecs[:, 1:k + 1]))
    sign = -1 + 2 * torch.randint(0, 2, (k,), generator=generator)
    pe *= sign
    return pe, edge_weight


def random_walk_pe(edge_index, num_nodes: int, walk_length: int) -> torch.Tensor:
    """Mirror of reference ``RandomWalkPE`` (``utils/PositionalEncoding.py:69-91``): float32 ``[num_nodes, walk_length]``."""
    ei = _undirected(edge_index)
    row, col = ei
    deg = np.maximum(np.bincount(row, minlength=num_nodes).astype(np.float32), 1.0)
    adj = sp.coo_matrix((1.0 / deg[row], (row, col)), shape=(num_nodes, num_nodes), dtype=np.float32).tocsr()
    out = adj
    cols = [out.diagonal()]
    for _ in range(walk_length - 1):
        out = out @ adj
        cols.append(out.diagonal())
    return torch.from_numpy(np.stack(cols, axis=-1).astype(np.float32))


# ---- the same two encodings computed on the GPU (torch device ops: this is set-up in front of the hot path, not a kernel of it) ----------
def random_walk_pe_device(edge_index, num_nodes: int, walk_length: int, device="cuda") -> torch.Tensor:
    """``random_walk_pe`` on the device: the random-walk matrix as a sparse CSR tensor, ``walk_length - 1`` sparse x sparse products, the
    diagonal of every power.  At the reference's input (<= 2 B edges of the first batch on N nodes) the powers stay sparse; the result is a
    device tensor ``[num_nodes, walk_length]`` float32 that the engine's ``initial_pe`` can take without a host round trip."""
    ei = edge_index.to(device=device, dtype=torch.int64) if isinstance(edge_index, torch.Tensor) else torch.as_tensor(np.asarray(edge_index), dtype=torch.int64, device=device)
    row, col = ei[0], ei[1]
    deg = torch.zeros(num_nodes, dtype=torch.float32, device=ei.device).index_add_(0, row, torch.ones_like(row, dtype=torch.float32)).clamp_(min=1.0)
    adj = torch.sparse_coo_tensor(ei, 1.0 / deg[row], (num_nodes, num_nodes)).coalesce()

    def diagonal(m):
        m = m.coalesce()
        idx, val = m.indices(), m.values()
        on = idx[0] == idx[1]
        return torch.zeros(num_nodes, dtype=torch.float32, device=ei.device).index_add_(0, idx[0][on], val[on])

    out = adj
    cols = [diagonal(out)]
    for _ in range(walk_length - 1):
        out = torch.sparse.mm(out, adj)
        cols.append(diagonal(out))
    return torch.stack(cols, dim=-1)


def laplacian_pe_device(edge_index, num_nodes: int, k: int, generator: torch.Generator = None, device="cuda", dense_limit: int = 16384):
    """``laplacian_pe`` with the eigen-decomposition on the device: the dense symmetric-normalised Laplacian (``num_nodes`` <= ``dense_limit``:
    1 GB in float64 at 11 k nodes, the reference's real datasets) through ``torch.linalg.eigh`` in float64 -- ALL eigenpairs, of which
    columns 1 .. k are kept, where ARPACK iterates for the k + 1 smallest.  Same definition, same sign convention (a random sign per
    column from ``torch.randint``); larger graphs take the host path (``laplacian_pe``)."""
    if num_nodes > dense_limit:
        pe, ew = laplacian_pe(edge_index, num_nodes, k, generator)
        return pe.to(device), ew.to(device)
    if not 0 < k < num_nodes - 1:
        raise ValueError("0 < k + 1 < num_nodes")
    ei = edge_index.to(device=device, dtype=torch.int64) if isinstance(edge_index, torch.Tensor) else torch.as_tensor(np.asarray(edge_index), dtype=torch.int64, device=device)
    row, col = ei[0], ei[1]
    keep = row != col
    row, col = row[keep], col[keep]
    w = torch.ones(row.numel(), dtype=torch.float64, device=ei.device)
    deg = torch.zeros(num_nodes, dtype=torch.float64, device=ei.device).index_add_(0, row, w)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0.0
    off = -dis[row] * w * dis[col]
    lap = torch.eye(num_nodes, dtype=torch.float64, device=ei.device)
    lap.index_put_((row, col), off, accumulate=True)
    vals, vecs = torch.linalg.eigh(lap)           # ascending eigenvalues
    pe = vecs[:, 1:k + 1].contiguous()
    sign = -1 + 2 * torch.randint(0, 2, (k,), generator=generator)
    pe = pe * sign.to(pe.device)
    edge_weight = torch.cat([off.to(torch.float32), torch.ones(num_nodes, dtype=torch.float32, device=ei.device)])
    return pe, edge_weight


def first_batch_edge_index(src, dst) -> torch.Tensor:
    """``edge_index`` exactly as the reference assembles it from the first batch (``train:181-182``): both directions, sources first."""
    src, dst = np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)
    return torch.from_numpy(np.stack([np.concatenate([src, dst]), np.concatenate([dst, src])]))

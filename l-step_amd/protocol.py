"""The per-batch caller contract of the L-STEP hot path, written against the method surface only.

The reference has no function for this: its train and eval batch loops are script bodies
(``train_LSTEP_link_prediction.py:204-311``, ``evaluate_model_utils.py:38-142``).  This module restates
those two loop bodies as functions that only touch ``backbone.fourier_transform_pe`` /
``combining_pe_raw_feat`` / ``update_pe`` and ``predictor(input_1=, input_2=)``, so the same driver runs

* the reference classes (``tests/golden/make_golden.py`` -> golden traces),
* the CPU oracle (``oracle/lstep_oracle.py``) and
* the HIP-backed :class:`lstep_amd.model.LSTEP`

and the three can be compared batch by batch.  The PE history is kept exactly as the reference keeps it
(a dense ``[N+1, t, P]`` tensor that grows by ``torch.cat`` and is trimmed to the last ``T`` snapshots);
the device-resident ring used by the fast harness lives in ``lstep_amd/engine.py``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F


def unique_batch_nodes(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """Sorted unique endpoints of a batch (``train:221-222,283-284``)."""
    return np.unique(np.concatenate([src, dst])).astype(np.int64)


@dataclass
class ProtocolState:
    """What survives from one batch to the next in the reference loops."""
    history: torch.Tensor  # [N+1, t, P]; t == 0 before the first batch of an epoch (train:197)
    initial_pe: torch.Tensor | None = None  # aliased and mutated in place at batch 0 (train:281,286)
    log: list = field(default_factory=list)


def splice_current_pe(backbone, history: torch.Tensor, batch_nodes: np.ndarray, batch_idx: int, num_fft_batches: int):
    """FFT-filter the batch rows and splice them into a copy of the last snapshot (``train:224-230``)."""
    if history.shape[1] > num_fft_batches:
        history = history[:, -num_fft_batches:, :].clone()
    filtered = backbone.fourier_transform_pe(batch_nodes, history, batch_idx)
    current = history[:, -1, :].clone()
    current[torch.from_numpy(batch_nodes).to(current.device)] = filtered
    return history, current


def link_probabilities(predictor, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``train:254-255``."""
    return predictor(input_1=a, input_2=b).squeeze(dim=-1).sigmoid().clamp(0, 1)


def train_iteration(backbone, predictor, optimizer, state: ProtocolState, batch_idx: int,
                    src: np.ndarray, dst: np.ndarray, times: np.ndarray, edge_ids: np.ndarray, neg_dst: np.ndarray,
                    num_neighbors: int, time_gap: int, num_fft_batches: int,
                    pe_weight: float = 0.5, neg_sample_weight: float = 0.3):
    """One batch of ``train_LSTEP_link_prediction.py:204-311``.  Returns a dict of python floats (losses) or None at batch 0."""
    dev = state.history.device if state.history.numel() else (state.initial_pe.device if state.initial_pe is not None else "cpu")
    out = None
    loss = None
    if batch_idx == 0:
        current = state.initial_pe  # aliased on purpose (train:281)
    else:
        batch_nodes = unique_batch_nodes(src, dst)
        state.history, current = splice_current_pe(backbone, state.history, batch_nodes, batch_idx, num_fft_batches)
        pos_src = backbone.combining_pe_raw_feat(pe=current, node_ids=src, node_interact_times=times,
                                                 num_neighbors=num_neighbors, time_gap=time_gap)
        pos_dst = backbone.combining_pe_raw_feat(pe=current, node_ids=dst, node_interact_times=times,
                                                 num_neighbors=num_neighbors, time_gap=time_gap)
        neg_src = pos_src  # train:245
        neg_dst_emb = backbone.combining_pe_raw_feat(pe=current, node_ids=neg_dst, node_interact_times=times,
                                                     num_neighbors=num_neighbors, time_gap=time_gap)
        p_pos = link_probabilities(predictor, pos_src, pos_dst)
        p_neg = link_probabilities(predictor, neg_src, neg_dst_emb)
        i_src = torch.from_numpy(src).to(dev)
        i_dst = torch.from_numpy(dst).to(dev)
        i_neg = torch.from_numpy(neg_dst).to(dev)
        predicts = torch.cat([p_pos, p_neg], dim=0)
        labels = torch.cat([torch.ones_like(p_pos), torch.zeros_like(p_neg)], dim=0)
        lp_loss = F.binary_cross_entropy(predicts, labels)
        pe_loss = F.mse_loss(current[i_src], current[i_dst]) - neg_sample_weight * F.mse_loss(current[i_src], current[i_neg])
        loss = (1.0 - pe_weight) * lp_loss + pe_weight * pe_loss
        out = {"lp_loss": float(lp_loss.item()), "pe_loss": float(pe_loss.item()), "loss": float(loss.item()),
               "predicts": predicts.detach().cpu().numpy()}

    batch_nodes = unique_batch_nodes(src, dst)
    new_pe = backbone.update_pe(pe=current, node_ids=batch_nodes, edge_ids=edge_ids, batch_src_node_ids=src,
                                batch_dst_node_ids=dst, node_interact_times=times, current_time=times.max(),
                                num_neighbors=num_neighbors, time_gap=time_gap)
    if batch_idx > 0:
        current = new_pe
    snap = current.unsqueeze(1)
    state.history = torch.cat([state.history.to(snap.device), snap], dim=1).detach()
    if loss is not None:
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
    return out


def eval_iteration(backbone, predictor, state: ProtocolState, batch_idx: int,
                   src: np.ndarray, dst: np.ndarray, times: np.ndarray, edge_ids: np.ndarray,
                   neg_src: np.ndarray, neg_dst: np.ndarray,
                   num_neighbors: int, time_gap: int, num_fft_batches: int):
    """One batch of ``evaluate_model_utils.py:38-142`` (caller wraps in ``torch.no_grad()`` and ``model.eval()``)."""
    batch_nodes = unique_batch_nodes(src, dst)
    state.history, current = splice_current_pe(backbone, state.history, batch_nodes, batch_idx, num_fft_batches)
    embs = [backbone.combining_pe_raw_feat(pe=current, node_ids=ids, node_interact_times=times,
                                           num_neighbors=num_neighbors, time_gap=time_gap)
            for ids in (src, dst, neg_src, neg_dst)]
    p_pos = link_probabilities(predictor, embs[0], embs[1])
    p_neg = link_probabilities(predictor, embs[2], embs[3])
    predicts = torch.cat([p_pos, p_neg], dim=0)
    labels = torch.cat([torch.ones_like(p_pos), torch.zeros_like(p_neg)], dim=0)
    current = backbone.update_pe(pe=current, node_ids=batch_nodes, edge_ids=edge_ids, batch_src_node_ids=src,
                                 batch_dst_node_ids=dst, node_interact_times=times, current_time=times.max(),
                                 num_neighbors=num_neighbors, time_gap=time_gap)
    state.history = torch.cat([state.history, current.unsqueeze(1)], dim=1)
    loss = F.binary_cross_entropy(predicts, labels)
    return {"loss": float(loss.item()), "predicts": predicts.detach().cpu().numpy()}

"""
Pair sharding across the GPUs of a node and the single collective of the path.

Song pairs are independent units (Serra09.py:161-192); the reference itself shards the pair list
over joblib processes (CoverAlgorithm.py:169-173) and over cluster jobs (:203-247).  Here: one
process per GPU, features replicated on every GPU, the pair list dealt to ranks, and ONE
all-gather (RCCL over xGMI under the "nccl" backend, gloo on CPU in the tests) of the per-rank
score vectors at the end.  No other exchange exists on the path.
"""
import numpy as np
import torch
import torch.distributed as dist


def pair_costs(frame_off, pairs, win=1):
    """Work estimate per pair: cells of its cross-recurrence matrix."""
    lens = np.diff(np.asarray(frame_off, dtype=np.int64))
    pairs = np.asarray(pairs).reshape(-1, 2)
    return (lens[pairs[:, 0]] - win + 1) * (lens[pairs[:, 1]] - win + 1)


def shard_indices(costs, world_size, rank):
    """
    Indices (into the pair list) owned by `rank`: pairs sorted by descending cost and dealt
    round-robin in a snake order, so ragged song lengths balance across ranks and every rank gets
    the same count up to one.  Deterministic; the shards of all ranks partition range(K).
    """
    costs = np.asarray(costs)
    K = len(costs)
    order = np.argsort(-costs, kind="stable")
    pos = np.arange(K)
    rnd, slot = pos // world_size, pos % world_size
    owner = np.where(rnd % 2 == 0, slot, world_size - 1 - slot)
    return np.sort(order[owner == rank])


def gather_scores(local_scores, local_idx, K, group=None, index_of_rank=None, force_collective=False):
    """
    All-gather of per-rank score vectors into the full length-K vector (on every rank): the path's ONE collective.
    local_scores: float tensor (n_local,) on this rank's device (GPU for nccl, CPU for gloo);
    local_idx: the int64 positions of those scores in the global pair list.
    index_of_rank: callable rank -> positions owned by that rank.  The shards of shard_indices() are a deterministic
    function of (costs, world_size, rank), so every rank can name every other rank's positions and only the scores
    travel; without it the positions ride along (as int64 bit patterns in a second half of the message).
    Shards are padded to a common length so that one all_gather_into_tensor moves everything.
    force_collective: run the collective even in a one-rank group (the one-GPU box's way of proving that RCCL loads, the
    communicator forms and the gather runs on device tensors: tests/test_gpu_rccl.py); needs an initialised group.
    """
    if not (dist.is_available() and dist.is_initialized()):
        if force_collective:
            raise RuntimeError("gather_scores(force_collective=True) needs an initialised torch.distributed group")
        world1 = True
    else:
        world1 = dist.get_world_size(group) == 1 and not force_collective
    if world1:
        out = torch.zeros(K, dtype=local_scores.dtype, device=local_scores.device)
        out[torch.as_tensor(local_idx, device=local_scores.device, dtype=torch.long)] = local_scores
        return out
    world = dist.get_world_size(group)
    dev = local_scores.device
    n_max = -(-K // world)
    n = local_scores.numel()
    out = torch.zeros(K, dtype=local_scores.dtype, device=dev)
    if index_of_rank is not None:
        send = torch.zeros(n_max, dtype=local_scores.dtype, device=dev)
        send[:n] = local_scores
        recv = torch.empty(world * n_max, dtype=local_scores.dtype, device=dev)
        dist.all_gather_into_tensor(recv, send, group=group)
        recv = recv.view(world, n_max)
        for r in range(world):
            idx = torch.as_tensor(np.asarray(index_of_rank(r)), device=dev, dtype=torch.long)
            out[idx] = recv[r, :idx.numel()]
        return out
    send = torch.zeros(2 * n_max, dtype=torch.float64, device=dev)
    send[:n] = local_scores.to(torch.float64)
    idx_part = send[n_max:].view(torch.int64)
    idx_part[:n] = torch.as_tensor(local_idx, device=dev, dtype=torch.int64)
    idx_part[n:] = -1
    recv = torch.empty(world * 2 * n_max, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, 2, n_max)
    idx = recv[:, 1, :].reshape(-1).view(torch.int64)
    val = recv[:, 0, :].reshape(-1)
    keep = idx >= 0
    out[idx[keep]] = val[keep].to(local_scores.dtype)
    return out


def scatter_to_matrix(pairs, scores, n_songs, symmetric=True):
    """Ds[i, j] = score, then Ds += Ds.T when symmetric (CoverAlgorithm.py:180-182)."""
    D = np.zeros((n_songs, n_songs), dtype=np.float32)
    pairs = np.asarray(pairs).reshape(-1, 2)
    D[pairs[:, 0], pairs[:, 1]] = np.asarray(scores, dtype=np.float32)
    if symmetric:
        D += D.T
    return D

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


# ---- round 5: shards without a sort and without the pair list ---------------------------------------------------------------------
# The pair list of all_pairwise is an enumeration (itertools.combinations / permutations order, CoverAlgorithm.py:166-168), so a
# position IS its pair: rank r takes positions r, r + world, r + 2 world, ... and computes its own pairs from them in closed form.
# Nothing is sorted, nothing of size K is materialised on any rank, every rank knows every other rank's positions without
# exchanging anything, and the gathered message un-interleaves by a transposition.  Balance: a rank samples every row of the
# triangle uniformly (a row's pairs go round the ranks, and the rows start at shifting residues), so the shard costs -- products
# of two song lengths -- agree to a fraction of a percent at any realistic K (tests/test_sharding.py: 1.0002 at 15 000 songs).

def n_pairs(n_songs, symmetric=True):
    n = int(n_songs)
    return n * (n - 1) // 2 if symmetric else n * (n - 1)


def strided_shard(K, world_size, rank):
    """Positions (into the pair enumeration) owned by `rank`."""
    return np.arange(int(rank), int(K), int(world_size), dtype=np.int64)


def pairs_of_positions(n_songs, pos, symmetric=True):
    """(i, j) of the given positions of the enumeration: combinations order (i < j, row after row) when symmetric, else
    permutations order (all j != i, row after row).  int64 array (len(pos), 2)."""
    n = int(n_songs)
    pos = np.asarray(pos, dtype=np.int64)
    if not symmetric:
        i = pos // (n - 1)
        jj = pos - i * (n - 1)
        return np.stack([i, jj + (jj >= i)], axis=1)
    # row i starts at off(i) = i (2 n - i - 1) / 2: invert with a float64 root (exact to within one row at these sizes), then
    # step to the exact row.  In place, one output array: 14 million positions (15 000 songs on 8 ranks) take about a second.
    out = np.empty((len(pos), 2), dtype=np.int64)
    i, j = out[:, 0], out[:, 1]
    b = 2.0 * n - 1.0
    t = pos.astype(np.float64)
    t *= -8.0
    t += b * b
    np.maximum(t, 0.0, out=t)
    np.sqrt(t, out=t)
    np.subtract(b, t, out=t)
    t *= 0.5
    np.floor(t, out=t)
    i[:] = t
    del t
    np.clip(i, 0, max(n - 2, 0), out=i)
    # off(i) into j; rows whose start lies behind the position step back, rows whose successor starts at or before it step on
    np.multiply(i, -1, out=j)
    j += 2 * n - 1
    j *= i
    j >>= 1
    back = j > pos
    i[back] -= 1
    np.multiply(i, -1, out=j)
    j += 2 * n - 2
    j *= i + 1
    j >>= 1                                                  # off(i + 1)
    fwd = j <= pos
    i[fwd] += 1
    np.multiply(i, -1, out=j)
    j += 2 * n - 1
    j *= i
    j >>= 1                                                  # off(i), final
    np.subtract(pos, j, out=j)
    j += i
    j += 1
    return out


def gather_strided(local_scores, K, group=None, force_collective=False):
    """The ONE collective of the path for strided shards: rank r holds the scores of positions r, r + world, ...; returns the
    full length-K vector on every rank.  One all_gather_into_tensor of equal-length messages, then a transposition."""
    if not (dist.is_available() and dist.is_initialized()):
        if force_collective:
            raise RuntimeError("gather_strided(force_collective=True) needs an initialised torch.distributed group")
        return local_scores
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return local_scores
    n_max = -(-int(K) // world)
    send = torch.zeros(n_max, dtype=local_scores.dtype, device=local_scores.device)
    send[:local_scores.numel()] = local_scores
    recv = torch.empty(world * n_max, dtype=local_scores.dtype, device=local_scores.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv.view(world, n_max).t().reshape(-1)[:int(K)]


def fill_matrix(D, scores, symmetric=True):
    """D[i, j] = score of pair (i, j), row after row of the enumeration -- no index arrays of size K."""
    n = D.shape[0]
    scores = np.asarray(scores)
    if symmetric:
        off = 0
        for i in range(n - 1):
            D[i, i + 1:] = scores[off:off + n - 1 - i]
            off += n - 1 - i
    else:
        for i in range(n):
            row = scores[i * (n - 1):(i + 1) * (n - 1)]
            D[i, :i] = row[:i]
            D[i, i + 1:] = row[i:]
    return D


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

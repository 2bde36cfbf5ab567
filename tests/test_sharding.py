"""Pair sharding and the path's single collective, on CPU with the gloo backend (world_size 2 and 3).
The same code runs under "nccl" (RCCL over xGMI) in bench.py and CoverAlgorithm.all_pairwise."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from acoss_amd import sharding, synth


def test_shards_partition_the_pair_list_and_balance_cost():
    rng = np.random.default_rng(0)
    lens = rng.integers(200, 1200, size=60)
    off = np.concatenate([[0], np.cumsum(lens)])
    pairs = synth.all_pairs(60)
    costs = sharding.pair_costs(off, pairs, win=9)
    assert costs[0] == (lens[0] - 8) * (lens[1] - 8)
    for world in (1, 2, 4, 8):
        shards = [sharding.shard_indices(costs, world, r) for r in range(world)]
        allidx = np.sort(np.concatenate(shards))
        assert np.array_equal(allidx, np.arange(len(pairs)))            # a partition
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 1
        loads = np.array([costs[s].sum() for s in shards], dtype=np.float64)
        assert loads.max() / loads.mean() < 1.01                        # ragged lengths balance


def test_world8_shards_balance_on_the_config3_length_distribution():
    """World size 8 (one node of MI355X) on the DA-TACOS-shaped length distribution of BASELINE config 3
    (synth.config3: lengths ~N(520, 120) clipped to [200, 1200]; here 5 000 songs = 12.5 M pairs of the 15 000):
    the eight shards partition the pair list, their sizes differ by at most one and their costs by under 1 %."""
    rng = np.random.default_rng(15000)
    lens = np.clip(rng.normal(520, 120, size=5000), 200, 1200).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    pairs = synth.all_pairs(len(lens))
    costs = sharding.pair_costs(off, pairs, win=9)
    owner = np.full(len(pairs), -1, dtype=np.int8)
    loads, sizes = [], []
    for r in range(8):
        idx = sharding.shard_indices(costs, 8, r)
        assert np.all(owner[idx] == -1)
        owner[idx] = r
        loads.append(float(costs[idx].sum()))
        sizes.append(len(idx))
    assert np.all(owner >= 0)                                               # a partition
    assert max(sizes) - min(sizes) <= 1
    assert max(loads) / min(loads) <= 1.01


def test_scatter_to_matrix_symmetrises_like_the_reference():
    pairs = synth.all_pairs(5)
    scores = np.arange(1, len(pairs) + 1, dtype=np.float64)
    D = sharding.scatter_to_matrix(pairs, scores, 5)
    assert D.dtype == np.float32 and np.array_equal(D, D.T) and np.all(np.diag(D) == 0)
    assert D[0, 1] == 1 and D[3, 4] == len(pairs)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, K, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        costs = (np.arange(K) * 7919) % 1000 + 1
        mine = sharding.shard_indices(costs, world, rank)
        truth = np.sin(np.arange(K)).astype(np.float32)                  # the "score" of pair k
        local = torch.from_numpy(truth[mine])
        full = sharding.gather_scores(local, mine, K)
        assert full.dtype == torch.float32
        assert np.array_equal(full.numpy(), truth), "rank %d gathered wrong scores" % rank
        np.save(os.path.join(tmp, "ok_%d.npy" % rank), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,K", [(2, 101), (3, 64), (2, 1), (8, 1003)])
def test_gather_scores_gloo(tmp_path, world, K):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, K, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok_%d.npy" % r)) for r in range(world))


def test_gather_scores_single_process():
    idx = np.array([4, 0, 2])
    out = sharding.gather_scores(torch.tensor([1.0, 2.0, 3.0]), idx, 6)
    assert out.tolist() == [2.0, 0.0, 3.0, 0.0, 1.0, 0.0]


def _worker_forced(rank, world, port, K, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        truth = np.cos(np.arange(K)).astype(np.float64)
        idx = np.arange(K)[::-1].copy()
        for kw in ({"index_of_rank": lambda r: idx}, {}):
            full = sharding.gather_scores(torch.from_numpy(truth[idx]), idx, K, force_collective=True, **kw)
            assert np.array_equal(full.numpy(), truth)
        np.save(os.path.join(tmp, "ok_forced.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


def test_gather_scores_forced_collective_in_a_one_rank_group(tmp_path):
    """force_collective=True runs the all-gather in a world of one (the mode tests/test_gpu_rccl.py uses under "nccl");
    without a process group it is an error, not a silent shortcut."""
    mp.spawn(_worker_forced, args=(1, _free_port(), 37, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok_forced.npy"))
    with pytest.raises(RuntimeError):
        sharding.gather_scores(torch.tensor([1.0]), np.array([0]), 1, force_collective=True)


# ---- round 5: strided shards (no sort, no pair list) ---------------------------------------------------------------------------------

def test_positions_are_pairs_in_the_reference_orders():
    """pairs_of_positions() inverts the enumerations of CoverAlgorithm.py:166-168 (itertools.combinations / permutations)."""
    import itertools
    for n in (2, 3, 7, 41):
        comb = np.array(list(itertools.combinations(range(n), 2)), dtype=np.int64)
        perm = np.array(list(itertools.permutations(range(n), 2)), dtype=np.int64)
        assert sharding.n_pairs(n, True) == len(comb) and sharding.n_pairs(n, False) == len(perm)
        assert np.array_equal(sharding.pairs_of_positions(n, np.arange(len(comb)), True), comb)
        assert np.array_equal(sharding.pairs_of_positions(n, np.arange(len(perm)), False), perm)
    # sizes where the float64 root is off by a row: the exact row is recovered
    for n in (100000, 1 << 20):
        K = sharding.n_pairs(n)
        rng = np.random.default_rng(n)
        pos = np.unique(np.concatenate([rng.integers(0, K, 100000), [0, 1, n - 2, n - 1, n, K - 2, K - 1]]))
        pr = sharding.pairs_of_positions(n, pos, True)
        assert np.all(pr[:, 0] < pr[:, 1]) and pr[:, 1].max() < n
        assert np.array_equal(pr[:, 0] * (2 * n - pr[:, 0] - 1) // 2 + pr[:, 1] - pr[:, 0] - 1, pos)


def test_strided_shards_of_the_config3_job_are_cheap_and_balanced():
    """BASELINE config 3 at full size: 15 000 songs = 112.5 M pairs on 8 ranks.  A rank's shard -- its positions and its pairs, built
    batch by batch as all_pairwise builds them -- takes seconds of host time and well under 1.5 GB, with no array of size K; the
    shards partition the enumeration and their costs (cells: products of the two song lengths) agree within 1 %."""
    import time
    import tracemalloc
    n, world = 15000, 8
    K = sharding.n_pairs(n)
    rng = np.random.default_rng(15000)
    cells = np.clip(rng.normal(520, 120, size=n), 200, 1200).astype(np.int64) - 8
    loads, counts = [], []
    for rank in range(world):
        n_mine = len(range(rank, K, world))
        if rank == 3:
            tracemalloc.start()
        load, t_build = 0, 0.0
        for lo in range(0, n_mine, 1 << 18):
            t0 = time.perf_counter()
            pos = rank + world * np.arange(lo, min(lo + (1 << 18), n_mine), dtype=np.int64)
            pairs = sharding.pairs_of_positions(n, pos, True)
            t_build += time.perf_counter() - t0
            load += int((cells[pairs[:, 0]] * cells[pairs[:, 1]]).sum())
            if lo == 0:
                assert np.array_equal(pairs[:, 0] * (2 * n - pairs[:, 0] - 1) // 2 + pairs[:, 1] - pairs[:, 0] - 1, pos)
        if rank == 3:
            peak = tracemalloc.get_traced_memory()[1]
            tracemalloc.stop()
            assert t_build <= 3.0, t_build                       # the whole shard of a rank: positions -> pairs
            assert peak <= 1.5 * 2 ** 30, peak
        loads.append(load)
        counts.append(n_mine)
    assert sum(counts) == K and max(counts) - min(counts) <= 1           # positions r, r + world, ...: a partition by construction
    assert max(loads) / min(loads) <= 1.01


def _worker_strided(rank, world, port, K, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        truth = np.sin(np.arange(K)).astype(np.float64)
        mine = sharding.strided_shard(K, world, rank)
        full = sharding.gather_strided(torch.from_numpy(truth[mine]), K, force_collective=True)
        assert np.array_equal(full.numpy(), truth), "rank %d gathered wrong scores" % rank
        np.save(os.path.join(tmp, "ok_%d.npy" % rank), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,K", [(2, 101), (3, 64), (2, 1), (8, 1003), (1, 5)])
def test_gather_strided_gloo(tmp_path, world, K):
    mp.spawn(_worker_strided, args=(world, _free_port(), K, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok_%d.npy" % r)) for r in range(world))


def test_fill_matrix_equals_the_index_form():
    for n in (1, 2, 5, 33):
        for symmetric in (True, False):
            K = sharding.n_pairs(n, symmetric)
            scores = np.arange(1, K + 1, dtype=np.float64)
            D = sharding.fill_matrix(np.zeros((n, n), dtype=np.float32), scores, symmetric)
            pairs = sharding.pairs_of_positions(n, np.arange(K), symmetric) if K else np.zeros((0, 2), dtype=np.int64)
            want = np.zeros((n, n), dtype=np.float32)
            want[pairs[:, 0], pairs[:, 1]] = scores
            assert np.array_equal(D, want)

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

"""The one-rank RCCL process of tests/test_gpu_rccl.py: forms an "nccl" (= RCCL on ROCm) process group of world size 1 on GPU 0
BEFORE any other GPU call and runs the path's collectives on device tensors -- sharding.gather_scores (both message forms),
the all_reduce(MAX) of bench.py's timing, a barrier, and the sharded all_pairwise driver -- writing what it saw as JSON."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, port = sys.argv[1], sys.argv[2]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)       # first GPU call of the process
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    from acoss_amd import sharding
    rng = np.random.default_rng(4)
    K = 5000
    local = torch.from_numpy(rng.random(K)).to(dev)
    idx = np.arange(K)
    a = sharding.gather_scores(local, idx, K, index_of_rank=lambda r: idx, force_collective=True)
    b = sharding.gather_scores(local, idx, K, force_collective=True)           # positions ride along
    perm = rng.permutation(K)
    c = sharding.gather_scores(local, perm, K, force_collective=True)
    res["gather_scores_index_of_rank_equal"] = bool(torch.equal(a, local) and a.is_cuda)
    res["gather_scores_with_positions_equal"] = bool(torch.equal(b, local) and b.is_cuda)
    res["gather_scores_permuted_equal"] = bool(torch.equal(c[torch.as_tensor(perm, device=dev)], local))
    t = torch.tensor([3.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    res["all_reduce_max"] = float(t.item())
    g = torch.empty(K, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(g, local.float())
    res["all_gather_into_tensor_equal"] = bool(torch.equal(g, local.float()))
    dist.barrier()
    with open("/proc/self/maps") as fh:
        libs = sorted(set(os.path.basename(ln.split()[-1]) for ln in fh if "rccl" in ln or "nccl" in ln.lower()))
    res["rccl_libraries_mapped"] = libs
    # the sharded driver (CoverAlgorithm.all_pairwise) through the same communicator: scores must equal the un-sharded call's
    from acoss_amd import engine, synth
    from acoss_amd.Serra09 import Serra09
    corpus = synth.make_corpus(4, 3, seed=11, lengths=lambda r: r.integers(90, 200))
    os.chdir(os.path.dirname(out_path))
    os.environ["ACOSS_FORCE_COLLECTIVE"] = "1"
    alg = Serra09(corpus, shortname="rccl1", do_memmaps=False, cachedir=os.path.join(os.path.dirname(out_path), "cache"))
    alg.all_pairwise(symmetric=True)
    pairs = synth.all_pairs(corpus.n_songs)
    dc = engine.DeviceCorpus(corpus.feats, corpus.frame_off, gchroma=corpus.gchroma)
    want = engine.serra09_scores(dc, pairs)
    res["all_pairwise_through_rccl_equal"] = bool(
        np.array_equal(alg.Ds["chroma_qmax"][pairs[:, 0], pairs[:, 1]], want["qmax"].astype(np.float32))
        and np.array_equal(alg.Ds["chroma_dmax"][pairs[:, 0], pairs[:, 1]], want["dmax"].astype(np.float32))
        and float(np.max(want["qmax"])) > 0.0)
    dist.destroy_process_group()
    with open(out_path, "w") as fh:
        json.dump(res, fh)


if __name__ == "__main__":
    main()

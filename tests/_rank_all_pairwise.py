"""One rank of the sharded-all_pairwise test (tests/test_gpu_sharded.py): started twice by torch.distributed.run, both
ranks on GPU 0 (the gloo rehearsal mode of a one-GPU box; with one GPU per rank the backend is "nccl" = RCCL)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    import torch
    import torch.distributed as dist
    backend = os.environ.get("ACOSS_TEST_DIST_BACKEND", "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist.init_process_group(backend)
    from acoss_amd import synth
    from acoss_amd.Serra09 import Serra09
    corpus = synth.config1()
    os.chdir(out_dir)
    alg = Serra09(corpus, shortname="config1_r%d" % dist.get_rank(), do_memmaps=True, cachedir=os.path.join(out_dir, "cache"))
    alg.all_pairwise(symmetric=True)
    np.savez(os.path.join(out_dir, "Ds_rank%d.npz" % dist.get_rank()),
             chroma_qmax=np.asarray(alg.Ds["chroma_qmax"]), chroma_dmax=np.asarray(alg.Ds["chroma_dmax"]),
             world=np.array([dist.get_world_size()]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

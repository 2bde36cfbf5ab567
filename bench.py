#!/usr/bin/env python
"""
bench.py -- pair-scores/sec of the Serra09 chroma_qmax hot path on MI355X.

Workload (BASELINE.json configs[1]): synthetic corpus of 1000 songs x 1000 frames x 12-bin HPCP
(float64, 250 cliques x 4 versions, seed 20260), Serra09 parameters m=9, kappa=0.095, OTI on.
One "step" = one batch of `--pairs-per-step` song pairs per GPU drawn from that corpus's
499 500-pair list, taken through the whole chain (OTI -> CSM -> sliding window -> mutual kNN
binarisation -> qmax -> /(M+N)), features already resident in HBM.  With N GPUs every rank works
on its own shard of the pair list (weak scaling, no data-path collective) and one all-gather of
the score vectors closes the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` describes the CSM kernel (the HBM-bound kernel the
north star grades), timed live with HIP events on the launch stream inside the timed region;
`cpu_baseline` is the CPU oracle's same chain timed on the host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs-per-step", type=int, default=2048)
    ap.add_argument("--songs", type=int, default=1000)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--path", choices=("staged",), default="staged")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="CPU baseline sample size (0 = auto)")
    return ap.parse_args()


class StagedRunner(object):
    """OTI -> CSM -> sliding -> binarise -> qmax through the stage kernels, all buffers
    preallocated, every launch on torch's current stream."""

    def __init__(self, corpus, batches, m, kappa):
        import torch
        from acoss_amd import engine
        self.engine, self.torch = engine, torch
        self.corpus, self.m, self.kappa = corpus, m, kappa
        dev = corpus.device
        tc = max(b.total_csm for b in batches)
        tr = max(b.total_crp for b in batches)
        self.C = torch.empty(tc, dtype=corpus.feats.dtype, device=dev)
        self.S = torch.empty(tr, dtype=torch.float64, device=dev)
        self.B = torch.zeros(tr, dtype=torch.uint8, device=dev)
        need = max(int(engine._lib.load().acoss_binarize_work_bytes(b.K, b.max_nx, b.max_ny, m)) for b in batches)
        self.work = torch.empty(need, dtype=torch.uint8, device=dev)
        self.plans = []
        for b in batches:
            mats, _ = b.mats()
            self.plans.append((b, mats, engine.to_device_bytes(mats, dev), int(mats["cols"].max())))
        self.csm_bytes = [float(np.sum(corpus.feats.element_size() *
                                       (b.descs["nx"].astype(np.float64) * b.descs["ny"] +
                                        corpus.d * (b.descs["nx"].astype(np.float64) + b.descs["ny"]))))
                          for b in batches]

    def step(self, i, scores_out, ev=None):
        e = self.engine
        b, mats, mats_dev, max_cols = self.plans[i]
        e.oti(self.corpus, b)
        if ev is not None:
            ev[0].record()
        e.csm(self.corpus, b, out=self.C)
        if ev is not None:
            ev[1].record()
        e.sliding(self.C, b, out=self.S)
        e.binarize(self.S, b, self.kappa, True, out=self.B, work=self.work)
        e.align("qmax", self.B, mats, mats_dev=mats_dev, max_cols=max_cols, scores=scores_out)


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from acoss_amd import engine, sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    m, kappa = 9, 0.095
    corpus_h = synth.config2(n_songs=args.songs, n_frames=args.frames)
    corpus = engine.DeviceCorpus(corpus_h.feats, corpus_h.frame_off, gchroma=corpus_h.gchroma, device=dev)
    all_pairs = synth.all_pairs(corpus_h.n_songs)
    mine = sharding.shard_indices(sharding.pair_costs(corpus_h.frame_off, all_pairs, m), world, rank)
    P = args.pairs_per_step
    n_steps = args.warmup + args.steps
    # deterministic walk over this rank's shard, wrapping around if the run is longer than the job
    step_idx = [mine[(np.arange(P) + s * P) % len(mine)] for s in range(n_steps)]
    batches = [engine.PairBatch(corpus.frame_off, all_pairs[ix], m, dev) for ix in step_idx]
    runner = StagedRunner(corpus, batches, m, kappa)
    scores = torch.zeros(n_steps, P, dtype=torch.float32, device=dev)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_steps)]

    def barrier():
        if world > 1:
            dist.barrier()

    for s in range(args.warmup):
        runner.step(s, scores[s])
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.warmup, n_steps):
        runner.step(s, scores[s], events[s])
    # the path's only collective: gather every rank's timed scores (RCCL all-gather over xGMI)
    timed_idx = np.concatenate(step_idx[args.warmup:])
    local = scores[args.warmup:].reshape(-1)
    if world > 1:
        gathered = torch.empty(world * local.numel(), dtype=local.dtype, device=dev)
        dist.all_gather_into_tensor(gathered, local)
    else:
        gathered = local
    host_scores = gathered.cpu().numpy()          # D2H of the results is inside the timed region
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_pairs = world * args.steps * P
    value = total_pairs / elapsed
    csm_ms = np.array([events[s][0].elapsed_time(events[s][1]) for s in range(args.warmup, n_steps)])
    csm_bytes = float(np.mean(runner.csm_bytes[args.warmup:]))
    achieved = csm_bytes / (float(np.mean(csm_ms)) * 1e-3) / 1e9

    out = {
        "metric": "pair-scores/sec (Serra09 qmax, 1000-frame HPCP)",
        "value": round(value, 1), "unit": "pair-scores/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "synthetic %d songs x %d frames x 12-bin HPCP (f64), Serra09 chroma_qmax "
                               "m=9 kappa=0.095 OTI, %d pairs/step/GPU of the %d-pair job"
                               % (args.songs, args.frames, P, len(all_pairs)),
                   "path": args.path, "pairs_per_step_per_gpu": P,
                   "parallelism": "pair-shard x%d, one all-gather" % world},
        "roofline": {"kernel": "csm_kernel<double,12> (materialising CSM, CRPUtils.py:67)", "bound": "hbm",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "bytes_per_launch": csm_bytes, "avg_launch_ms": round(float(np.mean(csm_ms)), 4)},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        threads = max(1, min(os.cpu_count() or 1, 16))
        n_cpu = args.cpu_pairs or 16 * threads
        sample = all_pairs[timed_idx[:n_cpu]]
        oracle.serra09_pairs(corpus_h.feats, corpus_h.frame_off, corpus_h.gchroma, sample[:threads],
                             m=m, kappa=kappa, nthreads=threads, want_dmax=False)      # warm the scratch
        t0 = time.perf_counter()
        q_cpu, _, used = oracle.serra09_pairs(corpus_h.feats, corpus_h.frame_off, corpus_h.gchroma, sample,
                                              m=m, kappa=kappa, nthreads=threads, want_dmax=False)
        cpu_s = time.perf_counter() - t0
        denom = 2.0 * (args.frames - m + 1)
        gpu_q = host_scores[:n_cpu].astype(np.float64) / denom
        out["cpu_baseline"] = {"value": round(n_cpu / cpu_s, 2), "unit": "pair-scores/s", "cores": int(used),
                               "kind": "port",
                               "sample": "%d pairs of the same workload through oracle/acoss_oracle.c "
                                         "(OpenMP over pairs, %.1f s)" % (n_cpu, cpu_s)}
        out["parity"] = {"checked_pairs": int(n_cpu), "identical": bool(np.array_equal(gpu_q, q_cpu))}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

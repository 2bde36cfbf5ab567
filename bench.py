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

Prints ONE JSON line on rank 0.  `roofline` describes the dominant HBM-bound kernel of the timed
path -- the cross-similarity kernel: fused with the sliding window in the default fast path,
materialising the CSM in `--path staged` -- timed live with HIP events on the launch stream inside
the timed region (`stage_ms` has every stage).  `roofline_csm_materialising` is the stand-alone
get_csm kernel on the same batch, measured outside the timed region.  `cpu_baseline` is the CPU
oracle's same chain timed on the host cores (rank 0, N=1 only); `parity` says whether the GPU scores
of the sampled pairs are identical to the oracle's.
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
    ap.add_argument("--pairs-per-step", type=int, default=4096,
                    help="pairs per launch batch and GPU (4096 x 3.9 MB of key high words = 16 GB of the 288 GB)")
    ap.add_argument("--songs", type=int, default=1000)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--path", choices=("fast32", "fast", "fast_f64", "fused", "staged"), default="fast32",
                    help="fast32 (the product path): CSM + sliding window in float32 on the matrix cores, float32 keys, "
                         "selection with exact float64 refinement of the rows / columns inside the error band (results "
                         "identical to float64); fast: the same chain with float64 windowed sums (key high words); "
                         "fast_f64: a float64 matrix in between; fused: masks from the band kernel, no matrix in HBM "
                         "(csrc/band_kernels.hip); staged: one kernel per reference function")
    ap.add_argument("--overlap", action="store_true",
                    help="fast path: run the alignment sweep of batch b on a second HIP stream while the main "
                         "stream computes batch b+1 (measured: no gain, the sweep's registers/LDS crowd the CUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="CPU baseline sample size (0 = auto)")
    return ap.parse_args()


STAGES = {"fused": ["oti", "pack_x", "crp", "mask_bits", "qmax_bits"],
          "fast": ["oti", "pack_x", "crp", "mask_bits", "qmax_bits"],
          "fast32": ["oti", "pack_x", "crp", "mask_bits", "qmax_bits"],
          "fast_f64": ["oti", "pack_x", "crp", "mask_bits", "qmax_bits"],
          "staged": ["oti", "csm", "sliding", "binarize", "qmax"]}


class Runner(object):
    """One step, fast path: OTI -> pack_x -> crp (fused CSM + sliding window, squared) -> mask_bits (row and
    column kNN selection emitting bit vectors by ballot, transposed and ANDed into a 124 KB bit mask per pair)
    -> qmax from the bits.  Staged path: OTI -> CSM -> sliding ->
    binarise (thresholds + mask) -> qmax, one kernel per reference function.  All buffers are
    preallocated and every launch goes to torch's current stream; HIP events between the stages give
    per-stage times of the timed steps."""

    def __init__(self, corpus, batches, m, kappa, path, overlap=True):
        import torch
        from acoss_amd import engine
        self.engine, self.torch, self.path = engine, torch, path
        # fast path: the alignment sweep (latency-bound: 992 serial row steps, ~15 % VALU) of batch b runs
        # on a second HIP stream while the main stream already computes batch b+1 (two sets of T/threshold
        # buffers, events both ways)
        self.overlap = False
        # "fast": T leaves the strip kernel as two uint32 planes (key high / low words) and the selections read
        # only the high-word plane; "fast_f64": T as float64, selections read 8 bytes per element
        # "fast32": the strip kernel computes a float32 approximation (float32 keys); rows / columns whose k-th smallest
        # has another value inside the error band are finished in float64: identical masks and scores
        self.p32 = path == "fast32"
        if self.p32:
            path = self.path = "fast"
        self.planar = path == "fast" and all(engine.planar_supported(corpus, b) for b in batches)
        self.p32 = self.p32 and self.planar
        if path == "fast_f64":
            path = self.path = "fast"
        self.corpus, self.m, self.kappa = corpus, m, kappa
        dev = corpus.device
        lib = engine._lib.load()
        tr = max(b.total_crp for b in batches)
        # the big intermediate: float64 sums (8 B / cell), or their key high words on the fast path (4 B / cell)
        s_elems = (tr // 2 + 32) if self.planar else (tr + 32)
        self.S = torch.empty(s_elems, dtype=torch.float64, device=dev)
        self.B = torch.zeros(tr, dtype=torch.uint8, device=dev) if path == "staged" else None
        if path == "staged":
            self.C = torch.empty(max(b.total_csm for b in batches), dtype=corpus.feats.dtype, device=dev)
        else:
            self.xp = torch.empty(max(int(lib.acoss_xpack_elems(b.K, b.max_nx)) for b in batches),
                                  dtype=corpus.feats.dtype, device=dev)
        if path == "fast":
            need = max(int(lib.acoss_mask_bits_work_bytes(b.K, b.max_nx, b.max_ny, m)) for b in batches)
            self.bits = torch.zeros(max(b.K * (b.max_nx - m + 1) * 16 for b in batches), dtype=torch.int64, device=dev)
        else:
            need = max(int(lib.acoss_binarize_work_bytes(b.K, b.max_nx, b.max_ny, m)) for b in batches)
        self.work = torch.empty(need, dtype=torch.uint8, device=dev)
        if self.p32:
            self.xp32 = torch.empty(self.xp.numel(), dtype=torch.float32, device=dev)
            self.bands = [engine.planar32_band(corpus, b) for b in batches]
            engine.float32_copy(corpus)
        # --overlap: the alignment sweep (latency-bound: 992 serial row steps, one wave per pair, 124 KB of mask per
        # pair) of batch b runs on a second HIP stream while the main stream already builds batch b + 1 (two mask
        # buffers, events both ways)
        self.overlap = bool(overlap) and self.planar
        if self.overlap:
            # two copies of the high-word matrix: the strip kernel of batch b + 1 (LDS / barrier-bound) runs on the main
            # stream while the selection + alignment kernels of batch b (HBM / latency-bound) run on the side stream
            self.S2 = [self.S, torch.empty(s_elems, dtype=torch.float64, device=dev)]
            self.side = torch.cuda.Stream(device=dev)
            self.ready = [torch.cuda.Event(), torch.cuda.Event()]
            self.free = [torch.cuda.Event(), torch.cuda.Event()]
        # The strip kernel's time depends on which allocation it writes (3.9 vs 4.4 ms for two
        # 16 GB buffers in one process, any offset inside either gives the same time).  Try a few placements for
        # the big intermediate once, before anything is timed, and keep the fastest.
        self.placement_ms = None
        if self.planar and not os.environ.get("ACOSS_BENCH_NO_PLACEMENT"):
            b0 = batches[0]
            engine.oti(corpus, b0)
            engine.pack_x(corpus, b0, out=self.xp)
            cands, times = [self.S], []
            try:
                for _ in range(2):
                    cands.append(torch.empty(s_elems, dtype=torch.float64, device=dev))
            except RuntimeError:
                pass
            for buf in cands:
                planes = buf.view(torch.int32)[:engine.planar_elems(b0)]
                best = 1e9
                for rep in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if self.p32:
                        engine.crp_planar32(corpus, b0, engine.pack_x32(corpus, b0, out=self.xp32), out=planes)
                    else:
                        engine.crp_planar(corpus, b0, self.xp, out=planes)
                    e1.record()
                    torch.cuda.synchronize()
                    if rep:
                        best = min(best, e0.elapsed_time(e1))
                times.append(best)
            self.S = cands[int(np.argmin(times))]
            if self.overlap:
                self.S2[0] = self.S
            self.placement_ms = [round(t, 3) for t in times]
            del cands
            torch.cuda.empty_cache()
        self.plans = []
        for b in batches:
            mats, _ = b.mats()
            self.plans.append((b, mats, engine.to_device_bytes(mats, dev), int(mats["cols"].max())))
        es = corpus.feats.element_size()
        nx = [b.descs["nx"].astype(np.float64) for b in batches]
        ny = [b.descs["ny"].astype(np.float64) for b in batches]
        # algorithmic bytes per launch of the cross-similarity kernel of each path
        self.csm_bytes = [float(np.sum(es * (x * y + corpus.d * (x + y)))) for x, y in zip(nx, ny)]
        # the fast path's strip kernel writes 4 bytes per cell (key high words), the float64 form 8
        cell = 4.0 if self.planar else 8.0
        fes = 4 if self.p32 else es
        self.crp_bytes = [float(np.sum(cell * (x - m + 1) * (y - m + 1) + fes * corpus.d * (x + y))) for x, y in zip(nx, ny)]

    def step(self, i, scores_out, ev=None):
        e = self.engine
        b, mats, mats_dev, max_cols = self.plans[i]

        def mark(k):
            if ev is not None:
                ev[k].record()
        if self.overlap:
            torch = self.torch
            slot = i & 1
            main = torch.cuda.current_stream()
            main.wait_event(self.free[slot])          # the selection that last read this copy has finished
            mark(0)
            e.oti(self.corpus, b)
            mark(1)
            e.pack_x(self.corpus, b, out=self.xp)
            mark(2)
            planes = self.S2[slot].view(torch.int32)[:e.planar_elems(b)]
            e.crp_planar(self.corpus, b, self.xp, out=planes)
            mark(3)
            self.ready[slot].record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ready[slot])
                e.mask_bits_planar(planes, self.corpus, b, self.kappa, True, out=self.bits, work=self.work)
                mark(4)
                e.align_bits("qmax", self.bits, b, scores=scores_out)
                mark(5)
                self.free[slot].record(self.side)
            return
        mark(0)
        e.oti(self.corpus, b)
        mark(1)
        if self.path == "fast":
            if self.p32:
                e.pack_x32(self.corpus, b, out=self.xp32)
            else:
                e.pack_x(self.corpus, b, out=self.xp)
            mark(2)
            if self.p32:
                planes = self.S.view(self.torch.int32)[:e.planar_elems(b)]
                e.crp_planar32(self.corpus, b, self.xp32, out=planes)
            elif self.planar:
                planes = self.S.view(self.torch.int32)[:e.planar_elems(b)]
                e.crp_planar(self.corpus, b, self.xp, out=planes)
            else:
                e.crp(self.corpus, b, self.xp, sqrt_out=False, out=self.S)
        else:
            e.csm(self.corpus, b, out=self.C)
            mark(2)
            e.sliding(self.C, b, out=self.S)
        mark(3)
        if self.path == "fast":
            if self.p32:
                e.mask_bits_planar32(planes, self.bands[i], self.corpus, b, self.kappa, True, out=self.bits, work=self.work)
            elif self.planar:
                e.mask_bits_planar(planes, self.corpus, b, self.kappa, True, out=self.bits, work=self.work)
            else:
                e.mask_bits(self.S, b, self.kappa, True, out=self.bits, work=self.work)
            mark(4)
            e.align_bits("qmax", self.bits, b, scores=scores_out)
        else:
            e.binarize(self.S, b, self.kappa, True, out=self.B, work=self.work)
            mark(4)
            e.align("qmax", self.B, mats, mats_dev=mats_dev, max_cols=max_cols, scores=scores_out)
        mark(5)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) through torch.distributed.run
    as a CHILD process -- nothing in this process has touched the GPU yet -- and pass rank 0's JSON line through."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    import torch
    import torch.distributed as dist
    from acoss_amd import engine, sharding, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    # one rank per GPU over RCCL (backend "nccl").  ACOSS_BENCH_DIST_BACKEND=gloo is a rehearsal mode for boxes with
    # fewer GPUs than ranks: ranks share the visible devices and the gather goes through host memory.
    backend = os.environ.get("ACOSS_BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    m, kappa = 9, 0.095
    corpus_h = synth.config2(n_songs=args.songs, n_frames=args.frames)
    corpus = engine.DeviceCorpus(corpus_h.feats, corpus_h.frame_off, gchroma=corpus_h.gchroma, device=dev)
    all_pairs = synth.all_pairs(corpus_h.n_songs)
    mine = sharding.shard_indices(sharding.pair_costs(corpus_h.frame_off, all_pairs, m), world, rank)
    P = args.pairs_per_step
    n_steps = args.warmup + args.steps
    # deterministic walk over this rank's shard, wrapping around if the run is longer than the job
    step_idx = [mine[(np.arange(P) + s * P) % len(mine)] for s in range(n_steps)]
    batches = [engine.PairBatch(corpus.frame_off, all_pairs[ix], m, dev, pitch_align=32 if args.path in ("fast", "fast32") else 16) for ix in step_idx]
    runner = Runner(corpus, batches, m, kappa, args.path, overlap=args.overlap)
    scores = torch.zeros(n_steps, P, dtype=torch.float32, device=dev)
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(n_steps)]

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
    if runner.overlap:
        torch.cuda.current_stream().wait_stream(runner.side)
    # the path's only collective: gather every rank's timed scores (RCCL all-gather over xGMI)
    timed_idx = np.concatenate(step_idx[args.warmup:])
    local = scores[args.warmup:].reshape(-1)
    if world > 1:
        src = local if backend == "nccl" else local.cpu()
        gathered = torch.empty(world * local.numel(), dtype=local.dtype, device=src.device)
        dist.all_gather_into_tensor(gathered, src)
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
    names = STAGES[args.path]
    stage_ms = {names[k]: float(np.mean([events[s][k].elapsed_time(events[s][k + 1])
                                          for s in range(args.warmup, n_steps)])) for k in range(5)}
    # dominant HBM-bound kernel of the path: the cross-similarity kernel (fused with the sliding
    # window in the fast path, materialising the CSM in the staged path)
    if args.path in ("fast", "fast32", "fast_f64"):
        kname, kms, kbytes = "crp_strip_kernel<12,9> (CRPUtils.py:67 + :24 fused, f64 MFMA)", stage_ms["crp"], runner.crp_bytes
    else:
        kname, kms, kbytes = "csm_kernel<double,12> (CRPUtils.py:67)", stage_ms["csm"], runner.csm_bytes
    kbytes = float(np.mean(kbytes[args.warmup:]))
    achieved = kbytes / (kms * 1e-3) / 1e9

    # HBM bytes per launch of that kernel from the committed PMC passes (profiles/README.md: WRITE_SIZE exact,
    # FETCH_SIZE x2 on gfx950) -- only quoted when the profile was taken on this very workload
    traffic, traffic_src = None, None
    pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_final_pmc.json")
    pmc_key = {"fast": "crp_strip_kernel<12, 9, false, 0, false, true>"}.get(args.path)
    if pmc_key and runner.planar and args.frames == 1000 and os.path.exists(pmc_file):
        with open(pmc_file) as fh:
            doc = json.load(fh)
        c = doc.get(pmc_key)
        if c and "hbm_write_GB" in c and doc.get("_pairs_per_step") == P:
            traffic = round((c["hbm_write_GB"] + c["hbm_fetch_GB_x2_corrected"]) * 1e9)
            traffic_src = "profiles/r01_final_pmc.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes on this workload)"
    if runner.planar:
        kname = "crp_strip_kernel<12,9,planar> (CRPUtils.py:67 + :24 fused, f64 MFMA, key high words out: 4 B / cell)"
    if runner.p32:
        kname = "crp_strip32_kernel<12> (CRPUtils.py:67 + :24 fused, f32 MFMA approximation, float32 keys out: 4 B / cell; exact f64 refinement in select_fix_planar_kernel)"
    out = {
        "metric": "pair-scores/sec (Serra09 qmax, 1000-frame HPCP)",
        "value": round(value, 1), "unit": "pair-scores/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 filter + f64 exact refinement (results identical to f64)" if runner.p32 else "f64",
        "data": "synthetic",
        "config": {"workload": "synthetic %d songs x %d frames x 12-bin HPCP (f64), Serra09 chroma_qmax "
                               "m=9 kappa=0.095 OTI, %d pairs/step/GPU of the %d-pair job"
                               % (args.songs, args.frames, P, len(all_pairs)),
                   "path": args.path, "pairs_per_step_per_gpu": P, "overlap_alignment_stream": bool(runner.overlap),
                   "output_placement_probe_ms": runner.placement_ms,
                   "parallelism": "pair-shard x%d, one all-gather" % world},
        "roofline": {"kernel": kname, "bound": "hbm",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "bytes_per_launch": kbytes, "avg_launch_ms": round(kms, 4)},
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
    }
    if rank == 0 and args.path in ("fast", "fast32", "fast_f64"):
        # get_csm as an API (the kernel the north star names) on the same batch, outside the timed
        # region, reported beside the path's own dominant kernel: the plain VALU kernel and the
        # persistent matrix-core strip kernel (bit-identical outputs)
        b = batches[-1]
        C = torch.empty(b.total_csm, dtype=corpus.feats.dtype, device=dev)
        cb = runner.csm_bytes[-1]

        def time_kernel(fn):
            ms = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
            return float(np.median(ms[1:]))
        xp = engine.pack_x(corpus, b, out=runner.xp)
        for key, kname2, fn in (("roofline_csm_materialising", "crp_strip_kernel<12,1,sqrt> as get_csm (CRPUtils.py:67), not on the fast path",
                                 lambda: engine.csm_strip(corpus, b, xp, out=C)),
                                ("roofline_csm_valu", "csm_kernel<double,12> (CRPUtils.py:67), not on the fast path",
                                 lambda: engine.csm(corpus, b, out=C))):
            cms = time_kernel(fn)
            out[key] = {"kernel": kname2, "bound": "hbm", "achieved": round(cb / cms / 1e6, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(cb / cms / 1e6 / HBM_PEAK_GBS, 4), "bytes_per_launch": cb,
                        "avg_launch_ms": round(cms, 4)}
        # context for those fractions: what a plain streaming store of the same byte count reaches on this GPU
        # (torch fill_ into the same buffer; 33.5 GB in one launch runs at ~4.7 TB/s, 16 GB at ~6.2 TB/s)
        fms = time_kernel(lambda: C.fill_(1.0))
        fb = C.numel() * C.element_size()
        out["hbm_write_ceiling"] = {"kernel": "torch fill_ of the CSM buffer, %d bytes (plain streaming stores)" % fb,
                                    "achieved": round(fb / fms / 1e6, 1), "unit": "GB/s", "avg_launch_ms": round(fms, 4)}
        del C
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        threads = max(1, min(os.cpu_count() or 1, 16))
        n_cpu = args.cpu_pairs or min(512 * threads, len(timed_idx))
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
        # beside it: the reference's own SequenceAlignment.c (compiled in place by oracle/Makefile into oracle/_ref,
        # -Ofast as in its setup.py) on the alignment step alone -- qmax_c over the same masks, one call per pair,
        # spread over the host threads (ctypes releases the GIL)
        ref = oracle.ref_lib("Ofast")
        if ref is not None:
            from concurrent.futures import ThreadPoolExecutor
            n_dp = min(8 * threads, n_cpu)
            masks = []
            for i, j in sample[:n_dp]:
                X, Y = corpus_h.song(int(i)), corpus_h.song(int(j))
                S = oracle.sliding_csm(oracle.get_csm(X, Y, oracle.get_oti(corpus_h.gchroma[i], corpus_h.gchroma[j])), m)
                masks.append(np.ascontiguousarray(oracle.csm_to_binary_mutual(S, kappa).flatten()))
            Mn = args.frames - m + 1

            def one(Bf):
                D = np.zeros(Mn * Mn, dtype=np.float32)
                return float(ref.qmax_c(oracle._u(Bf), oracle._f(D), Mn, Mn))
            reps = 4
            t0 = time.perf_counter()
            with ThreadPoolExecutor(threads) as ex:
                for _ in range(reps):
                    q_ref = list(ex.map(one, masks))
            dp_s = time.perf_counter() - t0
            out["cpu_baseline_alignment_only"] = {
                "value": round(reps * n_dp / dp_s, 1), "unit": "qmax_c calls/s", "cores": threads, "kind": "reference",
                "sample": "%d x %d calls of the reference's qmax_c (SequenceAlignment.c:113, -Ofast) on %dx%d masks, D zeroed per "
                          "call" % (reps, n_dp, Mn, Mn),
                "identical_to_gpu": bool(np.array_equal(np.array(q_ref) / denom, gpu_q[:n_dp]))}
    if rank == 0 and world == 1 and args.path == "fast" and runner.planar and not os.environ.get("ACOSS_BENCH_NO_FAST32"):
        # beside the float64 line above: the same steps with the float32-filter form of the strip kernel (`--path fast32`);
        # its scores must equal the float64 path's on every pair of the timed steps
        last = scores[args.warmup:].clone()
        del runner, events
        engine.release_scratch()
        r32 = Runner(corpus, batches, m, kappa, "fast32", overlap=False)
        s32 = torch.zeros(n_steps, P, dtype=torch.float32, device=dev)
        ev32 = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(n_steps)]
        for s_ in range(args.warmup):
            r32.step(s_, s32[s_])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s_ in range(args.warmup, n_steps):
            r32.step(s_, s32[s_], ev32[s_])
        h32 = s32[args.warmup:].reshape(-1).cpu().numpy()
        torch.cuda.synchronize()
        el32 = time.perf_counter() - t0
        names = STAGES["fast32"]
        out["fast32"] = {"value": round(args.steps * P / el32, 1), "unit": "pair-scores/s", "ms_per_step": round(1e3 * el32 / args.steps, 3),
                         "dtype": "f32 filter + f64 exact refinement",
                         "stage_ms": {names[k]: round(float(np.mean([ev32[s_][k].elapsed_time(ev32[s_][k + 1])
                                                                       for s_ in range(args.warmup, n_steps)])), 4) for k in range(5)},
                         "roofline": {"kernel": "crp_strip32_kernel<12> (CRPUtils.py:67 + :24 fused, f32 MFMA, float32 keys out: 4 B / cell)",
                                      "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "bytes_per_launch": float(np.mean(r32.crp_bytes[args.warmup:]))},
                         "scores_identical_to_f64_path": bool(torch.equal(s32[args.warmup:], last)),
                         "note": "crp_strip32_kernel + error-band check + float64 refinement (DESIGN.md section 4); same steps, same pairs"}
    if rank == 0 and "fast32" in out:
        rf = out["fast32"]["roofline"]
        rf["avg_launch_ms"] = out["fast32"]["stage_ms"]["crp"]
        rf["achieved"] = round(rf["bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9, 1)
        rf["frac"] = round(rf["achieved"] / HBM_PEAK_GBS, 4)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""
Mirror of the reference's plugin base class (benchmarking/CoverAlgorithm.py:12-418) for the
MI355X scoring path: same constructor, attributes, overridable hooks and drivers, so a plugin
written against the reference (`load_features(i)`, `similarity(idxs)`) keeps working, while

  * all_pairwise() hands the pair list to similarity() in large batches (one GPU launch chain
    per batch) instead of one pair at a time (CoverAlgorithm.py:176-177) or 45 joblib chunks
    (:169-173); under torch.distributed the pair list is sharded over the ranks and the score
    vectors are all-gathered once (sharding.py);
  * getEvalStatistics() returns exactly the reference's numbers but replaces its O(N^2)
    pure-Python rank loop (:367-390) by array operations;
  * persistence uses .npz instead of deepdish .h5 (deepdish / h5py / pytables are not available
    in this environment): feature files, the all-pairs matrix dump and the batch checkpoints.
    The on-disk field names are the reference's (preprocess/extractors.py:22-54).

Only what the scoring hot path needs is here; the reference's feature extraction is out of scope.
"""
import glob
import os
import time
import warnings
from itertools import chain

import numpy as np


def load_feature_file(path):
    """One song's feature dict from disk.  .npz natively; .h5 only if h5py happens to exist."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            feats = {k: z[k] for k in z.files}
        if "label" in feats:
            feats["label"] = str(feats["label"])
        return feats
    try:
        import h5py  # noqa: F401
    except ImportError:
        raise IOError("%s: no HDF5 reader (deepdish/h5py) in this environment; convert the "
                      "feature files to .npz with the same field names" % path)
    import h5py
    with h5py.File(path, "r") as f:
        feats = {k: (f[k][()] if not isinstance(f[k], h5py.Group) else {kk: f[k][kk][()] for kk in f[k]})
                 for k in f.keys()}
    if isinstance(feats.get("label"), bytes):
        feats["label"] = feats["label"].decode()
    return feats


class CoverAlgorithm(object):
    """
    Attributes
    ----------
    filepaths: list(string)
        List of paths to all files in the dataset
    cliques: {string: set}
        A dictionary of all cover cliques, where the cliques index into filepaths
    Ds: {string similarity type: ndarray(num files, num files)}
        A dictionary of pairwise similarity matrices, whose indices index into filepaths
    """

    def __init__(self, name="Generic", datapath="features_benchmark", shortname="full", cachedir="cache",
                 cache2dir="cache2", similarity_types=["main"], do_memmaps=True):
        """Same arguments and attributes as CoverAlgorithm.py:28-57 (`cache2dir` is accepted and, as there, unused until
        set_cache2dir).  `datapath` may also be an in-memory corpus (acoss_amd.synth.Corpus)."""
        self.name, self.shortname, self.cachedir = name, shortname, cachedir
        self.similarity_types, self.do_memmaps = similarity_types, do_memmaps
        self.cache2dir = None
        self.cliques = {}        # label -> set of song indices
        self.all_feats = {}      # per-song feature cache of the subclasses
        self.corpus, self.filepaths = self._discover(datapath)
        self.N = len(self.filepaths)
        if do_memmaps:
            self.Ds = {s: self._open_matrix(s) for s in similarity_types}
        print("Initialized %s algorithm on %i songs in dataset %s" % (name, self.N, shortname))

    @staticmethod
    def _discover(datapath):
        """(in-memory corpus or None, sorted song paths): the dataset directory holds one feature file per song,
        .h5 in the reference (CoverAlgorithm.py:41), .npz here when no .h5 is found."""
        if not isinstance(datapath, str):
            return datapath, ["<memory>/song_%06d" % i for i in range(datapath.n_songs)]
        for ext in ("h5", "npz"):
            found = sorted(glob.glob(os.path.join(datapath, "*." + ext)))
            if found:
                return None, found
        return None, []

    def _matrix_path(self, similarity_type):
        return "%s_%s_dmat" % (self.get_cacheprefix(), similarity_type)

    def _open_matrix(self, similarity_type):
        """The N x N float32 score matrix of one similarity type: file-backed under the cache prefix as in the
        reference (:52-55) on rank 0; in memory on the other ranks of a torch.distributed job, since every rank
        opening the same file with 'w+' would truncate it under the others."""
        shape = (self.N, self.N)
        if int(os.environ.get("RANK", "0")) != 0:
            return np.zeros(shape, dtype=np.float32)
        os.makedirs(self.cachedir, exist_ok=True)
        return np.memmap(self._matrix_path(similarity_type), dtype=np.float32, mode="w+", shape=shape)

    def set_cache2dir(self, cache2dir):
        self.cache2dir = cache2dir
        if not os.path.exists(cache2dir):
            os.mkdir(cache2dir)

    def get_cacheprefix(self):
        """Descriptive file prefix for cached features and distance matrices (CoverAlgorithm.py:59)."""
        return "%s/%s_%s" % (self.cachedir, self.name, self.shortname)

    def load_features(self, i):
        """
        Load the fields of song i and record its cover clique in self.cliques as a side effect
        (CoverAlgorithm.py:66-90).  In-memory corpora synthesise the dict.
        """
        if self.corpus is not None:
            x = self.corpus.song(i)
            feats = {"label": self.corpus.labels[i], "hpcp": x, "crema": x}
        else:
            feats = load_feature_file(self.filepaths[i])
        if not feats['label'] in self.cliques:
            self.cliques[feats['label']] = set([])
        self.cliques[feats['label']].add(i)
        return feats

    def _join_clique(self, label, i):
        self.cliques.setdefault(label, set()).add(int(i))

    def get_all_clique_ids(self, verbose=False):
        """Clique membership of every song (CoverAlgorithm.py:92-114): read from the "<index>,<label>" lines of
        <prefix>_clique_info.txt when that file exists, otherwise taken from every song's feature file (base-class
        loader, so a plugin's feature cache is not filled) and written there."""
        if self.corpus is not None:
            for i, label in enumerate(self.corpus.labels):
                self._join_clique(label, i)
            return
        listing = self.get_cacheprefix() + "_clique_info.txt"
        if os.path.exists(listing):
            with open(listing) as fin:
                for line in fin:
                    if line.strip():
                        i, label = line.split(",", 1)
                        self._join_clique(label.strip(), i)
            return
        os.makedirs(self.cachedir, exist_ok=True)
        with open(listing, "w") as fout:
            for i in range(len(self.filepaths)):
                label = CoverAlgorithm.load_features(self, i)["label"]      # joins the clique as a side effect
                fout.write("%i,%s\n" % (i, label))
                if verbose:
                    print(i)

    def similarity(self, idxs):
        """
        Scores for every row (i, j) of idxs, one array per similarity type; also stored in
        Ds[type][i, j] when do_memmaps.  The base class scores 0 (CoverAlgorithm.py:117-136).
        """
        idxs = np.asarray(idxs).reshape(-1, 2)
        if self.do_memmaps:
            self.Ds["main"][idxs[:, 0], idxs[:, 1]] = 0.0
        return {"main": np.zeros(idxs.shape[0])}

    # --------------------------------------------------------------------------------------
    def _pair_list(self, symmetric):
        n = len(self.filepaths)
        if symmetric:
            i, j = np.triu_indices(n, k=1)                       # itertools.combinations order (:166)
        else:
            i, j = np.nonzero(~np.eye(n, dtype=bool))            # itertools.permutations order (:168)
        return np.stack([i, j], axis=1).astype(np.int64)

    def all_pairwise(self, parallel=0, n_cores=12, symmetric=False, precomputed=False, batch_pairs=1 << 18):
        """
        All pairwise comparisons (CoverAlgorithm.py:138-184).  `parallel` / `n_cores` are accepted
        for call compatibility; the parallelism here is the GPU batch (and, when
        torch.distributed is initialised, one process per GPU over a sharded pair list).
        batch_pairs: pairs per similarity() call (the plugins split a call into launch batches of their own; a call costs
        a few milliseconds of host work and one synchronisation, so the default is large).
        """
        tic = time.time()
        dump = "%s_Ds.npz" % self.get_cacheprefix()
        if precomputed:
            with np.load(dump) as z:
                self.Ds = {k: z[k] for k in z.files}
            self.get_all_clique_ids()
        else:
            from . import sharding
            rank, world, dist, sharded = 0, 1, None, False
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    rank, world = dist.get_rank(), dist.get_world_size()
                    # ACOSS_FORCE_COLLECTIVE=1: shard and gather even in a one-rank group (how a one-GPU box runs the
                    # RCCL all-gather of this driver: tests/test_gpu_rccl.py)
                    sharded = world > 1 or os.environ.get("ACOSS_FORCE_COLLECTIVE", "0") == "1"
            except ImportError:
                dist = None
            # The pair list is an enumeration (:166-168): a rank takes every world-th POSITION and forms its own pairs from the
            # positions in closed form, batch by batch (sharding.py, round 5).  No list of all pairs, no sort, on any rank.
            K = sharding.n_pairs(self.N, symmetric)
            n_mine = len(range(rank, K, world)) if sharded else K
            local = {s: np.zeros(n_mine) for s in self.similarity_types}
            memmaps, self.do_memmaps = self.do_memmaps, False     # scatter once at the end instead
            t_sim = 0.0
            try:
                for lo in range(0, n_mine, batch_pairs):
                    hi = min(lo + batch_pairs, n_mine)
                    pos = (rank + world * np.arange(lo, hi, dtype=np.int64)) if sharded else np.arange(lo, hi, dtype=np.int64)
                    pairs = sharding.pairs_of_positions(self.N, pos, symmetric)
                    t0 = time.time()
                    res = self.similarity(pairs)
                    t_sim += time.time() - t0
                    for s in self.similarity_types:
                        local[s][lo:hi] = res[s]
            finally:
                self.do_memmaps = memmaps
            if not hasattr(self, "Ds"):
                self.Ds = {s: np.zeros((self.N, self.N), dtype=np.float32) for s in self.similarity_types}
            for s in self.similarity_types:
                full = local[s]
                if sharded:
                    import torch
                    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
                    full = sharding.gather_strided(torch.from_numpy(local[s]).to(dev), K, force_collective=True).cpu().numpy()
                # a matrix kept from an earlier call starts from zero again: `Ds += Ds.T` below would otherwise add the new
                # upper triangle to the old lower one (the reference's matrices are fresh 'w+' memmaps, CoverAlgorithm.py:52-55)
                self.Ds[s][:] = 0
                sharding.fill_matrix(self.Ds[s], full, symmetric)
            # where the call's time went: everything but the similarity() calls is host work (positions -> pairs, the gather,
            # the matrices); bench.py reports it as host_seconds_outside_kernels
            self.timing = {"similarity_seconds": t_sim, "pairs": int(K), "pairs_this_rank": int(n_mine)}
            self.get_all_clique_ids()
            if symmetric:
                for similarity_type in self.Ds:
                    self.Ds[similarity_type] += self.Ds[similarity_type].T     # :180-182
            if rank == 0:
                os.makedirs(self.cachedir, exist_ok=True)
                np.savez(dump, **{k: np.asarray(v) for k, v in self.Ds.items()})
        if hasattr(self, "timing") and not precomputed:
            self.timing["all_pairwise_seconds"] = time.time() - tic
        print("Elapsed Time All Pairwise: %.3g" % (time.time() - tic))

    def _pair_costs(self, pairs):
        return np.ones(len(pairs))

    def do_batch_features(self, n_batches, idx):
        """Precompute (and cache) the features of one slice of the dataset (CoverAlgorithm.py:186-201)."""
        N = len(self.filepaths)
        w = int(np.ceil(N / n_batches))
        for i in np.arange(w) + idx * w:
            if i < N:
                self.load_features(i)

    def do_batch_subbatch(self, w, idx, wsub, isub, jsub):
        """
        One wsub x wsub sub-block of block `idx` of the lower-triangular w x w block grid, pairs
        with row >= col, the diagonal included (CoverAlgorithm.py:203-247).
        """
        idxs = self.subbatch_pairs(len(self.filepaths), w, idx, wsub, isub, jsub)
        out = self.similarity(idxs)
        out['idxs'] = idxs
        return out

    @staticmethod
    def subbatch_pairs(n_songs, w, idx, wsub, isub, jsub):
        """The (row, col) song pairs of one sub-block, in the reference's enumeration order.  Blocks of the
        floor(n_songs / w)-wide block grid with block-row >= block-col are numbered column-major-within-row as
        the reference's meshgrid/flatten does (:208-212): block number idx sits at block-column c, block-row r
        where the numbering runs c = 0 (r = 0), c = 0..1 (r = 1) ... read along each grid row of the (J, I) mesh."""
        nb = n_songs // w
        # position idx in the sequence {(I, J) : meshgrid order, I >= J}: the mesh is indexed [a][b] with I = b, J = a
        order = [(b, a) for a in range(nb) for b in range(nb) if b >= a]
        bi, bj = order[idx]
        rows = np.arange(isub * wsub, min((isub + 1) * wsub, w)) + bi * w
        cols = np.arange(jsub * wsub, min((jsub + 1) * wsub, w)) + bj * w
        # the reference flattens meshgrid(pixi, pixj): the first index varies fastest inside each value of the second
        rr = np.tile(rows, len(cols))
        cc = np.repeat(cols, len(rows))
        keep = (rr < n_songs) & (cc < n_songs) & (rr >= cc)
        return np.stack([rr[keep], cc[keep]], axis=1)

    def _checkpoint_path(self, idx):
        return "%s_%s.npz" % (self.get_cacheprefix(), idx)

    @staticmethod
    def _read_checkpoint(path):
        """(accumulated result arrays, set of finished sub-blocks) of a block checkpoint; empty when the file is
        absent or unreadable (the reference recomputes in that case too, :262-267)."""
        if not os.path.exists(path):
            return {}, set()
        try:
            with np.load(path) as z:
                done = set((int(i), int(j)) for i, j in z["blocks_completed"])
                return {name[len("sim_"):]: z[name] for name in z.files if name.startswith("sim_")}, done
        except Exception:
            print("Error loading", path, ": recomputing")
            return {}, set()

    def _write_checkpoint(self, path, acc, done):
        os.makedirs(self.cachedir, exist_ok=True)
        arrays = {"sim_" + name: arr for name, arr in acc.items()}
        arrays["blocks_completed"] = np.array(sorted(done), dtype=np.int64).reshape(-1, 2)
        np.savez(path, **arrays)

    def do_batch(self, w, idx, wsub=-1):
        """
        Block `idx` of the w x w block grid, computed in wsub x wsub sub-blocks and checkpointed after each one in
        <prefix>_<idx>.npz (arrays "sim_<key>" + "blocks_completed"); sub-blocks the checkpoint already holds are
        skipped, so an interrupted job resumes (CoverAlgorithm.py:249-295).  Sub-block rows are walked boustrophedon
        (:294-295), which keeps the songs of the previous sub-block's edge in a plugin's cache.
        """
        path = self._checkpoint_path(idx)
        acc, done = self._read_checkpoint(path)
        per_side = int(w / (w if wsub == -1 else wsub))
        wsub = w if wsub == -1 else wsub
        for i in range(per_side):
            cols = range(per_side) if i % 2 == 0 else range(per_side - 1, -1, -1)
            for j in cols:
                if (i, j) in done:
                    continue
                tic = time.time()
                self.all_feats = {}                       # features of the finished sub-blocks are dropped (:282)
                part = self.do_batch_subbatch(w, idx, wsub, i, j)
                acc = dict(part) if not acc else {name: np.concatenate((acc[name], part[name])) for name in part}
                done.add((i, j))
                self._write_checkpoint(path, acc, done)
                print("Elapsed Time Sub-Batch %i_%i_%i: %.3g" % (idx, i, j, time.time() - tic), flush=True)
        return acc

    def load_batches(self, fileprefix):
        """Ds rebuilt from every block checkpoint "<fileprefix>*.npz": each pair's score is added at (i, j) and at
        (j, i) -- a diagonal pair therefore twice, as in CoverAlgorithm.py:297-317."""
        self.Ds = {name: np.zeros_like(D) for name, D in self.Ds.items()}
        for path in glob.glob(fileprefix + "*.npz"):
            with np.load(path) as z:
                if "sim_idxs" not in z.files:
                    continue
                rows, cols = z["sim_idxs"][:, 0], z["sim_idxs"][:, 1]
                for name, D in self.Ds.items():
                    scores = z["sim_" + name]
                    for a, b in ((rows, cols), (cols, rows)):
                        D[a, b] = D[a, b] + scores
        self.get_all_clique_ids()

    def cleanup_memmap(self):
        """Remove the memmap files behind Ds (the reference's rmtree on a file never succeeds, :319-328)."""
        for s in list(getattr(self, "Ds", {})):
            path = self._matrix_path(s)
            try:
                if os.path.exists(path):
                    os.remove(path)
            except OSError:
                print('Could not clean-up automatically.')

    # --------------------------------------------------------------------------------------
    def _mate_ranks_device(self, D, cliques):
        """Sorted ranks of every song's clique mates from the GPU (acoss_eval_ranks): {song: int32 array}."""
        from . import engine
        torch = engine.torch
        lib = engine._lib.load()
        engine.require_gpu()
        N = D.shape[0]
        dev = "cuda:%d" % torch.cuda.current_device()
        cid = -1 - np.arange(N, dtype=np.int32)                   # songs in no clique: unique ids, no mates
        mates = np.zeros(N, dtype=np.int64)
        for c, members in enumerate(cliques):
            cid[members] = c
            mates[members] = len(members) - 1
        off = np.zeros(N + 1, dtype=np.int64)
        off[1:] = np.cumsum(mates)
        Dd = torch.from_numpy(np.ascontiguousarray(D, dtype=np.float32)).to(dev)
        cid_d, off_d = torch.from_numpy(cid).to(dev), torch.from_numpy(off).to(dev)
        out = torch.zeros(max(int(off[-1]), 1), dtype=torch.int32, device=dev)
        engine.check(lib.acoss_eval_ranks(engine._ptr(Dd), N, N, engine._ptr(cid_d), engine._ptr(off_d),
                                          int(mates.max()) if N else 0, engine._ptr(out), engine._stream()), "eval_ranks")
        return out.cpu().numpy(), off

    def getEvalStatistics(self, similarity_type, topsidx=[1, 10, 100, 1000], verbose=True, write_csv=True,
                          on_gpu=False):
        """
        MR, MRR, MDR, MAP and Top-X of one similarity type -- the same numbers as
        CoverAlgorithm.py:330-418 (same permutation, same argsort calls and therefore the same
        tie-breaking), with the per-row rank loop (:367-390) done on arrays.

        on_gpu=True takes the ranks of the clique mates from the GPU instead of argsorting every row (the
        O(N^2 log N) part, hours of Python at N = 15000 in the reference): identical on rows without equal
        scores; equal scores rank in song-index order there, in introsort's order here.
        """
        D = np.array(self.Ds[similarity_type], dtype=np.float32)
        N = D.shape[0]
        cliques = [list(self.cliques[s]) for s in self.cliques]
        Ks = np.array([len(c) for c in cliques])
        order = np.argsort(-Ks)                                   # :349 (same call, same tie order)
        Ks = Ks[order]
        cliques = [cliques[i] for i in order]
        perm = np.array(list(chain(*cliques)), dtype=int)
        ranks = np.nan * np.ones(N)
        AllMap = np.nan * np.ones(N)
        starts = np.concatenate([[0], np.cumsum(Ks)[:-1]])
        if on_gpu:
            mate_ranks, off = self._mate_ranks_device(D, cliques)
        else:
            D = D[perm, :]
            D = D[:, perm]
            np.fill_diagonal(D, -np.inf)
            idx = np.argsort(-D, 1)                               # :362
            # rank (1-based position in its row's ordering) of every column
            pos = np.empty_like(idx)
            pos[np.arange(N)[:, None], idx] = np.arange(1, N + 1)[None, :]
        for start, K in zip(starts, Ks):
            if K < 2:
                break                                             # :372-375 cliques are sorted by size
            if on_gpu:
                block = np.stack([mate_ranks[off[s]:off[s + 1]] for s in perm[start:start + K]]).astype(np.int64)
            else:
                block = np.sort(pos[start:start + K, start:start + K], axis=1)[:, :-1]   # drop the last = self (:381)
            ranks[start:start + K] = block[:, 0]                  # :386
            j = np.arange(1, K, dtype=np.float64)[None, :]
            AllMap[start:start + K] = np.mean(j / block.astype(np.float64), axis=1)    # :388-390
        if np.all(np.isnan(AllMap)):
            warnings.warn("Recalling 0 songs: no clique with at least 2 songs")
        MAP = np.nanmean(AllMap)
        ranks = ranks[np.isnan(ranks) == 0]
        MR = np.mean(ranks)
        MRR = 1.0 / N * (np.sum(1.0 / ranks))                     # :395 (divides by ALL songs)
        MDR = np.median(ranks)
        if verbose:
            print("%s %s STATS\n-------------------------\nMR = %.3g\nMRR = %.3g\nMDR = %.3g\nMAP = %.3g"
                  % (self.name, similarity_type, MR, MRR, MDR, MAP))
        tops = np.array([np.sum(ranks <= t) for t in topsidx], dtype=np.float64)
        if verbose:
            for t, n in zip(topsidx, tops):
                print("Top-%i: %i" % (t, n))
        if write_csv:
            self._append_results_row(similarity_type, topsidx, (MR, MRR, MDR, MAP), tops)
        return (MR, MRR, MDR, MAP, tops)

    def _append_results_row(self, similarity_type, topsidx, stats, tops):
        """One line per (algorithm, similarity type) in results_<shortname>.csv, header written with the first line;
        columns and number format of CoverAlgorithm.py:404-417."""
        path = "results_%s.csv" % self.shortname
        lines = []
        if not os.path.exists(path):
            lines.append("name, MR, MRR, MDR, MAP" + "".join(",Top-%i" % t for t in topsidx))
        lines.append("%s_%s," % (self.name, similarity_type) + ", ".join("%.3g" % v for v in list(stats) + list(tops)))
        with open(path, "a") as fout:
            fout.write("\n".join(lines) + "\n")

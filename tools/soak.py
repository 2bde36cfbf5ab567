"""Randomised parity soak (dev tool): many ragged pairs through the product chain vs the CPU oracle, exact equality."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
from oracle import oracle
engine.require_gpu()
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
long_songs = len(sys.argv) > 3 and sys.argv[3] == "long"       # lengths 60 .. 2600: all three size classes of the chain
rng = np.random.default_rng(seed)
if long_songs:
    ch = synth.make_corpus(12, 4, seed=seed, singletons=8, lengths=lambda r: int(np.clip(r.normal(1300, 600), 60, 2600)))
else:
    ch = synth.make_corpus(40, 8, seed=seed, singletons=30, lengths=lambda r: int(np.clip(r.normal(520, 160), 60, 1032)))
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
allp = synth.all_pairs(ch.n_songs)
pairs = allp[rng.permutation(len(allp))[:n_pairs]]
pairs = np.where(rng.random((len(pairs), 1)) < 0.5, pairs, pairs[:, ::-1]).astype(np.int32)      # both orientations
t0 = time.time()
got = engine.serra09_scores(corpus, pairs, approx32=bool(os.environ.get("ACOSS_PLANAR32")) or None)
t1 = time.time()
threads = min(os.cpu_count() or 1, 16)
q, d, _ = oracle.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, pairs, nthreads=threads)
t2 = time.time()
bq, bd = np.flatnonzero(got["qmax"] != q), np.flatnonzero(got["dmax"] != d)
print("songs %d (lengths %d..%d), %d pairs: GPU %.2f s, oracle %.1f s on %d threads; qmax mismatches %d, dmax mismatches %d"
      % (ch.n_songs, np.diff(ch.frame_off).min(), np.diff(ch.frame_off).max(), len(pairs), t1 - t0, t2 - t1, threads, len(bq), len(bd)))
for t in list(bq[:5]) + list(bd[:5]):
    print("  pair", pairs[t], "gpu", got["qmax"][t], got["dmax"][t], "oracle", q[t], d[t])
sys.exit(1 if len(bq) or len(bd) else 0)

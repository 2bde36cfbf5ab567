"""Randomised soak of the float32-corpus filter path (round 4): engine.serra09_scores on ragged float32 corpora (MFCC-shaped random
walks, some songs with a large common offset, some with repeated frames) with the 16-bit-key filter against the same call with
approx32=False (crp_kernel<float> + float64 selection: the float32-input chain the filter must reproduce).  qmax, dmax, swc.
usage: python tools/soak_f32.py [songs] [seed] [kinds: 7 = offset, 8 = repeated frames, 9 = x 1e-6]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kinds = set(int(k) for k in sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] else {7, 8, 9}      # which adversarial families
rng = np.random.default_rng(seed)
songs = []
for s_ in range(S):
    n = int(rng.integers(60, 1033))
    w = np.cumsum(rng.standard_normal((n, 13)) * rng.uniform(0.2, 3.0), axis=0)
    kind = s_ % 10
    if kind == 7 and 7 in kinds:
        w = w + 150.0
    if kind == 8 and 8 in kinds:
        w = np.tile(w[:17], (n // 17 + 1, 1))[:n]
    if kind == 9 and 9 in kinds:
        w = w * 1e-6
    songs.append(w.astype(np.float32))
feats = np.concatenate(songs)
off = np.cumsum([0] + [len(s) for s in songs]).astype(np.int64)
corpus = engine.DeviceCorpus(feats, off)
pairs = synth.all_pairs(S)
engine.serra09_scores(corpus, pairs[:2000], do_oti=False, want=("qmax", "dmax", "swc"))                   # warm: scratch, module
engine.serra09_scores(corpus, pairs[:2000], do_oti=False, want=("qmax", "dmax", "swc"), approx32=False)
t0 = time.time(); a = engine.serra09_scores(corpus, pairs, do_oti=False, want=("qmax", "dmax", "swc")); t1 = time.time()
b = engine.serra09_scores(corpus, pairs, do_oti=False, want=("qmax", "dmax", "swc"), approx32=False); t2 = time.time()
bad = {k: int(np.sum(a[k] != b[k])) for k in a}
st = (engine.ctypes.c_int * 4)() if hasattr(engine, "ctypes") else None
print("%d songs, %d pairs: filter %.2f s, float32-input chain %.2f s, mismatches %s" % (S, len(pairs), t1 - t0, t2 - t1, bad))
assert not any(bad.values())

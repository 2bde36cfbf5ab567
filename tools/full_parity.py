"""Every pair of the headline workload (BASELINE config 2: 1000 songs x 1000 frames, 499 500 pairs) through the product chain and through
the CPU oracle, qmax and dmax compared exactly (dev tool; ~10 min of 16 host threads).  usage: python tools/full_parity.py [n_songs | config3]
(config3: the 2000-song DA-TACOS-shaped corpus of bench.py's job block, 1 999 000 ragged pairs)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acoss_amd import engine, synth
from oracle import oracle
engine.require_gpu()
c3 = len(sys.argv) > 1 and sys.argv[1] == "config3"
n = 2000 if c3 else (int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
ch = synth.config3(n_cliques=133, singletons=271) if c3 else synth.config2(n_songs=n, n_frames=1000)
corpus = engine.DeviceCorpus(ch.feats, ch.frame_off, gchroma=ch.gchroma)
pairs = synth.all_pairs(ch.n_songs)
t0 = time.time()
got = engine.serra09_scores(corpus, pairs)
t1 = time.time()
print("GPU: %d pairs in %.2f s" % (len(pairs), t1 - t0), flush=True)
threads = min(os.cpu_count() or 1, 16)
bad_q = bad_d = 0
step = 100000 if c3 else 20000
for a in range(0, len(pairs), step):
    q, d, _ = oracle.serra09_pairs(ch.feats, ch.frame_off, ch.gchroma, pairs[a:a + step], nthreads=threads)
    bad_q += int((got["qmax"][a:a + step] != q).sum()); bad_d += int((got["dmax"][a:a + step] != d).sum())
    print("  oracle %7d / %d pairs, %.0f s; mismatches so far: qmax %d dmax %d" % (min(a + step, len(pairs)), len(pairs), time.time() - t1, bad_q, bad_d), flush=True)
print(("config 3" if c3 else "config 2") + ", %d songs: %d pairs, qmax mismatches %d, dmax mismatches %d (oracle %d threads, %.0f s)" % (n, len(pairs), bad_q, bad_d, threads, time.time() - t1))

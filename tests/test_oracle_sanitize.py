"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer on the reference's fixtures (the
reference itself compiles its alignment kernel with bounds checks off, pySeqAlign.pyx:5-6).  CPU only; `make -C oracle
oracle_asan_check` builds oracle/asan_driver.c + acoss_oracle.c with -fsanitize=address,undefined."""
import os
import struct
import subprocess

import numpy as np

from conftest import ROOT


def _write_cases(golden, path):
    with open(path, "wb") as f:
        g = golden("dp_cases")
        for k in range(int(g["n_cases"])):
            p = "k%d_" % k
            S = np.ascontiguousarray(g[p + "S"], dtype=np.uint8)
            M, N = S.shape
            f.write(b"DP  " + struct.pack("<ii", M, N) + S.tobytes())
            for key in ("Dq", "Dd_reused", "Dd_fresh", "Dsw"):
                f.write(np.ascontiguousarray(g[p + key], dtype=np.float32).tobytes())
            f.write(np.ascontiguousarray(g[p + "scores"], dtype=np.float64).tobytes())
        g = golden("stages")
        for c in range(3):
            p = "c%d_" % c
            X, Y = np.ascontiguousarray(g[p + "X"], dtype=np.float64), np.ascontiguousarray(g[p + "Y"], dtype=np.float64)
            f.write(b"STG " + struct.pack("<iiiiid", X.shape[0], Y.shape[0], X.shape[1], int(g[p + "m"]), int(g[p + "oti"]),
                                           float(g[p + "kappa"])))
            for a in (X, Y, g[p + "gX"], g[p + "gY"], g[p + "CSM"], g[p + "S"]):
                f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
            for a in (g[p + "B1"], g[p + "B"]):
                f.write(np.ascontiguousarray(a, dtype=np.uint8).tobytes())
        f.write(b"END ")


def test_oracle_is_clean_under_address_and_ub_sanitizers(golden, tmp_path):
    build = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle_asan_check"], capture_output=True, text=True)
    assert build.returncode == 0, build.stdout + build.stderr
    cases = str(tmp_path / "cases.bin")
    _write_cases(golden, cases)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([os.path.join(ROOT, "oracle", "oracle_asan_check"), cases], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr
    n_dp = int(golden("dp_cases")["n_cases"])
    assert " 0 mismatches" in run.stdout and "%d alignment records, 3 stage records" % n_dp in run.stdout, run.stdout

"""
Builds libacoss_mi355x.so in-tree with hipcc for gfx950 (the only target).  The library has no
dependency on torch: it links the HIP runtime only.

    python -m acoss_amd.build            # build if sources are newer than the library
    python -m acoss_amd.build --force
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libacoss_mi355x.so")
SOURCES = ["capi.hip", "crp_kernels.hip", "fused_kernels.hip", "strip32_kernels.hip", "dp_kernels.hip", "planar_kernels.hip", "eval_kernels.hip", "ftm2d_kernels.hip", "snf_kernels.hip", "probe_kernels.hip"]
HEADERS = ["common.h", "wave_ops.h", "kernel_utils.h", "thresh_work.h", "gemm_f64.h", os.path.join("..", "..", "include", "acoss_mi355x.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off",      # every FMA in the kernels is written explicitly
         "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    out = subprocess.run(cmd, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("hipcc failed:\n%s\n%s" % (out.stdout, out.stderr))
    if verbose and out.stderr:
        print(out.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

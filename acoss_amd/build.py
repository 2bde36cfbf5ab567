"""
Builds libacoss_mi355x.so in-tree with hipcc for gfx950 (the only target).  The library has no
dependency on torch: it links the HIP runtime only.  Every source is compiled to its own object
(in parallel, rebuilt only when it or a header changed) and the objects are linked.

    python -m acoss_amd.build            # build what is out of date
    python -m acoss_amd.build --force
    python -m acoss_amd.build --probes   # also compile the development probe kernels (tools/*_probe.py)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "libacoss_mi355x.so")
SOURCES = ["capi.hip", "crp_kernels.hip", "fused_kernels.hip", "strip32_kernels.hip", "band_kernels.hip",
           "dp_kernels.hip", "planar_kernels.hip", "keys16_kernels.hip", "eval_kernels.hip", "ftm2d_kernels.hip", "snf_kernels.hip",
           "scorer.hip", "csm_rows_kernels.hip", "radix16_kernels.hip"]
PROBE_SOURCES = []          # (the store-pattern probes moved to tools/ubench/*.hip: stand-alone programs)
HEADERS = ["common.h", "wave_ops.h", "kernel_utils.h", "thresh_work.h", "gemm_f64.h", "gemm_f32.h", "planar_select.h", "keys16.h", "radix16.h",
           os.path.join("..", "..", "include", "acoss_mi355x.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
         "-ffp-contract=off",      # every FMA in the kernels is written explicitly
         "-Wall", "-Wno-unused-function"]


def _probes_wanted(probes):
    return bool(probes) or os.environ.get("ACOSS_BUILD_PROBES", "0") not in ("0", "", "no", "false")


def _sources(probes):
    return SOURCES + (PROBE_SOURCES if probes else [])


def _stamp(probes, extra_flags):
    return " ".join(FLAGS + list(extra_flags) + (["-DACOSS_PROBES"] if probes else []))


def _compile(src, obj, flags, verbose):
    cmd = [HIPCC] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd))
    out = subprocess.run(cmd, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, out.stdout, out.stderr))
    if verbose and out.stderr:
        print(out.stderr)


def build(force=False, verbose=False, extra_flags=(), probes=False):
    probes = _probes_wanted(probes)
    os.makedirs(OBJ, exist_ok=True)
    stamp_file = os.path.join(OBJ, "flags.txt")
    stamp = _stamp(probes, extra_flags)
    old = open(stamp_file).read() if os.path.exists(stamp_file) else None
    if old != stamp:
        force = True
    hdr_t = max([os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS] + [os.path.getmtime(os.path.abspath(__file__))])
    flags = FLAGS + list(extra_flags) + (["-DACOSS_PROBES"] if probes else [])
    todo, objs = [], []
    for s in _sources(probes):
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        src_t = os.path.getmtime(os.path.join(CSRC, s))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(src_t, hdr_t):
            todo.append((s, obj))
    if todo:
        jobs = max(1, min(len(todo), int(os.environ.get("ACOSS_BUILD_JOBS", "6"))))
        with ThreadPoolExecutor(jobs) as ex:
            for f in [ex.submit(_compile, s, o, flags, verbose) for s, o in todo]:
                f.result()
    if todo or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        out = subprocess.run(cmd, capture_output=True, text=True)
        if out.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (out.stdout, out.stderr))
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return LIB


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, probes="--probes" in sys.argv))

"""Dev tool: a second build of the library with extra compiler flags (-D switches), into tools/ab/lib_<name>.so, for
A/B runs in one process on the same buffers (tools/ab_k16.py).   usage: python tools/build_variant.py NAME [-DFLAG=1 ...]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from acoss_amd import build as b

name, flags = sys.argv[1], sys.argv[2:]
obj = os.path.join(ROOT, "tools", "ab", "obj_" + name)
os.makedirs(obj, exist_ok=True)
out = os.path.join(ROOT, "tools", "ab", "lib_%s.so" % name)


def cc(src):
    o = os.path.join(obj, src.replace(".hip", ".o"))
    subprocess.run([b.HIPCC] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", o], check=True)
    return o


with ThreadPoolExecutor(6) as ex:
    objs = list(ex.map(cc, b.SOURCES))
subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], check=True)
print(out)

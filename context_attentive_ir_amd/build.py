"""Build libneuroir_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m context_attentive_ir_amd.build [--force]

The .so is written IN-TREE next to this file (git-ignored, but it ships to the GPU box with the snapshot).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libneuroir_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# Instrumented / experimental builds for the tools (never the product library): NIR_VARIANT=name NIR_VARIANT_FLAGS="-D..." build into
# csrc/_obj_<name> and libneuroir_hip_<name>.so; tools load that file explicitly.
if os.environ.get("NIR_VARIANT"):
    _v = os.environ["NIR_VARIANT"]
    OBJ = os.path.join(HERE, "csrc", "_obj_" + _v)
    LIB = os.path.join(HERE, "libneuroir_hip_%s.so" % _v)
    FLAGS = FLAGS + os.environ.get("NIR_VARIANT_FLAGS", "").split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "neuroir_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime()):
        return obj, False
    subprocess.run([HIPCC] + FLAGS + ["-c", path, "-o", obj], check=True)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(_compile, _sources()))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

"""Builds libopental_hip.so for gfx950 with hipcc (no torch cpp_extension, no hipify).

    python -m opental_amd.csrc.build [--force]

The .so is written in-tree (opental_amd/lib/) so it travels with the repo snapshot to the GPU box.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
LIB_DIR = os.path.join(os.path.dirname(HERE), "lib")
LIB = os.path.join(LIB_DIR, "libopental_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-I" + os.path.join(REPO, "include"), "-I" + HERE, "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _includes(src, seen=None):
    """Files of this directory that `src` includes, directly or through another file (conv_gemm_half.hip includes
    conv_gemm.hip, which includes the .inc files)."""
    import re
    seen = set() if seen is None else seen
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(src).read(), flags=re.M):
        path = os.path.join(HERE, name)
        if os.path.exists(path) and path not in seen:
            seen.add(path)
            _includes(path, seen)
    return seen


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    headers = glob.glob(os.path.join(HERE, "*.h")) + glob.glob(os.path.join(HERE, "*.inc")) + glob.glob(os.path.join(REPO, "include", "*.h"))
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers + sorted(_includes(src))):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

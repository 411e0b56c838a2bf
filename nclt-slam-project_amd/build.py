"""Builds nclt-slam-project_amd/csrc/libreloc_hip.so with hipcc for gfx950 (cross-compiles
without a GPU).  In-tree output so the .so travels to the GPU box with the repository."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libreloc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    inc = os.path.join(HERE, "..", "include")
    hdrs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    objs, jobs = [], []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_variant(name: str, defines: list[str], verbose: bool = False) -> str:
    """Developer builds for kernel experiments: the whole library compiled with extra -D flags into
    build_variants/libreloc_hip_<name>.so (git-ignored, travels to the GPU box).  Never loaded by the product path;
    tests/dev/exp_scan_variants.py and tools/exp_*.py select one through RELOC_LIB (under RELOC_DEV=1)."""
    vdir = os.path.join(HERE, "..", "build_variants", name)
    os.makedirs(vdir, exist_ok=True)
    objs, jobs = [], []
    for s in sources():
        obj = os.path.join(vdir, s[:-4] + ".o")
        objs.append(obj)
        jobs.append([HIPCC] + FLAGS + [f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, s), "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    lib = os.path.join(vdir, "..", f"libreloc_hip_{name}.so")
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return os.path.abspath(lib)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""In-tree build of libradnerf_hip.so (hipcc, gfx950 only).

    python rad-nerf_amd/build.py [--force]

Every csrc/*.hip is compiled to an object next to it and linked into
rad-nerf_amd/lib/libradnerf_hip.so.  hipcc cross-compiles without a GPU, so this runs on
the CPU-only build container; the .so travels to the GPU box with the tree.

Flags: -ffp-contract=off (float expressions that feed integer indices must round exactly as
written -- see csrc/rn_common.h), -O3, wave64 (the gfx950 default).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
LIBDIR = os.path.join(ROOT, "lib")
LIB = os.path.join(LIBDIR, "libradnerf_hip.so")
INCLUDE = os.path.join(os.path.dirname(ROOT), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}", "-Wall",
            "-Wno-unused-function", f"-I{INCLUDE}"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(CSRC, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if force or _stale(obj, [path, __file__] + _headers()):
        cmd = [HIPCC] + CXXFLAGS + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build_all(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)

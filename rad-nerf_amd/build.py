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


BINDINGS_SRC = os.path.join(CSRC, "bindings", "radnerf_pybind.cpp")
BINDINGS_DIR = os.path.join(LIBDIR, "pybind")
BINDINGS_MODULES = ("_raymarching_face", "_gridencoder", "_shencoder", "_freqencoder")


def build_bindings(force=False, verbose=False):
    """The pybind11 modules with the reference's names (csrc/bindings/radnerf_pybind.cpp): ONE shared object, g++ against
    torch's headers and libradnerf_hip.so, installed under its four module names in rad-nerf_amd/lib/pybind/ (add that
    directory to sys.path and the reference's `import _gridencoder as _backend` finds it).  No device code in it."""
    import shutil
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(BINDINGS_DIR, exist_ok=True)
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    first = os.path.join(BINDINGS_DIR, BINDINGS_MODULES[0] + suffix)
    if force or _stale(first, [BINDINGS_SRC, __file__, LIB] + _headers()):
        tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
               f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"]
        cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}", "-I/opt/rocm/include", f"-I{INCLUDE}"]
        cmd += [BINDINGS_SRC, "-o", first, f"-L{tlib}", "-ltorch", "-ltorch_python", "-ltorch_cpu", "-lc10", "-lc10_hip",
                f"-L{LIBDIR}", "-lradnerf_hip", f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN/.."]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"pybind11 binding failed to build:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")
        for name in BINDINGS_MODULES[1:]:           # the same object under the other three module names (it exports every PyInit_)
            shutil.copyfile(first, os.path.join(BINDINGS_DIR, name + suffix))
    if verbose:
        print(f"built {BINDINGS_DIR}/{{{', '.join(BINDINGS_MODULES)}}}{suffix}")
    return BINDINGS_DIR


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
    if "--no-bindings" not in sys.argv:
        build_bindings(force="--force" in sys.argv, verbose=True)

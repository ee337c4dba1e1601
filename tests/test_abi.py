"""The C-ABI library loads on a CPU-only box and exports exactly what include/radnerf_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    decls = {}
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(ROOT, "include", fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b(rn_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
            args = m.group(2).strip()
            n = 0 if args in ("", "void") else len(args.split(","))
            decls[m.group(1)] = n
    return decls


def test_library_loads_and_exports_every_declared_symbol(hiplib):
    decls = _header_decls()
    assert len(decls) >= 25
    lib = ctypes.CDLL(hiplib.LIB_PATH)
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert set(hiplib.exported_symbols()) <= set(decls)


def test_ctypes_signatures_match_header_arity(hiplib):
    decls = _header_decls()
    for name, argtypes in hiplib._SIGNATURES.items():
        assert decls[name] == len(argtypes), f"{name}: header has {decls[name]} args, binding has {len(argtypes)}"


def test_version_and_error_channel(hiplib):
    assert hiplib.version() >= 100
    assert isinstance(hiplib.last_error(), str)
    # bad arguments are rejected on the host before anything touches a device
    rc = hiplib._lib.rn_sh_encode_forward(1, 1, 4, 2, 4, None, None)
    assert rc == -1 and "input dim == 3" in hiplib.last_error()
    rc = hiplib._lib.rn_grid_encode_forward(1, 1, 1, 1, 4, 3, 3, 16, 0.5, 16, None, 0, 0, 0, 0, 0, None)
    assert rc == -1 and "C must be 1, 2, 4, or 8" in hiplib.last_error()
    rc = hiplib._lib.rn_grid_encode_forward(1, 1, 1, 1, 4, 7, 2, 16, 0.5, 16, None, 0, 0, 0, 0, 0, None)
    assert rc == -1 and "D must be" in hiplib.last_error()


def test_no_oracle_or_cpu_fallback_in_product_tree():
    """The product package must never import, load or link anything under oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rad-nerf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"pyoracle|radnerf_oracle|libradnerf_oracle|orc_\w+\(", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad

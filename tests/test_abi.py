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


def test_round2_entry_points_refuse_bad_arguments(hiplib):
    """The argument checks of the training-side entry points run before anything touches a GPU: unsupported shapes and null
    pointers come back as RN_ERR_INVALID_ARG with a message (no launch, no CPU fallback)."""
    lib = ctypes.CDLL(hiplib.LIB_PATH)
    lib.rn_last_error.restype = ctypes.c_char_p
    c_u32, c_p, c_f = ctypes.c_uint32, ctypes.c_void_p, ctypes.c_float

    def err():
        return lib.rn_last_error().decode()
    lib.rn_mlp64_image_floats.restype = ctypes.c_size_t
    lib.rn_mlp64_image_floats.argtypes = [c_u32, c_u32, c_u32]
    assert lib.rn_mlp64_image_floats(96, 2, 3) > 0 and lib.rn_mlp64_image_floats(65, 65, 3) > 0 and lib.rn_mlp64_image_floats(84, 3, 2) > 0
    assert lib.rn_mlp64_image_floats(128, 2, 3) == 0 and lib.rn_mlp64_image_floats(65, 7, 3) == 0 and lib.rn_mlp64_image_floats(65, 65, 4) == 0
    lib.rn_mlp64_pack.argtypes = [c_p, c_u32, c_p, c_p, c_u32, c_u32, c_u32, c_p, c_p]
    assert lib.rn_mlp64_pack(None, 128, None, None, 128, 2, 3, None, None) != 0 and "unsupported shape" in err()
    assert lib.rn_mlp64_pack(None, 32, None, None, 64, 2, 3, None, None) != 0 and "ld0" in err()
    lib.rn_mlp64_forward.argtypes = [c_p, c_u32, c_p, c_p, c_u32, c_u32, c_u32, c_p, c_p, c_p, c_p]
    assert lib.rn_mlp64_forward(None, 0, None, None, 65, 65, 3, None, None, None, None) == 0        # M = 0: nothing to do
    assert lib.rn_mlp64_forward(None, 64, None, None, 65, 65, 3, None, None, None, None) != 0 and "null pointer" in err()
    lib.rn_adam_step.argtypes = [c_p, c_u32, c_f, c_f, c_f, c_p, c_p, c_p]
    assert lib.rn_adam_step(None, 0, 0.9, 0.99, 1e-15, None, None, None) != 0 and "step counter" in err()
    lib.rn_train_loss.argtypes = [c_p] * 6 + [c_u32] + [c_p] * 5
    assert lib.rn_train_loss(None, None, None, None, None, None, 0, None, None, None, None, None) != 0 and "N must be positive" in err()
    lib.rn_head_mid_forward.argtypes = [c_p, c_p, c_u32, c_u32, c_p, c_p, c_p]
    assert lib.rn_head_mid_forward(None, None, 10, 16, None, None, None) != 0 and "null pointer" in err()
    lib.rn_audio_encode_windows_backward.argtypes = [c_p, c_p, c_u32, c_p, c_p, c_p, c_p, c_p]
    assert lib.rn_audio_encode_windows_backward(None, None, 1, None, None, None, None, None) != 0 and "null weights" in err()
    lib.rn_march_rays_train_budget.argtypes = [c_p, c_p, c_p, c_f, c_f] + [c_u32] * 5 + [c_p] * 11
    assert lib.rn_march_rays_train_budget(None, None, None, 1.0, 0.0, 16, 64, 1, 128, 1000, *([None] * 11)) != 0 and "null pointer" in err()

"""The reference-side binding (radnerf_hip/compat_backend.py): `_backend` objects with the pybind modules'
function names and argument lists (raymarching.h:7-20, gridencoder.h:12-15, shencoder.h:9-10,
freqencoder.h:7-10), driven here exactly like the reference's Python wrappers drive them."""
import inspect

import numpy as np
import pytest
import torch

REFERENCE_SURFACE = {
    "raymarching_backend": {
        "near_far_from_aabb": 7, "sph_from_ray": 5, "morton3D": 3, "morton3D_invert": 3, "packbits": 4,
        "morton3D_dilation": 4, "march_rays_train": 18, "march_rays_train_backward": 8,
        "composite_rays_train_forward": 12, "composite_rays_train_backward": 17, "march_rays": 18, "composite_rays": 11},
    "gridencoder_backend": {"grid_encode_forward": 14, "grid_encode_backward": 16, "grad_total_variation": 13},
    "shencoder_backend": {"sh_encode_forward": 6, "sh_encode_backward": 7},
    "freqencoder_backend": {"freq_encode_forward": 6, "freq_encode_backward": 7},
}


def test_backend_objects_have_the_pybind_names_and_arities(hiplib):
    from radnerf_hip import compat_backend as cb
    for obj_name, fns in REFERENCE_SURFACE.items():
        obj = getattr(cb, obj_name)
        have = {n for n, f in vars(obj).items() if isinstance(f, staticmethod)}
        assert have == set(fns), (obj_name, have ^ set(fns))
        for fn, arity in fns.items():
            assert len(inspect.signature(getattr(obj, fn)).parameters) == arity, (obj_name, fn)


def _pybind_modules(hiplib):
    """The four pybind11 modules (csrc/bindings/radnerf_pybind.cpp), built on demand into rad-nerf_amd/lib/pybind/."""
    import importlib
    import importlib.util
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rn_build", os.path.join(root, "rad-nerf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sysconfig
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    d = mod.BINDINGS_DIR
    if os.environ.get("RN_REBUILD") == "1" or not all(os.path.exists(os.path.join(d, n + suffix)) for n in mod.BINDINGS_MODULES):
        d = mod.build_bindings()            # ~2.5 min of g++ on torch's headers; the built objects travel with the tree
    if d not in sys.path:
        sys.path.insert(0, d)
    return {name: importlib.import_module(name) for name in mod.BINDINGS_MODULES}


def test_pybind_modules_have_the_reference_names_and_arities(hiplib):
    """Module names of the reference's setup.py / bindings.cpp, every function, and the argument count pybind11 reports."""
    mods = _pybind_modules(hiplib)
    for mod_name, key in (("_raymarching_face", "raymarching_backend"), ("_gridencoder", "gridencoder_backend"),
                          ("_shencoder", "shencoder_backend"), ("_freqencoder", "freqencoder_backend")):
        have = {n for n in dir(mods[mod_name]) if not n.startswith("_")}
        assert have == set(REFERENCE_SURFACE[key]), (mod_name, have ^ set(REFERENCE_SURFACE[key]))
        for fn, arity in REFERENCE_SURFACE[key].items():
            sig = getattr(mods[mod_name], fn).__doc__.splitlines()[0]
            assert sig.count("arg") == arity, (mod_name, fn, sig)


@pytest.mark.gpu
@pytest.mark.parametrize("binding", ["ctypes", "pybind11"])
def test_reference_style_calls_through_the_backend(po, hiplib, rng, binding):
    if binding == "ctypes":
        from radnerf_hip.compat_backend import freqencoder_backend, gridencoder_backend, raymarching_backend, shencoder_backend
    else:
        mods = _pybind_modules(hiplib)
        raymarching_backend, gridencoder_backend = mods["_raymarching_face"], mods["_gridencoder"]
        shencoder_backend, freqencoder_backend = mods["_shencoder"], mods["_freqencoder"]
    from radnerf.scene import ellipsoid_bitfield
    from gridencoder.encoder import level_offsets
    dev = "cuda"
    N = 3000
    o = np.tile(np.array([[0.05, 3.3, -0.1]], np.float32), (N, 1))
    d = rng.uniform(-0.7, 0.7, (N, 3)).astype(np.float32) - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    bits, _ = ellipsoid_bitfield(128, 1.0, (0.4, 0.42, 0.4))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731

    # raymarching/raymarching.py:37-45
    rays_o, rays_d, taabb = t(o), t(d), t(aabb)
    nears = torch.empty(N, device=dev); fars = torch.empty(N, device=dev)
    raymarching_backend.near_far_from_aabb(rays_o, rays_d, taabb, N, 0.05, nears, fars)
    en, ef = po.near_far_from_aabb(o, d, aabb, 0.05)
    assert np.array_equal(nears.cpu().numpy(), en) and np.array_equal(fars.cpu().numpy(), ef)

    # raymarching/raymarching.py:380-395 (march_rays) and :433 (composite_rays)
    n_alive, n_step = N, 2
    M = n_alive * n_step
    M += 128 - (M % 128)
    xyzs = torch.zeros(M, 3, device=dev); dirs = torch.zeros(M, 3, device=dev); deltas = torch.zeros(M, 2, device=dev)
    noises = torch.zeros(n_alive, device=dev)
    alive = torch.arange(N, dtype=torch.int32, device=dev); rays_t = nears.clone(); tbits = t(bits)
    raymarching_backend.march_rays(n_alive, n_step, alive, rays_t, rays_o, rays_d, 1.0, 1 / 256, 16, 1, 128, tbits, nears, fars,
                                   xyzs, dirs, deltas, noises)
    ex, ed, edl = po.march_rays(n_alive, n_step, np.arange(N, dtype=np.int32), en, o, d, 1.0, 1 / 256, 16, 1, 128, bits, en, ef,
                                np.zeros(N, np.float32), M=M)
    assert np.array_equal(xyzs.cpu().numpy(), ex) and np.array_equal(deltas.cpu().numpy(), edl)

    # gridencoder/grid.py:47-54 ([L,B,C] outputs) and :75-84 (backward)
    D, C, L = 3, 2, 16
    pls = np.exp2(np.log2(2048 / 16) / 15)
    offsets = level_offsets(D, L, pls, 16, 16, False)
    emb = rng.uniform(-0.5, 0.5, (int(offsets[-1]), C)).astype(np.float32)
    B = 2000
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    tx, temb, toff = t(x), t(emb), t(offsets)
    outputs = torch.empty(L, B, C, device=dev)
    dy_dx = torch.empty(B, L * D * C, device=dev)
    S = np.log2(pls)
    gridencoder_backend.grid_encode_forward(tx, temb, toff, outputs, B, D, C, L, S, 16, dy_dx, 1, False, 0)
    eo, edy = po.grid_encode_forward(x, emb, offsets, B, D, C, L, float(S), 16, True, 1, False, 0)
    assert np.array_equal(outputs.cpu().numpy(), eo) and np.array_equal(dy_dx.cpu().numpy(), edy)
    grad = t(rng.standard_normal((L, B, C)).astype(np.float32))
    g_emb = torch.zeros_like(temb); g_in = torch.zeros(B, D, device=dev)
    gridencoder_backend.grid_encode_backward(grad, tx, temb, toff, g_emb, B, D, C, L, S, 16, dy_dx, g_in, 1, False, 0)
    ege, egi = po.grid_encode_backward(grad.cpu().numpy(), x, emb, offsets, B, D, C, L, float(S), 16, edy, 1, False, 0)
    np.testing.assert_allclose(g_emb.cpu().numpy(), ege, rtol=1e-4, atol=1e-4)
    assert np.array_equal(g_in.cpu().numpy(), egi)

    # shencoder/sphere_harmonics.py:24-32, freqencoder/freq.py:26-28
    v = t(d[:1000])
    sh = torch.empty(1000, 16, device=dev)
    shencoder_backend.sh_encode_forward(v, sh, 1000, 3, 4, None)
    np.testing.assert_allclose(sh.cpu().numpy(), po.sh_encode_forward(d[:1000], 4)[0], rtol=2e-6, atol=2e-6)
    fq = torch.empty(1000, 3 + 3 * 2 * 4, device=dev)
    freqencoder_backend.freq_encode_forward(v, 1000, 3, 4, 27, fq)
    np.testing.assert_allclose(fq.cpu().numpy(), po.freq_encode_forward(d[:1000], 4), rtol=0, atol=4e-5)

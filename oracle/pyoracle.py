"""ctypes front-end of the CPU oracle (oracle/libradnerf_oracle.so).

TEST INFRASTRUCTURE ONLY (see oracle/radnerf_oracle.h): importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the
product package rad-nerf_amd/.

All functions take/return C-contiguous numpy arrays; argument order follows
the reference's pybind entry points (raymarching/src/raymarching.h:7-20,
gridencoder/src/gridencoder.h:12-15, shencoder/src/shencoder.h:9-10,
freqencoder/src/freqencoder.h:7-10).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libradnerf_oracle.so")
_SRCS = ["orc_raymarching.c", "orc_grid.c", "orc_sh.c", "orc_freq.c", "orc_nerf.c",
         "radnerf_oracle.h", "Makefile"]


def build(force=False):
    """(Re)build the shared object with oracle/Makefile when sources are newer."""
    stale = force or not os.path.exists(_SO)
    if not stale:
        t = os.path.getmtime(_SO)
        stale = any(os.path.getmtime(os.path.join(_HERE, s)) > t for s in _SRCS)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_grid_index.restype = C.c_uint32
        _lib.orc_float_to_half.restype = C.c_uint16
        _lib.orc_float_to_half.argtypes = [C.c_float]
        _lib.orc_half_to_float.restype = C.c_float
        _lib.orc_half_to_float.argtypes = [C.c_uint16]
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


u32, f32, i32 = C.c_uint32, C.c_float, C.c_int

# ----------------------------------------------------------------- raymarching


def near_far_from_aabb(rays_o, rays_d, aabb, min_near):
    rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = rays_o.shape[0]
    nears, fars = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().orc_sph_from_ray(_p(rays_o), _p(rays_d), f32(radius), u32(N), _p(coords))
    return coords


def morton3D(coords):
    coords = _i32(coords).reshape(-1, 3)
    N = coords.shape[0]
    out = np.empty(N, np.int32)
    lib().orc_morton3D(_p(coords), u32(N), _p(out))
    return out


def morton3D_invert(indices):
    indices = _i32(indices).reshape(-1)
    N = indices.shape[0]
    out = np.empty((N, 3), np.int32)
    lib().orc_morton3D_invert(_p(indices), u32(N), _p(out))
    return out


def packbits(grid, thresh):
    grid = _f32(grid)
    N = grid.size // 8
    out = np.empty(N, np.uint8)
    lib().orc_packbits(_p(grid), u32(N), f32(thresh), _p(out))
    return out


def morton3D_dilation(grid):
    grid = _f32(grid)
    Cc, H3 = grid.shape
    H = int(round(H3 ** (1.0 / 3.0)))
    out = np.empty_like(grid)
    lib().orc_morton3D_dilation(_p(grid), u32(Cc), u32(H), _p(out))
    return out


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, Cc, H, M, nears, fars, noises,
                     counter=None):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    grid = np.ascontiguousarray(grid, dtype=np.uint8)
    nears, fars, noises = _f32(nears), _f32(fars), _f32(noises)
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    rays = np.empty((N, 3), np.int32)
    if counter is None:
        counter = np.zeros(2, np.int32)
    lib().orc_march_rays_train(_p(rays_o), _p(rays_d), _p(grid), f32(bound), f32(dt_gamma), u32(max_steps),
                               u32(N), u32(Cc), u32(H), u32(M), _p(nears), _p(fars), _p(xyzs), _p(dirs),
                               _p(deltas), _p(rays), _p(counter), _p(noises))
    return xyzs, dirs, deltas, rays, counter


def march_rays_train_backward(grad_xyzs, grad_dirs, rays, deltas):
    grad_xyzs, grad_dirs, deltas = _f32(grad_xyzs), _f32(grad_dirs), _f32(deltas)
    rays = _i32(rays)
    N, M = rays.shape[0], grad_xyzs.shape[0]
    go, gd = np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32)
    lib().orc_march_rays_train_backward(_p(grad_xyzs), _p(grad_dirs), _p(rays), _p(deltas), u32(N), u32(M),
                                        _p(go), _p(gd))
    return go, gd


def composite_rays_train_forward(sigmas, rgbs, ambient, deltas, rays, T_thresh):
    sigmas, rgbs, ambient, deltas = _f32(sigmas), _f32(rgbs), _f32(ambient), _f32(deltas)
    rays = _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ws, am, dp = np.empty(N, np.float32), np.empty(N, np.float32), np.empty(N, np.float32)
    im = np.empty((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(ambient), _p(deltas), _p(rays), u32(M),
                                           u32(N), f32(T_thresh), _p(ws), _p(am), _p(dp), _p(im))
    return ws, am, dp, im


def composite_rays_train_backward(g_ws, g_am, g_im, sigmas, rgbs, ambient, deltas, rays, ws, am, im, T_thresh):
    g_ws, g_am, g_im = _f32(g_ws), _f32(g_am), _f32(g_im)
    sigmas, rgbs, ambient, deltas = _f32(sigmas), _f32(rgbs), _f32(ambient), _f32(deltas)
    rays, ws, am, im = _i32(rays), _f32(ws), _f32(am), _f32(im)
    M, N = sigmas.shape[0], rays.shape[0]
    gs, gr, ga = np.zeros_like(sigmas), np.zeros_like(rgbs), np.zeros_like(ambient)
    lib().orc_composite_rays_train_backward(_p(g_ws), _p(g_am), _p(g_im), _p(sigmas), _p(rgbs), _p(ambient),
                                            _p(deltas), _p(rays), _p(ws), _p(am), _p(im), u32(M), u32(N),
                                            f32(T_thresh), _p(gs), _p(gr), _p(ga))
    return gs, gr, ga


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, Cc, H, grid,
               nears, fars, noises, M=None):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    rays_alive, rays_t = _i32(rays_alive), _f32(rays_t)
    grid = np.ascontiguousarray(grid, dtype=np.uint8)
    nears, fars, noises = _f32(nears), _f32(fars), _f32(noises)
    if M is None:
        M = n_alive * n_step
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    lib().orc_march_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d),
                         f32(bound), f32(dt_gamma), u32(max_steps), u32(Cc), u32(H), _p(grid), _p(nears),
                         _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(noises))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth,
                   image):
    """In place on rays_alive, rays_t, weights_sum, depth, image (must be writable C-contiguous arrays)."""
    for a, dt in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32),
                  (depth, np.float32), (image, np.float32)):
        assert a.dtype == dt and a.flags.c_contiguous and a.flags.writeable
    sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
    lib().orc_composite_rays(u32(n_alive), u32(n_step), f32(T_thresh), _p(rays_alive), _p(rays_t),
                             _p(sigmas), _p(rgbs), _p(deltas), _p(weights_sum), _p(depth), _p(image))


# ----------------------------------------------------------------- gridencoder


def _half_view(a):
    return np.ascontiguousarray(a, dtype=np.float16).view(np.uint16)


def grid_encode_forward(inputs, embeddings, offsets, B, D, Cc, L, S, H, calc_dy_dx, gridtype, align_corners,
                        interp, half=False):
    inputs = _f32(inputs).reshape(B, D)
    offsets = _i32(offsets)
    if half:
        emb = _half_view(embeddings)
        out = np.empty((L, B, Cc), np.uint16)
        dy = np.empty((B, L * D * Cc), np.uint16) if calc_dy_dx else None
    else:
        emb = _f32(embeddings)
        out = np.empty((L, B, Cc), np.float32)
        dy = np.empty((B, L * D * Cc), np.float32) if calc_dy_dx else None
    lib().orc_grid_encode_forward(_p(inputs), _p(emb), _p(offsets), _p(out), u32(B), u32(D), u32(Cc), u32(L),
                                  f32(S), u32(H), _p(dy), u32(gridtype), i32(int(align_corners)), u32(interp),
                                  i32(int(half)))
    if half:
        out = out.view(np.float16)
        dy = None if dy is None else dy.view(np.float16)
    return out, dy


def grid_encode_backward(grad, inputs, embeddings, offsets, B, D, Cc, L, S, H, dy_dx, gridtype, align_corners,
                         interp, half=False):
    inputs = _f32(inputs).reshape(B, D)
    offsets = _i32(offsets)
    if half:
        grad_ = _half_view(grad)
        emb = _half_view(embeddings)
        g_emb = np.zeros(emb.shape, np.uint16)
        dy = None if dy_dx is None else _half_view(dy_dx)
        g_in = None if dy_dx is None else np.zeros((B, D), np.uint16)
    else:
        grad_ = _f32(grad)
        emb = _f32(embeddings)
        g_emb = np.zeros(emb.shape, np.float32)
        dy = None if dy_dx is None else _f32(dy_dx)
        g_in = None if dy_dx is None else np.zeros((B, D), np.float32)
    lib().orc_grid_encode_backward(_p(grad_), _p(inputs), _p(emb), _p(offsets), _p(g_emb), u32(B), u32(D),
                                   u32(Cc), u32(L), f32(S), u32(H), _p(dy), _p(g_in), u32(gridtype),
                                   i32(int(align_corners)), u32(interp), i32(int(half)))
    if half:
        g_emb = g_emb.view(np.float16)
        g_in = None if g_in is None else g_in.view(np.float16)
    return g_emb, g_in


def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, Cc, L, S, H, gridtype,
                         align_corners):
    """Adds into `grad` in place."""
    inputs, embeddings, offsets = _f32(inputs), _f32(embeddings), _i32(offsets)
    assert grad.dtype == np.float32 and grad.flags.c_contiguous
    lib().orc_grad_total_variation(_p(inputs), _p(embeddings), _p(grad), _p(offsets), f32(weight), u32(B),
                                   u32(D), u32(Cc), u32(L), f32(S), u32(H), u32(gridtype),
                                   i32(int(align_corners)))


def grid_index(D, Cc, gridtype, align_corners, ch, hashmap_size, resolution, pos_grid):
    pg = np.ascontiguousarray(pos_grid, dtype=np.uint32)
    return int(lib().orc_grid_index(u32(D), u32(Cc), u32(gridtype), i32(int(align_corners)), u32(ch),
                                    u32(hashmap_size), u32(resolution), _p(pg)))


# ----------------------------------------------------------------- occupancy-grid maintenance (orc_occupancy.c)


def occupancy_points(Cc, H, bound, noise=None, seed=0):
    xyzs = np.empty((Cc * H ** 3, 3), np.float32)
    nz = None if noise is None else _f32(noise)
    lib().orc_occupancy_points(u32(Cc), u32(H), f32(bound), _p(nz), u32(seed), _p(xyzs))
    return xyzs


def occupancy_update(sigmas, density_scale, grid, Cc, H, decay, density_thresh):
    """In place on `grid` (float32 [C, H^3]); returns (bitfield uint8 [C*H^3/8], mean_density, threshold)."""
    assert grid.dtype == np.float32 and grid.flags.c_contiguous and grid.flags.writeable
    sigmas = _f32(sigmas)
    bits = np.empty(Cc * H ** 3 // 8, np.uint8)
    stats = np.zeros(2, np.float32)
    lib().orc_occupancy_update(_p(sigmas), f32(density_scale), _p(grid), u32(Cc), u32(H), f32(decay), f32(density_thresh), _p(bits),
                               _p(stats))
    return bits, float(stats[0]), float(stats[1])


def mark_untrained_grid(poses, intrinsic, Cc, H, bound, grid):
    """In place on `grid`."""
    assert grid.dtype == np.float32 and grid.flags.c_contiguous and grid.flags.writeable
    poses = _f32(poses)
    stride = poses.shape[1] * poses.shape[2]
    fx, fy, cx, cy = (float(v) for v in intrinsic)
    lib().orc_mark_untrained_grid(_p(poses), u32(poses.shape[0]), u32(stride), C.c_double(fx), C.c_double(fy), C.c_double(cx),
                                  C.c_double(cy), u32(Cc), u32(H), f32(bound), _p(grid))


def torso_grid_points(H, noise=None, seed=0):
    xys = np.empty((H * H, 2), np.float32)
    nz = None if noise is None else _f32(noise)
    lib().orc_torso_grid_points(u32(H), _p(nz), u32(seed), _p(xys))
    return xys


def torso_grid_update(alphas, grid, H, decay):
    """In place on `grid`; returns mean_density_torso."""
    assert grid.dtype == np.float32 and grid.flags.c_contiguous and grid.flags.writeable
    alphas = _f32(alphas)
    stats = np.zeros(1, np.float32)
    lib().orc_torso_grid_update(_p(alphas), _p(grid), u32(H), f32(decay), _p(stats))
    return float(stats[0])


# ----------------------------------------------------------------- sh / freq


def sh_encode_forward(inputs, degree, calc_dy_dx=False):
    inputs = _f32(inputs).reshape(-1, 3)
    B = inputs.shape[0]
    out = np.empty((B, degree * degree), np.float32)
    dy = np.empty((B, 3 * degree * degree), np.float32) if calc_dy_dx else None
    lib().orc_sh_encode_forward(_p(inputs), _p(out), u32(B), u32(3), u32(degree), _p(dy))
    return out, dy


def sh_encode_backward(grad, inputs, degree, dy_dx):
    grad, inputs, dy_dx = _f32(grad), _f32(inputs).reshape(-1, 3), _f32(dy_dx)
    B = inputs.shape[0]
    gi = np.zeros((B, 3), np.float32)
    lib().orc_sh_encode_backward(_p(grad), _p(inputs), u32(B), u32(3), u32(degree), _p(dy_dx), _p(gi))
    return gi


def freq_encode_forward(inputs, degree):
    inputs = _f32(inputs)
    B, D = inputs.shape
    Cc = D + D * 2 * degree
    out = np.empty((B, Cc), np.float32)
    lib().orc_freq_encode_forward(_p(inputs), u32(B), u32(D), u32(degree), u32(Cc), _p(out))
    return out


def freq_encode_backward(grad, outputs, D, degree):
    grad, outputs = _f32(grad), _f32(outputs)
    B, Cc = outputs.shape
    gi = np.empty((B, D), np.float32)
    lib().orc_freq_encode_backward(_p(grad), _p(outputs), u32(B), u32(D), u32(degree), u32(Cc), _p(gi))
    return gi


# ----------------------------------------------------------------- network / renderer


class _Grid(C.Structure):
    _fields_ = [("embeddings", C.c_void_p), ("offsets", C.c_void_p), ("D", u32), ("C", u32), ("L", u32),
                ("H", u32), ("S", f32), ("gridtype", u32), ("align_corners", i32), ("interp", u32)]


class _MLP(C.Structure):
    _fields_ = [("num_layers", u32), ("dim_in", u32), ("dim_hidden", u32), ("dim_out", u32),
                ("weights", C.c_void_p * 4)]


class _Model(C.Structure):
    _fields_ = [("enc_xyz", _Grid), ("enc_ambient", _Grid), ("ambient_net", _MLP), ("sigma_net", _MLP),
                ("color_net", _MLP), ("audio_dim", u32), ("ind_dim", u32), ("sh_degree", u32),
                ("has_eye", i32), ("bound", f32), ("enc_torso", _Grid), ("torso_deform_net", _MLP),
                ("torso_net", _MLP), ("ind_dim_torso", u32), ("torso_shrink", f32)]


class _RenderCfg(C.Structure):
    _fields_ = [("density_bitfield", C.c_void_p), ("cascade", u32), ("grid_size", u32), ("bound", f32),
                ("min_near", f32), ("aabb_infer", f32 * 6), ("dt_gamma", f32), ("max_steps", u32),
                ("T_thresh", f32), ("torso", i32), ("density_grid_torso", C.c_void_p),
                ("density_thresh_torso", f32), ("mean_density_torso", f32)]


class Model:
    """Holds numpy copies of every parameter + the C struct that points at them.

    `params` is a dict of numpy arrays keyed like the reference state_dict
    (encoder.embeddings, encoder.offsets, sigma_net.net.0.weight, ...) plus a
    `cfg` dict: {per_level_scale_xyz/_ambient/_torso, base_resolution, gridtype, bound,
    has_eye, ind_dim, ind_dim_torso, torso_shrink, sh_degree}.
    """

    def __init__(self, params, cfg):
        self._keep = []
        self.cfg = dict(cfg)
        m = _Model()
        m.enc_xyz = self._grid(params, "encoder", 3, cfg["per_level_scale_xyz"], cfg)
        m.enc_ambient = self._grid(params, "encoder_ambient", 2, cfg["per_level_scale_ambient"], cfg)
        m.ambient_net = self._mlp(params, "ambient_net")
        m.sigma_net = self._mlp(params, "sigma_net")
        m.color_net = self._mlp(params, "color_net")
        m.audio_dim = cfg.get("audio_dim", 64)
        m.ind_dim = cfg.get("ind_dim", 4)
        m.sh_degree = cfg.get("sh_degree", 4)
        m.has_eye = int(cfg.get("has_eye", True))
        m.bound = cfg.get("bound", 1.0)
        if "torso_encoder.embeddings" in params:
            m.enc_torso = self._grid(params, "torso_encoder", 2, cfg["per_level_scale_torso"], cfg)
            m.torso_deform_net = self._mlp(params, "torso_deform_net")
            m.torso_net = self._mlp(params, "torso_net")
        m.ind_dim_torso = cfg.get("ind_dim_torso", 8)
        m.torso_shrink = cfg.get("torso_shrink", 0.8)
        self.c = m

    def _grid(self, params, name, D, per_level_scale, cfg):
        emb = _f32(params[name + ".embeddings"])
        off = _i32(params[name + ".offsets"])
        self._keep += [emb, off]
        g = _Grid()
        g.embeddings, g.offsets = emb.ctypes.data, off.ctypes.data
        g.D, g.C, g.L = D, emb.shape[1], off.shape[0] - 1
        g.H = cfg.get("base_resolution", 16)
        g.S = float(np.log2(per_level_scale))
        g.gridtype = cfg.get("gridtype_" + name, cfg.get("gridtype", 1))
        g.align_corners = 0
        g.interp = 0
        return g

    def _mlp(self, params, name):
        ws = []
        l = 0
        while f"{name}.net.{l}.weight" in params:
            ws.append(_f32(params[f"{name}.net.{l}.weight"]))
            l += 1
        self._keep += ws
        mlp = _MLP()
        mlp.num_layers = len(ws)
        mlp.dim_in = ws[0].shape[1]
        mlp.dim_hidden = ws[0].shape[0]
        mlp.dim_out = ws[-1].shape[0]
        for i, w in enumerate(ws):
            mlp.weights[i] = w.ctypes.data
        return mlp


def mlp_forward(weights, x):
    ws = [_f32(w) for w in weights]
    x = _f32(x)
    mlp = _MLP()
    mlp.num_layers = len(ws)
    mlp.dim_in, mlp.dim_hidden, mlp.dim_out = ws[0].shape[1], ws[0].shape[0], ws[-1].shape[0]
    for i, w in enumerate(ws):
        mlp.weights[i] = w.ctypes.data
    out = np.empty((x.shape[0], mlp.dim_out), np.float32)
    lib().orc_mlp_forward(C.byref(mlp), _p(x), u32(x.shape[0]), _p(out))
    return out


def nerf_forward(model, xyzs, dirs, enc_a, ind_code, eye, mlp_dtype="f32"):
    """mlp_dtype="f16": the arithmetic of the opt-in 16-bit matrix-core kernel (orc_nerf_forward_mp16)."""
    xyzs, dirs = _f32(xyzs), _f32(dirs)
    enc_a, ind_code, eye = _f32(enc_a).reshape(-1), _f32(ind_code).reshape(-1), _f32(eye).reshape(-1)
    M = xyzs.shape[0]
    sigma, color, amb = np.empty(M, np.float32), np.empty((M, 3), np.float32), np.empty((M, 2), np.float32)
    fn = lib().orc_nerf_forward_mp16 if mlp_dtype == "f16" else lib().orc_nerf_forward
    fn(C.byref(model.c), _p(xyzs), _p(dirs), u32(M), _p(enc_a), _p(ind_code), _p(eye), _p(sigma), _p(color), _p(amb))
    return sigma, color, amb


def nerf_density(model, xyzs, enc_a, eye):
    xyzs, enc_a, eye = _f32(xyzs), _f32(enc_a).reshape(-1), _f32(eye).reshape(-1)
    M = xyzs.shape[0]
    sigma = np.empty(M, np.float32)
    lib().orc_nerf_density(C.byref(model.c), _p(xyzs), u32(M), _p(enc_a), _p(eye), _p(sigma))
    return sigma


def torso_forward(model, x, poses6, ind_code_torso):
    x, poses6, c = _f32(x), _f32(poses6).reshape(-1), _f32(ind_code_torso).reshape(-1)
    P = x.shape[0]
    alpha, color, dx = np.empty((P, 1), np.float32), np.empty((P, 3), np.float32), np.empty((P, 2), np.float32)
    lib().orc_torso_forward(C.byref(model.c), _p(x), u32(P), _p(poses6), _p(c), _p(alpha), _p(color), _p(dx))
    return alpha, color, dx


def render_frame(model, rcfg, rays_o, rays_d, enc_a, ind_code, eye, bg_coords, poses6, ind_code_torso,
                 bg_color):
    """rcfg: dict(density_bitfield, cascade, grid_size, bound, min_near, aabb_infer, dt_gamma, max_steps,
    T_thresh, torso, density_grid_torso, density_thresh_torso, mean_density_torso)."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    bits = np.ascontiguousarray(rcfg["density_bitfield"], dtype=np.uint8)
    tgrid = _f32(rcfg.get("density_grid_torso", np.zeros(rcfg["grid_size"] ** 2)))
    c = _RenderCfg()
    c.density_bitfield = bits.ctypes.data
    c.cascade, c.grid_size = rcfg["cascade"], rcfg["grid_size"]
    c.bound, c.min_near = rcfg["bound"], rcfg["min_near"]
    for i, v in enumerate(rcfg["aabb_infer"]):
        c.aabb_infer[i] = float(v)
    c.dt_gamma, c.max_steps, c.T_thresh = rcfg["dt_gamma"], rcfg["max_steps"], rcfg["T_thresh"]
    c.torso = int(rcfg.get("torso", True))
    c.density_grid_torso = tgrid.ctypes.data
    c.density_thresh_torso = rcfg.get("density_thresh_torso", 0.01)
    c.mean_density_torso = rcfg.get("mean_density_torso", 0.0)
    enc_a, ind_code, eye = _f32(enc_a).reshape(-1), _f32(ind_code).reshape(-1), _f32(eye).reshape(-1)
    bg_coords, poses6 = _f32(bg_coords).reshape(-1, 2), _f32(poses6).reshape(-1)
    ict = _f32(ind_code_torso).reshape(-1)
    bg_color = _f32(np.broadcast_to(bg_color, (N, 3)))
    image, depth = np.empty((N, 3), np.float32), np.empty(N, np.float32)
    stats = np.zeros(4, np.uint64)
    lib().orc_render_frame(C.byref(model.c), C.byref(c), _p(rays_o), _p(rays_d), u32(N), _p(enc_a), _p(ind_code),
                           _p(eye), _p(bg_coords), _p(poses6), _p(ict), _p(bg_color), _p(image), _p(depth),
                           _p(stats))
    return image, depth, dict(iterations=int(stats[0]), live_samples=int(stats[1]),
                              sample_slots=int(stats[2]), torso_pixels=int(stats[3]))


def num_threads():
    return int(lib().orc_num_threads())


def model_from_module(net):
    """Build an oracle Model from a NeRFNetwork-like torch module (reference's or this tree's):
    reads its state_dict (numpy copies) and the few scalar attributes the oracle needs."""
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    cfg = dict(per_level_scale_xyz=float(net.encoder.per_level_scale),
               per_level_scale_ambient=float(net.encoder_ambient.per_level_scale),
               base_resolution=int(net.encoder.base_resolution), gridtype=int(net.encoder_ambient.gridtype_id),
               gridtype_encoder=int(net.encoder.gridtype_id),
               bound=float(net.bound), has_eye=bool(net.exp_eye), ind_dim=int(net.individual_dim),
               audio_dim=int(net.audio_dim), sh_degree=int(net.encoder_dir.degree))
    if getattr(net, "torso", False):
        cfg.update(per_level_scale_torso=float(net.torso_encoder.per_level_scale),
                   ind_dim_torso=int(net.individual_dim_torso), torso_shrink=float(net.opt.torso_shrink))
    return Model(sd, cfg)


def render_cfg_from_module(net, dt_gamma, max_steps, T_thresh=1e-4):
    rc = dict(density_bitfield=net.density_bitfield.detach().cpu().numpy(), cascade=int(net.cascade),
              grid_size=int(net.grid_size), bound=float(net.bound), min_near=float(net.min_near),
              aabb_infer=net.aabb_infer.detach().cpu().numpy().tolist(), dt_gamma=float(dt_gamma),
              max_steps=int(max_steps), T_thresh=float(T_thresh), torso=bool(net.torso))
    if net.torso:
        rc.update(density_grid_torso=net.density_grid_torso.detach().cpu().numpy(),
                  density_thresh_torso=float(net.density_thresh_torso),
                  mean_density_torso=float(net.mean_density_torso))
    return rc

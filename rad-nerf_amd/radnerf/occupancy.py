"""Occupancy-grid maintenance on the device: what NeRFRenderer.update_extra_state and
mark_untrained_grid do (reference: nerf/renderer.py:318-499), as a handful of launches of libradnerf_hip.so with no Python
loop over blocks of cells, no index tensor, no scatter and no host read-back.

The cells are enumerated in MORTON order by the kernels (include/radnerf_fused.h, "occupancy-grid maintenance"), so slot i of
the probe-point buffer, of the sigma buffer and of `density_grid[c]` is the same cell:

    head   rn_occupancy_points -> density query (fused network kernel, sigma branch only; or `model.density` in chunks when the
           network is not of the fused shape) -> rn_occupancy_update (dilation, decayed max, mean, threshold, bit packing)
    torso  rn_torso_grid_points -> rn_torso_fused with the occupancy test disabled (alpha of every probe point)
           -> rn_torso_grid_update (5 x 5 max pool, decayed max, mean)
    mark   rn_mark_untrained_grid over all cameras at once

`mean_density` / `mean_density_torso` stay in device memory; the renderer reads them lazily the first time Python needs the
number (checkpoint, torso threshold of the next frame).
"""
import ctypes as C
import random

import torch

import radnerf_hip as hip

_lib = hip._lib
_u32, _f32, _ptr = C.c_uint32, C.c_float, C.c_void_p
_SIGS = {
    "rn_occupancy_points": [_u32, _u32, _f32, _ptr, _u32, _ptr, _ptr],
    "rn_occupancy_update": [_ptr, _f32, _ptr, _u32, _u32, _f32, _f32, _ptr, _ptr, _ptr, _ptr],
    "rn_mark_untrained_grid": [_ptr, _u32, _u32, C.c_double, C.c_double, C.c_double, C.c_double, _u32, _u32, _f32, _ptr, _ptr],
    "rn_torso_grid_points": [_u32, _ptr, _u32, _ptr, _ptr],
    "rn_torso_grid_update": [_ptr, _ptr, _u32, _f32, _ptr, _ptr],
    "rn_torso_mask": [_ptr, _u32, _ptr, _u32, _f32, _ptr, _ptr],
}
for _n, _a in _SIGS.items():
    getattr(_lib, _n).argtypes = _a
    getattr(_lib, _n).restype = C.c_int
_lib.rn_occupancy_workspace.argtypes = [_u32, _u32]
_lib.rn_occupancy_workspace.restype = C.c_size_t
_lib.rn_hash_u01_bits.argtypes = [_u32, _u32]
_lib.rn_hash_u01_bits.restype = _u32


def exported_symbols():
    return sorted(list(_SIGS) + ["rn_occupancy_workspace", "rn_hash_u01_bits"])


class _Scratch:
    """Per-model buffers of the refresh: probe points, sigmas, statistics, the partial-sum workspace."""

    def __init__(self, model):
        dev = model.density_bitfield.device
        n = model.cascade * model.grid_size ** 3
        self.xyzs = torch.empty(n, 3, dtype=torch.float32, device=dev)
        self.sigmas = torch.empty(n, dtype=torch.float32, device=dev)
        self.stats = torch.zeros(2, dtype=torch.float32, device=dev)
        self.ws = torch.zeros(int(_lib.rn_occupancy_workspace(model.cascade, model.grid_size)), dtype=torch.uint8, device=dev)
        if model.torso:
            g2 = model.grid_size ** 2
            self.xys = torch.empty(g2, 2, dtype=torch.float32, device=dev)
            self.alphas = torch.empty(g2, 1, dtype=torch.float32, device=dev)
            self.bg = torch.empty(g2, 3, dtype=torch.float32, device=dev)
            self.stats_torso = torch.zeros(1, dtype=torch.float32, device=dev)


def _scratch(model):
    sc = getattr(model, "_occ_scratch", None)
    if sc is None or sc.stats.device != model.density_bitfield.device:
        sc = _Scratch(model)
        object.__setattr__(model, "_occ_scratch", sc)
    return sc


def _jitter(noise, n, dims, dev):
    """(noise pointer or None, seed): the caller's uniform numbers, else the kernels' counter-based hash with a fresh seed."""
    if noise is None:
        return None, random.getrandbits(32)
    noise = noise.to(dev, torch.float32).contiguous()
    if noise.numel() != n * dims:
        raise ValueError(f"jitter noise must hold {n} x {dims} numbers")
    return noise, 0


def _density_into(model, xyzs, sigmas, enc_a, eye):
    """sigma of every probe point (nerf/renderer.py:438: self.density(...)['sigma'])."""
    from . import fused
    if getattr(model.opt, "grid_refresh_engine", "fused") == "fused" and fused.supported(model):
        fused.density_forward(model, xyzs, enc_a, eye, out=sigmas)
        return
    step = 1 << 19
    for lo in range(0, xyzs.shape[0], step):       # a network that is not of the fused shape: PyTorch layers, in chunks
        sigmas[lo:lo + step] = model.density(xyzs[lo:lo + step], enc_a, eye)["sigma"].reshape(-1).float()


@torch.no_grad()
def refresh_head(model, enc_a, eye, decay=0.95, noise=None):
    """The 3-D grid and its bitfield (nerf/renderer.py:408-449).  Everything is enqueued on the current stream; the new
    mean density stays on the device (model._mean_density_dev)."""
    sc = _scratch(model)
    Cc, H = int(model.cascade), int(model.grid_size)
    nz, seed = _jitter(noise, Cc * H ** 3, 3, sc.xyzs.device)
    s = hip.stream()
    hip.call("rn_occupancy_points", Cc, H, float(model.bound), hip.ptr(nz), seed, hip.ptr(sc.xyzs), s)
    _density_into(model, sc.xyzs, sc.sigmas, enc_a, eye)
    grid = model.density_grid
    if not grid.is_contiguous():
        raise RuntimeError("density_grid must be contiguous")
    hip.call("rn_occupancy_update", hip.ptr(sc.sigmas), float(model.density_scale), hip.ptr(grid), Cc, H, float(decay),
             float(model.density_thresh), hip.ptr(model.density_bitfield), hip.ptr(sc.stats), hip.ptr(sc.ws), s)
    return sc.stats


@torch.no_grad()
def refresh_torso(model, enc_a, pose6, ind_code, decay=0.95, noise=None):
    """The 2-D torso grid (nerf/renderer.py:451-490); its mean stays on the device."""
    from . import fused
    sc = _scratch(model)
    H = int(model.grid_size)
    nz, seed = _jitter(noise, H * H, 2, sc.xys.device)
    s = hip.stream()
    hip.call("rn_torso_grid_points", H, hip.ptr(nz), seed, hip.ptr(sc.xys), s)
    if fused.supported(model):
        # thresh = -1: every probe point passes the occupancy test, alpha_out = forward_torso(...)[0] of each
        fused.torso_forward(model, sc.xys, pose6, ind_code, thresh=-1.0, bg_out=sc.bg, alpha_out=sc.alphas)
    else:
        sc.alphas.copy_(model.forward_torso(sc.xys, pose6, enc_a, ind_code)[0].float())
    hip.call("rn_torso_grid_update", hip.ptr(sc.alphas), hip.ptr(model.density_grid_torso), H, float(decay),
             hip.ptr(sc.stats_torso), s)
    return sc.stats_torso


@torch.no_grad()
def mark_untrained(model, poses, intrinsic):
    """density_grid = -1 where no training camera sees the cell (nerf/renderer.py:318-379): all cascades, all cameras, one launch."""
    dev = model.density_bitfield.device
    poses = torch.as_tensor(poses, dtype=torch.float32).to(dev)
    if poses.dim() != 3 or poses.shape[1] not in (3, 4) or poses.shape[2] != 4:
        raise ValueError("poses must be [B, 4, 4] or [B, 3, 4] cam2world matrices")
    poses = poses.contiguous()
    fx, fy, cx, cy = (float(v) for v in intrinsic)
    hip.call("rn_mark_untrained_grid", hip.ptr(poses), int(poses.shape[0]), int(poses.shape[1] * 4), fx, fy, cx, cy,
             int(model.cascade), int(model.grid_size), float(model.bound), hip.ptr(model.density_grid), hip.stream())


def torso_pixels(model, bg_coords, thresh):
    """Indices of the pixels the torso layer covers (bilinear occupancy > thresh, nerf/renderer.py:281-283), ascending."""
    mask = torch.empty(bg_coords.shape[0], dtype=torch.uint8, device=bg_coords.device)
    coords = bg_coords.contiguous().float()
    hip.call("rn_torso_mask", hip.ptr(coords), int(coords.shape[0]), hip.ptr(model.density_grid_torso), int(model.grid_size),
             float(thresh), hip.ptr(mask), hip.stream())
    return torch.nonzero(mask).reshape(-1)

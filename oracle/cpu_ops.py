"""CPU operator set for the "reference-flow" CPU baseline (TEST / BENCH INFRASTRUCTURE ONLY -- see radnerf_oracle.h).

SURVEY 8(d) defines the CPU baseline as: the reference's Python control flow over CPU restatements of the kernels, with
PyTorch CPU GEMMs for the MLPs, all host cores.  The reference's files do not travel to the GPU box, so the control flow is
this tree's mirror of it (radnerf/network.py + radnerf/renderer.py, "ops" engine: the reference's loop shape, one operator
call per stage); this module supplies the operators on CPU tensors from the oracle and swaps them into a model:

    with cpu_ops.cpu_operators(model):      # model lives on the CPU
        out = model.render(...)

Only bench.py's cpu_baseline leg and tests use it.  Nothing in rad-nerf_amd/ imports it, and the product path has no CPU route.
"""
import contextlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import pyoracle as po


def _n(t):
    return t.detach().cpu().numpy()


class _RM:
    """`raymarching` functions the inference loop calls (nerf/renderer.py:183, 251, 256), oracle-backed."""

    @staticmethod
    def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
        n, f = po.near_far_from_aabb(_n(rays_o), _n(rays_d), _n(aabb), min_near)
        return torch.from_numpy(n), torch.from_numpy(f)

    @staticmethod
    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
                   perturb=False, dt_gamma=0, max_steps=1024):
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        x, d, dl = po.march_rays(n_alive, n_step, _n(rays_alive), _n(rays_t), _n(rays_o), _n(rays_d), bound, dt_gamma, max_steps, C, H,
                                 _n(density_bitfield), _n(near), _n(far), np.zeros(n_alive, np.float32), M=M)
        return torch.from_numpy(x), torch.from_numpy(d), torch.from_numpy(dl)

    @staticmethod
    def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
        po.composite_rays(n_alive, n_step, T_thresh, rays_alive.numpy(), rays_t.numpy(), _n(sigmas.float()), _n(rgbs.float()),
                          _n(deltas), weights_sum.numpy(), depth.numpy(), image.numpy())
        return tuple()


class _Grid(nn.Module):
    def __init__(self, enc):
        super().__init__()
        self.e = enc
        self.output_dim = enc.output_dim

    def forward(self, inputs, bound=1):
        e = self.e
        x = _n(((inputs + bound) / (2 * bound)).reshape(-1, e.input_dim).float())
        out, _ = po.grid_encode_forward(x, _n(e.embeddings), _n(e.offsets), x.shape[0], e.input_dim, e.level_dim, e.num_levels,
                                        float(np.log2(e.per_level_scale)), e.base_resolution, False, e.gridtype_id, e.align_corners, 0)
        return torch.from_numpy(np.ascontiguousarray(out.transpose(1, 0, 2)).reshape(x.shape[0], -1))


class _SH(nn.Module):
    def __init__(self, enc):
        super().__init__()
        self.degree, self.output_dim = enc.degree, enc.output_dim

    def forward(self, inputs, size=1):
        return torch.from_numpy(po.sh_encode_forward(_n((inputs / size).reshape(-1, 3).float()), self.degree)[0])


class _Freq(nn.Module):
    def __init__(self, enc):
        super().__init__()
        self.input_dim, self.degree, self.output_dim = enc.input_dim, enc.degree, enc.output_dim

    def forward(self, inputs, **kw):
        return torch.from_numpy(po.freq_encode_forward(_n(inputs.reshape(-1, self.input_dim).float()), self.degree))


def _torso_pixels(model, bg_coords, thresh):
    # nerf/renderer.py:281-283
    occ = F.grid_sample(model.density_grid_torso.view(1, 1, model.grid_size, model.grid_size), bg_coords.view(1, -1, 1, 2),
                        align_corners=True).view(-1)
    return torch.nonzero(occ > thresh).reshape(-1)


@contextlib.contextmanager
def cpu_operators(model):
    """Swap the model's encoders and the renderer's operator module for oracle-backed CPU versions; restored on exit."""
    import radnerf.occupancy as occ_mod
    import radnerf.renderer as ren_mod
    names = ["encoder", "encoder_ambient", "encoder_dir", "torso_encoder", "torso_deform_encoder", "pose_encoder"]
    saved = {n: getattr(model, n) for n in names if hasattr(model, n)}
    saved_rm, saved_tp = ren_mod.raymarching, occ_mod.torso_pixels
    try:
        for n, enc in saved.items():
            wrap = _Grid(enc) if n in ("encoder", "encoder_ambient", "torso_encoder") else (_SH(enc) if n == "encoder_dir" else _Freq(enc))
            setattr(model, n, wrap)
        ren_mod.raymarching, occ_mod.torso_pixels = _RM, _torso_pixels
        yield model
    finally:
        for n, enc in saved.items():
            setattr(model, n, enc)
        ren_mod.raymarching, occ_mod.torso_pixels = saved_rm, saved_tp

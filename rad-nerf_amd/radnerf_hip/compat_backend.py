"""`_backend` objects with the EXACT function names and argument lists of the reference's four pybind11
modules, implemented over libradnerf_hip.so.

This is the binding a maintainer of the reference adds to switch it to MI355X without touching its Python
wrappers: in each `<pkg>/backend.py` replace the `torch.utils.cpp_extension.load(...)` of the CUDA sources by

    from radnerf_hip.compat_backend import raymarching_backend as _backend     # raymarching/backend.py
    from radnerf_hip.compat_backend import gridencoder_backend as _backend     # gridencoder/backend.py
    from radnerf_hip.compat_backend import shencoder_backend as _backend       # shencoder/backend.py
    from radnerf_hip.compat_backend import freqencoder_backend as _backend     # freqencoder/backend.py

The reference's raymarching/raymarching.py, gridencoder/grid.py, shencoder/sphere_harmonics.py and
freqencoder/freq.py then run unchanged (they call `_backend.<fn>(tensor, ..., scalars)`).
Signatures: raymarching/src/raymarching.h:7-20, gridencoder/src/gridencoder.h:12-15,
shencoder/src/shencoder.h:9-10, freqencoder/src/freqencoder.h:7-10 (tensors by value, scalars, void return).
Differences that stay invisible to those callers: kernels launch on torch's current stream, the sample
reservation of march_rays_train is a deterministic scan, errors surface as RuntimeError.
"""
import torch

from . import RN_F16, RN_F32, RN_LAYOUT_LBC, call, host_offsets, ptr, stream, workspace, workspace_bytes


def _dt(t):
    if t.dtype == torch.float32:
        return RN_F32
    if t.dtype == torch.float16:
        return RN_F16
    raise RuntimeError(f"grid tables must be float32 or float16, got {t.dtype}")


class raymarching_backend:
    @staticmethod
    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        call("rn_near_far_from_aabb", ptr(rays_o), ptr(rays_d), ptr(aabb), N, min_near, ptr(nears), ptr(fars), stream())

    @staticmethod
    def sph_from_ray(rays_o, rays_d, radius, N, coords):
        call("rn_sph_from_ray", ptr(rays_o), ptr(rays_d), radius, N, ptr(coords), stream())

    @staticmethod
    def morton3D(coords, N, indices):
        call("rn_morton3D", ptr(coords), N, ptr(indices), stream())

    @staticmethod
    def morton3D_invert(indices, N, coords):
        call("rn_morton3D_invert", ptr(indices), N, ptr(coords), stream())

    @staticmethod
    def packbits(grid, N, density_thresh, bitfield):
        call("rn_packbits", ptr(grid), N, density_thresh, ptr(bitfield), stream())

    @staticmethod
    def morton3D_dilation(grid, C, H, grid_dilation):
        call("rn_morton3D_dilation", ptr(grid), C, H, ptr(grid_dilation), stream())

    @staticmethod
    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                         rays, counter, noises):
        ws = workspace(workspace_bytes("rn_march_rays_train_workspace", N), rays_o.device)
        call("rn_march_rays_train", ptr(rays_o), ptr(rays_d), ptr(grid), bound, dt_gamma, max_steps, N, C, H, M, ptr(nears),
             ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays), ptr(counter), ptr(noises), ptr(ws), stream())

    @staticmethod
    def march_rays_train_backward(grad_xyzs, grad_dirs, rays, deltas, N, M, grad_rays_o, grad_rays_d):
        call("rn_march_rays_train_backward", ptr(grad_xyzs), ptr(grad_dirs), ptr(rays), ptr(deltas), N, M,
             ptr(grad_rays_o), ptr(grad_rays_d), stream())

    @staticmethod
    def composite_rays_train_forward(sigmas, rgbs, ambient, deltas, rays, M, N, T_thresh, weights_sum, ambient_sum, depth,
                                     image):
        call("rn_composite_rays_train_forward", ptr(sigmas), ptr(rgbs), ptr(ambient), ptr(deltas), ptr(rays), M, N, T_thresh,
             ptr(weights_sum), ptr(ambient_sum), ptr(depth), ptr(image), stream())

    @staticmethod
    def composite_rays_train_backward(grad_weights_sum, grad_ambient_sum, grad_image, sigmas, rgbs, ambient, deltas, rays,
                                      weights_sum, ambient_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs, grad_ambient):
        call("rn_composite_rays_train_backward", ptr(grad_weights_sum), ptr(grad_ambient_sum), ptr(grad_image), ptr(sigmas),
             ptr(rgbs), ptr(ambient), ptr(deltas), ptr(rays), ptr(weights_sum), ptr(ambient_sum), ptr(image), M, N, T_thresh,
             ptr(grad_sigmas), ptr(grad_rgbs), ptr(grad_ambient), stream())

    @staticmethod
    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars,
                   xyzs, dirs, deltas, noises):
        call("rn_march_rays", n_alive, n_step, ptr(rays_alive), ptr(rays_t), ptr(rays_o), ptr(rays_d), bound, dt_gamma,
             max_steps, C, H, ptr(grid), ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(noises), None, stream())

    @staticmethod
    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        call("rn_composite_rays", n_alive, n_step, T_thresh, ptr(rays_alive), ptr(rays_t), ptr(sigmas), ptr(rgbs),
             ptr(deltas), ptr(weights_sum), ptr(depth), ptr(image), None, stream())


class gridencoder_backend:
    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gridtype, align_corners, interp):
        call("rn_grid_encode_forward_ws", ptr(inputs), ptr(embeddings), ptr(offsets), host_offsets(offsets), ptr(outputs), B, D, C, L,
             float(S), H, ptr(dy_dx), gridtype, int(bool(align_corners)), interp, _dt(embeddings), RN_LAYOUT_LBC, None, 0, stream())

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs,
                             gridtype, align_corners, interp):
        call("rn_grid_encode_backward", ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets), ptr(grad_embeddings), B, D, C,
             L, float(S), H, ptr(dy_dx), ptr(grad_inputs), gridtype, int(bool(align_corners)), interp, _dt(grad),
             RN_LAYOUT_LBC, stream())

    @staticmethod
    def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, C, L, S, H, gridtype, align_corners):
        call("rn_grad_total_variation", ptr(inputs), ptr(embeddings), ptr(grad), ptr(offsets), weight, B, D, C, L, float(S), H,
             gridtype, int(bool(align_corners)), stream())


class shencoder_backend:
    @staticmethod
    def sh_encode_forward(inputs, outputs, B, D, C, dy_dx):
        call("rn_sh_encode_forward", ptr(inputs), ptr(outputs), B, D, C, ptr(dy_dx), stream())

    @staticmethod
    def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
        call("rn_sh_encode_backward", ptr(grad), ptr(inputs), B, D, C, ptr(dy_dx), ptr(grad_inputs), stream())


class freqencoder_backend:
    @staticmethod
    def freq_encode_forward(inputs, B, D, deg, C, outputs):
        call("rn_freq_encode_forward", ptr(inputs), B, D, deg, C, ptr(outputs), stream())

    @staticmethod
    def freq_encode_backward(grad, outputs, B, D, deg, C, grad_inputs):
        call("rn_freq_encode_backward", ptr(grad), ptr(outputs), B, D, deg, C, ptr(grad_inputs), stream())

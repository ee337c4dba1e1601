"""`_backend` objects with the pybind11 signatures of the reference's four native modules
(raymarching/src/raymarching.h:7-20, gridencoder/src/gridencoder.h:12-15, shencoder/src/shencoder.h:9-10,
freqencoder/src/freqencoder.h:7-10), computing with the CPU oracle on CPU torch tensors, IN PLACE like the CUDA
entry points do (caller allocates, kernel writes).

TEST INFRASTRUCTURE (golden-vector generation only): tests/golden/make_golden.py registers these under the module
names the reference's wrappers import first (`_raymarching_face`, `_gridencoder`, `_shencoder`, `_freqencoder`,
raymarching/raymarching.py:9-13 etc.), so the reference's OWN, UNMODIFIED wrapper files -- raymarching/raymarching.py,
gridencoder/grid.py, shencoder/sphere_harmonics.py, freqencoder/freq.py -- run in the build container and their
padding / trimming / permute / autocast rules end up in the committed vectors."""
import ctypes as C
import types

import torch

import pyoracle as po

u32, f32, i32 = C.c_uint32, C.c_float, C.c_int


def _p(t, dtype=None):
    if t is None:
        return None
    assert isinstance(t, torch.Tensor) and t.device.type == "cpu", "oracle backend: CPU tensors only"
    assert t.is_contiguous(), "oracle backend: tensor must be contiguous (the CUDA kernels index raw data_ptr)"
    if dtype is not None:
        assert t.dtype == dtype, (t.dtype, dtype)
    return C.c_void_p(t.data_ptr())


F, I, U8 = torch.float32, torch.int32, torch.uint8


def _half_flag(t):
    assert t.dtype in (torch.float32, torch.float16), t.dtype
    return 1 if t.dtype == torch.float16 else 0


def raymarching_backend():
    m = types.ModuleType("_raymarching_face")
    L = po.lib()

    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        L.orc_near_far_from_aabb(_p(rays_o, F), _p(rays_d, F), _p(aabb, F), u32(N), f32(min_near), _p(nears, F), _p(fars, F))

    def sph_from_ray(rays_o, rays_d, radius, N, coords):
        L.orc_sph_from_ray(_p(rays_o, F), _p(rays_d, F), f32(radius), u32(N), _p(coords, F))

    def morton3D(coords, N, indices):
        L.orc_morton3D(_p(coords.contiguous(), I), u32(N), _p(indices, I))

    def morton3D_invert(indices, N, coords):
        L.orc_morton3D_invert(_p(indices.contiguous(), I), u32(N), _p(coords, I))

    def packbits(grid, N, density_thresh, bitfield):
        L.orc_packbits(_p(grid, F), u32(N), f32(density_thresh), _p(bitfield, U8))

    def morton3D_dilation(grid, Cc, H, grid_dilation):
        L.orc_morton3D_dilation(_p(grid, F), u32(Cc), u32(H), _p(grid_dilation, F))

    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, Cc, H, M, nears, fars, xyzs, dirs, deltas, rays,
                         counter, noises):
        L.orc_march_rays_train(_p(rays_o, F), _p(rays_d, F), _p(grid, U8), f32(bound), f32(dt_gamma), u32(max_steps), u32(N),
                               u32(Cc), u32(H), u32(M), _p(nears, F), _p(fars, F), _p(xyzs, F), _p(dirs, F), _p(deltas, F),
                               _p(rays, I), _p(counter, I), _p(noises, F))

    def march_rays_train_backward(grad_xyzs, grad_dirs, rays, deltas, N, M, grad_rays_o, grad_rays_d):
        L.orc_march_rays_train_backward(_p(grad_xyzs.contiguous(), F), _p(grad_dirs.contiguous(), F), _p(rays, I), _p(deltas, F),
                                        u32(N), u32(M), _p(grad_rays_o, F), _p(grad_rays_d, F))

    def composite_rays_train_forward(sigmas, rgbs, ambient, deltas, rays, M, N, T_thresh, weights_sum, ambient_sum, depth, image):
        L.orc_composite_rays_train_forward(_p(sigmas, F), _p(rgbs, F), _p(ambient, F), _p(deltas, F), _p(rays, I), u32(M), u32(N),
                                           f32(T_thresh), _p(weights_sum, F), _p(ambient_sum, F), _p(depth, F), _p(image, F))

    def composite_rays_train_backward(grad_weights_sum, grad_ambient_sum, grad_image, sigmas, rgbs, ambient, deltas, rays,
                                      weights_sum, ambient_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs, grad_ambient):
        L.orc_composite_rays_train_backward(_p(grad_weights_sum, F), _p(grad_ambient_sum, F), _p(grad_image, F), _p(sigmas, F),
                                            _p(rgbs, F), _p(ambient, F), _p(deltas, F), _p(rays, I), _p(weights_sum, F),
                                            _p(ambient_sum, F), _p(image, F), u32(M), u32(N), f32(T_thresh), _p(grad_sigmas, F),
                                            _p(grad_rgbs, F), _p(grad_ambient, F))

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, Cc, H, grid, nears, fars,
                   xyzs, dirs, deltas, noises):
        L.orc_march_rays(u32(n_alive), u32(n_step), _p(rays_alive, I), _p(rays_t, F), _p(rays_o, F), _p(rays_d, F), f32(bound),
                         f32(dt_gamma), u32(max_steps), u32(Cc), u32(H), _p(grid, U8), _p(nears, F), _p(fars, F), _p(xyzs, F),
                         _p(dirs, F), _p(deltas, F), _p(noises, F))

    def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        L.orc_composite_rays(u32(n_alive), u32(n_step), f32(T_thresh), _p(rays_alive, I), _p(rays_t, F), _p(sigmas.contiguous(), F),
                             _p(rgbs.contiguous(), F), _p(deltas, F), _p(weights_sum, F), _p(depth, F), _p(image, F))

    for fn in (near_far_from_aabb, sph_from_ray, morton3D, morton3D_invert, packbits, morton3D_dilation, march_rays_train,
               march_rays_train_backward, composite_rays_train_forward, composite_rays_train_backward, march_rays, composite_rays):
        setattr(m, fn.__name__, fn)
    return m


def gridencoder_backend():
    m = types.ModuleType("_gridencoder")
    L = po.lib()

    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, Cc, Lv, S, H, dy_dx, gridtype, align_corners, interp):
        half = _half_flag(embeddings)
        assert outputs.dtype == embeddings.dtype and (dy_dx is None or dy_dx.dtype == embeddings.dtype)
        L.orc_grid_encode_forward(_p(inputs, F), _p(embeddings), _p(offsets, I), _p(outputs), u32(B), u32(D), u32(Cc), u32(Lv),
                                  f32(S), u32(H), _p(dy_dx), u32(gridtype), i32(int(bool(align_corners))), u32(interp), i32(half))

    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, Lv, S, H, dy_dx, grad_inputs, gridtype,
                             align_corners, interp):
        half = _half_flag(grad)      # the CUDA entry point dispatches on grad.scalar_type() (gridencoder.cu:490)
        assert grad_embeddings.dtype == grad.dtype
        L.orc_grid_encode_backward(_p(grad), _p(inputs, F), _p(embeddings), _p(offsets, I), _p(grad_embeddings), u32(B), u32(D),
                                   u32(Cc), u32(Lv), f32(S), u32(H), _p(dy_dx), _p(grad_inputs), u32(gridtype),
                                   i32(int(bool(align_corners))), u32(interp), i32(half))

    def grad_total_variation(inputs, embeddings, grad, offsets, weight, B, D, Cc, Lv, S, H, gridtype, align_corners):
        L.orc_grad_total_variation(_p(inputs, F), _p(embeddings, F), _p(grad, F), _p(offsets, I), f32(weight), u32(B), u32(D),
                                   u32(Cc), u32(Lv), f32(S), u32(H), u32(gridtype), i32(int(bool(align_corners))))

    m.grid_encode_forward, m.grid_encode_backward, m.grad_total_variation = grid_encode_forward, grid_encode_backward, grad_total_variation
    return m


def shencoder_backend():
    m = types.ModuleType("_shencoder")
    L = po.lib()

    def sh_encode_forward(inputs, outputs, B, D, Cc, dy_dx):
        L.orc_sh_encode_forward(_p(inputs, F), _p(outputs, F), u32(B), u32(D), u32(Cc), _p(dy_dx, F) if dy_dx is not None else None)

    def sh_encode_backward(grad, inputs, B, D, Cc, dy_dx, grad_inputs):
        L.orc_sh_encode_backward(_p(grad, F), _p(inputs, F), u32(B), u32(D), u32(Cc), _p(dy_dx, F), _p(grad_inputs, F))

    m.sh_encode_forward, m.sh_encode_backward = sh_encode_forward, sh_encode_backward
    return m


def freqencoder_backend():
    m = types.ModuleType("_freqencoder")
    L = po.lib()

    def freq_encode_forward(inputs, B, D, deg, Cc, outputs):
        L.orc_freq_encode_forward(_p(inputs, F), u32(B), u32(D), u32(deg), u32(Cc), _p(outputs, F))

    def freq_encode_backward(grad, outputs, B, D, deg, Cc, grad_inputs):
        L.orc_freq_encode_backward(_p(grad, F), _p(outputs, F), u32(B), u32(D), u32(deg), u32(Cc), _p(grad_inputs, F))

    m.freq_encode_forward, m.freq_encode_backward = freq_encode_forward, freq_encode_backward
    return m

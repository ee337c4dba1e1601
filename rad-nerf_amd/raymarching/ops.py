"""Ray-marching operators over libradnerf_hip.so.

Same public names, positional orders, defaults and return values as the reference's
raymarching/raymarching.py (cited per op), so an unmodified nerf/renderer.py runs on top.
Differences are internal: kernels launch on torch's CURRENT stream, the three sample buffers of a
march come from one zero-filled allocation, and (additions, keyword-only) the inference ops can
take the live-ray count from device memory.
"""
import numpy as np
import torch
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

import radnerf_hip as hip

_f32 = torch.float32


def _rays(rays_o, rays_d):
    rays_o = hip.dev(rays_o).contiguous().view(-1, 3)
    rays_d = hip.dev(rays_d).contiguous().view(-1, 3)
    return rays_o, rays_d


def _sample_buffers(M, device):
    """xyzs [M,3], dirs [M,3], deltas [M,2] as views of ONE zero-filled block (one memset, not three).
    Zero-init is semantic: deltas == 0 marks a dead slot (raymarching.cu:982)."""
    buf = torch.zeros(M * 8, dtype=_f32, device=device)
    return buf[:M * 3].view(M, 3), buf[M * 3:M * 6].view(M, 3), buf[M * 6:].view(M, 2)


class _near_far_from_aabb(Function):
    # raymarching/raymarching.py:19-49
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        rays_o, rays_d = _rays(rays_o, rays_d)
        aabb = hip.dev(aabb).contiguous()
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=_f32, device=rays_o.device)
        fars = torch.empty(N, dtype=_f32, device=rays_o.device)
        hip.call("rn_near_far_from_aabb", hip.ptr(rays_o, _f32), hip.ptr(rays_d, _f32), hip.ptr(aabb, _f32), N,
                 float(min_near), hip.ptr(nears), hip.ptr(fars), hip.stream())
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    # raymarching/raymarching.py:52-80
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, rays_o, rays_d, radius):
        rays_o, rays_d = _rays(rays_o, rays_d)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=_f32, device=rays_o.device)
        hip.call("rn_sph_from_ray", hip.ptr(rays_o, _f32), hip.ptr(rays_d, _f32), float(radius), N,
                 hip.ptr(coords), hip.stream())
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    # raymarching/raymarching.py:83-104
    @staticmethod
    def forward(ctx, coords):
        coords = hip.dev(coords).int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        hip.call("rn_morton3D", hip.ptr(coords), N, hip.ptr(indices), hip.stream())
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    # raymarching/raymarching.py:106-126
    @staticmethod
    def forward(ctx, indices):
        indices = hip.dev(indices).int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        hip.call("rn_morton3D_invert", hip.ptr(indices), N, hip.ptr(coords), hip.stream())
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    # raymarching/raymarching.py:129-155
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, grid, thresh, bitfield=None):
        grid = hip.aligned(hip.dev(grid))
        C, H3 = grid.shape
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        hip.call("rn_packbits", hip.ptr(grid, _f32), N, float(thresh), hip.ptr(bitfield, torch.uint8), hip.stream())
        return bitfield


packbits = _packbits.apply


class _morton3D_dilation(Function):
    # raymarching/raymarching.py:158-181
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, grid):
        grid = hip.dev(grid).contiguous()
        C, H3 = grid.shape
        H = int(np.cbrt(H3))
        out = torch.empty_like(grid)
        hip.call("rn_morton3D_dilation", hip.ptr(grid, _f32), C, H, hip.ptr(out), hip.stream())
        return out


morton3D_dilation = _morton3D_dilation.apply


class _march_rays_train(Function):
    # raymarching/raymarching.py:187-281
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        rays_o, rays_d = _rays(rays_o, rays_d)
        density_bitfield = hip.dev(density_bitfield).contiguous()
        device = rays_o.device
        N = rays_o.shape[0]
        M = N * max_steps
        # running-average sample budget (raymarching.py:226-229); rays beyond it are dropped
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count

        xyzs, dirs, deltas = _sample_buffers(M, device)
        rays = torch.empty(N, 3, dtype=torch.int32, device=device)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=device)
        noises = torch.rand(N, dtype=_f32, device=device) if perturb else torch.zeros(N, dtype=_f32, device=device)

        ws = hip.workspace(hip.workspace_bytes("rn_march_rays_train_workspace", N), device)
        nears, fars = nears.contiguous(), fars.contiguous()  # named: must outlive the enqueue below
        hip.call("rn_march_rays_train", hip.ptr(rays_o, _f32), hip.ptr(rays_d, _f32),
                 hip.ptr(density_bitfield, torch.uint8), float(bound), float(dt_gamma), int(max_steps), N, int(C),
                 int(H), M, hip.ptr(nears, _f32), hip.ptr(fars, _f32), hip.ptr(xyzs),
                 hip.ptr(dirs), hip.ptr(deltas), hip.ptr(rays), hip.ptr(step_counter, torch.int32), hip.ptr(noises),
                 hip.ptr(ws), hip.stream())

        # first epochs only: trim to the used length (one device->host read, raymarching.py:249-255)
        if force_all_rays or mean_count <= 0:
            m = int(step_counter[0].item())
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]

        ctx.save_for_backward(rays, deltas)
        return xyzs, dirs, deltas, rays

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_xyzs, grad_dirs, grad_deltas, grad_rays):
        # only reached with --train_camera (raymarching.py:263-279)
        rays, deltas = ctx.saved_tensors
        N, M = rays.shape[0], grad_xyzs.shape[0]
        grad_rays_o = torch.zeros(N, 3, dtype=_f32, device=rays.device)
        grad_rays_d = torch.zeros(N, 3, dtype=_f32, device=rays.device)
        grad_xyzs, grad_dirs, deltas = grad_xyzs.contiguous(), grad_dirs.contiguous(), deltas.contiguous()
        hip.call("rn_march_rays_train_backward", hip.ptr(grad_xyzs, _f32),
                 hip.ptr(grad_dirs, _f32), hip.ptr(rays), hip.ptr(deltas, _f32), N, M,
                 hip.ptr(grad_rays_o), hip.ptr(grad_rays_d), hip.stream())
        return (grad_rays_o, grad_rays_d) + (None,) * 13


march_rays_train = _march_rays_train.apply


class _march_rays_train_budget(Function):
    """march_rays_train with the running-average sample budget ON THE DEVICE (not in the reference's surface; the captured
    training step of radnerf/train.py uses it): `capacity` rows are allocated, `budget` (int32 device scalar <= capacity,
    already aligned as raymarching.py:226-229 does) is what the drop rule compares with.  Same xyzs / dirs / deltas rows
    and compositor outputs as march_rays_train(mean_count = budget); rows past the budget stay zero."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter, budget, capacity, perturb=False,
                dt_gamma=0, max_steps=1024):
        rays_o, rays_d = _rays(rays_o, rays_d)
        density_bitfield = hip.dev(density_bitfield).contiguous()
        device = rays_o.device
        N, M = rays_o.shape[0], int(capacity)
        xyzs, dirs, deltas = _sample_buffers(M, device)
        rays = torch.empty(N, 3, dtype=torch.int32, device=device)
        noises = torch.rand(N, dtype=_f32, device=device) if perturb else torch.zeros(N, dtype=_f32, device=device)
        ws = hip.workspace(hip.workspace_bytes("rn_march_rays_train_workspace", N), device)
        nears, fars = nears.contiguous(), fars.contiguous()
        hip.call("rn_march_rays_train_budget", hip.ptr(rays_o, _f32), hip.ptr(rays_d, _f32), hip.ptr(density_bitfield, torch.uint8),
                 float(bound), float(dt_gamma), int(max_steps), N, int(C), int(H), M, hip.ptr(budget, torch.int32), hip.ptr(nears, _f32),
                 hip.ptr(fars, _f32), hip.ptr(xyzs), hip.ptr(dirs), hip.ptr(deltas), hip.ptr(rays), hip.ptr(step_counter, torch.int32),
                 hip.ptr(noises), hip.ptr(ws), hip.stream())
        ctx.save_for_backward(rays, deltas)
        return xyzs, dirs, deltas, rays

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_xyzs, grad_dirs, grad_deltas, grad_rays):
        return _march_rays_train.backward(ctx, grad_xyzs, grad_dirs, grad_deltas, grad_rays)[:14]


_STEP_STATE = {}


def _step_state(N, device):
    """Persistent exchange words of rn_march_rays_train_step for launches of N rays on `device` (zeroed once, then owned by the
    launches: see include/radnerf_hip.h)."""
    key = (device.index, int(N))
    st = _STEP_STATE.get(key)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            # memory allocated while capturing belongs to the graph and its zero-fill would be replayed: the launch epoch (and with
            # it the hash jitter) would restart on every replay
            raise RuntimeError("march_rays_train_step: create the launch state before capturing (raymarching.ops.step_marcher_prepare)")
        st = _STEP_STATE[key] = torch.zeros(int(hip._lib.rn_march_rays_train_step_state(int(N))) // 4, dtype=torch.int32, device=device)
    return st


def step_marcher_prepare(N, device):
    """Allocate the persistent launch state of march_rays_train_step for N rays on `device` (call before capturing a step that
    uses it in a graph; eager callers need not bother)."""
    _step_state(N, torch.device(device) if not isinstance(device, torch.device) else device)


def step_marcher_supported(N, device):
    """One workgroup of 256 rays per CU at most (the workgroups exchange their counts inside the launch)."""
    return device.type == "cuda" and (N + 255) // 256 <= torch.cuda.get_device_properties(device).multi_processor_count


class _march_rays_train_step(Function):
    """A training step's marcher as ONE launch (rn_march_rays_train_step; not in the reference's surface): near / far against
    `aabb`, count pass, ordered slices, write pass, counters -- what near_far_from_aabb + `step_counter.zero_()` +
    march_rays_train_budget do in five.  `step_counter` is SET to (samples, N).  perturb = "hash": the jitter comes from the
    launch's own hash instead of a torch.rand launch.  `zeroed`: hand out zero-filled sample buffers
    (a network pass that visits every row of the capacity needs them; the fused training pass stops at step_counter[0], and
    every row below it is written by the launch).  Returns nears, fars, xyzs, dirs, deltas, rays."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, rays_o, rays_d, aabb, min_near, bound, density_bitfield, C, H, step_counter, budget, capacity, perturb=False,
                dt_gamma=0, max_steps=1024, zeroed=True):
        rays_o, rays_d = _rays(rays_o, rays_d)
        density_bitfield = hip.dev(density_bitfield).contiguous()
        aabb = hip.dev(aabb).contiguous()
        device = rays_o.device
        N, M = rays_o.shape[0], int(capacity)
        if zeroed:
            xyzs, dirs, deltas = _sample_buffers(M, device)
        else:
            buf = torch.empty(M * 8, dtype=_f32, device=device)
            xyzs, dirs, deltas = buf[:M * 3].view(M, 3), buf[M * 3:M * 6].view(M, 3), buf[M * 6:].view(M, 2)
        rays = torch.empty(N, 3, dtype=torch.int32, device=device)
        nf = torch.empty(2, N, dtype=_f32, device=device)
        # perturb: False | True (torch.rand, what the reference draws) | "hash" (the launch's own counter-based hash: no launch)
        noises = torch.rand(N, dtype=_f32, device=device) if (perturb and perturb != "hash") else None
        seed = ((torch.initial_seed() & 0x7fffffff) | 1) if perturb == "hash" else 0
        state = _step_state(N, device)
        hip.call("rn_march_rays_train_step", hip.ptr(rays_o, _f32), hip.ptr(rays_d, _f32), hip.ptr(density_bitfield, torch.uint8),
                 hip.ptr(aabb, _f32), float(min_near), float(bound), float(dt_gamma), int(max_steps), N, int(C), int(H), M,
                 hip.ptr(budget, torch.int32), hip.ptr(noises), hip.ptr(nf[0]), hip.ptr(nf[1]), hip.ptr(xyzs), hip.ptr(dirs), hip.ptr(deltas),
                 hip.ptr(rays), hip.ptr(step_counter, torch.int32), hip.ptr(state), seed, hip.stream())
        ctx.save_for_backward(rays, deltas)
        ctx.mark_non_differentiable(nf, rays)
        return nf[0], nf[1], xyzs, dirs, deltas, rays

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_nears, g_fars, grad_xyzs, grad_dirs, grad_deltas, grad_rays):
        g = _march_rays_train.backward(ctx, grad_xyzs, grad_dirs, grad_deltas, grad_rays)
        return (g[0], g[1]) + (None,) * 13


def march_rays_train_step(rays_o, rays_d, aabb, min_near, bound, density_bitfield, C, H, step_counter, budget, capacity, perturb=False,
                          dt_gamma=0, max_steps=1024, zeroed=True):
    return _march_rays_train_step.apply(rays_o, rays_d, aabb, min_near, bound, density_bitfield, C, H, step_counter, budget, capacity,
                                        perturb, dt_gamma, max_steps, zeroed)


def march_rays_train_budget(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter, budget, capacity, perturb=False,
                            dt_gamma=0, max_steps=1024):
    return _march_rays_train_budget.apply(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter, budget, capacity,
                                          perturb, dt_gamma, max_steps)


class _composite_rays_train(Function):
    # raymarching/raymarching.py:284-342
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, sigmas, rgbs, ambient, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs, ambient, deltas = sigmas.contiguous(), rgbs.contiguous(), ambient.contiguous(), deltas.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=_f32, device=dev)
        ambient_sum = torch.empty(N, dtype=_f32, device=dev)
        depth = torch.empty(N, dtype=_f32, device=dev)
        image = torch.empty(N, 3, dtype=_f32, device=dev)
        hip.call("rn_composite_rays_train_forward", hip.ptr(sigmas, _f32), hip.ptr(rgbs, _f32), hip.ptr(ambient, _f32),
                 hip.ptr(deltas, _f32), hip.ptr(rays, torch.int32), M, N, float(T_thresh), hip.ptr(weights_sum),
                 hip.ptr(ambient_sum), hip.ptr(depth), hip.ptr(image), hip.stream())
        ctx.save_for_backward(sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum, image)
        ctx.dims = (M, N, T_thresh)
        ctx.set_materialize_grads(False)      # depth's gradient is ignored (raymarching.py:324): no memset for it when it is undefined
        return weights_sum, ambient_sum, depth, image

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_weights_sum, grad_ambient_sum, grad_depth, grad_image):
        # grad_depth is ignored, as in the reference (raymarching.py:324)
        sigmas, rgbs, ambient, deltas, rays, weights_sum, ambient_sum, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        # zero-initialised: samples that belong to no ray (rows past the counter) receive no gradient -- one memset for the three
        flat = torch.zeros(M * 5, dtype=_f32, device=sigmas.device)
        grad_sigmas, grad_ambient, grad_rgbs = flat[:M], flat[M:2 * M], flat[2 * M:].view(M, 3)
        if grad_weights_sum is None and grad_ambient_sum is None and grad_image is None:
            return grad_sigmas, grad_rgbs, grad_ambient, None, None, None
        zeros = lambda g, *shape: torch.zeros(*shape, dtype=_f32, device=sigmas.device) if g is None else g.contiguous()  # noqa: E731
        grad_weights_sum, grad_ambient_sum = zeros(grad_weights_sum, N), zeros(grad_ambient_sum, N)
        grad_image = zeros(grad_image, N, 3)
        hip.call("rn_composite_rays_train_backward", hip.ptr(grad_weights_sum, _f32),
                 hip.ptr(grad_ambient_sum, _f32), hip.ptr(grad_image, _f32), hip.ptr(sigmas),
                 hip.ptr(rgbs), hip.ptr(ambient), hip.ptr(deltas), hip.ptr(rays), hip.ptr(weights_sum),
                 hip.ptr(ambient_sum), hip.ptr(image), M, N, float(T_thresh), hip.ptr(grad_sigmas), hip.ptr(grad_rgbs),
                 hip.ptr(grad_ambient), hip.stream())
        return grad_sigmas, grad_rgbs, grad_ambient, None, None, None


composite_rays_train = _composite_rays_train.apply


def padded_samples(n_alive, n_step, align):
    """M of the inference marcher: always grows, even when already aligned (raymarching.py:380-383)."""
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    return M


class _march_rays(Function):
    # raymarching/raymarching.py:348-412
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                align=-1, perturb=False, dt_gamma=0, max_steps=1024):
        rays_o, rays_d = _rays(rays_o, rays_d)
        device = rays_o.device
        M = padded_samples(n_alive, n_step, align)
        xyzs, dirs, deltas = _sample_buffers(M, device)
        # perturb == False needs no noise tensor: the kernel treats NULL as all-zero noise
        noises = torch.rand(n_alive, dtype=_f32, device=device) if perturb else None
        hip.call("rn_march_rays", int(n_alive), int(n_step), hip.ptr(rays_alive, torch.int32), hip.ptr(rays_t, _f32),
                 hip.ptr(rays_o, _f32), hip.ptr(rays_d, _f32), float(bound), float(dt_gamma), int(max_steps), int(C),
                 int(H), hip.ptr(density_bitfield, torch.uint8), hip.ptr(near, _f32), hip.ptr(far, _f32),
                 hip.ptr(xyzs), hip.ptr(dirs), hip.ptr(deltas), hip.ptr(noises), None, hip.stream())
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    # raymarching/raymarching.py:415-437
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=_f32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                T_thresh=1e-2):
        sigmas, rgbs, deltas = sigmas.contiguous(), rgbs.contiguous(), deltas.contiguous()
        hip.call("rn_composite_rays", int(n_alive), int(n_step), float(T_thresh), hip.ptr(rays_alive, torch.int32),
                 hip.ptr(rays_t, _f32), hip.ptr(sigmas, _f32), hip.ptr(rgbs, _f32),
                 hip.ptr(deltas, _f32), hip.ptr(weights_sum, _f32), hip.ptr(depth, _f32),
                 hip.ptr(image, _f32), None, hip.stream())
        return tuple()


composite_rays = _composite_rays.apply


def compact_rays(rays_alive, n_alive=None, n_alive_dev=None, out=None, n_out=None):
    """Stable on-device `rays_alive[rays_alive >= 0]` (nerf/renderer.py:258) without a host sync.

    Returns (out, n_out): `out` has the capacity of the input, its first n_out[0] entries are the
    surviving ray ids in their original order; n_out is a device int32[1].
    """
    n = int(rays_alive.shape[0] if n_alive is None else n_alive)
    device = rays_alive.device
    if out is None:
        out = torch.empty_like(rays_alive)
    if n_out is None:
        n_out = torch.empty(1, dtype=torch.int32, device=device)
    ws = hip.workspace(hip.workspace_bytes("rn_compact_rays_workspace", max(n, 1)), device)
    hip.call("rn_compact_rays", hip.ptr(rays_alive, torch.int32), n, hip.ptr(n_alive_dev), hip.ptr(out),
             hip.ptr(n_out), hip.ptr(ws), hip.stream())
    return out, n_out

"""ctypes binding of libradnerf_hip.so -- the C ABI declared in include/radnerf_hip.h.

PyTorch is only plumbing here: it owns device memory and the current HIP stream; every kernel
of the render hot path lives in the shared object.  There is NO CPU or eager fallback: if the
library is missing this import fails, and calling an op with tensors that cannot be moved to a
GPU fails inside torch, exactly like the reference's `.cuda()` calls do
(raymarching/raymarching.py:34-35).
"""
import ctypes as C
import os

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_ROOT, "lib", "libradnerf_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP extension is not built. Run `python rad-nerf_amd/build.py` "
        "(or __graft_entry__.build()). There is no fallback path.")

_lib = C.CDLL(LIB_PATH)

_u32, _f32, _i32, _ptr, _sz = C.c_uint32, C.c_float, C.c_int, C.c_void_p, C.c_size_t

# name -> argtypes, in the order of include/radnerf_hip.h
_SIGNATURES = {
    "rn_near_far_from_aabb": [_ptr, _ptr, _ptr, _u32, _f32, _ptr, _ptr, _ptr],
    "rn_sph_from_ray": [_ptr, _ptr, _f32, _u32, _ptr, _ptr],
    "rn_morton3D": [_ptr, _u32, _ptr, _ptr],
    "rn_morton3D_invert": [_ptr, _u32, _ptr, _ptr],
    "rn_packbits": [_ptr, _u32, _f32, _ptr, _ptr],
    "rn_morton3D_dilation": [_ptr, _u32, _u32, _ptr, _ptr],
    "rn_march_rays_train": [_ptr, _ptr, _ptr, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr,
                            _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_march_rays_train_budget": [_ptr, _ptr, _ptr, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr,
                                   _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_march_rays_train_step": [_ptr, _ptr, _ptr, _ptr, _f32, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr,
                                 _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _ptr],
    "rn_march_rays_train_backward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _ptr, _ptr, _ptr],
    "rn_composite_rays_train_forward": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _f32, _ptr, _ptr, _ptr, _ptr,
                                        _ptr],
    "rn_composite_rays_train_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32,
                                         _u32, _f32, _ptr, _ptr, _ptr, _ptr],
    "rn_march_rays": [_u32, _u32, _ptr, _ptr, _ptr, _ptr, _f32, _f32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr,
                      _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_composite_rays": [_u32, _u32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_compact_rays": [_ptr, _u32, _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_grid_encode_forward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _f32, _u32, _ptr, _u32, _i32, _u32,
                               _i32, _i32, _ptr],
    "rn_grid_encode_forward_ws": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _f32, _u32, _ptr, _u32, _i32, _u32,
                                  _i32, _i32, _ptr, _sz, _ptr],
    "rn_grid_encode_forward_bound": [_ptr, _f32, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i32, _i32,
                                     _ptr, _sz, _ptr],
    "rn_grid_encode_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _f32, _u32, _ptr, _ptr,
                                _u32, _i32, _u32, _i32, _i32, _ptr],
    "rn_grad_total_variation": [_ptr, _ptr, _ptr, _ptr, _f32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i32,
                                _ptr],
    "rn_sh_encode_forward": [_ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr],
    "rn_sh_encode_backward": [_ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr],
    "rn_freq_encode_forward": [_ptr, _u32, _u32, _u32, _u32, _ptr, _ptr],
    "rn_freq_encode_backward": [_ptr, _ptr, _u32, _u32, _u32, _u32, _ptr, _ptr],
}

RN_F32, RN_F16 = 0, 1
RN_LAYOUT_LBC, RN_LAYOUT_BLC, RN_LAYOUT_BLC_LEVELMAJOR = 0, 1, 2

for _name, _args in _SIGNATURES.items():
    _fn = getattr(_lib, _name)
    _fn.argtypes = _args
    _fn.restype = C.c_int
_lib.rn_last_error.restype = C.c_char_p
_lib.rn_version.restype = C.c_int
_lib.rn_device_count.restype = C.c_int
_lib.rn_march_rays_train_workspace.restype = _sz
_lib.rn_march_rays_train_workspace.argtypes = [_u32]
_lib.rn_march_rays_train_step_state.restype = _sz
_lib.rn_march_rays_train_step_state.argtypes = [_u32]
_lib.rn_compact_rays_workspace.restype = _sz
_lib.rn_compact_rays_workspace.argtypes = [_u32]
_lib.rn_grid_encode_forward_workspace.restype = _sz
_lib.rn_grid_encode_forward_workspace.argtypes = [_u32, _u32, _u32, _i32]


_lib.rn_prof_enable.argtypes = [C.c_int]
_lib.rn_prof_enable.restype = C.c_int
_lib.rn_prof_pause.argtypes = [C.c_int]
_lib.rn_prof_pause.restype = C.c_int
_lib.rn_prof_collect.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
_lib.rn_prof_collect.restype = C.c_int


_lib.rn_prof_durations.argtypes = [C.POINTER(C.c_float), C.c_uint32]
_lib.rn_prof_durations.restype = C.c_int


def exported_symbols():
    """Every symbol include/radnerf_hip.h declares (used by the CPU-side load test)."""
    return sorted(list(_SIGNATURES) + ["rn_last_error", "rn_version", "rn_device_count", "rn_prof_enable", "rn_prof_pause",
                                       "rn_prof_collect", "rn_prof_durations", "rn_march_rays_train_workspace", "rn_march_rays_train_step_state",
                                       "rn_compact_rays_workspace",
                                       "rn_grid_encode_forward_workspace"])


def prof_enable(on=True):
    _lib.rn_prof_enable(1 if on else 0)


def prof_pause(paused=True):
    """Suspend / resume the kernel timing without dropping what was recorded."""
    _lib.rn_prof_pause(1 if paused else 0)


def prof_collect():
    """(launches, total_ms) of the fused per-sample kernel since prof_enable(True)."""
    n, ms = C.c_uint32(0), C.c_float(0.0)
    rc = _lib.rn_prof_collect(C.byref(n), C.byref(ms))
    if rc != 0:
        raise RuntimeError(f"rn_prof_collect failed ({rc}): {last_error()}")
    return int(n.value), float(ms.value)


def last_error():
    return _lib.rn_last_error().decode()


def version():
    return _lib.rn_version()


def device_count():
    return _lib.rn_device_count()


def stream():
    """Raw hipStream_t of torch's current stream (kernels are enqueued on it, never on the null stream)."""
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional HIP-event timing of selected C-ABI calls, on the stream they launch on (torch's current
    stream).  `select(name, args)` returns a key (or None to skip); durations are read after a sync."""

    def __init__(self, select):
        self.select = select
        self.events = {}

    def results(self):
        torch.cuda.synchronize()
        out = {}
        for key, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[key] = dict(launches=len(ms), total_ms=float(sum(ms)), avg_ms=float(sum(ms) / max(len(ms), 1)))
        return out


_timer = None


def set_timer(timer):
    global _timer
    _timer = timer


def call(name, *args):
    if _timer is not None:
        key = _timer.select(name, args)
        if key is not None:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = getattr(_lib, name)(*args)
            b.record()
            _timer.events.setdefault(key, []).append((a, b))
            if rc != 0:
                raise RuntimeError(f"{name} failed ({rc}): {last_error()}")
            return
    rc = getattr(_lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {last_error()}")


def prof_durations(capacity=1 << 16):
    """Per-launch durations (ms) of the fused per-sample kernel since prof_enable(True)."""
    buf = (C.c_float * capacity)()
    n = _lib.rn_prof_durations(buf, capacity)
    if n < 0:
        raise RuntimeError(f"rn_prof_durations failed ({n}): {last_error()}")
    return [float(buf[i]) for i in range(n)]


def workspace_bytes(name, n):
    return int(getattr(_lib, name)(n))


_HOST_OFFSETS = {}


def host_offsets(offsets):
    """Host copy (ctypes int32 array) of a grid's `offsets` buffer, made once per buffer (one synchronising copy, then
    cached on its address + version): the planned grid forward sizes its LDS staging from the level sizes."""
    # cached ON the tensor object (with its version): an address-keyed cache would hand a new buffer that happens to reuse a freed
    # one's address the OLD grid's level sizes
    hit = getattr(offsets, "_rn_host_offsets", None)
    if hit is not None and hit[0] == offsets._version:
        return hit[1]
    vals = offsets.detach().to("cpu", torch.int32).tolist()
    arr = (C.c_int32 * len(vals))(*vals)
    try:
        offsets._rn_host_offsets = (offsets._version, arr)
    except AttributeError:
        pass
    return arr


def workspace_bytes_grid(B, L, Cc, dtype_id):
    return int(_lib.rn_grid_encode_forward_workspace(int(B), int(L), int(Cc), int(dtype_id)))


def grid_forward_workspace(B, L, Cc, dtype_id, device):
    """Scratch for rn_grid_encode_forward_ws ([B, L*C] layout): one chunk of level-major features."""
    return workspace(int(_lib.rn_grid_encode_forward_workspace(int(B), int(L), int(Cc), int(dtype_id))), device)


def dev(t):
    """Reference wrappers move stray CPU tensors to the GPU (raymarching/raymarching.py:34-35)."""
    return t if t.is_cuda else t.cuda()


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA tensor (None -> NULL).

    The caller must keep `t` referenced until the C-ABI call has been made: a temporary such as
    ptr(x.contiguous()) is freed as soon as ptr() returns and its block can be handed out again
    before the kernel is even enqueued."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("radnerf_hip: expected a CUDA (ROCm) tensor; the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("radnerf_hip: tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"radnerf_hip: expected dtype {dtype}, got {t.dtype}")
    return t.data_ptr()


def mark_written(tensors):
    """Tell PyTorch that kernels wrote these tensors through raw pointers (the one-kernel Adam update, a replayed hipGraph):
    bump their version counters, which is what caches of derived data key on (FusedState's packed weight images,
    GridEncoder.half_table) -- otherwise they would keep serving the values of the first pack."""
    ts = [t for t in tensors if t is not None]
    if ts:
        torch._C._autograd._unsafe_set_version_counter(ts, [t._version + 1 for t in ts])


def aligned(t, nbytes=16):
    """Contiguous tensor whose base address is `nbytes`-aligned (vector loads in the kernels)."""
    t = t.contiguous()
    if t.data_ptr() % nbytes:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


_WS = {}


def workspace(nbytes, device):
    """Small per-device scratch buffer, grown on demand, reused across calls on one stream."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf

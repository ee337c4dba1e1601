"""Fused inference engine: NeRFRenderer.run_cuda's inference branch on the MI355X-native C ABI
(include/radnerf_fused.h).

Per frame the host enqueues, on torch's current stream and without reading anything back first:

    rn_nerf_frame_bias   broadcast inputs (audio code, eye, individual code) -> 3 x 64 bias values
    rn_head_begin        near/far + loop state
    rn_head_iterate      max_steps x {march, fused network (grid gathers + fp32 MFMA MLPs), composite, stable
                         compaction}; iterations after the loop has ended are device-side no-ops
    rn_torso_fused       torso occupancy test + deformation / torso MLPs + blend over the background
    rn_blend_frame       image + (1 - weights_sum) * bg, clamp, depth normalisation [, uint8]

Results equal the per-operator engine's (same DDA samples, same grid features, fp32 MLPs; parity tests in
tests/test_gpu_fused.py).
"""
import ctypes as C

import numpy as np
import os

import torch

import radnerf_hip as hip

_lib = hip._lib
_u32, _f32, _i32, _ptr = C.c_uint32, C.c_float, C.c_int, C.c_void_p


class GridT(C.Structure):
    _fields_ = [("embeddings", _ptr), ("offsets", _ptr), ("D", _u32), ("L", _u32), ("H", _u32), ("S", _f32),
                ("gridtype", _u32), ("dtype", _i32)]


class NerfWeightsT(C.Structure):
    _fields_ = [("amb_w0", _ptr), ("amb_w1", _ptr), ("amb_w2", _ptr), ("sig_w0", _ptr), ("sig_w1", _ptr),
                ("sig_w2", _ptr), ("col_w0", _ptr), ("col_w1", _ptr), ("audio_dim", _u32), ("has_eye", _u32),
                ("ind_dim", _u32)]


class TorsoWeightsT(C.Structure):
    _fields_ = [("def_w0", _ptr), ("def_w1", _ptr), ("def_w2", _ptr), ("tor_w0", _ptr), ("tor_w1", _ptr),
                ("tor_w2", _ptr), ("ind_dim", _u32)]


class HeadT(C.Structure):
    _fields_ = [("rays_o", _ptr), ("rays_d", _ptr), ("N", _u32), ("aabb", _ptr), ("min_near", _f32),
                ("bitfield", _ptr), ("bound", _f32), ("dt_gamma", _f32), ("max_steps", _u32), ("cascade", _u32),
                ("grid_size", _u32), ("T_thresh", _f32), ("nears", _ptr), ("fars", _ptr), ("weights_sum", _ptr),
                ("depth", _ptr), ("image", _ptr), ("rays_alive_a", _ptr), ("rays_alive_b", _ptr), ("rays_t", _ptr),
                ("xyzs", _ptr), ("dirs", _ptr), ("deltas", _ptr), ("sigmas", _ptr), ("rgbs", _ptr), ("state", _ptr),
                ("block_counts", _ptr), ("live_slots", _ptr), ("order_w", _u32)]


RN_HEAD_STATE_INTS = 64
ST_HIST = 32
ST_UNFINISHED = 19
ST_STALLED = 22
ST_ACTIVE, ST_ITERS, ST_LIVE, ST_SLOTS = 4, 16, 17, 18

_SIGS = {
    "rn_nerf_pack_weights": [C.POINTER(NerfWeightsT), _ptr, _ptr],
    "rn_nerf_pack_weights_h16": [C.POINTER(NerfWeightsT), _ptr, _ptr],
    "rn_nerf_pack_weights_split": [C.POINTER(NerfWeightsT), _ptr, _ptr],
    "rn_nerf_frame_bias": [C.POINTER(NerfWeightsT), _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_nerf_fused_forward": [_ptr, _ptr, _ptr, _u32, _ptr, C.POINTER(GridT), C.POINTER(GridT), _ptr, _ptr, _f32, _ptr,
                              _ptr, _ptr, C.c_int, _ptr],
    "rn_head_begin": [C.POINTER(HeadT), _ptr],
    "rn_head_iterate": [C.POINTER(HeadT), C.POINTER(GridT), C.POINTER(GridT), _ptr, _ptr, _u32, _u32, C.c_int, _ptr],
    "rn_head_iterate_ex": [C.POINTER(HeadT), C.POINTER(GridT), C.POINTER(GridT), _ptr, _ptr, _u32, _u32, C.c_int, _u32, _ptr],
    "rn_frame_begin": [C.POINTER(HeadT), _ptr, _f32, _f32, _f32, _f32, _u32, _ptr],
    "rn_torso_blend_frame": [_ptr, _u32, _ptr, _u32, _f32, _ptr, _ptr, _f32, C.POINTER(TorsoWeightsT), _ptr, C.POINTER(GridT),
                             _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_nerf_frame_bias_batch": [C.POINTER(NerfWeightsT), _ptr, _u32, _ptr, _ptr, _ptr, _ptr],
    "rn_head_reschedule": [C.POINTER(HeadT), _u32, _u32, _ptr, _ptr],
    "rn_head_check_done": [C.POINTER(HeadT), _u32, _ptr],
    "rn_get_rays": [_ptr, _f32, _f32, _f32, _f32, _u32, _u32, _ptr, _ptr, _ptr],
    "rn_get_bg_coords": [_u32, _u32, _ptr, _ptr],
    "rn_convert_poses": [_ptr, _u32, _ptr, _ptr],
    "rn_torso_pack_weights": [C.POINTER(TorsoWeightsT), _ptr, _ptr],
    "rn_torso_fused": [_ptr, _u32, _ptr, _u32, _f32, _ptr, _ptr, _f32, C.POINTER(TorsoWeightsT), _ptr, C.POINTER(GridT),
                       _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_blend_frame": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _ptr, _ptr],
}
for _n, _a in _SIGS.items():
    _f = getattr(_lib, _n)
    _f.argtypes = _a
    _f.restype = C.c_int
_lib.rn_nerf_packed_floats.restype = C.c_size_t
_lib.rn_nerf_packed_floats_h16.restype = C.c_size_t
_lib.rn_nerf_packed_floats_split.restype = C.c_size_t
RN_F32_SPLIT = 2
RN_LOOP_FIRST_MARCHED, RN_LOOP_CLOSE_FRAME, RN_LOOP_COOP = 1, 2, 4
_lib.rn_nerf_bias_floats.restype = C.c_size_t
_lib.rn_torso_packed_floats.restype = C.c_size_t


def exported_symbols():
    return sorted(list(_SIGS) + ["rn_nerf_packed_floats", "rn_nerf_packed_floats_h16", "rn_nerf_packed_floats_split", "rn_nerf_bias_floats",
                                 "rn_torso_packed_floats"])


def _grid_desc(enc, table):
    g = GridT()
    g.embeddings, g.offsets = table.data_ptr(), enc.offsets.data_ptr()
    g.D, g.L, g.H = enc.input_dim, enc.num_levels, enc.base_resolution
    g.S = float(np.log2(enc.per_level_scale))
    g.gridtype = enc.gridtype_id
    g.dtype = hip.RN_F16 if table.dtype == torch.float16 else hip.RN_F32
    return g


def supported(model):
    """True when the model has the shape the fused kernels are built for (else: use the 'ops' engine)."""
    try:
        ok = (model.encoder.num_levels == 16 and model.encoder.level_dim == 2 and model.encoder.input_dim == 3
              and model.encoder_ambient.num_levels == 16 and model.encoder_ambient.level_dim == 2
              and model.encoder_ambient.input_dim == 2 and model.hidden_dim == 64 and model.geo_feat_dim == 64
              and model.hidden_dim_ambient == 64 and model.hidden_dim_color == 64 and model.num_layers == 3
              and model.num_layers_ambient == 3 and model.num_layers_color == 2 and model.ambient_dim == 2
              and model.encoder_dir.degree == 4 and not model.encoder.align_corners and model.encoder.interp_id == 0)
        if model.torso:
            ok = ok and (model.torso_encoder.num_levels == 16 and model.torso_encoder.level_dim == 2
                         and model.torso_deform_net.dim_hidden == 64 and model.torso_net.dim_hidden == 32
                         and model.torso_deform_in_dim == 42 and model.pose_in_dim == 54)
        return bool(ok)
    except AttributeError:
        return False


class FusedState:
    """Device-side constants + scratch of one model: packed weights (re-packed when parameters change),
    grid descriptors, per-frame bias block, loop scratch for N rays."""

    def __init__(self, model):
        if not supported(model):
            raise RuntimeError("fused engine: unsupported network shape; use engine='ops'")
        self.model = model
        self.dev = model.density_bitfield.device
        n_packed = max(int(_lib.rn_nerf_packed_floats()), int(_lib.rn_nerf_packed_floats_h16()),
                       int(_lib.rn_nerf_packed_floats_split()))
        self.packed = torch.empty(n_packed, dtype=torch.float32, device=self.dev)
        self.mlp_dtype = hip.RN_F32
        self.loop_hint = None
        self.bias = torch.empty(int(_lib.rn_nerf_bias_floats()), dtype=torch.float32, device=self.dev)
        self.tpacked = (torch.empty(int(_lib.rn_torso_packed_floats()), dtype=torch.float32, device=self.dev)
                        if model.torso else None)
        self._versions = None
        self._N = 0
        self._zero_eye = torch.zeros(1, dtype=torch.float32, device=self.dev)

    # -- weights --------------------------------------------------------------------------------------
    def _weights(self):
        m = self.model
        ws = [l.weight for l in m.ambient_net.net] + [l.weight for l in m.sigma_net.net] + [l.weight for l in m.color_net.net]
        if m.torso:
            ws += [l.weight for l in m.torso_deform_net.net] + [l.weight for l in m.torso_net.net]
        return ws

    def refresh(self):
        ws = self._weights()
        tables = [self.model.encoder.embeddings, self.model.encoder_ambient.embeddings]
        if self.model.torso:
            tables.append(self.model.torso_encoder.embeddings)
        # opt.mlp_dtype = "f16": contractions on the 16-bit matrix cores (fp32 accumulate), the reference's -O arithmetic
        # "f32x2": fp32-grade products from two fp16 halves per operand on the same matrix cores (include/radnerf_fused.h)
        mlp = {"f32": hip.RN_F32, "f16": hip.RN_F16, "f32x2": RN_F32_SPLIT}[
            getattr(getattr(self.model, "opt", None), "mlp_dtype", "f32")]
        # opt.half_tables (load_checkpoint(half_tables=True) sets it): the kernels read persistent fp16 copies of the grid tables
        # (GridEncoder.half_table, re-cast only when the parameter changes) -- the reference's -O mode casts every table on every
        # call (gridencoder/grid.py:43-44: 7.2 + 4.4 + 4.4 MB per loop iteration)
        half = bool(getattr(getattr(self.model, "opt", None), "half_tables", False))
        versions = tuple((w._version, w.data_ptr()) for w in ws + tables) + (mlp, half)
        if versions == self._versions:
            return
        m = self.model
        for w in ws:
            if w.dtype != torch.float32 or not w.is_contiguous():
                raise RuntimeError("fused engine: MLP weights must be contiguous float32")
        self.nw = NerfWeightsT()
        (self.nw.amb_w0, self.nw.amb_w1, self.nw.amb_w2, self.nw.sig_w0, self.nw.sig_w1, self.nw.sig_w2, self.nw.col_w0,
         self.nw.col_w1) = [w.data_ptr() for w in ws[:8]]
        self.nw.audio_dim, self.nw.has_eye, self.nw.ind_dim = m.audio_dim, int(m.exp_eye), m.individual_dim
        assert tuple(ws[0].shape) == (64, 32 + m.audio_dim) and tuple(ws[3].shape) == (64, 64 + int(m.exp_eye))
        assert tuple(ws[5].shape) == (65, 64) and tuple(ws[6].shape) == (64, 80 + m.individual_dim)
        self.mlp_dtype = mlp
        packer = {hip.RN_F32: "rn_nerf_pack_weights", hip.RN_F16: "rn_nerf_pack_weights_h16",
                  RN_F32_SPLIT: "rn_nerf_pack_weights_split"}[mlp]
        hip.call(packer, C.byref(self.nw), hip.ptr(self.packed), hip.stream())
        if m.torso:
            self.tw = TorsoWeightsT()
            (self.tw.def_w0, self.tw.def_w1, self.tw.def_w2, self.tw.tor_w0, self.tw.tor_w1,
             self.tw.tor_w2) = [w.data_ptr() for w in ws[8:14]]
            self.tw.ind_dim = m.individual_dim_torso
            assert tuple(ws[8].shape) == (64, 96 + m.individual_dim_torso) and tuple(ws[11].shape) == (32, 128 + m.individual_dim_torso)
            hip.call("rn_torso_pack_weights", C.byref(self.tw), hip.ptr(self.tpacked), hip.stream())
        # tables: fp32 parameters are read in place; with opt.half_tables their persistent fp16 copies
        if half:
            encs = [m.encoder, m.encoder_ambient] + ([m.torso_encoder] if m.torso else [])
            self.tables = [hip.aligned(e.half_table()) for e in encs]
        else:
            self.tables = [hip.aligned(t.detach()) for t in tables]
        self.gx = _grid_desc(m.encoder, self.tables[0])
        self.gw = _grid_desc(m.encoder_ambient, self.tables[1])
        self.gt = _grid_desc(m.torso_encoder, self.tables[2]) if m.torso else None
        self._versions = versions

    # -- scratch --------------------------------------------------------------------------------------
    def scratch(self, N):
        if N == self._N:
            return
        d, f32, i32 = self.dev, torch.float32, torch.int32
        self.nears = torch.empty(N, dtype=f32, device=d)
        self.fars = torch.empty(N, dtype=f32, device=d)
        self.rays_alive = torch.empty(2, N, dtype=i32, device=d)
        self.rays_t = torch.empty(N, dtype=f32, device=d)
        self.samples = torch.empty(N * 8, dtype=f32, device=d)   # xyzs | dirs | deltas
        self.sigmas = torch.empty(N, dtype=f32, device=d)
        self.rgbs = torch.empty(N, 3, dtype=f32, device=d)
        self.state = torch.zeros(RN_HEAD_STATE_INTS, dtype=i32, device=d)
        self.stats_prev = [0, 0, 0]
        self.block_counts = torch.zeros(3 * ((N + 255) // 256 + 1), dtype=i32, device=d)
        self.live_slots = torch.empty(N, dtype=i32, device=d)   # live-sample list of the iteration in flight
        self._N = N


class LoopStats(dict):
    """Loop statistics of the frame just enqueued, read back lazily (first access synchronises the stream).
    The device counters accumulate across frames; a frame's numbers are the difference to the previous read."""

    def __init__(self, st):
        super().__init__()
        self._st, self._loaded = st, False

    def _load(self):
        if not self._loaded:
            st = self._st
            cur = st.state[ST_ITERS:ST_SLOTS + 1].cpu().tolist()
            prev = st.stats_prev
            st.stats_prev = cur
            super().update(iterations=(cur[0] - prev[0]) & 0xFFFFFFFF, live_samples=(cur[1] - prev[1]) & 0xFFFFFFFF,
                           sample_slots=(cur[2] - prev[2]) & 0xFFFFFFFF)
            self._loaded = True

    def __getitem__(self, k):
        self._load()
        return super().__getitem__(k)

    def get(self, k, default=None):
        self._load()
        return super().get(k, default)

    def __contains__(self, k):
        self._load()
        return super().__contains__(k)

    # iteration / copying / comparison / printing also see the loaded numbers (dict(stats) would otherwise copy nothing)
    def __iter__(self):
        self._load()
        return super().__iter__()

    def keys(self):
        self._load()
        return super().keys()

    def items(self):
        self._load()
        return super().items()

    def values(self):
        self._load()
        return super().values()

    def __len__(self):
        self._load()
        return super().__len__()

    def __eq__(self, other):
        self._load()
        return dict.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        self._load()
        return super().__repr__()


def _all_states(model):
    """Every FusedState of the model: one per HIP stream frames are enqueued on (a renderer that keeps two frames in flight
    uses two streams, and each needs its own loop state and scratch)."""
    states = getattr(model, "_fused_states", None)
    return [st for st in states.values() if st.model is model and st._N] if states else []


def loop_counters(model):
    """Cumulative device-side loop counters (iterations, live samples, sample slots), summed over the model's streams;
    synchronises."""
    states = _all_states(model)
    if not states:
        return None
    tot = [0, 0, 0]
    for st in states:
        cur = st.state[ST_ITERS:ST_SLOTS + 1].cpu().tolist()
        tot = [(a + b) & 0xFFFFFFFF for a, b in zip(tot, cur)]
    return tot


def loop_history(model, n):
    """Device view of the live-ray count entering each of the first n loop iterations of the frame just enqueued (0 once the
    loop is over); see RN_HEAD_ST_HIST."""
    st = _state(model)
    return st.state[ST_HIST:ST_HIST + n]


def unfinished_frames(model):
    """Frames (cumulative) whose loop was cut short by a speculative iteration count (set_loop_hint); synchronises."""
    return sum(int(st.state[ST_UNFINISHED].item()) for st in _all_states(model))


def stalled_workgroups(model):
    """Workgroups (cumulative) that gave up at the in-launch barrier of the one-launch loop step (RN_LOOP_COOP); must be 0 --
    a frame rendered while this moved is invalid.  Synchronises."""
    return sum(int(st.state[ST_STALLED].item()) for st in _all_states(model))


def loop_flags(model):
    """Flags of rn_head_iterate_ex for this model: `opt.loop_launch` = "split" (default: a launch each for the compositor and
    for compaction + next march, 3 launches per iteration) or "coop" (both in one launch with a grid-wide barrier inside, 2
    launches per iteration; measured 2 % SLOWER on MI355X -- the barrier costs more than the kernel boundary it replaces,
    DESIGN.md section 3 -- and limited to three streams per device, see RN_LOOP_COOP in radnerf_fused.h)."""
    coop = getattr(model.opt, "loop_launch", "split") == "coop"
    return RN_LOOP_COOP if coop else 0


def set_loop_hint(model, iterations):
    """Enqueue only `iterations` loop iterations per frame from now on (None: all max_steps, the default).  The device
    records frames for which that was not enough (unfinished_frames); the caller checks it where it synchronises and
    renders those frames again.  Meant for streams whose iteration count is known from earlier frames."""
    hint = None if iterations is None else max(1, int(iterations))
    object.__setattr__(model, "_fused_loop_hint", hint)
    for st in (getattr(model, "_fused_states", None) or {}).values():
        st.loop_hint = hint


def _state(model):
    """The model's state for the CURRENT stream (created on first use)."""
    states = getattr(model, "_fused_states", None)
    if states is None:
        states = {}
        object.__setattr__(model, "_fused_states", states)
    key = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else 0
    st = states.get(key)
    if st is None or st.model is not model:
        st = FusedState(model)
        st.loop_hint = getattr(model, "_fused_loop_hint", None)
        states[key] = st
    return st


def network_forward(model, xyzs, dirs, enc_a, ind_code, eye, deltas=None, want_ambient=True):
    """NeRFNetwork.forward through the fused kernel: (sigma [M], color [M,3], ambient [M,2])."""
    st = _state(model)
    st.refresh()
    xyzs, dirs = xyzs.contiguous().float(), dirs.contiguous().float()
    M = xyzs.shape[0]
    enc_a = enc_a.reshape(-1).contiguous().float()
    eye_t = eye.reshape(-1).contiguous().float() if eye is not None else st._zero_eye
    ind = ind_code.detach().reshape(-1).contiguous().float() if ind_code is not None else None
    hip.call("rn_nerf_frame_bias", C.byref(st.nw), hip.ptr(enc_a), hip.ptr(eye_t), hip.ptr(ind), hip.ptr(st.bias), hip.stream())
    sigmas = torch.empty(M, dtype=torch.float32, device=xyzs.device)
    rgbs = torch.empty(M, 3, dtype=torch.float32, device=xyzs.device)
    ambient = torch.empty(M, 2, dtype=torch.float32, device=xyzs.device) if want_ambient else None
    if deltas is not None:
        deltas = deltas.contiguous()
    hip.call("rn_nerf_fused_forward", hip.ptr(xyzs), hip.ptr(dirs), hip.ptr(deltas), M, None, C.byref(st.gx), C.byref(st.gw),
             hip.ptr(st.packed), hip.ptr(st.bias), float(model.bound), hip.ptr(sigmas), hip.ptr(rgbs), hip.ptr(ambient),
             st.mlp_dtype, hip.stream())
    return sigmas, rgbs, ambient


def density_forward(model, xyzs, enc_a, eye, out=None):
    """NeRFNetwork.density (nerf/network.py:286-325) through the fused kernel's sigma branch: no geo_feat layer, no SH, no
    colour network.  xyzs [M,3] -> sigma [M] (written into `out` when given)."""
    st = _state(model)
    st.refresh()
    xyzs = xyzs.contiguous().float()
    M = xyzs.shape[0]
    enc_a = enc_a.reshape(-1).contiguous().float()
    eye_t = eye.reshape(-1).contiguous().float() if eye is not None else st._zero_eye
    ind = model.individual_codes[0].detach().reshape(-1).contiguous().float() if model.individual_dim > 0 else None
    hip.call("rn_nerf_frame_bias", C.byref(st.nw), hip.ptr(enc_a), hip.ptr(eye_t), hip.ptr(ind), hip.ptr(st.bias), hip.stream())
    sigmas = out if out is not None else torch.empty(M, dtype=torch.float32, device=xyzs.device)
    hip.call("rn_nerf_fused_forward", hip.ptr(xyzs), None, None, M, None, C.byref(st.gx), C.byref(st.gw), hip.ptr(st.packed),
             hip.ptr(st.bias), float(model.bound), hip.ptr(sigmas), None, None, st.mlp_dtype, hip.stream())
    return sigmas


def torso_forward(model, bg_coords, poses6, ind_code_torso, thresh, bg_in=None, bg_out=None, alpha_out=None, deform_out=None):
    """The torso layer over N pixels (nerf/renderer.py:269-299 + nerf/network.py:188-219): occupancy test against `thresh`,
    deformation / torso networks, blend over bg_in (None = white).  Returns (bg_out [N,3], alpha_out [N,1])."""
    st = _state(model)
    st.refresh()
    N = bg_coords.shape[0]
    dev = bg_coords.device
    bg_coords = bg_coords.contiguous().float()
    poses6 = poses6.reshape(-1).contiguous().float()
    ict = ind_code_torso.detach().reshape(-1).contiguous().float() if ind_code_torso is not None else None
    if bg_out is None:
        bg_out = torch.empty(N, 3, dtype=torch.float32, device=dev)
    if alpha_out is None:
        alpha_out = torch.empty(N, 1, dtype=torch.float32, device=dev)
    hip.call("rn_torso_fused", hip.ptr(bg_coords), N, hip.ptr(model.density_grid_torso), int(model.grid_size), float(thresh),
             hip.ptr(poses6), hip.ptr(ict), float(model.opt.torso_shrink), C.byref(st.tw), hip.ptr(st.tpacked), C.byref(st.gt),
             hip.ptr(bg_in), hip.ptr(bg_out), hip.ptr(alpha_out), hip.ptr(deform_out), hip.stream())
    return bg_out, alpha_out


def frame_bias_batch(model, codes, eye, ind_code):
    """Per-frame bias blocks of n consecutive frames' audio codes [n, audio_dim] -> [n, 192] (one launch)."""
    st = _state(model)
    st.refresh()
    codes = codes.contiguous().float()
    n = codes.shape[0]
    eye_t = eye.reshape(-1).contiguous().float() if eye is not None else st._zero_eye
    ind = ind_code.detach().reshape(-1).contiguous().float() if ind_code is not None else None
    out = torch.empty(n, int(_lib.rn_nerf_bias_floats()), dtype=torch.float32, device=codes.device)
    hip.call("rn_nerf_frame_bias_batch", C.byref(st.nw), hip.ptr(codes), n, hip.ptr(eye_t), hip.ptr(ind), hip.ptr(out), hip.stream())
    return out


def render_frame(model, rays_o, rays_d, enc_a, ind_code, eye, bg_coords, poses, ind_code_torso, bg_color, dt_gamma,
                 max_steps, T_thresh, want_u8=False, ray_source=None, frame_bias=None):
    """Inference frame: returns dict(image [N,3], depth [N], weights_sum [N], nears, fars[, image_u8]).

    Launches per frame (whole-frame renders): ONE prologue (rays from `ray_source` = (pose [4,4] on the device, intrinsics,
    W) when given -- rays_o / rays_d are then output buffers --, near/far, loop initialisation, march of iteration 0), per
    loop iteration {network, composite, compaction + next march}, ONE epilogue (torso pass + blend [+ uint8]); plus the
    per-frame bias fold unless the caller hands in `frame_bias` (frame_bias_batch)."""
    st = _state(model)
    st.refresh()
    N = rays_o.shape[0]
    st.scratch(N)
    dev = rays_o.device
    s = hip.stream()

    if frame_bias is None:
        enc_a = enc_a.reshape(-1).contiguous().float()
        eye_t = eye.reshape(-1).contiguous().float() if eye is not None else st._zero_eye
        ind = ind_code.detach().reshape(-1).contiguous().float() if ind_code is not None else None
        hip.call("rn_nerf_frame_bias", C.byref(st.nw), hip.ptr(enc_a), hip.ptr(eye_t), hip.ptr(ind), hip.ptr(st.bias), s)
        bias = st.bias
    else:
        bias = frame_bias.reshape(-1)
        assert bias.numel() == st.bias.numel() and bias.is_contiguous() and bias.dtype == torch.float32

    weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
    depth = torch.empty(N, dtype=torch.float32, device=dev)
    image = torch.empty(N, 3, dtype=torch.float32, device=dev)

    h = HeadT()
    h.rays_o, h.rays_d, h.N = rays_o.data_ptr(), rays_d.data_ptr(), N
    h.aabb, h.min_near = model.aabb_infer.data_ptr(), float(model.min_near)
    h.bitfield, h.bound, h.dt_gamma = model.density_bitfield.data_ptr(), float(model.bound), float(dt_gamma)
    h.max_steps, h.cascade, h.grid_size, h.T_thresh = int(max_steps), int(model.cascade), int(model.grid_size), float(T_thresh)
    h.nears, h.fars = st.nears.data_ptr(), st.fars.data_ptr()
    h.weights_sum, h.depth, h.image = weights_sum.data_ptr(), depth.data_ptr(), image.data_ptr()
    h.rays_alive_a, h.rays_alive_b = st.rays_alive[0].data_ptr(), st.rays_alive[1].data_ptr()
    h.rays_t = st.rays_t.data_ptr()
    h.xyzs = st.samples.data_ptr()
    h.dirs = st.samples.data_ptr() + N * 3 * 4
    h.deltas = st.samples.data_ptr() + N * 6 * 4
    h.sigmas, h.rgbs = st.sigmas.data_ptr(), st.rgbs.data_ptr()
    h.state, h.block_counts = st.state.data_ptr(), st.block_counts.data_ptr()
    h.live_slots = st.live_slots.data_ptr() if (getattr(model.opt, "live_list", True) and os.environ.get("RN_LIVE_LIST", "1") != "0") else None
    # image width, if the caller said the rays are the row-major pixels of an image (SyntheticScene does): the loop then
    # walks the rays in 8 x 8 pixel blocks (more shared grid rows per wave; no pixel changes)
    h.order_w = int(getattr(model, "ray_order_width", 0) or 0)

    # The whole <= max_steps loop is enqueued without reading anything back: iterations past the end of the loop
    # are no-ops decided on the device (a few microseconds each), so the host can run ahead of the GPU.
    shard = getattr(model, "shard_schedule", None)
    merged = shard is None and getattr(model.opt, "frame_kernels", "merged") == "merged"
    n_iters = int(max_steps) if st.loop_hint is None else min(int(max_steps), st.loop_hint)
    if merged:
        if ray_source is not None:
            pose, (fx, fy, cx, cy), W = ray_source
            pose = pose.reshape(-1, 4)[:4].contiguous().float()
            hip.call("rn_frame_begin", C.byref(h), hip.ptr(pose), float(fx), float(fy), float(cx), float(cy), int(W), s)
        else:
            hip.call("rn_frame_begin", C.byref(h), None, 0.0, 0.0, 0.0, 0.0, 0, s)
        hip.call("rn_head_iterate_ex", C.byref(h), C.byref(st.gx), C.byref(st.gw), hip.ptr(st.packed), hip.ptr(bias), 0, n_iters,
                 st.mlp_dtype, RN_LOOP_FIRST_MARCHED | RN_LOOP_CLOSE_FRAME | loop_flags(model), s)
    else:
        if ray_source is not None:
            pose, (fx, fy, cx, cy), W = ray_source
            pose = pose.reshape(-1, 4)[:4].contiguous().float()
            hip.call("rn_get_rays", hip.ptr(pose), float(fx), float(fy), float(cx), float(cy), N // int(W), int(W), hip.ptr(rays_o),
                     hip.ptr(rays_d), s)
        hip.call("rn_head_begin", C.byref(h), s)
        if shard is None:
            hip.call("rn_head_iterate_ex", C.byref(h), C.byref(st.gx), C.byref(st.gw), hip.ptr(st.packed), hip.ptr(bias), 0,
                     n_iters, st.mlp_dtype, loop_flags(model), s)
        else:
            # This call renders a shard of a frame (tile-parallel): the step schedule must be the whole frame's, so the
            # live-ray counts are summed over the ranks between iterations -- still without the host reading anything.
            group, n_total = shard
            total = torch.empty(1, dtype=torch.int32, device=dev)
            for it in range(n_iters):
                hip.call("rn_head_iterate_ex", C.byref(h), C.byref(st.gx), C.byref(st.gw), hip.ptr(st.packed), hip.ptr(bias),
                         it, 1, st.mlp_dtype, loop_flags(model), s)
                bank = ((it + 1) & 1) * 8
                total.copy_(st.state[bank:bank + 1])
                group.all_reduce(total)
                hip.call("rn_head_reschedule", C.byref(h), it, int(n_total), hip.ptr(total), hip.stream())
        if n_iters < int(max_steps):
            hip.call("rn_head_check_done", C.byref(h), n_iters, s)

    # torso layer over the background, final blend
    bg_in = None
    if torch.is_tensor(bg_color):
        bg_in = bg_color.reshape(-1, 3).contiguous().float()
        if bg_in.shape[0] != N:
            bg_in = bg_in.expand(N, 3).contiguous()
    elif bg_color is not None and float(bg_color) != 1.0:
        bg_in = torch.full((N, 3), float(bg_color), dtype=torch.float32, device=dev)
    results = {}
    model.last_stats = LoopStats(st)
    u8 = torch.empty(N, 3, dtype=torch.uint8, device=dev) if want_u8 else None
    if model.torso and merged:
        thresh = min(model.density_thresh_torso, model.mean_density_torso)
        keep = getattr(model.opt, "keep_torso_layer", True)      # results["torso_color"] / ["torso_alpha"] (the reference returns them)
        talpha = torch.empty(N, 1, dtype=torch.float32, device=dev) if keep else None
        bg_final = torch.empty(N, 3, dtype=torch.float32, device=dev) if keep else None
        coords = bg_coords.contiguous().float()
        p6 = poses.reshape(-1).contiguous().float()
        ict = ind_code_torso.detach().reshape(-1).contiguous().float() if ind_code_torso is not None else None
        hip.call("rn_torso_blend_frame", hip.ptr(coords), N, hip.ptr(model.density_grid_torso), int(model.grid_size), float(thresh),
                 hip.ptr(p6), hip.ptr(ict), float(model.opt.torso_shrink), C.byref(st.tw), hip.ptr(st.tpacked), C.byref(st.gt),
                 hip.ptr(bg_in), hip.ptr(bg_final), hip.ptr(talpha), hip.ptr(image), hip.ptr(weights_sum), hip.ptr(depth),
                 hip.ptr(st.nears), hip.ptr(st.fars), hip.ptr(u8), s)
        if keep:
            results["torso_alpha"], results["torso_color"] = talpha, bg_final
    else:
        bg_final = bg_in
        if model.torso:
            thresh = min(model.density_thresh_torso, model.mean_density_torso)
            bg_final, talpha = torso_forward(model, bg_coords, poses, ind_code_torso, thresh, bg_in=bg_in)
            results["torso_alpha"], results["torso_color"] = talpha, bg_final
        hip.call("rn_blend_frame", hip.ptr(image), hip.ptr(weights_sum), hip.ptr(bg_final), hip.ptr(depth), hip.ptr(st.nears),
                 hip.ptr(st.fars), N, hip.ptr(u8), s)
    results.update(image=image, depth=depth, weights_sum=weights_sum, nears=st.nears, fars=st.fars)
    if want_u8:
        results["image_u8"] = u8
    return results


def get_rays(pose, intrinsics, H, W):
    """Full-image rays of one cam2world pose [4,4] (or [1,4,4]) on the device: dict(rays_o, rays_d) of shape [1, H*W, 3]
    (get_rays with N = -1, nerf/utils.py:249-333) in one launch."""
    pose = pose.reshape(-1, 4)[:4].contiguous().float()
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    rays_o = torch.empty(1, H * W, 3, dtype=torch.float32, device=pose.device)
    rays_d = torch.empty(1, H * W, 3, dtype=torch.float32, device=pose.device)
    hip.call("rn_get_rays", hip.ptr(pose), fx, fy, cx, cy, int(H), int(W), hip.ptr(rays_o), hip.ptr(rays_d), hip.stream())
    return {"rays_o": rays_o, "rays_d": rays_d}


def get_bg_coords(H, W, device):
    """[1, H*W, 2] (nerf/utils.py:240-245) in one launch."""
    out = torch.empty(1, int(H) * int(W), 2, dtype=torch.float32, device=device)
    hip.call("rn_get_bg_coords", int(H), int(W), hip.ptr(out), hip.stream())
    return out


def convert_poses(poses):
    """[B, 4, 4] cam2world -> [B, 6] (nerf/utils.py:231-237) in one launch."""
    p = poses.contiguous().float()
    out = torch.empty(p.shape[0], 6, dtype=torch.float32, device=p.device)
    hip.call("rn_convert_poses", hip.ptr(p), p.shape[0], hip.ptr(out), hip.stream())
    return out

"""NeRFNetwork.forward under autograd as one forward and one backward kernel (csrc/rn_train_head.hip, C ABI
include/radnerf_train.h).

Reference: NeRFNetwork.forward (nerf/network.py:222-283) in the train branch of NeRFRenderer.run_cuda
(nerf/renderer.py:206-223), differentiated by torch.autograd in Trainer.train_step (nerf/utils.py:718-806).  One call of
`head_forward` is 2 launches (weight images + the network), its backward 6 (network, weight gradients + reduction + constant
columns, one table scatter per grid) plus the two memsets of the table gradients -- against ~130 launches of grid / MLP /
activation / concatenation kernels.  Differentiable in both grid tables, the eight weight matrices, the audio code, the eye
value and the individual code; the sample positions and directions receive no gradient (as in the reference: the grid inputs
do not require grad unless --train_camera, which keeps the per-operator path).
"""
import ctypes as C

import torch

import radnerf_hip as hip

from .fused import GridT, NerfWeightsT, _grid_desc

_lib = hip._lib
_ptr, _u32, _f32 = C.c_void_p, C.c_uint32, C.c_float


class ScatterJobT(C.Structure):
    _fields_ = [("grad", _ptr), ("inputs", _ptr), ("grid", C.POINTER(GridT)), ("offsets_host", _ptr), ("grad_table", _ptr)]


class HeadGradsT(C.Structure):
    _fields_ = [(n, _ptr) for n in ("amb_w0", "amb_w1", "amb_w2", "sig_w0", "sig_w1", "sig_w2", "col_w0", "col_w1", "enc_a", "eye", "ind_code")]


_SIGS = {
    "rn_train_head_pack": [C.POINTER(NerfWeightsT), _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_train_head_pack_row": [C.POINTER(NerfWeightsT), _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_train_head_weight_grads_row": [C.POINTER(NerfWeightsT), _ptr, _ptr, _ptr, _ptr, _u32, _u32, _ptr, _ptr, C.POINTER(HeadGradsT), _ptr,
                                       _ptr],
    "rn_train_head_forward": [_ptr, _ptr, _u32, _ptr, C.POINTER(GridT), C.POINTER(GridT), _ptr, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                              _ptr, _ptr],
    "rn_train_head_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_train_head_weight_grads": [C.POINTER(NerfWeightsT), _ptr, _ptr, _ptr, _u32, _ptr, _ptr, C.POINTER(HeadGradsT), _ptr, _ptr],
    "rn_grid_scatter_lbc": [_ptr, _ptr, _u32, _ptr, C.POINTER(GridT), _ptr, _ptr],
    "rn_train_head_loss": [_ptr, _ptr, _ptr, _ptr, _u32, _ptr, _u32, _ptr, _u32, _ptr, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_train_batch_gather": [_ptr, _u32, _ptr, _u32, C.POINTER(_u32), _u32, _ptr, _ptr],
    "rn_grid_scatter_binned": [_ptr, _ptr, _u32, _ptr, C.POINTER(GridT), _ptr, _ptr, _ptr, C.c_size_t, _ptr],
    "rn_grid_scatter_jobs": [C.POINTER(ScatterJobT), _u32, _u32, _ptr, _ptr, C.c_size_t, _ptr],
}
for _n, _a in _SIGS.items():
    getattr(_lib, _n).argtypes = _a
    getattr(_lib, _n).restype = C.c_int
for _n in ("rn_train_head_image_floats", "rn_train_head_workspace_floats", "rn_train_head_wgrad_workspace"):
    getattr(_lib, _n).restype = C.c_size_t
_lib.rn_train_head_workspace_floats.argtypes = [_u32]
_lib.rn_grid_scatter_workspace.restype = C.c_size_t
_lib.rn_grid_scatter_workspace.argtypes = [_u32, C.POINTER(GridT), _ptr]
_lib.rn_grid_scatter_binned_levels.restype = C.c_uint32
_lib.rn_grid_scatter_binned_levels.argtypes = [C.POINTER(GridT), _ptr]


def exported_symbols():
    return sorted(list(_SIGS) + ["rn_train_head_image_floats", "rn_train_head_workspace_floats", "rn_train_head_wgrad_workspace",
                                 "rn_grid_scatter_workspace", "rn_grid_scatter_binned_levels"])


def supported(model):
    """The network shape the kernels are built for (= the fused inference engine's), fp32 tables, no --emb / --train_camera."""
    from . import fused
    return (fused.supported(model) and model.encoder.embeddings.dtype == torch.float32 and model.encoder_ambient.embeddings.dtype == torch.float32
            and model.audio_dim > 0 and not model.train_camera)


def _weights_of(model):
    return [l.weight for l in model.ambient_net.net] + [l.weight for l in model.sigma_net.net] + [l.weight for l in model.color_net.net]


def _weights_desc(ws, audio_dim, has_eye, ind_dim):
    nw = NerfWeightsT()
    (nw.amb_w0, nw.amb_w1, nw.amb_w2, nw.sig_w0, nw.sig_w1, nw.sig_w2, nw.col_w0, nw.col_w1) = [w.data_ptr() for w in ws]
    nw.audio_dim, nw.has_eye, nw.ind_dim = audio_dim, has_eye, ind_dim
    return nw


class _HeadTrain(torch.autograd.Function):
    """(sigma [M], rgb [M,3], ambient [M,2], |ambient|.sum(-1) [M]) = NeRFNetwork.forward(xyzs, dirs, enc_a, ind_code, eye)."""

    @staticmethod
    def forward(ctx, xyzs, dirs, enc_a, eye, ind_code, m_dev, meta, table_x, table_w, *ws):
        enc_x, enc_w, bound, ind_index = meta
        dev = xyzs.device
        M = xyzs.shape[0]
        xyzs, dirs = xyzs.contiguous(), dirs.contiguous()
        ws = [w.contiguous() for w in ws]
        audio_dim, has_eye, ind_dim = ws[0].shape[1] - 32, ws[3].shape[1] - 64, ws[6].shape[1] - 80
        enc_a_c = enc_a.reshape(-1).contiguous().float()
        eye_c = eye.reshape(-1).contiguous().float() if has_eye else None
        if ind_index is not None:       # ind_code is the TABLE individual_codes; the row is picked on the device (no index_select)
            ind_c = ind_code.detach().contiguous()
            assert ind_dim and ind_c.dtype == torch.float32 and ind_c.dim() == 2 and ind_c.shape[1] == ind_dim
            ind_index = ind_index.reshape(-1)[:1].contiguous()
            assert ind_index.dtype == torch.int64 and ind_index.is_cuda
        else:
            ind_c = ind_code.reshape(-1).contiguous().float() if ind_dim else None
        tx, tw = hip.aligned(table_x.detach(), 64), hip.aligned(table_w.detach(), 64)
        nw = _weights_desc(ws, audio_dim, has_eye, ind_dim)
        gx, gw = _grid_desc(enc_x, tx), _grid_desc(enc_w, tw)
        s = hip.stream()
        image = torch.empty(int(_lib.rn_train_head_image_floats()), dtype=torch.float32, device=dev)
        if ind_index is not None:
            hip.call("rn_train_head_pack_row", C.byref(nw), hip.ptr(enc_a_c), hip.ptr(eye_c), hip.ptr(ind_c), hip.ptr(ind_index), hip.ptr(image), s)
        else:
            hip.call("rn_train_head_pack", C.byref(nw), hip.ptr(enc_a_c), hip.ptr(eye_c), hip.ptr(ind_c), hip.ptr(image), s)
        work = torch.empty(int(_lib.rn_train_head_workspace_floats(M)), dtype=torch.float32, device=dev)
        # one block for the outputs.  Rows past the live count are not written: they belong to no ray, so the compositor never
        # reads them and the backward kernel skips them (RN_TRAIN_HEAD_ZERO=1 zero-fills the block, for tools that look at all rows)
        import os
        out = (torch.zeros if (m_dev is not None and os.environ.get("RN_TRAIN_HEAD_ZERO") == "1") else torch.empty)(
            M, 12, dtype=torch.float32, device=dev)
        flat = out.view(-1)
        sigmas, amb_abs = flat[0:M], flat[M:2 * M]
        rgbs, ambient = flat[2 * M:5 * M].view(M, 3), flat[5 * M:7 * M].view(M, 2)
        xn, wn = flat[7 * M:10 * M].view(M, 3), flat[10 * M:12 * M].view(M, 2)
        if M:
            hip.call("rn_train_head_forward", hip.ptr(xyzs), hip.ptr(dirs), M, hip.ptr(m_dev), C.byref(gx), C.byref(gw), hip.ptr(image),
                     float(bound), sigmas.data_ptr(), rgbs.data_ptr(), ambient.data_ptr(), amb_abs.data_ptr(), xn.data_ptr(), wn.data_ptr(),
                     hip.ptr(work), s)
        ctx.set_materialize_grads(False)      # an output nobody differentiates (ambient) costs no [M, 2] memset
        ctx.save_for_backward(image, work, out, m_dev, tx, tw, enc_a_c, eye_c, ind_c, *ws)
        ctx.meta = (enc_x, enc_w, M, audio_dim, has_eye, ind_dim, enc_a.shape, None if eye is None else eye.shape,
                    None if ind_code is None else ind_code.shape, table_x.dtype)
        ctx.ind_index = ind_index
        return sigmas, rgbs, ambient, amb_abs

    @staticmethod
    def backward(ctx, g_sigma, g_rgb, g_ambient, g_amb_abs):
        image, work, out, m_dev, tx, tw, enc_a_c, eye_c, ind_c, *ws = ctx.saved_tensors
        enc_x, enc_w, M, audio_dim, has_eye, ind_dim, enc_a_shape, eye_shape, ind_shape, _ = ctx.meta
        dev = image.device
        s = hip.stream()
        flat = out.view(-1)
        rgbs, ambient = flat[2 * M:5 * M], flat[5 * M:7 * M]
        xn, wn = flat[7 * M:10 * M], flat[10 * M:12 * M]

        def dense(g, shape):
            if g is None:
                return None
            g = g.contiguous()
            return g if g.dtype == torch.float32 else g.float()
        g_sigma = dense(g_sigma, (M,))
        g_rgb = dense(g_rgb, (M, 3))
        if g_sigma is None:
            g_sigma = torch.zeros(M, dtype=torch.float32, device=dev)
        if g_rgb is None:
            g_rgb = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        g_ambient, g_amb_abs = dense(g_ambient, (M, 2)), dense(g_amb_abs, (M,))
        grads = [torch.empty_like(w) for w in ws]
        g_enc_a = torch.empty(audio_dim, dtype=torch.float32, device=dev)
        g_eye = torch.empty(1, dtype=torch.float32, device=dev) if has_eye else None
        ind_index = ctx.ind_index
        if ind_index is not None:       # the whole gradient table of individual_codes, written by the constants' launch
            g_ind = torch.empty(ind_shape, dtype=torch.float32, device=dev)
        else:
            g_ind = torch.empty(ind_dim, dtype=torch.float32, device=dev) if ind_dim else None
        if M:
            g_feat = torch.empty(2, 16, M, 2, dtype=torch.float32, device=dev)
            hip.call("rn_train_head_backward", hip.ptr(g_sigma), hip.ptr(g_rgb), hip.ptr(g_ambient), hip.ptr(g_amb_abs), rgbs.data_ptr(),
                     ambient.data_ptr(), M, hip.ptr(m_dev), hip.ptr(image), hip.ptr(work), g_feat[0].data_ptr(), g_feat[1].data_ptr(), s)
            # The table gradients need only the feature gradients the kernel above has written; the weight gradients, the
            # constants' gradients and whatever autograd runs after this function (the audio nets' backward) need nothing of the
            # tables.  RN_TRAIN_OVERLAP=1: the scatter launches go to a side stream and the optimizer waits for them
            # (take_pending_events), so they run beside those kernels instead of in front of them.
            side = _side_stream(dev) if overlap_enabled() else None
            joined = None
            if side is not None:
                main = torch.cuda.current_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    g_tx, g_tw = zero_table_gradient(enc_x, tx), torch.zeros_like(tw)
                    gx, gw = _grid_desc(enc_x, tx), _grid_desc(enc_w, tw)
                    grid_scatter([(g_feat[0], xn, enc_x, gx, g_tx), (g_feat[1], wn, enc_w, gw, g_tw)], M, m_dev)
                    ev = torch.cuda.Event()
                    ev.record(side)
                for t in (g_tx, g_tw, g_feat, out):
                    t.record_stream(side)
                g_tx.record_stream(main)
                g_tw.record_stream(main)
                if DEFER_JOIN:
                    _PENDING.append((ev, g_tx.data_ptr(), g_tw.data_ptr()))
                else:
                    joined = ev
            nw = _weights_desc(ws, audio_dim, has_eye, ind_dim)
            hg = HeadGradsT()
            (hg.amb_w0, hg.amb_w1, hg.amb_w2, hg.sig_w0, hg.sig_w1, hg.sig_w2, hg.col_w0, hg.col_w1) = [g.data_ptr() for g in grads]
            hg.enc_a, hg.eye, hg.ind_code = g_enc_a.data_ptr(), hip.ptr(g_eye), hip.ptr(g_ind)
            wsp = hip.workspace(int(_lib.rn_train_head_wgrad_workspace()), dev)
            if ind_index is not None:
                hip.call("rn_train_head_weight_grads_row", C.byref(nw), hip.ptr(enc_a_c), hip.ptr(eye_c), hip.ptr(ind_c), hip.ptr(ind_index),
                         int(ind_shape[0]), M, hip.ptr(m_dev), hip.ptr(work), C.byref(hg), hip.ptr(wsp), s)
            else:
                hip.call("rn_train_head_weight_grads", C.byref(nw), hip.ptr(enc_a_c), hip.ptr(eye_c), hip.ptr(ind_c), M, hip.ptr(m_dev), hip.ptr(work),
                         C.byref(hg), hip.ptr(wsp), s)
            if side is None:
                g_tx, g_tw = zero_table_gradient(enc_x, tx), torch.zeros_like(tw)
                gx, gw = _grid_desc(enc_x, tx), _grid_desc(enc_w, tw)
                grid_scatter([(g_feat[0], xn, enc_x, gx, g_tx), (g_feat[1], wn, enc_w, gw, g_tw)], M, m_dev)
            elif joined is not None:            # nobody else will: the table gradients are complete when this function returns
                torch.cuda.current_stream(dev).wait_event(joined)
        else:
            g_tx, g_tw = torch.zeros_like(tx), torch.zeros_like(tw)
            for g in grads:
                g.zero_()
            g_enc_a.zero_()
            if g_eye is not None:
                g_eye.zero_()
            if g_ind is not None:
                g_ind.zero_()
        return (None, None, g_enc_a.view(enc_a_shape), g_eye.view(eye_shape) if g_eye is not None else None,
                g_ind.view(ind_shape) if g_ind is not None else None, None, None, g_tx, g_tw, *grads)


_SCATTER_WS = {}
_SIDE = {}
_PENDING = []


DEFER_JOIN = False      # set by a caller whose optimizer waits for take_pending_events() (radnerf.train.Trainer with HipAdam)


def overlap_enabled():
    """RN_TRAIN_OVERLAP (default on): the table-gradient scatter of the backward pass runs on a side stream beside the weight
    gradients (and, when the caller's optimizer takes over the join -- DEFER_JOIN -- beside everything autograd runs after this
    function: the audio nets' backward).  Measured on config 2: 1266 -> 1351 steps/s.  (Round 2 had found the same overlap
    slower, 1.35 -> 1.45 ms per step: the scatter was then 6.4 M memory-side atomic requests, which slowed whatever ran beside it;
    it is now mostly bucket sums in LDS.)"""
    import os
    return os.environ.get("RN_TRAIN_OVERLAP", "1") == "1"


def side_stream(dev, which=0):
    """Side streams of the training step: 0 = table-gradient scatter (backward), 1 = audio nets (forward; their backward follows
    them there, autograd runs an op's backward on its forward's stream)."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), which)
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=dev)
    return st


def _side_stream(dev):
    return side_stream(dev, 0)


def take_pending_events():
    """Events of table-gradient scatters still running on the side stream (RN_TRAIN_OVERLAP=1): whoever reads the table gradients
    next (the optimizer) makes its stream wait for them."""
    evs = list(_PENDING)
    _PENDING.clear()
    return evs


class deferred_join:
    """with deferred_join(): ... loss.backward(); optimizer.step() -- the optimizer (HipAdam.step) joins the side stream."""

    def __enter__(self):
        global DEFER_JOIN
        self.prev, DEFER_JOIN = DEFER_JOIN, True

    def __exit__(self, *exc):
        global DEFER_JOIN
        DEFER_JOIN = self.prev


def binning_active():
    """RN_SCATTER=binned: the hashed levels of the first grid are summed by table region (no global atomics on them: 0.4 -- 1.2 M
    atomic requests per xyz launch instead of 3.0 -- 4.1 M).  Default: the per-workgroup line merge for every level -- measured
    FASTER end to end (1 380 against 1 340 steps/s on config 2): the bucket passes are bound by the LDS float-atomic rate, cost two
    more launches, and compete with the kernels the scatter overlaps with, while the line merge mostly waits for its atomics."""
    import os
    return os.environ.get("RN_SCATTER", "lbc") == "binned"


def zero_table_gradient(enc, table):
    """A gradient buffer for `table` ready for grid_scatter as its FIRST job: the rows of binned levels are written by the scatter
    (every one of them, by the workgroup that owns its region), so only the other levels are cleared -- for the T = 2^19 xyz table
    2.8 MB instead of 49 MB."""
    gd = _grid_desc(enc, table)
    mask = int(_lib.rn_grid_scatter_binned_levels(C.byref(gd), hip.host_offsets(enc.offsets))) if binning_active() else 0
    if not mask:
        return torch.zeros_like(table)
    g = torch.empty_like(table)
    off = enc.offsets.tolist() if not hasattr(enc, "_offsets_list") else enc._offsets_list
    try:
        enc._offsets_list = off
    except AttributeError:
        pass
    lo = None
    for l in range(len(off) - 1):            # runs of consecutive non-binned levels: one fill each
        if not (mask >> l) & 1:
            lo = off[l] if lo is None else lo
        elif lo is not None:
            g[lo:off[l]].zero_()
            lo = None
    if lo is not None:
        g[lo:off[-1]].zero_()
    return g


def grid_scatter(jobs, M, m_dev):
    """grad_table += the table gradient, for one or two grids: jobs = [(grad_lbc [L, M, 2] level-major feature gradients, inputs
    [M, D] normalised coordinates, GridEncoder, its descriptor, grad_table), ...].  The first grid's hashed levels that are large
    enough are summed by table region (two launches, no global atomics: the T = 2^19 xyz table), every other level of both grids
    goes through the per-workgroup line merge in ONE launch (rn_grid_scatter_jobs) -- with RN_SCATTER=binned; by default every
    level of both grids takes the line merge (one launch; see binning_active)."""
    import os
    s = hip.stream()
    grad0, _, enc0, gd0, table0 = jobs[0]
    arr = (ScatterJobT * len(jobs))()
    keep = []
    for i, (grad, inputs, enc, gd, table) in enumerate(jobs):
        arr[i].grad, arr[i].inputs, arr[i].grid, arr[i].grad_table = hip.ptr(grad), hip.ptr(inputs), C.pointer(gd), hip.ptr(table)
        off = hip.host_offsets(enc.offsets)        # lets the library tell hashed levels (straight to memory) from dense ones
        arr[i].offsets_host = C.cast(off, _ptr)
        keep += [gd, off]
    ws, ws_bytes = None, 0
    if binning_active():
        off_host = hip.host_offsets(enc0.offsets)
        if int(_lib.rn_grid_scatter_workspace(M, C.byref(gd0), off_host)) > 256:
            # bucket workspace: per device and grid, sized for a multiple of 65 536 rows (a step's changing sample budget does not
            # re-allocate), zeroed once -- the bucket cursors at its start are left zero by every call
            dev = table0.device
            need = int(_lib.rn_grid_scatter_workspace(-(-M // 65536) * 65536, C.byref(gd0), off_host))
            key = (dev.index if dev.index is not None else torch.cuda.current_device(), enc0.offsets.data_ptr(), int(enc0.offsets[-1]) if False else 0)
            buf = _SCATTER_WS.get(key)
            if buf is None or buf.numel() < need:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("grid_scatter: the bucket workspace must exist before a step is captured (take one eager step first)")
                buf = _SCATTER_WS[key] = torch.zeros(need, dtype=torch.uint8, device=dev)
            ws, ws_bytes = buf, buf.numel()
            arr[0].offsets_host = C.cast(off_host, _ptr)
    hip.call("rn_grid_scatter_jobs", arr, len(jobs), M, hip.ptr(m_dev), hip.ptr(ws), ws_bytes, s)


def head_forward(model, xyzs, dirs, enc_a, ind_code, eye, m_dev=None, ind_index=None):
    """-> (sigma, rgb, ambient, ambient_abs): NeRFNetwork.forward + `ambient.abs().sum(-1)` (nerf/renderer.py:216) through the
    fused training kernels.  m_dev: optional int32 device scalar, the number of live sample rows (the marcher's counter).
    ind_index (int64 device tensor, one element) instead of ind_code: the code is model.individual_codes[ind_index], picked by the
    kernels; the gradient of individual_codes comes back whole from the backward pass (no index_select / memset / index_add)."""
    ws = _weights_of(model)
    ind = None
    if model.individual_dim > 0:
        ind = model.individual_codes if ind_index is not None else ind_code
    else:
        ind_index = None
    e = eye if model.exp_eye else None
    return _HeadTrain.apply(xyzs, dirs, enc_a, e, ind, m_dev, (model.encoder, model.encoder_ambient, float(model.bound), ind_index),
                            model.encoder.embeddings, model.encoder_ambient.embeddings, *ws)


def usable(model, x, enc_a):
    """Training call of the supported shape on the GPU in fp32 (autocast keeps the per-operator path)."""
    import os
    return (os.environ.get("RN_TRAIN_HEAD", "fused") == "fused" and x.is_cuda and torch.is_grad_enabled() and x.dim() == 2
            and x.dtype == torch.float32 and not torch.is_autocast_enabled() and enc_a is not None and not x.requires_grad
            and getattr(model, "_train_head_ok", None) is not False and _check(model))


def _check(model):
    ok = getattr(model, "_train_head_ok", None)
    if ok is None:
        ok = model._train_head_ok = supported(model)
    return ok


class _HeadLoss(torch.autograd.Function):
    """Blend over the background, clamp and the head loss of Trainer.train_step in one kernel; returns (loss, pred)."""

    @staticmethod
    def forward(ctx, image, weights_sum, ambient, bg, target, face, w_amb):
        N = image.shape[0]
        dev = image.device
        image, weights_sum, ambient = image.contiguous(), weights_sum.contiguous(), ambient.contiguous()
        out = torch.empty(N * 8 + 1, dtype=torch.float32, device=dev)
        loss, pred = out[N * 8:], out[0:3 * N].view(N, 3)
        g_image, g_ws, g_amb = out[3 * N:6 * N].view(N, 3), out[6 * N:7 * N], out[7 * N:8 * N]
        hip.call("rn_train_head_loss", hip.ptr(image), hip.ptr(weights_sum), hip.ptr(ambient), bg.data_ptr(), bg.stride(0), target.data_ptr(),
                 target.stride(0), face.data_ptr(), face.stride(0), hip.ptr(w_amb), N, loss.data_ptr(), pred.data_ptr(), g_image.data_ptr(),
                 g_ws.data_ptr(), g_amb.data_ptr(), hip.stream())
        ctx.save_for_backward(out)
        ctx.N = N
        grads = out[3 * N:8 * N]
        ctx.mark_non_differentiable(pred, grads)
        ctx.set_materialize_grads(False)          # no [N, 3] memset for pred's (never defined) gradient
        return loss.view(()), pred, grads

    @staticmethod
    def backward(ctx, g, _g_pred, _g_grads):
        (out,) = ctx.saved_tensors
        N = ctx.N
        if g is None:
            return (None,) * 7
        scaled = out[3 * N:8 * N] * g            # one kernel for the three gradients
        return scaled[0:3 * N].view(N, 3), scaled[3 * N:4 * N], scaled[4 * N:5 * N], None, None, None, None


def loss_usable(*tensors):
    return all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


def head_loss(image, weights_sum, ambient, bg, target, face, w_amb):
    """image [N,3], weights_sum [N], ambient [N] (composited), bg / target [N,3] and face [N] (0/1 floats; rows may be strided
    views of one batch table), w_amb: device scalar.  -> (loss, pred [N,3])."""
    def rows(t, width):
        t = t.reshape(-1, width) if width > 1 else t.reshape(-1)
        if t.dim() == 2 and t.stride(1) != 1:
            t = t.contiguous()
        return t
    loss, pred, grads = _HeadLoss.apply(image, weights_sum, ambient, rows(bg, 3), rows(target, 3), rows(face, 1), w_amb.reshape(1))
    # d loss / d (image, weights_sum, ambient) were written by the same kernel: backward(loss) below hands them to autograd as
    # they are instead of going through `ones_like(loss)` and a multiplication by it (two launches)
    N = image.shape[0]
    loss._rn_direct = ((image, weights_sum, ambient), (grads[0:3 * N].view(N, 3), grads[3 * N:4 * N], grads[4 * N:5 * N]))
    return loss, pred


def backward(loss):
    """loss.backward() for a loss of head_loss(): its input gradients exist already (upstream gradient 1)."""
    direct = getattr(loss, "_rn_direct", None)
    if direct is None or not all(t.requires_grad for t in direct[0]):
        loss.backward()
    else:
        torch.autograd.backward(direct[0], direct[1])


def batch_gather(table, idx, widths):
    """[n, w0] | [n, w1] | ... = the column sections of table[idx] ([n_px, sum(widths)] fp32, idx int64), each contiguous, as views
    of ONE flat buffer; returns (flat, [section views])."""
    n, row = idx.shape[0], table.shape[1]
    assert sum(widths) == row and table.is_contiguous() and table.dtype == torch.float32 and idx.dtype == torch.int64
    flat = torch.empty(n * row, dtype=torch.float32, device=table.device)
    w = (_u32 * len(widths))(*widths)
    hip.call("rn_train_batch_gather", hip.ptr(table), row, hip.ptr(idx), n, w, len(widths), hip.ptr(flat), hip.stream())
    return flat, split_sections(flat, n, widths)


def split_sections(flat, n, widths):
    out, at = [], 0
    for wd in widths:
        out.append(flat[at:at + n * wd].view(n, wd))
        at += n * wd
    return out

"""Host side of the one-kernel audio code (include/radnerf_fused.h, "audio code"; csrc/rn_audio.hip).

`encode_windows` is NeRFNetwork.encode_audio (nerf/network.py:170-185) for a batch of attention windows,
`encode_stream` cuts the windows from a feature stream on the device (get_audio_features, nerf/utils.py:56-72),
`smooth_` is the lip-smoothing recurrence of nerf/renderer.py:190-194 applied in place to the model's state.
"""
import ctypes as C

import torch

import radnerf_hip as hip

_lib = hip._lib
_ptr, _u32 = C.c_void_p, C.c_uint32


class AudioWeightsT(C.Structure):
    _fields_ = [("conv_w", _ptr * 4), ("conv_b", _ptr * 4), ("fc_w", _ptr * 2), ("fc_b", _ptr * 2),
                ("att_conv_w", _ptr * 5), ("att_conv_b", _ptr * 5), ("att_fc_w", _ptr), ("att_fc_b", _ptr),
                ("dim_in", _u32), ("dim_aud", _u32), ("has_att", _u32)]


class AudioGradsT(C.Structure):
    _fields_ = [("conv_w", _ptr * 4), ("conv_b", _ptr * 4), ("fc_w", _ptr * 2), ("fc_b", _ptr * 2),
                ("att_conv_w", _ptr * 5), ("att_conv_b", _ptr * 5), ("att_fc_w", _ptr), ("att_fc_b", _ptr)]


_SIGS = {
    "rn_audio_encode_windows": [C.POINTER(AudioWeightsT), _ptr, _u32, _ptr, _ptr, _ptr],
    "rn_audio_encode_windows_backward": [C.POINTER(AudioWeightsT), _ptr, _u32, _ptr, _ptr, C.POINTER(AudioGradsT), _ptr, _ptr],
    "rn_audio_encode_windows_train": [C.POINTER(AudioWeightsT), _ptr, _u32, _ptr, _ptr, _ptr, _ptr],
    "rn_audio_encode_windows_backward_acts": [C.POINTER(AudioWeightsT), _ptr, _u32, _ptr, _ptr, C.POINTER(AudioGradsT), _ptr, _ptr, _ptr],
    "rn_audio_encode_stream": [C.POINTER(AudioWeightsT), _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr],
    "rn_audio_smooth": [_ptr, _u32, _u32, C.c_float, _ptr, C.c_int, _ptr],
    "rn_audio_smooth_seq": [_ptr, _u32, _u32, C.c_float, _ptr, C.c_int, _ptr, _ptr],
}
for _n, _a in _SIGS.items():
    getattr(_lib, _n).argtypes = _a
    getattr(_lib, _n).restype = C.c_int
_lib.rn_audio_train_acts_floats.argtypes = [_u32, C.c_int]
_lib.rn_audio_train_acts_floats.restype = C.c_size_t


def exported_symbols():
    return sorted(list(_SIGS) + ["rn_audio_train_acts_floats"])


def supported(model):
    """The kernel covers the reference's shapes: AudioNet(dim_in <= 64, 64) [+ AudioAttNet(64, 8)], no --emb."""
    try:
        if getattr(model, "emb", False) or model.audio_dim > 64 or model.audio_in_dim > 64:
            return False
        convs = [m for m in model.audio_net.encoder_conv if isinstance(m, torch.nn.Conv1d)]
        ok = [c.out_channels for c in convs] == [32, 32, 64, 64] and model.audio_net.win_size == 16
        if model.att > 0:
            ok = ok and model.audio_att_net.seq_len == 8
        return ok and all(p.dtype == torch.float32 for p in model.audio_net.parameters())
    except AttributeError:
        return False


def _weights(model):
    """(struct, keep-alive list); rebuilt per call: 20 pointer reads, no device work."""
    w = AudioWeightsT()
    keep = []

    def p(t):
        t = t.detach()
        if not t.is_contiguous():
            t = t.contiguous()
        keep.append(t)
        return t.data_ptr()
    convs = [m for m in model.audio_net.encoder_conv if isinstance(m, torch.nn.Conv1d)]
    for i, c in enumerate(convs):
        w.conv_w[i], w.conv_b[i] = p(c.weight), p(c.bias)
    fcs = [m for m in model.audio_net.encoder_fc1 if isinstance(m, torch.nn.Linear)]
    for i, f in enumerate(fcs):
        w.fc_w[i], w.fc_b[i] = p(f.weight), p(f.bias)
    w.dim_in, w.dim_aud, w.has_att = int(model.audio_in_dim), int(model.audio_dim), int(model.att > 0)
    if model.att > 0:
        aconvs = [m for m in model.audio_att_net.attentionConvNet if isinstance(m, torch.nn.Conv1d)]
        for i, c in enumerate(aconvs):
            w.att_conv_w[i], w.att_conv_b[i] = p(c.weight), p(c.bias)
        lin = model.audio_att_net.attentionNet[0]
        w.att_fc_w, w.att_fc_b = p(lin.weight), p(lin.bias)
    return w, keep


def encode_windows(model, auds):
    """auds: [n, 8, dim_in, 16] (or [8, dim_in, 16] for one window; [n, 1, dim_in, 16] when att == 0) -> [n, dim_aud]."""
    if auds.dim() == 3:
        auds = auds.unsqueeze(0)
    auds = auds.contiguous().float()
    frames = 8 if model.att > 0 else 1
    if tuple(auds.shape[1:]) != (frames, model.audio_in_dim, 16):
        raise RuntimeError(f"audio windows must be [n, {frames}, {model.audio_in_dim}, 16], got {tuple(auds.shape)}")
    n = auds.shape[0]
    enc = torch.empty(n, model.audio_dim, dtype=torch.float32, device=auds.device)
    ws = torch.empty(n * 8, model.audio_dim, dtype=torch.float32, device=auds.device)
    w, keep = _weights(model)
    hip.call("rn_audio_encode_windows", C.byref(w), hip.ptr(auds), n, hip.ptr(enc), hip.ptr(ws), hip.stream())
    return enc


def _parameters(model):
    """The audio nets' parameters in the order of the weight struct: 4 x (conv w, b), 2 x (fc w, b) [, 5 x (att conv w, b), att fc w, b]."""
    ps = []
    for c in (m for m in model.audio_net.encoder_conv if isinstance(m, torch.nn.Conv1d)):
        ps += [c.weight, c.bias]
    for f in (m for m in model.audio_net.encoder_fc1 if isinstance(m, torch.nn.Linear)):
        ps += [f.weight, f.bias]
    if model.att > 0:
        for c in (m for m in model.audio_att_net.attentionConvNet if isinstance(m, torch.nn.Conv1d)):
            ps += [c.weight, c.bias]
        lin = model.audio_att_net.attentionNet[0]
        ps += [lin.weight, lin.bias]
    return ps


class _EncodeWindows(torch.autograd.Function):
    """encode_audio with gradients for the audio nets' parameters (the input features are data): forward = the two forward
    kernels, backward = rn_audio_encode_windows_backward (two kernels) instead of ~100 torch / MIOpen launches."""

    @staticmethod
    def forward(ctx, model, auds, *params):
        n = auds.shape[0]
        enc = torch.empty(n, model.audio_dim, dtype=torch.float32, device=auds.device)
        codes = torch.empty(n * 8, model.audio_dim, dtype=torch.float32, device=auds.device)
        w, keep = _weights(model)
        # the forward keeps every layer's output per frame; the backward starts from them (no second forward inside it)
        acts = torch.empty(int(_lib.rn_audio_train_acts_floats(n, 1 if model.att > 0 else 0)), dtype=torch.float32, device=auds.device)
        hip.call("rn_audio_encode_windows_train", C.byref(w), hip.ptr(auds), n, hip.ptr(enc), hip.ptr(codes), hip.ptr(acts), hip.stream())
        ctx.model = model
        ctx.save_for_backward(auds, codes, acts)
        return enc

    @staticmethod
    def backward(ctx, grad_enc):
        model = ctx.model
        auds, codes, acts = ctx.saved_tensors
        params = _parameters(model)
        sizes = [p.numel() for p in params]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=auds.device)      # one memset for all gradient buffers
        views, at = [], 0
        for p, k in zip(params, sizes):
            views.append(flat[at:at + k].view(p.shape))
            at += k
        g = AudioGradsT()
        it = iter(v.data_ptr() for v in views)
        for i in range(4):
            g.conv_w[i], g.conv_b[i] = next(it), next(it)
        for i in range(2):
            g.fc_w[i], g.fc_b[i] = next(it), next(it)
        if model.att > 0:
            for i in range(5):
                g.att_conv_w[i], g.att_conv_b[i] = next(it), next(it)
            g.att_fc_w, g.att_fc_b = next(it), next(it)
        w, keep = _weights(model)
        ge = grad_enc.contiguous().float()
        scratch = torch.empty_like(codes)
        hip.call("rn_audio_encode_windows_backward_acts", C.byref(w), hip.ptr(auds), auds.shape[0], hip.ptr(codes), hip.ptr(ge), C.byref(g),
                 hip.ptr(scratch), hip.ptr(acts), hip.stream())
        return (None, None, *views)


def encode_windows_train(model, auds):
    """encode_windows, differentiable in the audio nets' parameters (training step)."""
    if auds.dim() == 3:
        auds = auds.unsqueeze(0)
    auds = auds.contiguous().float()
    frames = 8 if model.att > 0 else 1
    if tuple(auds.shape[1:]) != (frames, model.audio_in_dim, 16):
        raise RuntimeError(f"audio windows must be [n, {frames}, {model.audio_in_dim}, 16], got {tuple(auds.shape)}")
    return _EncodeWindows.apply(model, auds, *_parameters(model))


def encode_stream(model, feats, first, n):
    """Codes of the n consecutive frames (first + i) mod T of feats [T, dim_in, 16] (att_mode 2 windows) -> [n, dim_aud]."""
    feats = feats.contiguous().float()
    T = feats.shape[0]
    enc = torch.empty(n, model.audio_dim, dtype=torch.float32, device=feats.device)
    ws = torch.empty(n * 8, model.audio_dim, dtype=torch.float32, device=feats.device)
    w, keep = _weights(model)
    hip.call("rn_audio_encode_stream", C.byref(w), hip.ptr(feats), T, int(first) % T, int(n), hip.ptr(enc), hip.ptr(ws),
             hip.stream())
    return enc


def smooth_(model, enc, lam=0.35):
    """Fold enc [n, dim_aud] (in order) into model.enc_a, the lip-smoothing state (nerf/renderer.py:190-194)."""
    n, dim = enc.shape
    valid = model.enc_a is not None
    if not valid:
        model.enc_a = torch.empty(1, dim, dtype=torch.float32, device=enc.device)
    state = model.enc_a
    hip.call("rn_audio_smooth", hip.ptr(enc), n, dim, float(lam), hip.ptr(state), int(valid), hip.stream())
    return state


def smooth_seq_(model, enc, lam=0.35):
    """Like smooth_, but returns the state after EVERY code: [n, dim_aud] (row i = model.enc_a as frame i would see it)."""
    n, dim = enc.shape
    valid = model.enc_a is not None
    if not valid:
        model.enc_a = torch.empty(1, dim, dtype=torch.float32, device=enc.device)
    out = torch.empty(n, dim, dtype=torch.float32, device=enc.device)
    hip.call("rn_audio_smooth_seq", hip.ptr(enc), n, dim, float(lam), hip.ptr(model.enc_a), int(valid), hip.ptr(out), hip.stream())
    return out

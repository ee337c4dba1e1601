"""Elementwise glue of the training step as single kernels (C ABI rn_head_mid_*, rn_abs_sum2_*, rn_train_loss; csrc/rn_train.hip).

The expressions are the reference's (nerf/network.py:266-276, nerf/renderer.py:216, nerf/utils.py:772-803); PyTorch evaluates
each as 3 - 15 kernels per direction, which at 2 - 5 us apiece is a third of a 1.5 ms training step.  RN_TRAIN_GLUE=torch keeps
the PyTorch expressions (the parity tests compare the two)."""
import ctypes as C
import os

import torch

import radnerf_hip as hip

_lib = hip._lib
_ptr, _u32 = C.c_void_p, C.c_uint32
_SIGS = {
    "rn_head_mid_forward": [_ptr, _ptr, _u32, _u32, _ptr, _ptr, _ptr],
    "rn_head_mid_backward": [_ptr, _ptr, _ptr, _u32, _u32, _ptr, _ptr],
    "rn_abs_sum2_forward": [_ptr, _u32, _ptr, _ptr],
    "rn_abs_sum2_backward": [_ptr, _ptr, _u32, _ptr, _ptr],
    "rn_train_loss": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _ptr, _ptr, _ptr, _ptr, _ptr],
}
for _n, _a in _SIGS.items():
    getattr(_lib, _n).argtypes = _a
    getattr(_lib, _n).restype = C.c_int


def exported_symbols():
    return sorted(_SIGS)


def enabled(*tensors):
    """The kernels apply to CUDA fp32 tensors under autograd, outside autocast."""
    return os.environ.get("RN_TRAIN_GLUE", "hip") != "torch" and torch.is_grad_enabled() and not torch.is_autocast_enabled() and \
        all(t.is_cuda and t.dtype == torch.float32 for t in tensors)


class _HeadMid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, enc_d):
        h, enc_d = h.contiguous(), enc_d.contiguous()
        M, n_sh = h.shape[0], enc_d.shape[1]
        sigma = torch.empty(M, dtype=torch.float32, device=h.device)
        x_color = torch.empty(M, n_sh + 64, dtype=torch.float32, device=h.device)
        hip.call("rn_head_mid_forward", hip.ptr(h), hip.ptr(enc_d), M, n_sh, hip.ptr(sigma), hip.ptr(x_color), hip.stream())
        ctx.save_for_backward(h)
        ctx.n_sh = n_sh
        return sigma, x_color

    @staticmethod
    def backward(ctx, g_sigma, g_x_color):
        (h,) = ctx.saved_tensors
        M = h.shape[0]
        g_sigma, g_x_color = g_sigma.contiguous(), g_x_color.contiguous()
        g_h = torch.empty_like(h)
        hip.call("rn_head_mid_backward", hip.ptr(h), hip.ptr(g_sigma), hip.ptr(g_x_color), M, ctx.n_sh, hip.ptr(g_h), hip.stream())
        return g_h, None


def head_mid(h, enc_d):
    """(trunc_exp(h[:, 0]), cat[enc_d, h[:, 1:]]) for h [M, 65], enc_d [M, n_sh]; the SH features receive no gradient."""
    return _HeadMid.apply(h, enc_d)


class _AbsSum2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        a = a.contiguous()
        out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
        hip.call("rn_abs_sum2_forward", hip.ptr(a), a.shape[0], hip.ptr(out), hip.stream())
        ctx.save_for_backward(a)
        return out

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        g = g.contiguous()
        ga = torch.empty_like(a)
        hip.call("rn_abs_sum2_backward", hip.ptr(a), hip.ptr(g), a.shape[0], hip.ptr(ga), hip.stream())
        return ga


def abs_sum2(a):
    """a.abs().sum(-1) for a [M, 2]."""
    return _AbsSum2.apply(a)


class _TrainLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weights_sum, ambient, face, w_amb):
        pred, target = pred.contiguous(), target.contiguous()
        weights_sum, ambient, face = weights_sum.contiguous(), ambient.contiguous(), face.contiguous()
        N, dev = weights_sum.numel(), pred.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        g_pred, g_ws, g_amb = torch.empty_like(pred), torch.empty_like(weights_sum), torch.empty_like(ambient)
        hip.call("rn_train_loss", hip.ptr(pred), hip.ptr(target), hip.ptr(weights_sum), hip.ptr(ambient), hip.ptr(face), hip.ptr(w_amb), N,
                 hip.ptr(loss), hip.ptr(g_pred), hip.ptr(g_ws), hip.ptr(g_amb), hip.stream())
        ctx.save_for_backward(g_pred, g_ws, g_amb)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        g_pred, g_ws, g_amb = ctx.saved_tensors
        return g * g_pred, None, g * g_ws, g * g_amb, None, None


def train_loss(pred, target, weights_sum, ambient, face, w_amb):
    """Head loss of Trainer.train_step (nerf/utils.py:772-803); face: 0 / 1 floats [N], w_amb: device scalar."""
    return _TrainLoss.apply(pred.reshape(-1, 3), target.reshape(-1, 3), weights_sum.reshape(-1), ambient.reshape(-1), face.reshape(-1),
                            w_amb.reshape(1))

"""The path's 64-wide MLPs with hand-written forward / backward kernels for training (csrc/rn_mlp.hip, C ABI rn_mlp64_*).

Reference: `MLP` (nerf/network.py:69-88) under autograd in the training step (nerf/utils.py:718-806).  One MLP = pack +
forward (2 launches) and backward-data + weight gradients + their reduction (3 launches) instead of a GEMM, a ReLU and
three more kernels per layer; hidden activations are kept in the matrix-core register layout between the passes.
Used by radnerf.network.MLP when the input is a CUDA fp32 matrix and gradients are on; shapes outside what the kernels
are built for (supported()) keep the nn.Linear path."""
import ctypes as C

import torch

import radnerf_hip as hip

_lib = hip._lib
_ptr, _u32 = C.c_void_p, C.c_uint32
_SIGS = {
    "rn_mlp64_pack": [_ptr, _u32, _ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr],
    "rn_mlp64_forward": [_ptr, _u32, _ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr],
    "rn_mlp64_backward": [_ptr, _u32, _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "rn_mlp64_weight_grads": [_ptr, _ptr, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _ptr, _ptr, _ptr, _ptr, _ptr],
}
for _n, _a in _SIGS.items():
    getattr(_lib, _n).argtypes = _a
    getattr(_lib, _n).restype = C.c_int
for _n in ("rn_mlp64_image_floats", "rn_mlp64_tile_floats", "rn_mlp64_wgrad_workspace"):
    getattr(_lib, _n).restype = C.c_size_t
_lib.rn_mlp64_image_floats.argtypes = [_u32, _u32, _u32]
_lib.rn_mlp64_tile_floats.argtypes = [_u32]
_lib.rn_mlp64_wgrad_workspace.argtypes = [_u32]


def exported_symbols():
    return sorted(list(_SIGS) + ["rn_mlp64_image_floats", "rn_mlp64_tile_floats", "rn_mlp64_wgrad_workspace"])


def supported(dim_in, dim_out, dim_hidden, num_layers):
    """Hidden width 64 natively; width 32 (torso_net, nerf/network.py:165) runs on the same kernels with its hidden layers
    zero-padded to 64 units (fused_mlp): a padded unit's pre-activation is 0, its ReLU output 0 and its weight gradients 0."""
    return dim_hidden in (32, 64) and num_layers in (2, 3) and int(_lib.rn_mlp64_image_floats(dim_in, dim_out, num_layers)) > 0 and \
        (dim_out, num_layers) in ((65, 3), (64, 3), (2, 3), (3, 2), (4, 3), (1, 3))


class _FusedMLP64(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias0, *weights):
        n_layers = len(weights)
        w0, w_last = weights[0], weights[-1]
        M, in_dim, ld0, out_dim = x.shape[0], x.shape[1], w0.shape[1], w_last.shape[0]
        dev = x.device
        in_pad = (in_dim + 3) & ~3
        xp = x.contiguous() if in_pad == in_dim else torch.nn.functional.pad(x, (0, in_pad - in_dim))
        ws = [w.contiguous() for w in weights]
        b0 = bias0.contiguous().float() if bias0 is not None else None
        image = torch.empty(int(_lib.rn_mlp64_image_floats(in_dim, out_dim, n_layers)), dtype=torch.float32, device=dev)
        s = hip.stream()
        hip.call("rn_mlp64_pack", hip.ptr(ws[0]), ld0, hip.ptr(ws[1]) if n_layers == 3 else None, hip.ptr(ws[-1]), in_dim, out_dim, n_layers,
                 hip.ptr(image), s)
        tile = int(_lib.rn_mlp64_tile_floats(M))
        h0 = torch.empty(tile, dtype=torch.float32, device=dev)
        h1 = torch.empty(tile, dtype=torch.float32, device=dev) if n_layers == 3 else None
        out = torch.empty(M, out_dim, dtype=torch.float32, device=dev)
        if M:
            hip.call("rn_mlp64_forward", hip.ptr(xp), M, hip.ptr(image), hip.ptr(b0), in_dim, out_dim, n_layers, hip.ptr(out), hip.ptr(h0),
                     hip.ptr(h1), s)
        ctx.save_for_backward(xp, image, h0, *([h1] if h1 is not None else []))
        ctx.dims = (M, in_dim, in_pad, ld0, out_dim, n_layers, bias0 is not None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        M, in_dim, in_pad, ld0, out_dim, n_layers, has_bias = ctx.dims
        saved = ctx.saved_tensors
        xp, image, h0 = saved[0], saved[1], saved[2]
        h1 = saved[3] if n_layers == 3 else None
        dev = xp.device
        go = grad_out.contiguous().float()
        s = hip.stream()
        gx = torch.empty(M, in_pad, dtype=torch.float32, device=dev)
        dz0 = torch.empty_like(h0)
        dz1 = torch.empty_like(h0) if n_layers == 3 else None
        # only the first in_dim columns of w0 belong to this MLP: the rest (constant inputs) get their gradient through bias0
        gw0 = (torch.empty if ld0 == in_dim else torch.zeros)(64, ld0, dtype=torch.float32, device=dev)
        gw1 = torch.empty(64, 64, dtype=torch.float32, device=dev) if n_layers == 3 else None
        gwl = torch.empty(out_dim, 64, dtype=torch.float32, device=dev)
        gb0 = torch.empty(64, dtype=torch.float32, device=dev) if has_bias else None
        if M:
            hip.call("rn_mlp64_backward", hip.ptr(go), M, hip.ptr(image), in_dim, out_dim, n_layers, hip.ptr(h0), hip.ptr(h1), hip.ptr(gx),
                     hip.ptr(dz0), hip.ptr(dz1), s)
            wsp = torch.empty(int(_lib.rn_mlp64_wgrad_workspace(n_layers)), dtype=torch.uint8, device=dev)
            hip.call("rn_mlp64_weight_grads", hip.ptr(xp), hip.ptr(go), M, in_dim, out_dim, n_layers, hip.ptr(h0), hip.ptr(h1), hip.ptr(dz0),
                     hip.ptr(dz1), hip.ptr(gw0), ld0, hip.ptr(gw1), hip.ptr(gwl), hip.ptr(gb0), hip.ptr(wsp), s)
        else:
            gx.zero_(), gw0.zero_(), gwl.zero_()
            if gw1 is not None:
                gw1.zero_()
            if gb0 is not None:
                gb0.zero_()
        grads = [gw0] + ([gw1] if n_layers == 3 else []) + [gwl]
        return (gx[:, :in_dim] if in_pad != in_dim else gx, gb0, *grads)


def fused_mlp(x, weights, constants=None):
    """y = MLP(cat[x, constants repeated for every row]) for x [M, in_x] (CUDA fp32), `constants` [in_c] / [1, in_c] (optional:
    inputs that are the same for every sample -- audio code, eye value, individual code) and the nn.Linear weights of the stack
    (weights[0]: [64, in_x + in_c]); differentiable in x, the constants and the weights.  The constants are never repeated:
    they enter as a bias of the first layer."""
    weights = list(weights)
    hidden = weights[0].shape[0]
    if hidden != 64:
        # width-32 stack on the 64-wide kernels: hidden units 32..63 are all-zero rows / columns (differentiable pads, so the
        # gradients of the real weights come back through the slices)
        pad = 64 - hidden
        F = torch.nn.functional
        weights[0] = F.pad(weights[0], (0, 0, 0, pad))
        if len(weights) == 3:
            weights[1] = F.pad(weights[1], (0, pad, 0, pad))
        weights[-1] = F.pad(weights[-1], (0, pad))
    bias0 = None
    if constants is not None:
        in_x = x.shape[1]
        bias0 = torch.nn.functional.linear(constants.reshape(1, -1).float(), weights[0][:, in_x:]).reshape(-1)
    return _FusedMLP64.apply(x, bias0, *weights)

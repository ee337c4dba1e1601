"""Spherical-harmonics direction encoder over libradnerf_hip.so.

Public surface of the reference's shencoder/sphere_harmonics.py: `sh_encode` (autograd Function,
always float32) and `SHEncoder(input_dim=3, degree=4)`.
"""
import torch
import torch.nn as nn
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

import radnerf_hip as hip


class _sh_encoder(Function):
    # shencoder/sphere_harmonics.py:14-56
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        inputs = hip.dev(inputs).contiguous()
        B, input_dim = inputs.shape
        output_dim = degree ** 2
        outputs = torch.empty(B, output_dim, dtype=inputs.dtype, device=inputs.device)
        dy_dx = (torch.empty(B, input_dim * output_dim, dtype=inputs.dtype, device=inputs.device)
                 if calc_grad_inputs else None)
        hip.call("rn_sh_encode_forward", hip.ptr(inputs, torch.float32), hip.ptr(outputs), B, input_dim, int(degree),
                 hip.ptr(dy_dx), hip.stream())
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = (B, input_dim, degree)
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        if dy_dx is None:
            return None, None, None
        B, input_dim, degree = ctx.dims
        grad = grad.contiguous().float()
        grad_inputs = torch.zeros_like(inputs)  # the kernel accumulates (shencoder.cu:378)
        hip.call("rn_sh_encode_backward", hip.ptr(grad), hip.ptr(inputs), B, input_dim, int(degree), hip.ptr(dy_dx),
                 hip.ptr(grad_inputs), hip.stream())
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    # shencoder/sphere_harmonics.py:61-86
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert self.degree > 0 and self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        # inputs: [..., 3] in [-size, size] -> [..., degree^2]
        inputs = inputs / size
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix_shape + [self.output_dim])

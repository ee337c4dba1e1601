"""Host logic of radnerf/train.py that needs no GPU: the loss terms of nerf/utils.py:745-806 and the Adam groups of
main.py:204 / nerf/network.py:328-357, driven through a stand-in model."""
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))


def _load_train():
    import importlib.util
    spec = importlib.util.spec_from_file_location("radnerf_train_host", os.path.join(ROOT, "rad-nerf_amd", "radnerf", "train.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Stub(torch.nn.Module):
    def __init__(self, out):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(1))
        self.out = out
        self.kwargs = None

    def render(self, *a, **kw):
        self.kwargs = kw
        return {k: v * self.w for k, v in self.out.items()}

    def get_params(self, lr, lr_net, wd=0):
        return [{"params": [self.w], "lr": lr_net}]


def test_head_loss_terms():
    tr = _load_train()
    N = 64
    g = torch.Generator().manual_seed(0)
    out = dict(image=torch.rand(1, N, 3, generator=g), weights_sum=torch.rand(N, generator=g), ambient=torch.rand(N, generator=g))
    data = dict(rays_o=None, rays_d=None, auds=None, bg_coords=None, poses=None, eye=None, index=[0], bg_color=None,
                images=torch.rand(1, N, 3, generator=g), face_mask=torch.rand(1, N, generator=g) > 0.5)
    opt = types.SimpleNamespace(torso=False, dt_gamma=1 / 256, max_steps=16)
    m = _Stub(out)
    pred, rgb, loss = tr.train_step(m, data, opt, global_step=50000, iters=200000, lambda_amb=0.1)
    assert m.kwargs["perturb"] is True and m.kwargs["force_all_rays"] is False and m.kwargs["staged"] is False
    a = out["weights_sum"].clamp(1e-5, 1 - 1e-5).double()
    ent = (-a * torch.log2(a) - (1 - a) * torch.log2(1 - a)).mean()
    mse = ((out["image"] - data["images"]).double() ** 2).mean()
    amb = (out["ambient"].double() * (~data["face_mask"].view(-1))).mean()
    expect = mse + 1e-4 * ent + 0.25 * 0.1 * amb
    assert abs(float(loss) - float(expect)) < 1e-6
    assert rgb is data["images"] and torch.equal(pred.detach(), out["image"])


def test_torso_loss_uses_torso_outputs():
    tr = _load_train()
    N = 32
    out = dict(torso_color=torch.full((1, N, 3), 0.5), torso_alpha=torch.full((N, 1), 0.5), image=torch.zeros(1, N, 3))
    data = dict(rays_o=None, rays_d=None, auds=None, bg_coords=None, poses=None, eye=None, index=[0], bg_color=None,
                bg_torso_color=torch.full((1, N, 3), 0.25), face_mask=torch.zeros(1, N, dtype=torch.bool))
    opt = types.SimpleNamespace(torso=True, dt_gamma=0, max_steps=16)
    _, _, loss = tr.train_step(_Stub(out), data, opt)
    assert abs(float(loss) - (0.0625 + 1e-4 * 1.0)) < 1e-7      # entropy(0.5) = 1 bit


def test_optimizer_is_the_references_adam():
    tr = _load_train()
    o = tr.make_optimizer(_Stub({}), lr=5e-3, lr_net=5e-4)
    assert isinstance(o, torch.optim.Adam)
    grp = o.param_groups[0]
    assert grp["betas"] == (0.9, 0.99) and grp["eps"] == 1e-15 and math.isclose(grp["lr"], 5e-4)
    assert np.allclose(float(tr.entropy_of(torch.tensor([0.0]))), float(tr.entropy_of(torch.tensor([1e-5]))))

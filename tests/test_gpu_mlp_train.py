"""The training kernels of the path's MLPs (csrc/rn_mlp.hip: rn_mlp64_forward / _backward / _weight_grads behind
radnerf.mlp_train.fused_mlp) against the reference's formulation -- the bias-free nn.Linear stack with ReLU of
nerf/network.py:69-88 under torch autograd, in fp32.  Same fp32 products; only the summation order differs, hence the
tolerances: 2e-5 of the largest magnitude for activations and input gradients, 1e-4 for the weight gradients (sums over all
samples)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(96, 2, 3), (65, 65, 3), (84, 3, 2), (64, 2, 3), (33, 65, 3)]   # ambient_net, sigma_net, color_net, + two odd ones


def _reference(x, ws):
    for l, w in enumerate(ws):
        x = F.linear(x, w)
        if l != len(ws) - 1:
            x = F.relu(x)
    return x


def _weights(dim_in, dim_out, n_layers, gen):
    dims = [dim_in] + [64] * (n_layers - 1) + [dim_out]
    return [((torch.rand(dims[l + 1], dims[l], device="cuda", generator=gen) * 2 - 1) / np.sqrt(dims[l])).requires_grad_(True)
            for l in range(n_layers)]


def _close(a, b, tol):
    scale = float(b.detach().abs().max()) + 1e-30
    err = float((a.detach() - b.detach()).abs().max())
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("M", [1, 31, 32, 33, 4097, 58181])
def test_fused_mlp_matches_torch_autograd(hiplib, shape, M):
    from radnerf import mlp_train
    dim_in, dim_out, n_layers = shape
    assert mlp_train.supported(dim_in, dim_out, 64, n_layers)
    gen = torch.Generator(device="cuda").manual_seed(M + dim_in)
    ws = _weights(dim_in, dim_out, n_layers, gen)
    x = (torch.rand(M, dim_in, device="cuda", generator=gen) * 2 - 1).requires_grad_(True)
    gy = torch.rand(M, dim_out, device="cuda", generator=gen) * 2 - 1
    # ReLU is discontinuous in its derivative: a pre-activation within rounding distance of 0 may land on either side of it
    # depending on the summation order, and the whole gradient term of that unit flips.  Such samples (a few in 10^5) get a
    # zero upstream gradient, so they test nothing and break nothing.
    with torch.no_grad():
        z = x.double()
        ambiguous = torch.zeros(M, dtype=torch.bool, device="cuda")
        for w in ws[:-1]:
            z = z @ w.double().t()
            ambiguous |= (z.abs() < 1e-5).any(1)
            z = z.clamp_min(0)
        gy[ambiguous] = 0
        assert int(ambiguous.sum()) <= max(2, M // 50)
    y = mlp_train.fused_mlp(x, ws)
    grads = torch.autograd.grad(y, [x] + ws, gy)
    xr = x.detach().clone().requires_grad_(True)
    wr = [w.detach().clone().requires_grad_(True) for w in ws]
    yr = _reference(xr, wr)
    ref = torch.autograd.grad(yr, [xr] + wr, gy)
    _close(y, yr, 2e-5)
    _close(grads[0], ref[0], 2e-5)
    for g, r in zip(grads[1:], ref[1:]):
        assert g.shape == r.shape
        _close(g, r, 1e-4)


@pytest.mark.parametrize("in_x,in_c,dim_out,n_layers", [(32, 64, 2, 3), (64, 1, 65, 3), (80, 4, 3, 2), (30, 7, 65, 3)])
@pytest.mark.parametrize("M", [33, 20011])
def test_constant_inputs_enter_as_a_bias(hiplib, in_x, in_c, dim_out, n_layers, M):
    """fused_mlp(x, weights, constants): the columns that are the same for every sample (audio code, eye, individual code --
    nerf/network.py:236, 262, 274 repeat them N times and concatenate) are folded into a first-layer bias; outputs and ALL
    gradients (x, the constants, every weight incl. the constant columns of the first layer) equal the concatenated formulation."""
    from radnerf import mlp_train
    gen = torch.Generator(device="cuda").manual_seed(M + in_x)
    ws = _weights(in_x + in_c, dim_out, n_layers, gen)
    x = (torch.rand(M, in_x, device="cuda", generator=gen) * 2 - 1).requires_grad_(True)
    c = (torch.rand(1, in_c, device="cuda", generator=gen) * 2 - 1).requires_grad_(True)
    gy = torch.rand(M, dim_out, device="cuda", generator=gen) * 2 - 1
    with torch.no_grad():
        z = torch.cat([x, c.repeat(M, 1)], 1).double()
        ambiguous = torch.zeros(M, dtype=torch.bool, device="cuda")
        for w in ws[:-1]:
            z = z @ w.double().t()
            ambiguous |= (z.abs() < 1e-5).any(1)
            z = z.clamp_min(0)
        gy[ambiguous] = 0
    y = mlp_train.fused_mlp(x, ws, c)
    grads = torch.autograd.grad(y, [x, c] + ws, gy)
    xr, cr = x.detach().clone().requires_grad_(True), c.detach().clone().requires_grad_(True)
    wr = [w.detach().clone().requires_grad_(True) for w in ws]
    yr = _reference(torch.cat([xr, cr.repeat(M, 1)], 1), wr)
    ref = torch.autograd.grad(yr, [xr, cr] + wr, gy)
    _close(y, yr, 2e-5)
    _close(grads[0], ref[0], 2e-5)
    for g, r in zip(grads[1:], ref[1:]):
        assert g.shape == r.shape
        _close(g, r, 1e-4)


def test_mlp_module_uses_the_kernels_and_trains_like_the_linear_stack(hiplib, monkeypatch):
    """radnerf.network.MLP routes CUDA fp32 training batches through the kernels; RN_MLP_TRAIN=torch keeps nn.Linear.  A few SGD
    steps on a regression target end at the same weights (1e-4)."""
    from radnerf.network import MLP
    torch.manual_seed(0)
    a, b = MLP(65, 65, 64, 3).cuda(), MLP(65, 65, 64, 3).cuda()
    b.load_state_dict(a.state_dict())
    x = torch.rand(8192, 65, device="cuda") * 2 - 1
    target = torch.sin(3 * x)
    for step in range(5):
        for m, env in ((a, "hip"), (b, "torch")):
            monkeypatch.setenv("RN_MLP_TRAIN", env)
            loss = ((m(x) - target) ** 2).mean()
            grads = torch.autograd.grad(loss, list(m.parameters()))
            with torch.no_grad():
                for p, g in zip(m.parameters(), grads):
                    p -= 0.05 * g
    for p, q in zip(a.parameters(), b.parameters()):
        _close(p.detach(), q.detach(), 1e-4)


def test_unsupported_shapes_fall_back_to_linear_layers(hiplib):
    from radnerf import mlp_train
    from radnerf.network import MLP
    assert not mlp_train.supported(136, 4, 32, 3) and not mlp_train.supported(128, 2, 64, 3) and not mlp_train.supported(65, 7, 64, 3)
    m = MLP(136, 4, 32, 3).cuda()                              # torso_net's shape: hidden width 32
    x = torch.rand(2048, 136, device="cuda", requires_grad=True)
    y = m(x)
    assert y.shape == (2048, 4) and torch.autograd.grad(y.sum(), x)[0].shape == x.shape


def test_torso_branch_trains_on_the_mlp_kernels(hiplib, monkeypatch):
    """nerf/network.py:188-219 under autograd (the 200 k torso iterations of the reference's schedule): torso_deform_net (width 64)
    and torso_net (width 32, zero-padded to the kernels' 64) through rn_mlp64_* give the loss and gradients of the nn.Linear
    formulation; the launches of the HIP MLP path really are these kernels."""
    from radnerf.scene import SyntheticScene, default_opt
    from radnerf.train import SyntheticTrainStream, train_step
    import radnerf_hip as hip

    def run(mode):
        monkeypatch.setenv("RN_MLP_TRAIN", mode)
        torch.manual_seed(0)
        scene = SyntheticScene(H=64, W=64, n_frames=8, device="cuda", opt=default_opt(engine="ops", torso=True, smooth_lips=False))
        m = scene.model
        m.train()
        stream = SyntheticTrainStream(scene, n_rays=4096, seed=2)
        called = []
        real = hip.call

        def spy(name, *a):
            called.append(name)
            return real(name, *a)
        monkeypatch.setattr(hip, "call", spy)
        _, _, loss = train_step(m, stream.batch(), scene.opt, global_step=1)
        loss.backward()
        monkeypatch.setattr(hip, "call", real)
        names = ("torso_net", "torso_deform_net", "torso_encoder", "individual_codes_torso")
        return float(loss), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None and n.startswith(names)}, called
    l_t, g_t, _ = run("torch")
    l_h, g_h, called = run("hip")
    assert called.count("rn_mlp64_forward") == 2 and called.count("rn_mlp64_backward") == 2
    assert abs(l_t - l_h) <= 1e-5 * max(abs(l_t), 1e-3)
    assert set(g_t) == set(g_h) and any(n.startswith("torso_net") for n in g_t)
    for n in g_t:
        a, b = g_h[n], g_t[n]
        scale = float(b.abs().max()) + 1e-20
        cos = float(torch.nn.functional.cosine_similarity(a.reshape(1, -1).double(), b.reshape(1, -1).double()))
        # upstream of the 2-D torso grid (deformation net, its codes) a pixel on a cell boundary may take the neighbour's derivative
        up = n.startswith(("torso_deform_net", "individual_codes_torso"))
        assert float((a - b).abs().max()) / scale < (3e-2 if up else 2e-3) and cos > (0.9995 if up else 0.99999), (n, cos)

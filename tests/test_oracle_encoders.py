"""Pins of the oracle's SH / frequency encoders and of its PyTorch-arithmetic restatements (MLP, network,
torso, grid_sample) against independent computations: scipy's complex spherical harmonics, finite
differences, numpy sin/cos, torch CPU modules."""
import numpy as np
import pytest
import scipy.special
import torch


def real_sh_reference_convention(v, degree):
    """Real SH in the reference's sign convention (shencoder.cu:50-68: Y1 = (-y, z, -x) * 0.4886): for m > 0
    sqrt(2) Re Y_l^m, for m < 0 sqrt(2) Im Y_l^|m|, with scipy's Condon-Shortley complex harmonics; index l^2+l+m."""
    x, y, z = v[:, 0].astype(np.float64), v[:, 1].astype(np.float64), v[:, 2].astype(np.float64)
    theta = np.arccos(np.clip(z, -1, 1))
    phi = np.arctan2(y, x)
    out = np.zeros((v.shape[0], degree * degree))
    for l in range(degree):
        for m in range(-l, l + 1):
            Y = scipy.special.sph_harm_y(l, abs(m), theta, phi)
            val = Y.real if m == 0 else (np.sqrt(2) * Y.real if m > 0 else np.sqrt(2) * Y.imag)
            out[:, l * l + l + m] = val
    return out


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_values_against_scipy(po, rng, degree):
    v = rng.standard_normal((2000, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    out, _ = po.sh_encode_forward(v.astype(np.float32), degree)
    np.testing.assert_allclose(out, real_sh_reference_convention(v.astype(np.float32), degree), rtol=0, atol=3e-5)


@pytest.mark.parametrize("degree", [2, 4, 8])
def test_sh_jacobian_against_finite_differences(po, rng, degree):
    v = rng.uniform(-0.9, 0.9, (300, 3)).astype(np.float32)  # polynomials are defined off the sphere too
    _, dy = po.sh_encode_forward(v, degree, True)
    dy = dy.reshape(-1, 3, degree * degree)
    h = 1e-3
    for d in range(3):
        e = np.zeros(3, np.float32); e[d] = h
        fp, _ = po.sh_encode_forward(v + e, degree)
        fm, _ = po.sh_encode_forward(v - e, degree)
        np.testing.assert_allclose(dy[:, d], (fp - fm) / (2 * h), rtol=2e-2, atol=2e-2)
    g = rng.standard_normal((300, degree * degree)).astype(np.float32)
    gi = po.sh_encode_backward(g, v, degree, dy.reshape(300, -1))
    np.testing.assert_allclose(gi, np.einsum("bc,bdc->bd", g, dy), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("D,deg", [(2, 10), (6, 4), (3, 6)])
def test_freq_against_numpy(po, rng, D, deg):
    x = rng.uniform(-1, 1, (500, D)).astype(np.float32)
    out = po.freq_encode_forward(x, deg)
    assert out.shape == (500, D + 2 * D * deg)
    cols = [x.astype(np.float64)]
    for f in range(deg):
        cols += [np.sin(2.0 ** f * x.astype(np.float64)), np.cos(2.0 ** f * x.astype(np.float64))]
    np.testing.assert_allclose(out, np.concatenate(cols, 1), rtol=0, atol=1e-6 * 2 ** deg)
    # backward vs autograd of the same formula
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tcols = [tx] + [fn(2.0 ** f * tx) for f in range(deg) for fn in (torch.sin, torch.cos)]
    tout = torch.cat(tcols, 1)
    g = rng.standard_normal(out.shape).astype(np.float32)
    (tout * torch.tensor(g, dtype=torch.float64)).sum().backward()
    gi = po.freq_encode_backward(g, out, D, deg)
    np.testing.assert_allclose(gi, tx.grad.numpy(), rtol=1e-3, atol=1e-3 * 2 ** deg * 1e-1)


def test_mlp_against_torch_linear(po, rng):
    dims = [(96, 64), (64, 64), (64, 2)]
    ws = [rng.standard_normal((o, i)).astype(np.float32) * 0.2 for i, o in dims]
    x = rng.standard_normal((1000, 96)).astype(np.float32)
    out = po.mlp_forward(ws, x)
    h = torch.tensor(x, dtype=torch.float64)
    for l, w in enumerate(ws):
        h = h @ torch.tensor(w, dtype=torch.float64).t()
        if l != len(ws) - 1:
            h = torch.relu(h)
    np.testing.assert_allclose(out, h.numpy(), rtol=1e-4, atol=1e-5)


def test_grid_sample_restatement_against_torch(po, hiplib, rng):
    """The torso-mask sampling in orc_render_frame equals F.grid_sample(align_corners=True) (renderer.py:282):
    checked through the torso pixel count of a frame whose head never hits (empty bitfield)."""
    import torch.nn.functional as F
    from radnerf.scene import SyntheticScene, default_opt
    scene = SyntheticScene(H=48, W=48, n_frames=8, device="cpu", opt=default_opt())
    m = scene.model
    g = rng.uniform(0, 0.03, 128 * 128).astype(np.float32)
    m.density_grid_torso.copy_(torch.from_numpy(g))
    m.density_bitfield.zero_()
    f = scene.frame(0)
    occ = F.grid_sample(m.density_grid_torso.view(1, 1, 128, 128), f["bg_coords"].view(1, -1, 1, 2), align_corners=True).view(-1)
    expect = int((occ > min(m.density_thresh_torso, m.mean_density_torso)).sum())
    om = po.model_from_module(m)
    rc = po.render_cfg_from_module(m, 1 / 256, 16)
    with torch.no_grad():
        enc_a = m.encode_audio(f["auds"])
    _, _, stats = po.render_frame(om, rc, f["rays_o"].numpy(), f["rays_d"].numpy(), enc_a.numpy(),
                                  m.individual_codes[0].detach().numpy(), f["eye"].numpy(), f["bg_coords"].numpy(),
                                  f["poses"].numpy(), m.individual_codes_torso[0].detach().numpy(),
                                  f["bg_color"].reshape(-1, 3).numpy())
    assert stats["live_samples"] == 0
    assert abs(stats["torso_pixels"] - expect) <= 1 and expect > 100

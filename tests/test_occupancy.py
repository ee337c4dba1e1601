"""Occupancy-grid maintenance (reference nerf/renderer.py:318-499).

CPU: the oracle's restatement (oracle/orc_occupancy.c) against independent numpy / torch formulations of the cited lines --
     dense 3-D dilation after a morton -> xyz reorder, numpy.packbits, torch max_pool2d, a float64 frustum test.
GPU: the kernels of rad-nerf_amd/csrc/rn_occupancy.hip through the C ABI and through NeRFRenderer.update_extra_state /
     mark_untrained_grid against that oracle: probe points, grid values and bitfield bit for bit (jitter pinned), sigma of
     the density query rel 2e-4, torso alphas abs 5e-5.
"""
import ctypes as C

import numpy as np
import pytest
import torch


def _morton_np(x, y, z):
    from radnerf.scene import morton3d_np
    return morton3d_np(x, y, z).astype(np.int64)


def _dense_from_morton(v, H):
    idx = np.arange(H)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    return v[_morton_np(X.reshape(-1), Y.reshape(-1), Z.reshape(-1))].reshape(H, H, H)


# ------------------------------------------------------------------------------------------------------------ CPU
def test_builtin_jitter_hash_is_the_same_on_both_sides_and_uniform(po, hiplib):
    import radnerf.occupancy  # noqa: F401  (declares the ctypes signature)
    lib = po.lib()
    lib.orc_hash_u01_bits.restype = C.c_uint32
    vals = []
    for seed in (0, 1, 0xDEADBEEF):
        for idx in (0, 1, 2, 12345, 6291455, 0xFFFFFFFF):
            a = lib.orc_hash_u01_bits(C.c_uint32(seed), C.c_uint32(idx))
            assert a == hiplib._lib.rn_hash_u01_bits(seed, idx) and a < (1 << 24)
            vals.append(a)
    u = np.array([lib.orc_hash_u01_bits(C.c_uint32(7), C.c_uint32(i)) for i in range(20000)]) / 2.0 ** 24
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005 and u.min() >= 0 and u.max() < 1


def test_oracle_probe_points_follow_the_cited_expressions(po):
    Cc, H, bound = 2, 16, 2.0
    rng = np.random.default_rng(0)
    noise = rng.uniform(0, 1, (Cc * H ** 3, 3)).astype(np.float32)
    got = po.occupancy_points(Cc, H, bound, noise)
    idx = np.arange(H, dtype=np.int32)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    coords = np.stack([X.reshape(-1), Y.reshape(-1), Z.reshape(-1)], 1)
    mo = _morton_np(coords[:, 0], coords[:, 1], coords[:, 2])
    for cas in range(Cc):
        b = min(2 ** cas, bound)
        half = b / H
        # renderer.py:421-430 in torch, element by element
        xyzs = 2 * torch.from_numpy(coords).float() / (H - 1) - 1
        cas_xyzs = xyzs * (b - half)
        nz = torch.from_numpy(noise[cas * H ** 3 + mo])
        cas_xyzs += (nz * 2 - 1) * half
        assert np.array_equal(got[cas * H ** 3 + mo], cas_xyzs.numpy())
    # the built-in jitter stays inside the cell
    own = po.occupancy_points(1, H, 1.0, None, seed=5).reshape(-1, 3)
    centre = po.occupancy_points(1, H, 1.0, np.full((H ** 3, 3), 0.5, np.float32))
    assert np.abs(own - centre).max() <= 1.0 / H + 1e-6 and np.abs(own - centre).max() > 0.9 / H


def test_oracle_grid_update_matches_numpy(po):
    Cc, H = 2, 16
    rng = np.random.default_rng(1)
    sig = rng.uniform(0, 30, (Cc, H ** 3)).astype(np.float32)
    grid = rng.uniform(0, 20, (Cc, H ** 3)).astype(np.float32)
    grid[rng.uniform(size=grid.shape) < 0.2] = -1.0                     # untrained cells
    g0 = grid.copy()
    bits, mean, thresh = po.occupancy_update(sig, 1.0, grid, Cc, H, 0.95, 10.0)
    for cas in range(Cc):
        d = _dense_from_morton(sig[cas], H)
        pad = np.pad(d, 1, constant_values=-np.inf)
        dil = np.max(np.stack([pad[1:-1, 1:-1, 1:-1], pad[2:, 1:-1, 1:-1], pad[:-2, 1:-1, 1:-1], pad[1:-1, 2:, 1:-1],
                               pad[1:-1, :-2, 1:-1], pad[1:-1, 1:-1, 2:], pad[1:-1, 1:-1, :-2]]), 0)
        old = _dense_from_morton(g0[cas], H)
        want = np.where(old >= 0, np.maximum(old * np.float32(0.95), dil), old)
        assert np.array_equal(_dense_from_morton(grid[cas], H), want)
    m = np.float32(np.clip(grid, 0, None).astype(np.float64).mean())
    assert mean == m and thresh == min(m, np.float32(10.0))
    assert np.array_equal(bits, np.packbits(grid.reshape(-1) > thresh, bitorder="little"))
    assert abs(float(torch.from_numpy(grid).clamp(min=0).mean()) - mean) <= 4e-7 * mean       # torch.mean (float32 sum): a few ulps


def test_oracle_mark_untrained_matches_a_float64_frustum_test(po):
    from radnerf.rays import orbit_pose
    Cc, H, bound = 2, 32, 2.0
    poses = np.stack([orbit_pose(3.35, yaw, pitch) for yaw, pitch in ((0, 0), (25, 5), (-30, -8), (10, 20))]).astype(np.float32)
    fx = fy = 400.0
    cx, cy = 96.0, 128.0
    grid = np.zeros((Cc, H ** 3), np.float32)
    po.mark_untrained_grid(poses, (fx, fy, cx, cy), Cc, H, bound, grid)
    assert set(np.unique(grid)) <= {0.0, -1.0} and 0.02 < (grid < 0).mean() < 0.98
    idx = np.arange(H)
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    coords = np.stack([X.reshape(-1), Y.reshape(-1), Z.reshape(-1)], 1).astype(np.float64)
    mo = _morton_np(X.reshape(-1), Y.reshape(-1), Z.reshape(-1))
    wrong = total = 0
    for cas in range(Cc):
        b = min(2 ** cas, bound)
        half = b / H
        w = (2 * coords / (H - 1) - 1) * (b - half)
        seen = np.zeros(len(w), bool)
        sure = np.ones(len(w), bool)
        for P in poses.astype(np.float64):
            cam = (w - P[:3, 3]) @ P[:3, :3]
            mx = cx / fx * cam[:, 2] + half * 2 - np.abs(cam[:, 0])
            my = cy / fy * cam[:, 2] + half * 2 - np.abs(cam[:, 1])
            seen |= (cam[:, 2] > 0) & (mx > 0) & (my > 0)
            sure &= (np.abs(cam[:, 2]) > 1e-4) & (np.abs(mx) > 1e-4) & (np.abs(my) > 1e-4)    # away from float32 knife edges
        got_seen = grid[cas, mo] == 0
        wrong += int((got_seen != seen)[sure].sum())
        total += int(sure.sum())
    assert wrong == 0 and total > 0.95 * Cc * H ** 3


def test_oracle_torso_grid_update_matches_torch_max_pool(po):
    import torch.nn.functional as F
    H = 32
    rng = np.random.default_rng(3)
    alphas = rng.uniform(0, 1, H * H).astype(np.float32)
    grid = rng.uniform(0, 1, H * H).astype(np.float32)
    g0 = torch.from_numpy(grid.copy())
    mean = po.torso_grid_update(alphas, grid, H, 0.95)
    pooled = F.max_pool2d(torch.from_numpy(alphas).view(1, 1, H, H), kernel_size=5, stride=1, padding=2).view(-1)
    want = torch.maximum(g0 * 0.95, pooled)                                   # renderer.py:486-489
    assert np.array_equal(grid, want.numpy()) and abs(mean - float(want.mean())) <= 2e-7
    pts = po.torso_grid_points(H, np.full((H * H, 2), 0.5, np.float32))
    assert np.allclose(pts[1], [(2 * 1 / (H - 1) - 1) * (1 - 1 / H), -(1 - 1 / H)], atol=1e-7)   # element 1 = column 1, row 0


# ------------------------------------------------------------------------------------------------------------ GPU
DEV = "cuda"


def _scene(torso, **kw):
    from radnerf.scene import SyntheticScene, default_opt
    sc = SyntheticScene(H=32, W=32, n_frames=8, device=DEV, opt=default_opt(engine="fused", torso=torso, **kw))
    m = sc.model
    m.aud_features, m.poses = sc.aud_features, sc.poses
    m.eye_area = torch.full((sc.n_frames, 1), 0.25, device=DEV)
    return sc


@pytest.mark.gpu
@pytest.mark.parametrize("Cc,H,bound", [(1, 128, 1.0), (2, 32, 2.0)])
def test_probe_points_bit_exact(po, hiplib, Cc, H, bound):
    import radnerf_hip as hip
    from radnerf import occupancy  # noqa: F401
    n = Cc * H ** 3
    noise = torch.rand(n, 3, device=DEV)
    out = torch.empty(n, 3, device=DEV)
    hip.call("rn_occupancy_points", Cc, H, bound, hip.ptr(noise), 0, hip.ptr(out), hip.stream())
    assert np.array_equal(out.cpu().numpy(), po.occupancy_points(Cc, H, bound, noise.cpu().numpy()))
    hip.call("rn_occupancy_points", Cc, H, bound, None, 1234, hip.ptr(out), hip.stream())
    assert np.array_equal(out.cpu().numpy(), po.occupancy_points(Cc, H, bound, None, seed=1234))
    xy = torch.empty(H * H, 2, device=DEV)
    hip.call("rn_torso_grid_points", H, None, 99, hip.ptr(xy), hip.stream())
    assert np.array_equal(xy.cpu().numpy(), po.torso_grid_points(H, None, seed=99))


@pytest.mark.gpu
@pytest.mark.parametrize("Cc,H", [(1, 128), (2, 32)])
def test_grid_update_bit_exact(po, hiplib, Cc, H):
    import radnerf_hip as hip
    from radnerf import occupancy
    rng = np.random.default_rng(Cc)
    sig = rng.uniform(0, 30, (Cc, H ** 3)).astype(np.float32)
    grid = rng.uniform(0, 20, (Cc, H ** 3)).astype(np.float32)
    grid[rng.uniform(size=grid.shape) < 0.2] = -1.0
    want = grid.copy()
    bits, mean, thresh = po.occupancy_update(sig, 1.0, want, Cc, H, 0.95, 10.0)
    g, s = torch.from_numpy(grid).to(DEV), torch.from_numpy(sig).to(DEV)
    bf = torch.zeros(Cc * H ** 3 // 8, dtype=torch.uint8, device=DEV)
    stats = torch.zeros(2, device=DEV)
    ws = torch.zeros(int(occupancy._lib.rn_occupancy_workspace(Cc, H)), dtype=torch.uint8, device=DEV)
    for _ in range(2):                                   # second round: the arrival counter was left at zero
        g.copy_(torch.from_numpy(grid))
        hip.call("rn_occupancy_update", hip.ptr(s), 1.0, hip.ptr(g), Cc, H, 0.95, 10.0, hip.ptr(bf), hip.ptr(stats), hip.ptr(ws),
                 hip.stream())
        assert np.array_equal(g.cpu().numpy(), want)
        assert stats.cpu().tolist() == [mean, thresh]
        assert np.array_equal(bf.cpu().numpy(), bits)


@pytest.mark.gpu
def test_mark_untrained_grid_bit_exact(po, hiplib):
    sc = _scene(False)
    m = sc.model
    m.density_grid.zero_()
    intr = (400.0, 400.0, 96.0, 128.0)
    m.mark_untrained_grid(sc.poses[:6], intr)
    want = np.zeros((m.cascade, m.grid_size ** 3), np.float32)
    po.mark_untrained_grid(sc.poses[:6].cpu().numpy(), intr, m.cascade, m.grid_size, float(m.bound), want)
    assert np.array_equal(m.density_grid.cpu().numpy(), want) and 0.05 < (want < 0).mean() < 0.95
    m.density_grid.zero_()
    m.mark_untrained_grid(sc.poses[:6, :3].contiguous(), intr)            # [B,3,4] matrices
    assert np.array_equal(m.density_grid.cpu().numpy(), want)
    m.density_grid.zero_()
    m.mark_untrained_grid(sc.poses[:6].cpu().numpy(), intr)               # numpy input, as the reference's callers pass it
    assert np.array_equal(m.density_grid.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("mlp", ["f32", "f32x2"])
def test_density_query_is_the_sigma_of_the_full_forward(po, hiplib, mlp):
    from radnerf import fused
    sc = _scene(False, mlp_dtype=mlp)
    m = sc.model
    x = torch.rand(70001, 3, device=DEV) * 1.6 - 0.8
    enc_a = torch.randn(1, 64, device=DEV)
    eye = torch.tensor([[0.25]], device=DEV)
    with torch.no_grad():
        sig = fused.density_forward(m, x, enc_a, eye)
        d = torch.zeros_like(x)
        d[:, 2] = 1
        full = fused.network_forward(m, x, d, enc_a, m.individual_codes[0], eye, want_ambient=False)[0]
    assert torch.equal(sig, full)                                         # same instructions up to sigma
    want = po.nerf_density(po.model_from_module(m), x.cpu().numpy(), enc_a.cpu().numpy(), eye.cpu().numpy())
    np.testing.assert_allclose(sig.cpu().numpy(), want, rtol=2e-4, atol=1e-6)


@pytest.mark.gpu
def test_update_extra_state_head_matches_oracle_chain(po, hiplib):
    import random
    sc = _scene(False)
    m = sc.model
    m.train()
    g = torch.Generator(device=DEV).manual_seed(4)
    noise = torch.rand(m.cascade * m.grid_size ** 3, 3, device=DEV, generator=g)
    with torch.no_grad():
        m.density_grid.uniform_(0, 2.0)
        m.density_grid[0, ::7] = -1.0
    g0 = m.density_grid.cpu().numpy().copy()
    random.seed(11)
    m.update_extra_state(noise=noise)
    # oracle chain with the same random audio window
    random.seed(11)
    pick = random.randint(0, m.aud_features.shape[0] - 1)
    from radnerf.rays import get_audio_features
    with torch.no_grad():
        enc_a = m.encode_audio(get_audio_features(m.aud_features, m.att, pick)).cpu().numpy()
    pts = po.occupancy_points(m.cascade, m.grid_size, float(m.bound), noise.cpu().numpy())
    sig = po.nerf_density(po.model_from_module(m), pts, enc_a, np.array([[0.25]], np.float32))
    want = g0.copy()
    bits, mean, thresh = po.occupancy_update(sig, 1.0, want, m.cascade, m.grid_size, 0.95, float(m.density_thresh))
    got = m.density_grid.cpu().numpy()
    assert np.array_equal(got < 0, want < 0)
    np.testing.assert_allclose(got, want, rtol=3e-4, atol=1e-6)            # sigma differs by summation order inside the MFMA tiles
    assert abs(m.mean_density - mean) <= 3e-4 * mean
    assert float((np.unpackbits(m.density_bitfield.cpu().numpy()) != np.unpackbits(bits)).mean()) < 2e-4   # cells on the threshold
    assert m.iter_density == 1 and m.local_step == 0
    # the same sigmas into both updates: bit for bit
    from radnerf import occupancy
    sg = occupancy._scratch(m).sigmas.cpu().numpy()
    again = g0.copy()
    bits2, mean2, _ = po.occupancy_update(sg, 1.0, again, m.cascade, m.grid_size, 0.95, float(m.density_thresh))
    assert np.array_equal(got, again) and np.array_equal(m.density_bitfield.cpu().numpy(), bits2) and m.mean_density == mean2


@pytest.mark.gpu
def test_update_extra_state_torso_matches_oracle_chain(po, hiplib):
    import random
    sc = _scene(True)
    m = sc.model
    m.train()
    H = m.grid_size
    noise = torch.rand(H * H, 2, device=DEV)
    g0 = m.density_grid_torso.cpu().numpy().copy()
    random.seed(5)
    m.update_extra_state(noise=noise)
    random.seed(5)
    random.randint(0, m.aud_features.shape[0] - 1)
    pick = random.randint(0, m.poses.shape[0] - 1)
    from radnerf.rays import convert_poses
    pose6 = convert_poses(m.poses[[pick]]).cpu().numpy()
    pts = po.torso_grid_points(H, noise.cpu().numpy())
    alpha, _, _ = po.torso_forward(po.model_from_module(m), pts, pose6, m.individual_codes_torso[pick].detach().cpu().numpy())
    want = g0.copy()
    mean = po.torso_grid_update(alpha.reshape(-1), want, H, 0.95)
    np.testing.assert_allclose(m.density_grid_torso.cpu().numpy(), want, rtol=0, atol=5e-5)
    assert abs(m.mean_density_torso - mean) <= 5e-5
    from radnerf import occupancy
    al = occupancy._scratch(m).alphas.cpu().numpy().reshape(-1)
    again = g0.copy()
    mean2 = po.torso_grid_update(al, again, H, 0.95)
    assert np.array_equal(m.density_grid_torso.cpu().numpy(), again) and m.mean_density_torso == mean2


@pytest.mark.gpu
def test_torso_layer_training_formulation_equals_the_fused_pass(hiplib):
    """The differentiable gather / index_copy formulation (training) and the fused torso kernel (inference) blend the same
    background; gradients reach the torso networks."""
    sc = _scene(True)
    m = sc.model
    f = sc.frame(0)
    coords = f["bg_coords"].reshape(-1, 2)
    with torch.no_grad():
        enc_a = m.encode_audio(f["auds"])
        res_a = {}
        bg_a = m._torso_layer(coords, f["poses"], enc_a, 0, 1, res_a)
    res_b = {}
    bg_b = m._torso_layer(coords, f["poses"], enc_a, 0, 1, res_b)           # grad enabled -> PyTorch layers
    assert (bg_a - bg_b).abs().max().item() <= 5e-5 and (res_a["torso_alpha"] - res_b["torso_alpha"]).abs().max().item() <= 5e-5
    assert (res_b["torso_alpha"] > 0).sum().item() > 100
    bg_b.sum().backward()
    assert m.torso_net.net[0].weight.grad.abs().sum().item() > 0 and m.torso_encoder.embeddings.grad.abs().sum().item() > 0

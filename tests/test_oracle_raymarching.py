"""Pins of the CPU oracle's ray-marching functions against independent numpy derivations
(the reference ships no tests or goldens -- SURVEY §4/§8c -- so these are the pins)."""
import numpy as np
import pytest

AABB = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)


def _rays(rng, N, spread=0.7):
    o = np.tile(np.array([[0.05, 3.3, -0.1]], np.float32), (N, 1))
    tgt = rng.uniform(-spread, spread, (N, 3)).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def _bits():
    from radnerf.scene import ellipsoid_bitfield
    return ellipsoid_bitfield(128, 1.0, (0.40, 0.42, 0.40))


def test_morton_against_naive_bit_interleave_all_cells(po):
    H = 128
    idx = np.arange(H, dtype=np.int64)
    c = np.stack(np.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3)
    naive = np.zeros(len(c), np.int64)
    for b in range(7):
        naive |= ((c[:, 0] >> b) & 1) << (3 * b)
        naive |= ((c[:, 1] >> b) & 1) << (3 * b + 1)
        naive |= ((c[:, 2] >> b) & 1) << (3 * b + 2)
    m = po.morton3D(c.astype(np.int32))
    assert np.array_equal(m.astype(np.int64), naive)
    assert np.array_equal(po.morton3D_invert(m), c.astype(np.int32))
    # 10-bit coordinates (the documented range of the bit tricks)
    big = np.array([[1023, 0, 0], [0, 1023, 0], [0, 0, 1023], [1023, 1023, 1023]], np.int32)
    assert po.morton3D(big).astype(np.uint32).tolist() == [0x09249249, 0x12492492, 0x24924924, 0x3FFFFFFF]


def test_scene_morton_matches_oracle(po, hiplib):
    from radnerf.scene import morton3d_np
    rng = np.random.default_rng(0)
    c = rng.integers(0, 128, (10000, 3)).astype(np.int32)
    assert np.array_equal(morton3d_np(c[:, 0], c[:, 1], c[:, 2]).astype(np.int32), po.morton3D(c))


def test_packbits_against_numpy(po, rng):
    g = rng.uniform(0, 1, (2, 4096)).astype(np.float32)
    assert np.array_equal(po.packbits(g, 0.5), np.packbits(g.reshape(-1) > 0.5, bitorder="little"))
    assert po.packbits(np.full((1, 8), 0.5, np.float32), 0.5)[0] == 0  # strict >


def test_dilation_against_dense_6_neighbour_max(po, rng):
    H = 16
    g = rng.uniform(-1, 3, (2, H ** 3)).astype(np.float32)
    out = po.morton3D_dilation(g)
    idx = np.arange(H)
    c = np.stack(np.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    m = po.morton3D(c)
    for cas in range(2):
        dense = np.full((H + 2, H + 2, H + 2), -np.inf, np.float32)
        dense[1:-1, 1:-1, 1:-1][c[:, 0], c[:, 1], c[:, 2]] = g[cas, m]
        ctr = dense[1:-1, 1:-1, 1:-1]
        mx = np.maximum.reduce([ctr, dense[2:, 1:-1, 1:-1], dense[:-2, 1:-1, 1:-1], dense[1:-1, 2:, 1:-1],
                                dense[1:-1, :-2, 1:-1], dense[1:-1, 1:-1, 2:], dense[1:-1, 1:-1, :-2]])
        assert np.array_equal(out[cas, m], mx[c[:, 0], c[:, 1], c[:, 2]])


def test_near_far_against_numpy_slab(po, rng):
    N = 20000
    o, d = _rays(rng, N, spread=2.0)
    nears, fars = po.near_far_from_aabb(o, d, AABB, 0.05)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (AABB[:3] - o) / d
        t1 = (AABB[3:] - o) / d
    lo, hi = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
    hit = lo <= hi
    fm = np.finfo(np.float32).max
    assert np.array_equal(nears == fm, ~hit) and np.array_equal(fars == fm, ~hit)
    np.testing.assert_allclose(nears[hit], np.maximum(lo[hit], 0.05), rtol=1e-6)
    np.testing.assert_allclose(fars[hit], hi[hit], rtol=1e-6)
    assert hit.any() and (~hit).any()
    # published trace sanity (raymarching/raymarching.py:440-446): camera at y~3.35 -> near~2.9, far~3.9
    assert 2.7 < nears[hit].min() and fars[hit].max() < 4.3


def test_march_rays_invariants_and_step_size(po, rng):
    N, n_step = 4000, 5
    o, d = _rays(rng, N, spread=1.5)
    nears, fars = po.near_far_from_aabb(o, d, AABB, 0.05)
    bits, dens = _bits()
    alive = np.arange(N, dtype=np.int32)
    xyzs, dirs, deltas = po.march_rays(N, n_step, alive, nears, o, d, 1.0, 1 / 256, 16, 1, 128, bits, nears, fars,
                                       np.zeros(N, np.float32))
    live = deltas[:, 0] > 0
    assert live.any()
    # constant step at cascade=1, H=128, max_steps=16: dt = 2*sqrt(3)/128
    assert np.allclose(deltas[live, 0], 2 * np.sqrt(3) / 128, rtol=1e-6)
    # every emitted sample lies in an occupied cell and inside the aabb
    p = xyzs[live]
    cell = np.clip((0.5 * (p.astype(np.float64) + 1) * 128).astype(np.float32), 0, 127).astype(np.int32)
    occ = np.unpackbits(bits, bitorder="little")[po.morton3D(cell)]
    assert occ.all()
    assert (np.abs(p) <= 1).all()
    # live slots of a ray are a prefix of its n_step slots; deltas[1] increases by dt
    lv = live.reshape(N, n_step)
    assert np.array_equal(lv, np.sort(lv, axis=1)[:, ::-1])
    d1 = deltas[:, 1].reshape(N, n_step)
    inc = np.diff(d1, axis=1)[lv[:, 1:]]
    assert (inc >= 2 * np.sqrt(3) / 128 - 1e-5).all()
    # dirs of live samples equal the ray direction, dead slots stay zero
    assert np.array_equal(dirs[live], np.repeat(d, n_step, 0)[live])
    assert not xyzs[~live].any() and not deltas[~live].any()
    # rays that miss the box emit nothing
    miss = nears == np.finfo(np.float32).max
    assert not lv[miss].any()


def test_march_rays_train_matches_inference_marcher(po, rng):
    N = 1500
    o, d = _rays(rng, N)
    nears, fars = po.near_far_from_aabb(o, d, AABB, 0.05)
    bits, _ = _bits()
    noises = rng.uniform(0, 1, N).astype(np.float32)
    xyzs, dirs, deltas, rays, cnt = po.march_rays_train(o, d, bits, 1.0, 1 / 256, 16, 1, 128, N * 16, nears, fars, noises)
    assert cnt[1] == N and cnt[0] == rays[:, 2].sum()
    assert np.array_equal(rays[:, 0], np.arange(N))
    assert np.array_equal(rays[:, 1], np.concatenate([[0], np.cumsum(rays[:, 2])[:-1]]))
    ix, idr, idl = po.march_rays(N, 16, np.arange(N, dtype=np.int32), nears, o, d, 1.0, 1 / 256, 16, 1, 128, bits, nears,
                                 fars, noises)
    for r in (0, 7, 100, N - 1):
        off, k = rays[r, 1], rays[r, 2]
        assert np.array_equal(xyzs[off:off + k], ix[r * 16:r * 16 + k])
        assert np.array_equal(deltas[off:off + k], idl[r * 16:r * 16 + k])
    # budget overflow drops whole rays and leaves their slots zero
    M = int(cnt[0]) // 2
    x2, _, dl2, rays2, cnt2 = po.march_rays_train(o, d, bits, 1.0, 1 / 256, 16, 1, 128, M, nears, fars, noises)
    assert np.array_equal(cnt2, cnt)
    dropped = rays2[:, 1] + rays2[:, 2] > M
    assert dropped.any()
    assert np.array_equal(x2[:M][: rays2[~dropped, 1].max()], xyzs[: rays2[~dropped, 1].max()])


def test_composite_rays_against_cumprod(po, rng):
    N, n_step = 500, 8
    sig = rng.uniform(0, 30, (N, n_step)).astype(np.float32)
    rgb = rng.uniform(0, 1, (N, n_step, 3)).astype(np.float32)
    dt = np.full((N, n_step), 0.027, np.float32)
    tt = 3 + np.cumsum(dt, 1)
    nlive = rng.integers(0, n_step + 1, N)
    mask = np.arange(n_step)[None] < nlive[:, None]
    deltas = np.stack([dt * mask, tt * mask], -1).astype(np.float32)
    alive = np.arange(N, dtype=np.int32)
    rays_t = np.full(N, 3, np.float32)
    ws, dp, im = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    po.composite_rays(N, n_step, 0.0, alive, rays_t, sig.reshape(-1), rgb.reshape(-1, 3), deltas.reshape(-1, 2), ws, dp, im)
    alpha = (1 - np.exp(-sig.astype(np.float64) * dt)) * mask
    T = np.cumprod(np.concatenate([np.ones((N, 1)), 1 - alpha[:, :-1]], 1), 1)
    w = alpha * T
    np.testing.assert_allclose(ws, w.sum(1), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(im, (w[..., None] * rgb).sum(1), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(dp, (w * tt).sum(1), rtol=2e-5, atol=1e-5)
    assert np.array_equal(alive < 0, nlive < n_step)          # a ray with an empty slot is terminated
    assert np.array_equal(rays_t[nlive == n_step], tt[nlive == n_step, -1].astype(np.float32))
    assert (ws <= 1 + 1e-6).all()


def test_composite_rays_early_termination_semantics(po):
    """T is tested BEFORE adding the sample (raymarching.cu:992,1006): one extra sample is blended."""
    n_step = 4
    sig = np.array([1000, 1000, 1000, 1000], np.float32)
    rgb = np.ones((4, 3), np.float32)
    deltas = np.array([[0.1, 1.1], [0.1, 1.2], [0.1, 1.3], [0.1, 1.4]], np.float32)
    alive = np.array([0], np.int32); rays_t = np.array([1.0], np.float32)
    ws, dp, im = np.zeros(1, np.float32), np.zeros(1, np.float32), np.zeros((1, 3), np.float32)
    po.composite_rays(1, n_step, 1e-4, alive, rays_t, sig, rgb, deltas, ws, dp, im)
    assert alive[0] == -1 and rays_t[0] == 1.0  # terminated early, rays_t untouched
    assert abs(ws[0] - 1.0) < 1e-6


def test_composite_train_forward_against_cumprod_and_backward_against_autograd(po, rng):
    import torch
    N, S = 64, 12
    counts = rng.integers(0, S + 1, N)
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    M = int(counts.sum()) + 5
    rays = np.stack([rng.permutation(N), offs, counts], 1).astype(np.int32)
    sig = rng.uniform(0, 20, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    amb = rng.uniform(0, 1, M).astype(np.float32)
    deltas = np.stack([np.full(M, 0.027), rng.uniform(3, 4, M)], 1).astype(np.float32)
    ws, am, dp, im = po.composite_rays_train_forward(sig, rgb, amb, deltas, rays, 0.0)

    ts = torch.tensor(sig, dtype=torch.float64, requires_grad=True)
    tr = torch.tensor(rgb, dtype=torch.float64, requires_grad=True)
    tws = torch.zeros(N, dtype=torch.float64); tim = torch.zeros(N, 3, dtype=torch.float64)
    tdp = torch.zeros(N, dtype=torch.float64); tam = np.zeros(N)
    for i in range(N):
        idx, o, k = rays[i]
        if k == 0:
            continue
        a = 1 - torch.exp(-ts[o:o + k] * torch.tensor(deltas[o:o + k, 0], dtype=torch.float64))
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1 - a[:-1]]), 0)
        w = a * T
        tws[idx] = w.sum(); tim[idx] = (w[:, None] * tr[o:o + k]).sum(0)
        tdp[idx] = (w * torch.tensor(deltas[o:o + k, 1], dtype=torch.float64)).sum()
        tam[idx] = amb[o:o + k].sum()
    np.testing.assert_allclose(ws, tws.detach().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(im, tim.detach().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(dp, tdp.detach().numpy(), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(am, tam, rtol=1e-5)

    g_ws = rng.standard_normal(N).astype(np.float32); g_im = rng.standard_normal((N, 3)).astype(np.float32)
    g_am = rng.standard_normal(N).astype(np.float32)
    ((tws * torch.tensor(g_ws, dtype=torch.float64)).sum() + (tim * torch.tensor(g_im, dtype=torch.float64)).sum()).backward()
    gs, gr, ga = po.composite_rays_train_backward(g_ws, g_am, g_im, sig, rgb, amb, deltas, rays, ws, am, im, 0.0)
    used = np.zeros(M, bool)
    for i in range(N):
        used[rays[i, 1]:rays[i, 1] + rays[i, 2]] = True
    np.testing.assert_allclose(gr[used], tr.grad.numpy()[used], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gs[used], ts.grad.numpy()[used], rtol=2e-3, atol=2e-4)
    for i in range(N):
        o, k = rays[i, 1], rays[i, 2]
        assert np.all(ga[o:o + k] == g_am[rays[i, 0]])


def test_march_rays_train_backward_against_autograd(po, rng):
    N, S = 40, 6
    counts = rng.integers(0, S + 1, N)
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
    M = int(counts.sum())
    rays = np.stack([np.arange(N), offs, counts], 1).astype(np.int32)
    deltas = rng.uniform(0.1, 4, (M, 2)).astype(np.float32)
    gx = rng.standard_normal((M, 3)).astype(np.float32); gd = rng.standard_normal((M, 3)).astype(np.float32)
    go, gdd = po.march_rays_train_backward(gx, gd, rays, deltas)
    # xyz = o + t d, dir = d  =>  d/do = sum gx, d/dd = sum (t gx + gd)
    for i in range(N):
        o, k = offs[i], counts[i]
        np.testing.assert_allclose(go[i], gx[o:o + k].sum(0), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(gdd[i], (gx[o:o + k] * deltas[o:o + k, 1:2] + gd[o:o + k]).sum(0), rtol=1e-5, atol=1e-5)

"""GPU parity: every drop-in operator (called through the C ABI by the Python wrappers) against the
CPU oracle on the same seeded inputs.

Bars: integer / index / byte outputs bit-exact; float outputs bit-exact where the op order is the same
on both sides (-ffp-contract=off: DDA, grid interpolation in fp32 and fp16), otherwise a stated tolerance
(fast exp in compositing, sinf in the frequency encoder, atomics order in the grid backward).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(DEV)


def n(x):
    return x.detach().cpu().numpy()


_KEEP = []


def dp(a):
    """Upload a numpy array and return its device pointer; the tensor is kept alive for the test's duration
    (a temporary would be freed -- and its block reused -- before the kernel is enqueued)."""
    import radnerf_hip as hip
    x = t(a)
    _KEEP.append(x)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return hip.ptr(x)


def make_rays(rng, N, spread=0.35):
    """Camera on +y looking at the origin (OrbitCamera convention), plus a few degenerate rays."""
    o = np.tile(np.array([[0.05, 3.3, -0.1]], np.float32), (N, 1))
    tgt = rng.uniform(-spread * 2, spread * 2, (N, 3)).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    if N >= 8:
        d[0] = (0, -1, 0)          # axis-parallel: 1/dx = inf
        d[1] = (1, 0, 0)           # misses the box
        o[2] = (0, 0, 0)           # origin inside the box
        d[3] = (0, 1, 0)           # looks away
    return o, d


def ellipsoid_bits(H=128, axes=(0.40, 0.42, 0.40)):
    from radnerf.scene import ellipsoid_bitfield
    bits, dens = ellipsoid_bitfield(H, 1.0, axes)
    return bits, dens


AABB = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)

# ------------------------------------------------------------------------------------------------ utils


@pytest.mark.parametrize("N", [1, 63, 64, 1000, 262144])
def test_near_far_bit_exact(po, hiplib, rng, N):
    import raymarching
    o, d = make_rays(rng, N)
    nears, fars = raymarching.near_far_from_aabb(t(o), t(d), t(AABB), 0.05)
    rn_, rf_ = po.near_far_from_aabb(o, d, AABB, 0.05)
    assert np.array_equal(n(nears), rn_) and np.array_equal(n(fars), rf_)
    assert (rn_ == np.finfo(np.float32).max).any() or N < 8


def test_near_far_empty(hiplib):
    import raymarching
    nears, fars = raymarching.near_far_from_aabb(torch.empty(0, 3, device=DEV), torch.empty(0, 3, device=DEV), t(AABB), 0.05)
    assert nears.shape == (0,) and fars.shape == (0,)


def test_sph_from_ray(po, hiplib, rng):
    import raymarching
    o = rng.uniform(-0.3, 0.3, (5000, 3)).astype(np.float32)
    d = rng.standard_normal((5000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    got = n(raymarching.sph_from_ray(t(o), t(d), 2.0))
    np.testing.assert_allclose(got, po.sph_from_ray(o, d, 2.0), rtol=0, atol=2e-6)


def test_morton_roundtrip_full_grid_bit_exact(po, hiplib):
    import raymarching
    H = 128
    idx = np.arange(H, dtype=np.int32)
    c = np.stack(np.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3)
    m = raymarching.morton3D(t(c))
    assert np.array_equal(n(m), po.morton3D(c))
    back = raymarching.morton3D_invert(m)
    assert np.array_equal(n(back), c)
    assert np.array_equal(np.sort(n(m)), np.arange(H ** 3, dtype=np.int32))  # a permutation of the cells


def test_packbits_bit_exact(po, hiplib, rng):
    import raymarching
    g = rng.uniform(0, 20, (1, 128 ** 3)).astype(np.float32)
    g[0, :64] = 10.0  # equal to the threshold -> not set (strict >)
    got = raymarching.packbits(t(g), 10.0)
    assert np.array_equal(n(got), po.packbits(g, 10.0))
    assert np.array_equal(n(got), np.packbits(g.reshape(-1) > 10.0, bitorder="little"))
    # in-place variant writes into the caller's buffer
    buf = torch.zeros(128 ** 3 // 8, dtype=torch.uint8, device=DEV)
    out = raymarching.packbits(t(g), 5.0, buf)
    assert out.data_ptr() == buf.data_ptr() and np.array_equal(n(buf), po.packbits(g, 5.0))


def test_dilation_bit_exact(po, hiplib, rng):
    import raymarching
    g = rng.uniform(-1, 5, (2, 32 ** 3)).astype(np.float32)
    assert np.array_equal(n(raymarching.morton3D_dilation(t(g))), po.morton3D_dilation(g))
    g = rng.uniform(-1, 5, (1, 128 ** 3)).astype(np.float32)
    assert np.array_equal(n(raymarching.morton3D_dilation(t(g))), po.morton3D_dilation(g))


# ------------------------------------------------------------------------------------------------ inference march


def _march_case(po, rng, N, n_step, perturb=False, dt_gamma=1 / 256, max_steps=16, bits=None):
    o, d = make_rays(rng, N)
    nears, fars = po.near_far_from_aabb(o, d, AABB, 0.05)
    if bits is None:
        bits, _ = ellipsoid_bits()
    keep = rng.permutation(N)[: max(1, (N * 3) // 4)].astype(np.int32)
    keep.sort()
    rays_t = nears.copy()
    rays_t[keep[::3]] += 0.3  # some rays already advanced
    noises = rng.uniform(0, 1, keep.shape[0]).astype(np.float32) if perturb else np.zeros(keep.shape[0], np.float32)
    return o, d, nears, fars, bits, keep, rays_t, noises, dt_gamma, max_steps


@pytest.mark.parametrize("N,n_step", [(1, 1), (100, 1), (5000, 3), (5000, 8), (100000, 2)])
def test_march_rays_bit_exact(po, hiplib, rng, N, n_step):
    import radnerf_hip as hip
    o, d, nears, fars, bits, alive, rays_t, noises, dt_gamma, max_steps = _march_case(po, rng, N, n_step, perturb=True)
    n_alive = alive.shape[0]
    M = n_alive * n_step + 128 - (n_alive * n_step) % 128
    ex, ed, edl = po.march_rays(n_alive, n_step, alive, rays_t, o, d, 1.0, dt_gamma, max_steps, 1, 128, bits, nears, fars, noises, M=M)
    xyzs = torch.zeros(M, 3, device=DEV); dirs = torch.zeros(M, 3, device=DEV); deltas = torch.zeros(M, 2, device=DEV)
    hip.call("rn_march_rays", n_alive, n_step, dp(alive), dp(rays_t), dp(o), dp(d), 1.0,
             dt_gamma, max_steps, 1, 128, dp(bits), dp(nears), dp(fars), hip.ptr(xyzs),
             hip.ptr(dirs), hip.ptr(deltas), dp(noises), None, hip.stream())
    assert np.array_equal(n(xyzs), ex) and np.array_equal(n(dirs), ed) and np.array_equal(n(deltas), edl)
    assert (edl[:, 0] > 0).sum() > 0 or N < 10


def test_march_rays_python_wrapper_and_cascades(po, hiplib, rng):
    """Wrapper path (padding rule, NULL noise) + bound=2 / cascade=2 (mip levels)."""
    import raymarching
    N, n_step = 4000, 4
    o, d = make_rays(rng, N, spread=0.8)
    aabb = np.array([-2, -1, -2, 2, 1, 2], np.float32)
    nears, fars = po.near_far_from_aabb(o, d, aabb, 0.05)
    bits = rng.integers(0, 256, 2 * 128 ** 3 // 8).astype(np.uint8) & rng.integers(0, 256, 2 * 128 ** 3 // 8).astype(np.uint8)
    alive = np.arange(N, dtype=np.int32)
    for dt_gamma in (0.0, 1 / 64):
        xyzs, dirs, deltas = raymarching.march_rays(N, n_step, t(alive), t(nears), t(o), t(d), 2.0, t(bits), 2, 128,
                                                    t(nears), t(fars), 128, False, dt_gamma, 64)
        M = N * n_step + 128 - (N * n_step) % 128
        assert xyzs.shape == (M, 3) and deltas.shape == (M, 2)
        ex, ed, edl = po.march_rays(N, n_step, alive, nears, o, d, 2.0, dt_gamma, 64, 2, 128, bits, nears, fars,
                                    np.zeros(N, np.float32), M=M)
        assert np.array_equal(n(xyzs), ex) and np.array_equal(n(dirs), ed) and np.array_equal(n(deltas), edl)


def test_composite_rays_parity(po, hiplib, rng):
    import raymarching
    N, n_step = 6000, 4
    o, d, nears, fars, bits, alive, rays_t, noises, dt_gamma, max_steps = _march_case(po, rng, N, n_step)
    n_alive = alive.shape[0]
    xyzs, dirs, deltas = po.march_rays(n_alive, n_step, alive, rays_t, o, d, 1.0, dt_gamma, max_steps, 1, 128, bits, nears, fars, noises)
    M = xyzs.shape[0]
    sig = rng.uniform(0, 400, M).astype(np.float32)  # opaque enough to trigger T < T_thresh
    sig[rng.uniform(size=M) < 0.5] *= 0.01
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    ws = rng.uniform(0, 0.3, N).astype(np.float32); dp = rng.uniform(0, 1, N).astype(np.float32)
    im = rng.uniform(0, 0.3, (N, 3)).astype(np.float32)
    e_alive, e_t, e_ws, e_dp, e_im = alive.copy(), rays_t.copy(), ws.copy(), dp.copy(), im.copy()
    po.composite_rays(n_alive, n_step, 1e-4, e_alive, e_t, sig, rgb, deltas, e_ws, e_dp, e_im)
    g_alive, g_t, g_ws, g_dp, g_im = t(alive), t(rays_t), t(ws), t(dp), t(im)
    out = raymarching.composite_rays(n_alive, n_step, g_alive, g_t, t(sig), t(rgb), t(deltas), g_ws, g_dp, g_im, 1e-4)
    assert out == tuple()
    # which rays die is decided by deltas == 0 / T < T_thresh: must agree exactly except on knife-edge T
    agree = (n(g_alive) == e_alive)
    assert agree.mean() > 0.999
    assert (e_alive < 0).any() and (e_alive >= 0).any()
    m = agree
    np.testing.assert_allclose(n(g_ws)[alive[m]], e_ws[alive[m]], rtol=2e-5, atol=2e-6)  # __expf vs expf
    np.testing.assert_allclose(n(g_im)[alive[m]], e_im[alive[m]], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(n(g_dp)[alive[m]], e_dp[alive[m]], rtol=2e-5, atol=1e-5)
    np.testing.assert_array_equal(n(g_t)[alive[m]], e_t[alive[m]])


@pytest.mark.parametrize("nn", [0, 1, 64, 1023, 1024, 1025, 70000, 262144])
def test_compact_rays_stable(hiplib, rng, nn):
    import raymarching
    a = rng.integers(0, 1 << 20, nn).astype(np.int32)
    a[rng.uniform(size=nn) < 0.6] = -1
    src = t(a) if nn else torch.empty(0, dtype=torch.int32, device=DEV)
    out, n_out = raymarching.compact_rays(src)
    k = int(n_out.item())
    assert k == int((a >= 0).sum())
    assert np.array_equal(n(out)[:k], a[a >= 0])
    if nn > 10:  # device-side count smaller than the launch bound
        nd = torch.tensor([nn // 2], dtype=torch.int32, device=DEV)
        out2, n2 = raymarching.compact_rays(src, n_alive_dev=nd)
        k2 = int(n2.item())
        assert np.array_equal(n(out2)[:k2], a[: nn // 2][a[: nn // 2] >= 0])


# ------------------------------------------------------------------------------------------------ training path


@pytest.mark.parametrize("N,mean_count,force", [(1, -1, False), (4096, -1, False), (4096, 8000, False), (5000, 8000, True)])
def test_march_rays_train_bit_exact(po, hiplib, rng, N, mean_count, force):
    import raymarching
    o, d = make_rays(rng, N)
    nears, fars = po.near_far_from_aabb(o, d, AABB, 0.05)
    bits, _ = ellipsoid_bits()
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    torch.manual_seed(7)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(t(o), t(d), 1.0, t(bits), 1, 128, t(nears), t(fars), counter,
                                                            mean_count, True, 128, force, 1 / 256, 16)
    torch.manual_seed(7)
    noises = n(torch.rand(N, dtype=torch.float32, device=DEV))
    M = N * 16
    if not force and mean_count > 0:
        M = mean_count + 128 - mean_count % 128
    ex, ed, edl, erays, ecnt = po.march_rays_train(o, d, bits, 1.0, 1 / 256, 16, 1, 128, M, nears, fars, noises)
    assert np.array_equal(n(counter), ecnt)
    assert np.array_equal(n(rays), erays)
    m = xyzs.shape[0]
    if force or mean_count <= 0:
        total = int(ecnt[0])
        assert m == min(M, total + 128 - total % 128)  # slicing past the buffer keeps what exists
    assert np.array_equal(n(xyzs), ex[:m]) and np.array_equal(n(dirs), ed[:m]) and np.array_equal(n(deltas), edl[:m])
    if mean_count > 0 and not force and N > 1:
        assert int(ecnt[0]) > M  # the budget really dropped rays


def test_composite_rays_train_fwd_bwd(po, hiplib, rng):
    import raymarching
    N = 3000
    o, d = make_rays(rng, N)
    nears, fars = po.near_far_from_aabb(o, d, AABB, 0.05)
    bits, _ = ellipsoid_bits()
    M = 20096
    xyzs, dirs, deltas, rays, cnt = po.march_rays_train(o, d, bits, 1.0, 1 / 256, 16, 1, 128, M, nears, fars, np.zeros(N, np.float32))
    sig = rng.uniform(0, 60, M).astype(np.float32)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    amb = rng.uniform(0, 1, M).astype(np.float32)
    ts, tr, ta = t(sig).requires_grad_(), t(rgb).requires_grad_(), t(amb).requires_grad_()
    ws, am, dp, im = raymarching.composite_rays_train(ts, tr, ta, t(deltas), t(rays), 1e-4)
    e_ws, e_am, e_dp, e_im = po.composite_rays_train_forward(sig, rgb, amb, deltas, rays, 1e-4)
    for g, e in ((ws, e_ws), (am, e_am), (dp, e_dp), (im, e_im)):
        np.testing.assert_allclose(n(g), e, rtol=3e-5, atol=3e-6)
    g_ws = rng.standard_normal(N).astype(np.float32); g_am = rng.standard_normal(N).astype(np.float32)
    g_im = rng.standard_normal((N, 3)).astype(np.float32)
    (ws * t(g_ws)).sum().add((am * t(g_am)).sum()).add((im * t(g_im)).sum()).backward()
    e_gs, e_gr, e_ga = po.composite_rays_train_backward(g_ws, g_am, g_im, sig, rgb, amb, deltas, rays, e_ws, e_am, e_im, 1e-4)
    np.testing.assert_allclose(n(ts.grad), e_gs, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(n(tr.grad), e_gr, rtol=3e-5, atol=3e-6)
    np.testing.assert_allclose(n(ta.grad), e_ga, rtol=0, atol=0)


def test_march_rays_train_backward(po, hiplib, rng):
    import radnerf_hip as hip
    N = 2000
    o, d = make_rays(rng, N)
    nears, fars = po.near_far_from_aabb(o, d, AABB, 0.05)
    bits, _ = ellipsoid_bits()
    M = N * 16
    xyzs, dirs, deltas, rays, cnt = po.march_rays_train(o, d, bits, 1.0, 1 / 256, 16, 1, 128, M, nears, fars, np.zeros(N, np.float32))
    gx = rng.standard_normal((M, 3)).astype(np.float32); gd = rng.standard_normal((M, 3)).astype(np.float32)
    e_go, e_gd = po.march_rays_train_backward(gx, gd, rays, deltas)
    go = torch.zeros(N, 3, device=DEV); gdd = torch.zeros(N, 3, device=DEV)
    hip.call("rn_march_rays_train_backward", dp(gx), dp(gd), dp(rays), dp(deltas), N, M,
             hip.ptr(go), hip.ptr(gdd), hip.stream())
    np.testing.assert_array_equal(n(go), e_go)
    np.testing.assert_array_equal(n(gdd), e_gd)


# ------------------------------------------------------------------------------------------------ grid encoder


def _grid_case(rng, D, C, L, log2T, gridtype, B, desired=2048, base=16):
    from gridencoder.encoder import level_offsets
    pls = np.exp2(np.log2(desired / base) / (L - 1))
    offsets = level_offsets(D, L, pls, base, log2T, False)
    emb = rng.uniform(-0.5, 0.5, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = 0.0
    if B > 4:
        x[1] = 1.0
        x[2, 0] = -0.01   # out of range -> zeros
        x[3, D - 1] = 1.01
    return offsets, emb, x, float(np.log2(pls))


GRID_CASES = [
    (3, 2, 16, 16, 1, 5000),   # the shipped xyz grid (tiled, T=2^16)
    (2, 2, 16, 16, 1, 5000),   # ambient / torso grid
    (3, 2, 16, 19, 0, 5000),   # BASELINE config 1: hash, T=2^19
    (3, 1, 8, 14, 0, 777),
    (3, 4, 8, 14, 0, 777),
    (3, 8, 4, 12, 1, 300),
    (4, 2, 6, 14, 0, 500),
    (5, 2, 4, 12, 0, 300),
    (2, 1, 16, 16, 1, 1),
]


@pytest.mark.parametrize("D,C,L,log2T,gridtype,B", GRID_CASES)
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("interp", [0, 1])
def test_grid_forward_fp32_bit_exact(po, hiplib, rng, D, C, L, log2T, gridtype, B, layout, interp):
    import radnerf_hip as hip
    offsets, emb, x, S = _grid_case(rng, D, C, L, log2T, gridtype, B)
    e_out, e_dy = po.grid_encode_forward(x, emb, offsets, B, D, C, L, S, 16, True, gridtype, False, interp)
    out = torch.empty((L, B, C) if layout == 0 else (B, L * C), device=DEV)
    dy = torch.empty(B, L * D * C, device=DEV)
    hip.call("rn_grid_encode_forward", dp(x), dp(emb), dp(offsets), hip.ptr(out), B, D, C, L, S, 16,
             hip.ptr(dy), gridtype, 0, interp, hip.RN_F32, layout, hip.stream())
    got = n(out) if layout == 0 else n(out).reshape(B, L, C).transpose(1, 0, 2)
    assert np.array_equal(got, e_out)
    assert np.array_equal(n(dy), e_dy)
    # without dy_dx the outputs must not change
    out2 = torch.empty_like(out)
    hip.call("rn_grid_encode_forward", dp(x), dp(emb), dp(offsets), hip.ptr(out2), B, D, C, L, S, 16,
             None, gridtype, 0, interp, hip.RN_F32, layout, hip.stream())
    assert torch.equal(out, out2)


PLANNED_CASES = [  # D, C, L, log2T, gridtype, B
    (3, 2, 16, 19, 0, 70001),   # BASELINE config 1 (hash, T=2^19): levels 0-1 in LDS, 2-4 coarse from L2, 5-15 level-major
    (3, 2, 16, 16, 1, 70001),   # the shipped xyz grid
    (2, 2, 16, 16, 1, 33333),   # ambient / torso grid: all of levels 0..5 are dense and small
    (3, 4, 8, 14, 0, 9000),
    (3, 2, 16, 19, 0, 3000),    # below the planning threshold: per-level kernels, same entry point
]


@pytest.mark.parametrize("D,C,L,log2T,gridtype,B", PLANNED_CASES)
@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("layout,ws_samples", [(0, 0), (1, None), (1, 4096 + 256)])
def test_grid_forward_planned_bit_exact(po, hiplib, rng, D, C, L, log2T, gridtype, B, dtype, layout, ws_samples):
    """rn_grid_encode_forward_ws: LDS-staged coarse pass + level-major pass (+ chunked transposition for [B, L*C]) gives the
    bits of the oracle -- with the recommended workspace and with one so small that the samples go through in many chunks."""
    import ctypes as C_
    import radnerf_hip as hip
    offsets, emb, x, S = _grid_case(rng, D, C, L, log2T, gridtype, B)
    half = dtype == "f16"
    tab = emb.astype(np.float16) if half else emb
    e_out, _ = po.grid_encode_forward(x, tab, offsets, B, D, C, L, S, 16, False, gridtype, False, 0, half=half)
    tdt = torch.half if half else torch.float32
    out = torch.full((L, B, C) if layout == 0 else (B, L * C), float("nan"), device=DEV, dtype=tdt)
    oh = (C_.c_int32 * len(offsets))(*[int(v) for v in offsets])
    did = hip.RN_F16 if half else hip.RN_F32
    esz = 2 if half else 4
    if layout == 0:
        ws, ws_bytes = None, 0
    else:
        ws_bytes = hip.workspace_bytes_grid(B, L, C, did) if ws_samples is None else ws_samples * L * C * esz
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    xd, td, od = dp(x), dp(tab), dp(offsets)
    hip.call("rn_grid_encode_forward_ws", xd, td, od, oh, hip.ptr(out), B, D, C, L, S, 16, None, gridtype, 0, 0, did, layout,
             hip.ptr(ws), ws_bytes, hip.stream())
    got = n(out) if layout == 0 else n(out).reshape(B, L, C).transpose(1, 0, 2)
    view = np.uint16 if half else np.uint32
    assert np.array_equal(np.ascontiguousarray(got).view(view), np.ascontiguousarray(e_out).view(view))


def test_grid_encoder_module_uses_planned_path_and_matches_oracle(po, hiplib, rng):
    """gridencoder.GridEncoder.forward (the operator surface) at a size where the planned path is taken."""
    from gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048,
                      gridtype="hash").cuda()
    enc.embeddings.data.uniform_(-0.5, 0.5)
    B = 50001
    x = torch.rand(B, 3, device=DEV) * 2.2 - 1.1                 # some points outside [-1, 1]
    with torch.no_grad():
        y = enc(x, bound=1)
    x01 = ((x + 1) / 2).cpu().numpy()
    e_out, _ = po.grid_encode_forward(x01, enc.embeddings.detach().cpu().numpy(), enc.offsets.cpu().numpy(), B, 3, 2, 16,
                                      float(np.log2(enc.per_level_scale)), 16, False, 0, False, 0)
    assert np.array_equal(y.cpu().numpy(), e_out.transpose(1, 0, 2).reshape(B, -1))


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_grid_rows_equal_the_transposition_pass_at_full_size(hiplib, dtype):
    """B = 2^22 + 777 samples (the size the lookup is benchmarked at, one full chunk and an odd tail chunk), hash T = 2^19: the
    [B, L*C] rows written by the last level launch are, bit for bit, the rows the transposition pass builds (that path is
    pinned by the oracle at the sizes the oracle finishes in seconds), and rows of the level-major output transposed on the host."""
    import ctypes as C_
    import os
    import radnerf_hip as hip
    from gridencoder.encoder import level_offsets
    D, C, L, log2T = 3, 2, 16, 19
    B = (1 << 22) + 777
    scale = float(np.exp2(np.log2(2048 / 16) / (L - 1)))
    offsets = level_offsets(D, L, scale, 16, log2T, False)
    g = torch.Generator(device="cuda").manual_seed(5)
    table = (torch.rand(int(offsets[-1]), C, device="cuda", generator=g) - 0.5)
    if dtype == "f16":
        table = table.half()
    x = torch.rand(B, D, device="cuda", generator=g)
    x[:5] = 1.5                                                    # outside [0, 1]: zero rows
    S = float(np.log2(scale))
    oh = (C_.c_int32 * len(offsets))(*[int(v) for v in offsets])
    od = torch.from_numpy(offsets).cuda()
    did = hip.RN_F16 if dtype == "f16" else hip.RN_F32
    ws = torch.empty(hip.workspace_bytes_grid(B, L, C, did), dtype=torch.uint8, device="cuda")

    def run(layout, rows):
        out = torch.full((L, B, C) if layout == 0 else (B, L * C), float("nan"), device="cuda", dtype=table.dtype)
        if rows is not None:
            os.environ["RN_GRID_ROWS"] = rows
        try:
            hip.call("rn_grid_encode_forward_ws", hip.ptr(x), hip.ptr(table), hip.ptr(od, torch.int32), oh, hip.ptr(out), B, D, C, L, S, 16, None,
                     0, 0, 0, did, layout, hip.ptr(ws), ws.numel(), hip.stream())
            torch.cuda.synchronize()
        finally:
            os.environ.pop("RN_GRID_ROWS", None)
        return out
    rows = run(1, None)
    transposed = run(1, "0")
    assert torch.equal(rows, transposed)
    del transposed
    lbc = run(0, None)
    assert torch.equal(rows.view(B, L, C), lbc.permute(1, 0, 2))
    assert not torch.isnan(rows).any() and float(rows[:5].abs().max()) == 0.0


@pytest.mark.parametrize("gridtype,log2T", [("hash", 19), ("tiled", 16)])
def test_grid_encoder_module_folds_the_bound_and_writes_rows_without_a_transposition(hiplib, gridtype, log2T):
    """GridEncoder.forward with no gradient wanted = ONE library call (rn_grid_encode_forward_bound: grid.py:149's normalisation
    inside the coordinate load, [B, L*C] rows written by the last level launch).  Bit-identical to (a) the autograd path, which
    normalises with torch ops as the reference does, for bounds whose doubles are and are not powers of two, and (b) the same
    call with the transposition pass (RN_GRID_ROWS=0), fp32 and fp16 tables."""
    import os
    from gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=log2T, desired_resolution=2048,
                      gridtype=gridtype).cuda()
    enc.embeddings.data.uniform_(-0.5, 0.5)
    B = 70001
    for bound in (1, 2, 1.5, 0.7):
        x = (torch.rand(B, 3, device=DEV) * 2.2 - 1.1) * bound     # some points outside [-bound, bound]
        with torch.no_grad():
            fast = enc(x, bound=bound)
        slow = enc(x, bound=bound)                                # embeddings.requires_grad: torch normalisation + autograd Function
        assert slow.requires_grad and not fast.requires_grad
        assert torch.equal(fast, slow.detach()), bound
        os.environ["RN_GRID_ROWS"] = "0"
        try:
            with torch.no_grad():
                transposed = enc(x, bound=bound)
                with torch.autocast("cuda", dtype=torch.half):
                    transposed_h = enc(x, bound=bound)
        finally:
            del os.environ["RN_GRID_ROWS"]
        assert torch.equal(fast, transposed), bound
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.half):
            fast_h = enc(x, bound=bound)
        assert fast_h.dtype == torch.half and torch.equal(fast_h, transposed_h), bound


@pytest.mark.parametrize("D,C,L,log2T,gridtype,B", [c for c in GRID_CASES if c[1] % 2 == 0][:4])
@pytest.mark.parametrize("layout", [0, 1])
def test_grid_forward_fp16_bit_exact(po, hiplib, rng, D, C, L, log2T, gridtype, B, layout):
    import radnerf_hip as hip
    offsets, emb, x, S = _grid_case(rng, D, C, L, log2T, gridtype, B)
    emb16 = emb.astype(np.float16)
    e_out, e_dy = po.grid_encode_forward(x, emb16, offsets, B, D, C, L, S, 16, True, gridtype, False, 0, half=True)
    out = torch.empty((L, B, C) if layout == 0 else (B, L * C), device=DEV, dtype=torch.half)
    dy = torch.empty(B, L * D * C, device=DEV, dtype=torch.half)
    hip.call("rn_grid_encode_forward", dp(x), dp(emb16), dp(offsets), hip.ptr(out), B, D, C, L, S, 16,
             hip.ptr(dy), gridtype, 0, 0, hip.RN_F16, layout, hip.stream())
    got = n(out) if layout == 0 else n(out).reshape(B, L, C).transpose(1, 0, 2)
    assert np.array_equal(got.view(np.uint16), e_out.view(np.uint16))
    assert np.array_equal(n(dy).view(np.uint16), e_dy.view(np.uint16))


def test_grid_forward_align_corners(po, hiplib, rng):
    import radnerf_hip as hip
    from gridencoder.encoder import level_offsets
    D, C, L, B = 3, 2, 8, 2000
    pls = 1.5
    offsets = level_offsets(D, L, pls, 16, 15, True)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    S = float(np.log2(pls))
    e_out, _ = po.grid_encode_forward(x, emb, offsets, B, D, C, L, S, 16, False, 0, True, 0)
    out = torch.empty(L, B, C, device=DEV)
    hip.call("rn_grid_encode_forward", dp(x), dp(emb), dp(offsets), hip.ptr(out), B, D, C, L, S, 16,
             None, 0, 1, 0, hip.RN_F32, 0, hip.stream())
    assert np.array_equal(n(out), e_out)


@pytest.mark.parametrize("D,C,L,log2T,gridtype,B", GRID_CASES[:6])
@pytest.mark.parametrize("layout", [0, 1])
def test_grid_backward_fp32(po, hiplib, rng, D, C, L, log2T, gridtype, B, layout):
    import radnerf_hip as hip
    offsets, emb, x, S = _grid_case(rng, D, C, L, log2T, gridtype, B)
    _, dy = po.grid_encode_forward(x, emb, offsets, B, D, C, L, S, 16, True, gridtype, False, 0)
    grad = rng.standard_normal((L, B, C)).astype(np.float32)
    e_ge, e_gi = po.grid_encode_backward(grad, x, emb, offsets, B, D, C, L, S, 16, dy, gridtype, False, 0)
    g_in = grad if layout == 0 else np.ascontiguousarray(grad.transpose(1, 0, 2)).reshape(B, L * C)
    ge = torch.zeros(emb.shape, device=DEV); gi = torch.zeros(B, D, device=DEV)
    hip.call("rn_grid_encode_backward", dp(g_in), dp(x), dp(emb), dp(offsets), hip.ptr(ge), B, D,
             C, L, S, 16, dp(dy), hip.ptr(gi), gridtype, 0, 0, hip.RN_F32, layout, hip.stream())
    # scatter-add order differs (atomics): tolerance scaled by the number of colliding adds
    np.testing.assert_allclose(n(ge), e_ge, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(e_ge).max()))
    np.testing.assert_array_equal(n(gi), e_gi)  # sequential per (b, d): same order -> exact


@pytest.mark.parametrize("pattern", ["clustered", "ray_runs", "ragged_tail", "with_oob"])
@pytest.mark.parametrize("D,C", [(2, 2), (3, 2), (3, 1), (2, 4)])
def test_grid_backward_colliding_samples(po, hiplib, rng, D, C, pattern):
    """The scatter-add pre-reduces runs of equal rows inside a wave (ambient coordinates cluster around one cell,
    ray-ordered samples share coarse cells): the sums must still be the reference's (gridencoder.cu:247-339)."""
    import radnerf_hip as hip
    from gridencoder.encoder import level_offsets
    L, log2T, gridtype = 16, 16, 1
    B = {"clustered": 5000, "ray_runs": 4096, "ragged_tail": 1000 + 37, "with_oob": 3000}[pattern]
    pls = np.exp2(np.log2(2048 / 16) / (L - 1))
    offsets = level_offsets(D, L, pls, 16, log2T, False)
    S = float(np.log2(pls))
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    if pattern == "clustered":
        x = (0.5 + 1e-4 * rng.standard_normal((B, D))).astype(np.float32)
    elif pattern == "ray_runs":
        o = rng.uniform(0.2, 0.8, (B // 16, 1, D)); d = rng.standard_normal((B // 16, 1, D)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
        x = np.clip(o + d * (0.0135 * np.arange(16))[None, :, None], 0, 1).reshape(B, D).astype(np.float32)
    else:
        x = np.repeat(rng.uniform(0, 1, (B // 8 + 1, D)), 8, axis=0)[:B].astype(np.float32)
        if pattern == "with_oob":
            x[::7] = 1.5          # skipped samples inside the runs
    grad = rng.standard_normal((L, B, C)).astype(np.float32)
    e_ge, _ = po.grid_encode_backward(grad, x, emb, offsets, B, D, C, L, S, 16, None, gridtype, False, 0)
    ge = torch.zeros(emb.shape, device=DEV)
    hip.call("rn_grid_encode_backward", dp(grad), dp(x), dp(emb), dp(offsets), hip.ptr(ge), B, D, C, L, S, 16, None, None,
             gridtype, 0, 0, hip.RN_F32, 0, hip.stream())
    np.testing.assert_allclose(n(ge), e_ge, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(e_ge).max()))


def test_grid_backward_fp16(po, hiplib, rng):
    import radnerf_hip as hip
    D, C, L, log2T, gridtype, B = 3, 2, 16, 16, 1, 3000
    offsets, emb, x, S = _grid_case(rng, D, C, L, log2T, gridtype, B)
    emb16 = emb.astype(np.float16)
    grad = (rng.standard_normal((L, B, C)) * 0.1).astype(np.float16)
    e_ge, _ = po.grid_encode_backward(grad, x, emb16, offsets, B, D, C, L, S, 16, None, gridtype, False, 0, half=True)
    ge = torch.zeros(emb.shape, device=DEV, dtype=torch.half)
    hip.call("rn_grid_encode_backward", dp(grad), dp(x), dp(emb16), dp(offsets), hip.ptr(ge), B,
             D, C, L, S, 16, None, None, gridtype, 0, 0, hip.RN_F16, 0, hip.stream())
    # half accumulation in arrival order: compare against the fp32 truth with a half-precision tolerance
    t_ge, _ = po.grid_encode_backward(grad.astype(np.float32), x, emb, offsets, B, D, C, L, S, 16, None, gridtype, False, 0)
    scale = np.abs(t_ge).max()
    assert np.abs(n(ge).astype(np.float32) - t_ge).max() < 2e-2 * scale
    assert np.abs(e_ge.astype(np.float32) - t_ge).max() < 2e-2 * scale


def test_grid_module_autograd_and_autocast(po, hiplib, rng):
    """GridEncoder module: forward/backward through autograd, fp32 and autocast(fp16), vs the oracle."""
    from gridencoder import GridEncoder
    enc = GridEncoder(input_dim=2, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16,
                      desired_resolution=2048, gridtype="tiled").to(DEV)
    enc.embeddings.data.uniform_(-0.5, 0.5)
    B = 4000
    x = t(rng.uniform(-1, 1, (B, 2)).astype(np.float32)).requires_grad_()
    out = enc(x, bound=1)
    assert out.shape == (B, 32) and out.dtype == torch.float32
    w = t(rng.standard_normal((B, 32)).astype(np.float32))
    (out * w).sum().backward()
    S = float(np.log2(enc.per_level_scale))
    xin = (n(x) + 1) / 2
    e_out, e_dy = po.grid_encode_forward(xin, n(enc.embeddings), n(enc.offsets), B, 2, 2, 16, S, 16, True, 1, False, 0)
    assert np.array_equal(n(out), e_out.transpose(1, 0, 2).reshape(B, 32))
    grad_lbc = np.ascontiguousarray(n(w).reshape(B, 16, 2).transpose(1, 0, 2))
    e_ge, e_gi = po.grid_encode_backward(grad_lbc, xin, n(enc.embeddings), n(enc.offsets), B, 2, 2, 16, S, 16, e_dy, 1, False, 0)
    np.testing.assert_allclose(n(enc.embeddings.grad), e_ge, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(n(x.grad), e_gi / 2, rtol=1e-6, atol=1e-7)  # d((x+1)/2)/dx = 1/2
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        o16 = enc(x.detach(), bound=1)
        o16b = enc(x.detach(), bound=1)  # cached half table
    assert o16.dtype == torch.float16 and torch.equal(o16, o16b)
    e16, _ = po.grid_encode_forward(xin, n(enc.embeddings).astype(np.float16), n(enc.offsets), B, 2, 2, 16, S, 16, False, 1,
                                    False, 0, half=True)
    assert np.array_equal(n(o16).view(np.uint16), e16.transpose(1, 0, 2).reshape(B, 32).view(np.uint16))


def test_grad_total_variation(po, hiplib, rng):
    from gridencoder import GridEncoder
    enc = GridEncoder(input_dim=3, num_levels=8, level_dim=2, base_resolution=16, log2_hashmap_size=14,
                      desired_resolution=512, gridtype="hash").to(DEV)
    enc.embeddings.data.uniform_(-0.5, 0.5)
    B = 3000
    x = rng.uniform(-1, 1, (B, 3)).astype(np.float32)
    enc.embeddings.grad = torch.zeros_like(enc.embeddings)
    enc.grad_total_variation(weight=1e-2, inputs=t(x), bound=1)
    e = np.zeros(tuple(enc.embeddings.shape), np.float32)
    po.grad_total_variation((x + 1) / 2, n(enc.embeddings), e, n(enc.offsets), 1e-2, B, 3, 2, 8, float(np.log2(enc.per_level_scale)), 16, 0, False)
    np.testing.assert_allclose(n(enc.embeddings.grad), e, rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------ SH / freq


@pytest.mark.parametrize("degree", [1, 2, 3, 4, 5, 6, 7, 8])
def test_sh_encoder(po, hiplib, rng, degree):
    from shencoder import SHEncoder
    B = 3001
    v = rng.standard_normal((B, 3)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    enc = SHEncoder(degree=degree)
    x = t(v).requires_grad_()
    out = enc(x)
    e_out, e_dy = po.sh_encode_forward(v, degree, True)
    np.testing.assert_allclose(n(out), e_out, rtol=2e-6, atol=2e-6)
    g = rng.standard_normal(e_out.shape).astype(np.float32)
    (out * t(g)).sum().backward()
    e_gi = po.sh_encode_backward(g, v, degree, e_dy)
    np.testing.assert_allclose(n(x.grad), e_gi, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("D,deg", [(2, 10), (6, 4), (3, 6), (1, 1)])
def test_freq_encoder(po, hiplib, rng, D, deg):
    from freqencoder import FreqEncoder
    B = 2500
    v = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    enc = FreqEncoder(input_dim=D, degree=deg)
    assert enc.output_dim == D + 2 * D * deg
    x = t(v).requires_grad_()
    out = enc(x)
    e_out = po.freq_encode_forward(v, deg)
    np.testing.assert_allclose(n(out), e_out, rtol=0, atol=2e-6 * 2 ** deg)  # sinf of 2^f x
    g = rng.standard_normal(e_out.shape).astype(np.float32)
    (out * t(g)).sum().backward()
    e_gi = po.freq_encode_backward(g, e_out, D, deg)
    np.testing.assert_allclose(n(x.grad), e_gi, rtol=1e-4, atol=1e-3 * 2 ** deg * 1e-2)


def test_get_encoder_and_trunc_exp(hiplib):
    from activation import trunc_exp
    from encoding import get_encoder
    enc, dim = get_encoder("spherical_harmonics")
    assert dim == 16
    enc, dim = get_encoder("frequency", input_dim=6, multires=4)
    assert dim == 54
    enc, dim = get_encoder("hashgrid", input_dim=3)
    assert dim == 32 and enc.gridtype == "hash" and enc.log2_hashmap_size == 19
    with pytest.raises(NotImplementedError):
        get_encoder("nope")
    x = torch.tensor([-20.0, 0.0, 3.0, 20.0], device=DEV, requires_grad=True)
    y = trunc_exp(x)
    y.sum().backward()
    assert torch.allclose(y, torch.exp(x.detach()))
    assert torch.allclose(x.grad, torch.exp(x.detach().clamp(-15, 15)))

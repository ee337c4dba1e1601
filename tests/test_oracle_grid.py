"""Pins of the CPU oracle's grid encoder against independent derivations: a Python big-int restatement of
get_grid_index (incl. the z-dropping quirk of tiled levels), a numpy multilinear interpolation on dense
levels, torch autograd of a pure-torch formulation for the backward and dy_dx."""
import numpy as np
import pytest
import torch

PRIMES = [1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737]


def py_grid_index(D, C, gridtype, align_corners, ch, hashmap_size, resolution, pos_grid):
    """gridencoder.cu:66-84 with Python ints, wrapping to uint32 explicitly."""
    M = 1 << 32
    stride, index, d = 1, 0, 0
    while d < D and stride <= hashmap_size:
        index = (index + pos_grid[d] * stride) % M
        stride = (stride * (resolution if align_corners else resolution + 1)) % M
        d += 1
    if gridtype == 0 and stride > hashmap_size:
        index = 0
        for i in range(D):
            index ^= (pos_grid[i] * PRIMES[i]) % M
    return (index % hashmap_size) * C + ch


def offsets_for(D, L, log2T, desired=2048, base=16, align=False):
    from gridencoder.encoder import level_offsets
    pls = np.exp2(np.log2(desired / base) / (L - 1))
    return level_offsets(D, L, pls, base, log2T, align), pls


def test_offsets_match_the_reference_model_tables(hiplib):
    # published in gridencoder/grid.py:129 (xyz grid) and encoding.py:40 (torso grid: 555520 rows)
    off3, pls = offsets_for(3, 16, 16)
    assert off3.tolist() == [0, 4920, 18744, 51512, 117048, 182584, 248120, 313656, 379192, 444728, 510264, 575800,
                             641336, 706872, 772408, 837944, 903480]
    assert abs(pls - 1.381912879967776) < 1e-12
    off2, _ = offsets_for(2, 16, 16)
    assert off2[-1] == 555520
    off19, _ = offsets_for(3, 16, 19)
    assert off19[-1] == 6119864


@pytest.mark.parametrize("D", [2, 3, 4, 5])
@pytest.mark.parametrize("gridtype", [0, 1])
def test_grid_index_against_bigint_restatement(po, rng, D, gridtype):
    for _ in range(300):
        res = int(rng.integers(2, 3000))
        hs = int(rng.integers(1, 1 << 19)) * 8
        pg = [int(v) for v in rng.integers(0, res + 2, D)]
        for ac in (False, True):
            for ch in (0, 1):
                assert po.grid_index(D, 2, gridtype, ac, ch, hs, res, pg) == py_grid_index(D, 2, gridtype, ac, ch, hs, res, pg)


def test_tiled_levels_drop_z_once_the_stride_passes_the_table_size(po):
    """with T=2^16 the 3-D tiled grid ignores z on levels with (res+1)^2 > T."""
    hs, res = 65536, 295  # (296)^2 = 87616 > 65536
    a = po.grid_index(3, 2, 1, False, 0, hs, res, [10, 20, 5])
    b = po.grid_index(3, 2, 1, False, 0, hs, res, [10, 20, 250])
    assert a == b == ((10 + 20 * 296) % hs) * 2
    hs, res = 65536, 100  # 101^2 = 10201 <= 65536: z participates, then wraps modulo the table
    a = po.grid_index(3, 2, 1, False, 0, hs, res, [10, 20, 5])
    assert a == ((10 + 20 * 101 + 5 * 101 * 101) % hs) * 2


def numpy_multilinear(x, table, res, D):
    """Independent dense-lattice interpolation: pos = x*scale + 0.5 on a (res+1)^D lattice, x fastest."""
    out = np.zeros((x.shape[0], table.shape[1]), np.float64)
    return out


def test_dense_levels_against_independent_numpy_interpolation(po, rng):
    D, C, L = 3, 2, 3
    off, pls = offsets_for(D, L, 19, desired=64, base=16)  # all levels dense
    S = float(np.log2(pls))
    emb = rng.uniform(-1, 1, (int(off[-1]), C)).astype(np.float32)
    B = 4000
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    out, _ = po.grid_encode_forward(x, emb, off, B, D, C, L, S, 16, False, 1, False, 0)
    for l in range(L):
        scale = np.float32(np.exp2(np.float32(l * np.float32(S)))) * np.float32(16) - np.float32(1)
        res = int(np.ceil(scale)) + 1
        assert (res + 1) ** 3 <= off[l + 1] - off[l]
        pos = x.astype(np.float64) * float(scale) + 0.5
        p0 = np.floor(pos).astype(np.int64)
        fr = pos - p0
        acc = np.zeros((B, C))
        for corner in range(8):
            w = np.ones(B)
            idx = np.zeros(B, np.int64)
            stride = 1
            for d in range(3):
                bit = (corner >> d) & 1
                w *= fr[:, d] if bit else 1 - fr[:, d]
                idx += (p0[:, d] + bit) * stride
                stride *= res + 1
            acc += w[:, None] * emb[off[l] + idx]
        np.testing.assert_allclose(out[l], acc, rtol=0, atol=3e-5)


def torch_grid(x, emb, off, S, H, D, C, L, gridtype, interp):
    """Pure-torch float64 formulation (differentiable in emb and x) using the oracle's integer indices."""
    outs = []
    for l in range(L):
        scale = float(np.float32(np.exp2(np.float32(np.float32(l) * np.float32(S)))) * np.float32(H) - np.float32(1))
        res = int(np.ceil(scale)) + 1
        hs = int(off[l + 1] - off[l])
        pos = x * scale + 0.5
        p0 = torch.floor(pos).detach()
        fr = pos - p0
        if interp == 1:
            fr = fr * fr * (3 - 2 * fr)
        acc = 0
        p0n = p0.long().numpy()
        for corner in range(1 << D):
            w = 1
            pg = p0n.copy()
            for d in range(D):
                bit = (corner >> d) & 1
                w = w * (fr[:, d] if bit else 1 - fr[:, d])
                pg[:, d] += bit
            rows = np.array([py_grid_index(D, 1, gridtype, False, 0, hs, res, [int(v) for v in r]) for r in pg])
            acc = acc + w[:, None] * emb[off[l] + torch.from_numpy(rows)]
        outs.append(acc)
    return torch.stack(outs, 0)  # [L,B,C]


@pytest.mark.parametrize("D,gridtype,interp", [(3, 1, 0), (3, 0, 0), (2, 1, 0), (3, 0, 1)])
def test_forward_backward_and_dydx_against_torch_autograd(po, rng, D, gridtype, interp):
    C, L, B = 2, 6, 120
    off, pls = offsets_for(D, L, 12, desired=512, base=16)
    S = float(np.log2(pls))
    emb = rng.uniform(-1, 1, (int(off[-1]), C)).astype(np.float32)
    x = rng.uniform(0.02, 0.98, (B, D)).astype(np.float32)
    out, dy = po.grid_encode_forward(x, emb, off, B, D, C, L, S, 16, True, gridtype, False, interp)
    te = torch.tensor(emb, dtype=torch.float64, requires_grad=True)
    tx = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    tout = torch_grid(tx, te, off, S, 16, D, C, L, gridtype, interp)
    np.testing.assert_allclose(out, tout.detach().numpy(), rtol=0, atol=1e-4)  # fp32 vs fp64 lattice position
    g = rng.standard_normal(out.shape).astype(np.float32)
    (tout * torch.tensor(g, dtype=torch.float64)).sum().backward()
    ge, gi = po.grid_encode_backward(g, x, emb, off, B, D, C, L, S, 16, dy, gridtype, False, interp)
    np.testing.assert_allclose(ge, te.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gi, tx.grad.numpy(), rtol=2e-3, atol=2e-2)  # finite lattice: fp32 products of scale ~ 500


def test_out_of_range_inputs_give_zero_rows(po, rng):
    D, C, L = 3, 2, 4
    off, pls = offsets_for(D, L, 12, desired=128)
    emb = rng.uniform(-1, 1, (int(off[-1]), C)).astype(np.float32)
    x = np.array([[0.5, 0.5, -1e-6], [1.0000001, 0.2, 0.2], [0.0, 1.0, 0.5], [0.3, 0.3, 0.3]], np.float32)
    out, dy = po.grid_encode_forward(x, emb, off, 4, D, C, L, float(np.log2(pls)), 16, True, 0, False, 0)
    assert not out[:, 0].any() and not out[:, 1].any() and not dy[:2].any()
    assert out[:, 2].any() and out[:, 3].any()
    g = np.ones_like(out)
    ge, gi = po.grid_encode_backward(g, x, emb, off, 4, D, C, L, float(np.log2(pls)), 16, dy, 0, False, 0)
    assert not gi[:2].any()


def test_half_mode_rounds_like_c10_half(po, rng):
    """fp16 tables: outputs equal a step-by-step numpy float16 emulation of `scalar_t += float * scalar_t`."""
    D, C, L, B = 2, 2, 5, 300
    off, pls = offsets_for(D, L, 12, desired=256)
    S = float(np.log2(pls))
    emb = rng.uniform(-1, 1, (int(off[-1]), C)).astype(np.float16)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    out, _ = po.grid_encode_forward(x, emb, off, B, D, C, L, S, 16, False, 1, False, 0, half=True)
    assert out.dtype == np.float16
    for l in range(L):
        scale = np.float32(np.exp2(np.float32(np.float32(l) * np.float32(S)))) * np.float32(16) - np.float32(1)
        res = int(np.ceil(scale)) + 1
        hs = int(off[l + 1] - off[l])
        pos = x * scale + np.float32(0.5)
        p0 = np.floor(pos)
        fr = (pos - p0).astype(np.float32)
        acc = np.zeros((B, C), np.float16)
        for corner in range(4):
            w = np.ones(B, np.float32)
            pg = p0.astype(np.int64).copy()
            for d in range(2):
                bit = (corner >> d) & 1
                w = (w * (fr[:, d] if bit else np.float32(1) - fr[:, d])).astype(np.float32)
                pg[:, d] += bit
            rows = np.array([py_grid_index(2, 1, 1, False, 0, hs, res, [int(v) for v in r]) for r in pg])
            acc = (acc.astype(np.float32) + w[:, None] * emb[off[l] + rows].astype(np.float32)).astype(np.float16)
        assert np.array_equal(out[l].view(np.uint16), acc.view(np.uint16))


def test_total_variation_gradient_is_the_gradient_of_the_tv_energy(po, rng):
    """kernel_grad_tv accumulates w * sum_nb (c - nb) / sqrt(sum (c - nb)^2) per visited cell."""
    D, C, L = 2, 1, 2
    off, pls = offsets_for(D, L, 12, desired=32)
    S = float(np.log2(pls))
    emb = rng.uniform(-1, 1, (int(off[-1]), C)).astype(np.float32)
    x = np.array([[0.5, 0.5]], np.float32)
    grad = np.zeros_like(emb)
    po.grad_total_variation(x, emb, grad, off, 1.0, 1, D, C, L, S, 16, 1, False)
    for l in range(L):
        scale = np.float32(np.exp2(np.float32(np.float32(l) * np.float32(S)))) * np.float32(16) - np.float32(1)
        res = int(np.ceil(scale)) + 1
        p = np.floor(x[0] * scale + 0.5).astype(int)
        idx = lambda a, b: off[l] + (a + b * (res + 1))
        c = emb[idx(p[0], p[1]), 0]
        nb = [emb[idx(p[0] + 1, p[1]), 0], emb[idx(p[0] - 1, p[1]), 0], emb[idx(p[0], p[1] + 1), 0], emb[idx(p[0], p[1] - 1), 0]]
        diffs = np.array([c - v for v in nb], np.float64)
        expect = (1.0 / 4) * diffs.sum() / np.sqrt((diffs ** 2).sum() + 1e-9)
        assert abs(grad[idx(p[0], p[1]), 0] - expect) < 1e-5

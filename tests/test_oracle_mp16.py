"""The oracle's emulation of the opt-in 16-bit matrix-core arithmetic (orc_nerf_forward_mp16, the checker of
mlp_dtype = RN_F16) against a plain numpy restatement with explicit np.float16 roundings and float64 sums: fp16
operands where they enter a contraction, unrounded fp32 for the folded broadcast inputs and the narrow output rows."""
import numpy as np

from test_oracle_grid import offsets_for


def _h(v):
    return np.asarray(v, np.float32).astype(np.float16).astype(np.float32)


def _mlp_mp(ws, x, n_var, n_fp32_rows):
    a = x.astype(np.float32)
    for l, w in enumerate(ws):
        last = l == len(ws) - 1
        if l == 0:
            c = _h(a[:, :n_var]).astype(np.float64) @ _h(w[:, :n_var]).astype(np.float64).T
            c = c + a[:, n_var:].astype(np.float64) @ w[:, n_var:].astype(np.float64).T
        else:
            c = _h(a).astype(np.float64) @ _h(w).astype(np.float64).T
            if last:
                c[:, :n_fp32_rows] = a.astype(np.float64) @ w[:n_fp32_rows].astype(np.float64).T
        a = c.astype(np.float32)
        if not last:
            a = np.maximum(a, 0)
    return a


def _encode(po, x01, emb, off, D, S):
    B = x01.shape[0]
    out, _ = po.grid_encode_forward(x01, emb, off, B, D, 2, 16, S, 16, False, 1, False, 0)
    return np.ascontiguousarray(out.transpose(1, 0, 2)).reshape(B, 32)


def test_mp16_forward_against_numpy(po, rng):
    off3, pls3 = offsets_for(3, 16, 16)
    off2, pls2 = offsets_for(2, 16, 16)
    P = {"encoder.embeddings": rng.uniform(-0.5, 0.5, (int(off3[-1]), 2)).astype(np.float32), "encoder.offsets": off3,
         "encoder_ambient.embeddings": rng.uniform(-0.5, 0.5, (int(off2[-1]), 2)).astype(np.float32),
         "encoder_ambient.offsets": off2}

    def lin(o, i):
        return rng.uniform(-1, 1, (o, i)).astype(np.float32) / np.float32(np.sqrt(i))
    shapes = {"ambient_net": [(64, 96), (64, 64), (2, 64)], "sigma_net": [(64, 65), (64, 64), (65, 64)],
              "color_net": [(64, 84), (3, 64)]}
    for name, ls in shapes.items():
        for l, (o, i) in enumerate(ls):
            P[f"{name}.net.{l}.weight"] = lin(o, i)
    cfg = dict(per_level_scale_xyz=float(pls3), per_level_scale_ambient=float(pls2), base_resolution=16, gridtype=1, bound=1.0,
               has_eye=True, ind_dim=4, audio_dim=64, sh_degree=4)
    om = po.Model(P, cfg)
    M = 300
    x = rng.uniform(-0.9, 0.9, (M, 3)).astype(np.float32)
    d = rng.standard_normal((M, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    enc_a = rng.standard_normal(64).astype(np.float32)
    ind = rng.standard_normal(4).astype(np.float32)
    eye = np.array([0.3], np.float32)
    sig, col, amb = po.nerf_forward(om, x, d, enc_a, ind, eye, mlp_dtype="f16")

    S3, S2 = float(np.log2(pls3)), float(np.log2(pls2))
    enc_x = _encode(po, ((x + 1) / 2).astype(np.float32), P["encoder.embeddings"], off3, 3, S3)
    wa = [P[f"ambient_net.net.{l}.weight"] for l in range(3)]
    e_amb = np.tanh(_mlp_mp(wa, np.concatenate([enc_x, np.tile(enc_a, (M, 1))], 1), 32, 2))
    np.testing.assert_allclose(amb, e_amb, rtol=0, atol=2e-4)
    # continue from the oracle's own ambient so that a rounding flip upstream cannot move the 2-D lookup
    enc_w = _encode(po, ((amb + 1) / 2).astype(np.float32), P["encoder_ambient.embeddings"], off2, 2, S2)
    ws = [P[f"sigma_net.net.{l}.weight"] for l in range(3)]
    hs = _mlp_mp(ws, np.concatenate([enc_x, enc_w, np.tile(eye, (M, 1))], 1), 64, 1)
    np.testing.assert_allclose(sig, np.exp(hs[:, 0]), rtol=2e-3)
    sh = po.sh_encode_forward(d, 4)[0] if hasattr(po, "sh_encode_forward") else None
    wc = [P[f"color_net.net.{l}.weight"] for l in range(2)]
    hc = _mlp_mp(wc, np.concatenate([sh, hs[:, 1:], np.tile(ind, (M, 1))], 1), 80, 3)
    np.testing.assert_allclose(col, 1 / (1 + np.exp(-hc)), rtol=0, atol=5e-4)
    # and the mode is a small perturbation of the fp32 forward, not a different function
    s32, c32, a32 = po.nerf_forward(om, x, d, enc_a, ind, eye)
    assert 0 < np.abs(col - c32).max() < 2e-2 and np.abs(amb - a32).max() < 2e-2

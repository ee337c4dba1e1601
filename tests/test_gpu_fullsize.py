"""Parity at the sizes BASELINE.json names (not at toy sizes): the HIP path against the oracle's orc_render_frame.

  config[1]  512 x 512, xyz grid = hash, T = 2^19, max 16 steps/ray, torso on -- the three arithmetic variants of the
             fused network kernel, two consecutive frames of the pose stream (the second sees the lip-smoothing EMA).
  config[0]  one 256 x 256 frame of the shipped (tiled, T = 2^16) model, both engines.
  config[4]  1024 x 1024 split in interleaved 8-row bands over 8 ranks; ranks 0 and 7 are emulated here, one after
             the other (the gather itself is covered by the gloo tests).

What full size exercises that 32..160 px frames do not: N = 262 144 / 131 072 rays give 1 024 / 512 survivor-count
blocks per compaction launch, n_step = N // n_alive sees the real ratios, and the flat indices n * 3 reach 3 M.

Tolerances: |dRGB| <= 2e-3 (north-star fp32 bar; fp32 and fp32x2 kernels), <= 4e-3 (one 8-bit step; the f16 kernel =
the reference's -O arithmetic); depth 1e-3 where defined; loop iterations, live samples and sample slots EXACT
(regime B: no ray terminates on opacity, so the integer statistics do not depend on the MLP arithmetic)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HASH19 = dict(xyz_grid="hashgrid", xyz_log2_hashmap_size=19)


def _scene(size, engine, n_frames=8, **kw):
    from radnerf.scene import SyntheticScene, default_opt
    return SyntheticScene(H=size, W=size, n_frames=n_frames, device="cuda", opt=default_opt(engine=engine, **kw))


def _oracle(po, scene, rays_o, rays_d, bg_coords, bg_color, f, enc_a):
    m = scene.model
    om = po.model_from_module(m)
    rc = po.render_cfg_from_module(m, scene.opt.dt_gamma, scene.opt.max_steps)
    return po.render_frame(om, rc, rays_o.cpu().numpy(), rays_d.cpu().numpy(), enc_a.cpu().numpy(),
                           m.individual_codes[0].detach().cpu().numpy(), f["eye"].cpu().numpy(), bg_coords.cpu().numpy(),
                           f["poses"].cpu().numpy(), m.individual_codes_torso[0].detach().cpu().numpy(),
                           bg_color.reshape(-1, 3).cpu().numpy())


def _compare(out, model, img, dep, stats, rgb_tol, exact_counts=True):
    got = out["image"].reshape(-1, 3).cpu().numpy()
    err = float(np.abs(got - img).max())
    assert err <= rgb_tol, err
    gd = out["depth"].reshape(-1).cpu().numpy()
    assert np.array_equal(np.isnan(gd), np.isnan(dep))
    ok = ~np.isnan(dep)
    assert float(np.abs(gd[ok] - dep[ok]).max()) <= 1e-3
    st = model.last_stats
    if exact_counts and st is not None and "live_samples" in st:
        assert st["iterations"] == stats["iterations"], (st["iterations"], stats["iterations"])
        assert st["live_samples"] == stats["live_samples"], (st["live_samples"], stats["live_samples"])
        if "sample_slots" in st and model.engine == "fused":
            # the oracle counts the wrapper's padded slots (M += 128 - M % 128, raymarching.py:380-383), the device loop
            # the n_alive * n_step it really fills: they differ by 1..128 per iteration
            pad = stats["sample_slots"] - st["sample_slots"]
            assert st["iterations"] <= pad <= 128 * st["iterations"], pad
    return err


@pytest.mark.parametrize("mlp,tol", [("f32", 2e-3), ("f32x2", 2e-3), ("f16", 4e-3)])
def test_config1_512_hash19_frames_match_oracle(po, hiplib, mlp, tol):
    scene = _scene(512, "fused", mlp_dtype=mlp, **HASH19)
    assert scene.model.encoder.gridtype == "hash" and scene.model.encoder.embeddings.shape[0] == 6119864
    for i in (0, 1):
        f = scene.frame(i)
        with torch.no_grad():
            out = scene.render(i)
        img, dep, stats = _oracle(po, scene, f["rays_o"], f["rays_d"], f["bg_coords"], f["bg_color"], f, scene.model.enc_a)
        assert stats["live_samples"] > 1_000_000          # the benchmark's regime: ~1.24 M live samples per frame
        _compare(out, scene.model, img, dep, stats, tol)


def test_config1_512_hash19_ops_engine_matches_oracle(po, hiplib):
    """The per-operator engine (reference loop shape over the C-ABI operators) at full size."""
    scene = _scene(512, "ops", **HASH19)
    f = scene.frame(0)
    scene.model.count_samples = True
    with torch.no_grad():
        out = scene.render(0)
    img, dep, stats = _oracle(po, scene, f["rays_o"], f["rays_d"], f["bg_coords"], f["bg_color"], f, scene.model.enc_a)
    _compare(out, scene.model, img, dep, stats, 2e-3)


@pytest.mark.parametrize("engine", ["fused", "ops"])
def test_config0_256_frame_matches_oracle(po, hiplib, engine):
    """BASELINE config[0]: one 256 x 256 frame of the shipped model (tiled grids, T = 2^16)."""
    scene = _scene(256, engine)
    assert scene.model.encoder.gridtype == "tiled"
    f = scene.frame(0)
    if engine == "ops":
        scene.model.count_samples = True
    with torch.no_grad():
        out = scene.render(0)
    img, dep, stats = _oracle(po, scene, f["rays_o"], f["rays_d"], f["bg_coords"], f["bg_color"], f, scene.model.enc_a)
    _compare(out, scene.model, img, dep, stats, 2e-3)


@pytest.mark.parametrize("rank", [0, 7])
def test_config4_1024_bands_of_rank_match_oracle(po, hiplib, rank):
    """BASELINE config[4]: 1024 x 1024, world 8, interleaved 8-row bands; the band a rank renders equals the reference
    semantics applied to that band's 131 072 rays (schedule="band": the rank's own n_step policy)."""
    from radnerf.parallel import TileParallelRenderer
    scene = _scene(1024, "fused", **HASH19)
    m = scene.model
    tpr = TileParallelRenderer(scene, rank, 8, None, band=8, schedule="band")
    assert tpr.pix.numel() == 131072
    with torch.no_grad():
        m.enc_a = None
        f, (rays_o, rays_d), (bg_coords, bg_color) = tpr._inputs(0)
        out = m.render(rays_o, rays_d, f["auds"], bg_coords, f["poses"], eye=f["eye"], index=f["index"], bg_color=bg_color,
                       **scene.render_kwargs())
    img, dep, stats = _oracle(po, scene, rays_o, rays_d, bg_coords, bg_color, f, m.enc_a)
    _compare(out, m, img, dep, stats, 2e-3)
    band = tpr.render_local(1)                                  # the uint8 rows that would enter the gather
    assert tuple(band.shape) == (128, 1024, 3) and band.dtype == torch.uint8


def _config2_run(monkeypatch, product, steps):
    """`steps` optimizer steps of BASELINE config 2 (512 x 512 frame, 4 096 rays per step, hash T = 2^19 xyz grid, head model)
    with the occupancy refresh every 16 steps; returns losses, per-step sample counts, the running-average budget after each
    step and the final occupancy bitfield."""
    import random
    from radnerf.scene import SyntheticScene, default_opt
    from radnerf.train import GraphedTrainer, SyntheticTrainStream, Trainer
    for k, v in (("RN_TRAIN_HEAD", "fused" if product else "ops"), ("RN_TRAIN_LOSS", "fused" if product else "torch"),
                 ("RN_TRAIN_MARCH", "step" if product else "ops"), ("RN_TRAIN_NOISE", "torch")):
        monkeypatch.setenv(k, v)
    torch.manual_seed(0)
    random.seed(0)
    scene = SyntheticScene(H=512, W=512, n_frames=8, device="cuda", opt=default_opt(engine="ops", torso=False, smooth_lips=False, **HASH19))
    stream = SyntheticTrainStream(scene, n_rays=4096, seed=1)
    m = scene.model
    trainer = (GraphedTrainer if product else Trainer)(m, scene.opt)
    losses, counts, budgets = [], [], []
    for _ in range(steps):
        losses.append(float(trainer.step(stream.batch())))
        counts.append(int(m.step_counter[(m.local_step - 1) % 16, 0]))
        budgets.append(int(m.mean_count))
    if product:
        assert trainer.captures >= 1 and trainer.replays >= steps - 17
    return np.array(losses), counts, budgets, m.density_bitfield.clone()


def test_config2_training_steps_full_size(hiplib, monkeypatch):
    """BASELINE config 2 at its own size, 36 steps across two occupancy refreshes: the product path (hipGraph replay of the
    one-launch marcher + fused forward / backward + loss kernels + HipAdam) against this tree's per-operator chain of the reference's
    calls (near_far_from_aabb, march_rays_train, grid / SH / MLP operators, composite_rays_train, torch loss), same seeds, same
    jitter draws.  The marcher is bit-exact, so until the first refresh that sees trained weights the per-step sample counts are
    IDENTICAL; afterwards the two occupancy grids may differ in cells at the threshold (the weights agree to rounding, Adam with
    eps = 1e-15 amplifies it): counts within 0.1 %, bitfields within 0.01 % of their bits, losses within 2 % throughout (achieved:
    5e-5, 44 bits of 2 M, 1.2e-3 .. 5.8e-3 from run to run; 2.3e-4 over the first 16 steps)."""
    la, ca, ba, bits_a = _config2_run(monkeypatch, True, 36)
    lb, cb, bb, bits_b = _config2_run(monkeypatch, False, 36)
    assert np.all(np.isfinite(la)) and np.all(np.isfinite(lb))
    assert ca[:16] == cb[:16], (ca[:16], cb[:16])
    assert ba[:16] == bb[:16]
    assert max(abs(x - y) / max(y, 1) for x, y in zip(ca, cb)) <= 1e-3, (ca, cb)       # achieved 5e-5
    np.testing.assert_allclose(la, lb, rtol=2e-2, atol=1e-7)                            # achieved 1.2e-3 .. 5.8e-3 (run to run: atomics)
    np.testing.assert_allclose(la[:16], lb[:16], rtol=2e-3, atol=1e-7)                  # achieved 2.3e-4
    differing = int(np.unpackbits((bits_a ^ bits_b).cpu().numpy()).sum())
    assert differing <= 1e-4 * bits_a.numel() * 8, differing                            # achieved 44 of 2 097 152
    assert la[-1] < la[0]                                   # and it trains
    print("config 2 full size: loss rel. diff max %.2e (first 16: %.2e), count rel. diff max %.2e, differing occupancy bits %d of %d, "
          "samples per step %d .. %d" % (float(np.abs(la / lb - 1).max()), float(np.abs(la[:16] / lb[:16] - 1).max()),
                                         max(abs(x - y) / max(y, 1) for x, y in zip(ca, cb)), differing, bits_a.numel() * 8, min(ca), max(ca)))

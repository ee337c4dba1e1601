"""The reference's `-O` mode (fp16 autocast; main.py:111-120, nerf/utils.py:944) pinned by the reference's own Python:
tests/golden/reference_mixed.npz holds NeRFNetwork.forward / density / forward_torso and whole frames produced by the
UNMODIFIED nerf/network.py + nerf/renderer.py under fp16 autocast on the CPU (generator: tests/golden/make_golden.py
`mixed`: torch's CPU autocast in float16 for nn.Linear / nn.Conv1d, the legacy CUDA autocast flag for the reference's own
`torch.is_autocast_enabled()` / `custom_fwd` rules -- gridencoder/grid.py:41-44, nerf/network.py:246).

Two implementations of that arithmetic are checked against it: the oracle's orc_nerf_forward_mp16 (CPU) and the opt-in 16-bit
matrix-core kernel k_nerf_fused_h16 (GPU, opt.mlp_dtype = "f16").  All three round to fp16 at slightly different points (the
reference rounds every layer's OUTPUT to fp16, the kernels round the operands entering the matrix instruction and keep fp32
accumulators between layers), so they agree to 16-bit rounding noise, not bit for bit.  Stated tolerance of the mode:
sigma 1e-3 relative, rgb 1e-3, ambient 1e-4 per sample (measured 3.7e-4 / 2.5e-4 / 3e-5); frames 4e-3 = one 8-bit step."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import cases  # noqa: E402

SIGMA_RTOL, RGB_ATOL, AMB_ATOL, FRAME_ATOL = 1e-3, 1e-3, 1e-4, 4e-3


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "reference_mixed.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def flow():
    return np.load(os.path.join(HERE, "golden", "reference_flow.npz"), allow_pickle=False)


def _scene32(device, **kw):
    from radnerf.scene import SyntheticScene, default_opt
    return SyntheticScene(H=32, W=32, n_frames=8, device=device, opt=default_opt(**kw))


def test_reference_autocast_dtypes_are_the_cuda_rule(gold):
    """What leaves the reference's forward under autocast: fp32 where network.py / the custom_fwd rules force it (ambient:
    `.float()`, network.py:246; sigma: trunc_exp casts to fp32, activation.py:8), fp16 elsewhere."""
    dt = dict(s.split(":") for s in gold["dtypes"])
    assert dt["sigma"] == "torch.float32" and dt["ambient"] == "torch.float32"
    assert dt["color"] == "torch.float16" and dt["torso_alpha"] == "torch.float16" and dt["enc_a"] == "torch.float16"
    # and the mode is a perturbation of the fp32 forward at 16-bit rounding level, not something else
    assert np.abs(gold["net_sigma"] / gold["net_sigma_fp32"] - 1).max() < 2e-3
    assert np.abs(gold["net_color"] - gold["net_color_fp32"]).max() < 2e-3


def test_oracle_mp16_forward_matches_reference_autocast(po, gold, flow, hiplib):
    scene = _scene32("cpu")
    m = scene.model
    om = po.model_from_module(m)
    ind, eye = m.individual_codes[0].detach().numpy(), scene.eye.numpy()
    sig, col, amb = po.nerf_forward(om, flow["net_x"], flow["net_d"], gold["net_enc_a"], ind, eye, mlp_dtype="f16")
    np.testing.assert_allclose(sig, gold["net_sigma"], rtol=SIGMA_RTOL, atol=1e-6)
    np.testing.assert_allclose(col, gold["net_color"], rtol=0, atol=RGB_ATOL)
    np.testing.assert_allclose(amb, gold["net_ambient"], rtol=0, atol=AMB_ATOL)
    np.testing.assert_allclose(sig, gold["net_density"], rtol=SIGMA_RTOL, atol=1e-6)          # NeRFNetwork.density = the same sigma


def test_oracle_frames_within_one_8bit_step_of_reference_autocast(po, gold, hiplib):
    """Whole frames: the fp32 oracle frame against the reference's autocast frame (the loop policy, compaction order, torso
    layer and blend are identical; the 16-bit layers move a pixel by less than one 8-bit step)."""
    from test_golden_frames import _hash19_scene, _oracle_frame
    scene = _scene32("cpu")
    img, dep, _ = _oracle_frame(po, scene, scene.frame(0), gold["tiled16_frame0_enc_a"])
    np.testing.assert_allclose(img, gold["tiled16_frame0_image"], rtol=0, atol=FRAME_ATOL)
    scene = _hash19_scene("cpu")
    img, dep, stats = _oracle_frame(po, scene, scene.frame(0), gold["hash19_frame0_enc_a"])
    assert stats["live_samples"] > 10000
    np.testing.assert_allclose(img, gold["hash19_frame0_image"], rtol=0, atol=FRAME_ATOL)


@pytest.mark.gpu
def test_hip_f16_kernel_matches_reference_autocast(gold, flow, hiplib):
    from radnerf import fused
    scene = _scene32("cuda", engine="fused", mlp_dtype="f16")
    m = scene.model
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    with torch.no_grad():
        sig, col, amb = fused.network_forward(m, t(flow["net_x"]), t(flow["net_d"]), t(gold["net_enc_a"]), m.individual_codes[0].detach(),
                                              scene.eye)
    np.testing.assert_allclose(sig.cpu().numpy(), gold["net_sigma"], rtol=SIGMA_RTOL, atol=1e-6)
    np.testing.assert_allclose(col.cpu().numpy(), gold["net_color"], rtol=0, atol=RGB_ATOL)
    np.testing.assert_allclose(amb.cpu().numpy(), gold["net_ambient"], rtol=0, atol=AMB_ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize("tables", ["fp32", "persistent fp16"])
def test_hip_f16_frames_match_reference_autocast(gold, hiplib, tables):
    """The fused engine in the reference's -O arithmetic (mlp_dtype = f16; `persistent fp16`: the grid tables are kept as fp16
    copies the kernels read directly, the cast gridencoder/grid.py:43-44 repeats per call) against the reference's autocast frames."""
    from test_golden_frames import _hash19_scene
    half = tables != "fp32"
    scene = _scene32("cuda", engine="fused", mlp_dtype="f16", half_tables=half)
    with torch.no_grad():
        out = scene.render(0)
    np.testing.assert_allclose(scene.model.enc_a.cpu().numpy(), gold["tiled16_frame0_enc_a"], rtol=0, atol=5e-4)   # fp16 audio convs there
    np.testing.assert_allclose(out["image"].reshape(-1, 3).cpu().numpy(), gold["tiled16_frame0_image"], rtol=0, atol=FRAME_ATOL)
    scene = _hash19_scene("cuda", "fused", mlp_dtype="f16", half_tables=half)
    with torch.no_grad():
        out = scene.render(0)
    np.testing.assert_allclose(out["image"].reshape(-1, 3).cpu().numpy(), gold["hash19_frame0_image"], rtol=0, atol=FRAME_ATOL)

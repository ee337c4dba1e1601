"""GPU parity of the one-kernel audio code (rn_audio_encode_windows / _stream / rn_audio_smooth) against the PyTorch
modules it replaces -- AudioNet + AudioAttNet (nerf/network.py:10-67, 170-185), whose arithmetic IS the reference's
(third-party torch ops); the modules themselves are pinned to the reference's outputs by tests/golden.
Tolerance: fp32 sums in a different order than MIOpen / rocBLAS, |d| <= 2e-5 on codes of magnitude ~1."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(**kw):
    from radnerf.scene import SyntheticScene, default_opt
    return SyntheticScene(H=32, W=32, n_frames=24, device="cuda", opt=default_opt(engine="fused", **kw))


def _torch_code(m, auds):
    """The nn.Modules themselves (NeRFNetwork.encode_audio routes CUDA tensors through the kernels under test)."""
    with torch.no_grad():
        enc = m.audio_net(auds)
        return m.audio_att_net(enc.unsqueeze(0)) if m.att > 0 else enc


@pytest.mark.parametrize("asr", ["cpierse/wav2vec2-large-xlsr-53-esperanto", "deepspeech", "other"])   # dim_in 44 / 29 / 32
def test_windows_match_torch_modules(hiplib, asr):
    from radnerf import audio
    from radnerf.rays import get_audio_features
    scene = _scene(asr_model=asr)
    m = scene.model
    feats = torch.randn(24, m.audio_in_dim, 16, device="cuda") * 3
    wins = torch.stack([get_audio_features(feats, 2, i) for i in (0, 3, 11, 20, 23)])       # incl. zero-padded ends
    got = audio.encode_windows(m, wins)
    want = torch.cat([_torch_code(m, w) for w in wins])
    assert got.shape == want.shape == (5, 64)
    assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    one = audio.encode_windows(m, wins[2])
    assert torch.equal(one, got[2:3])


def test_stream_windows_are_cut_like_get_audio_features(hiplib):
    from radnerf import audio
    from radnerf.rays import get_audio_features
    scene = _scene()
    m, T = scene.model, 24
    feats = scene.aud_features
    got = audio.encode_stream(m, feats, 20, 9)                                           # frames 20..23, 0..4 (wraps)
    ids = [(20 + i) % T for i in range(9)]
    want = audio.encode_windows(m, torch.stack([get_audio_features(feats, 2, i) for i in ids]))
    assert torch.equal(got, want)


def test_no_attention_variant(hiplib):
    from radnerf import audio
    scene = _scene(att=0)
    m = scene.model
    a = torch.randn(1, m.audio_in_dim, 16, device="cuda")
    got = audio.encode_windows(m, a[None])
    assert (got - _torch_code(m, a)).abs().max().item() <= 2e-5


def test_smoothing_recurrence_and_render_path(po, hiplib):
    """The fused engine's frames use the kernel path; the state it carries equals the reference recurrence
    0.35 * prev + 0.65 * code (nerf/renderer.py:190-194) evaluated with the torch modules."""
    from radnerf.rays import get_audio_features
    scene = _scene()
    m = scene.model
    assert m.fused_audio_enabled()
    ref = None
    for i in range(4):
        with torch.no_grad():
            scene.render(i)
        code = _torch_code(m, get_audio_features(scene.aud_features, 2, i))
        ref = code if ref is None else 0.35 * ref + (1 - 0.35) * code
        assert (m.enc_a - ref).abs().max().item() <= 3e-5
    # switching the kernel path off gives the same frames
    other = _scene(audio_engine="torch")
    assert not other.model.fused_audio_enabled()
    with torch.no_grad():
        a = scene.render(5)["image"]
        for i in range(4):
            other.render(i)
        b = other.render(5)["image"]
    assert (a - b).abs().max().item() <= 1e-4


def test_frame_parallel_advance_uses_one_launch(hiplib):
    """A rank that skips frames folds their codes with encode_stream + smooth_: same state as rendering every frame."""
    from radnerf.parallel import FrameParallelRenderer
    seq, par = _scene(), _scene()
    with torch.no_grad():
        for i in range(9):
            seq.render(i)
        fpr = FrameParallelRenderer(par, rank=2, world=3, dist=None, gather=False)
        for s in range(3):                      # frames 2, 5, 8
            fpr.step(s)
    assert (seq.model.enc_a - par.model.enc_a).abs().max().item() <= 1e-6


@pytest.mark.parametrize("asr,att", [("cpierse/wav2vec2-large-xlsr-53-esperanto", 2), ("deepspeech", 2), ("other", 0)])
def test_backward_kernels_match_torch_autograd(hiplib, monkeypatch, asr, att):
    """rn_audio_encode_windows_backward (radnerf.audio.encode_windows_train, what NeRFNetwork.encode_audio uses under autograd on the
    GPU) against torch autograd through the nn.Modules: every parameter gradient of AudioNet (+ AudioAttNet) within 1e-4 of its
    largest magnitude (fp32 sums in a different order; the gradient of one code is the sum over 8 frames and 16 positions)."""
    from radnerf.rays import get_audio_features
    scene = _scene(asr_model=asr, att=att)
    m = scene.model
    m.train()
    feats = torch.randn(24, m.audio_in_dim, 16, device="cuda") * 2
    a = get_audio_features(feats, 2, 7) if att > 0 else torch.randn(1, m.audio_in_dim, 16, device="cuda")
    gy = torch.randn(1, 64, device="cuda")
    params = [p for mod in ([m.audio_net] + ([m.audio_att_net] if att > 0 else [])) for p in mod.parameters()]
    grads = {}
    for env in ("hip", "torch"):
        monkeypatch.setenv("RN_AUDIO_TRAIN", env)
        enc = m.encode_audio(a)
        assert enc.shape == (1, 64) and enc.requires_grad
        grads[env] = (enc.detach().clone(), torch.autograd.grad(enc, params, gy))
    assert (grads["hip"][0] - grads["torch"][0]).abs().max().item() <= 2e-5 * max(1.0, grads["torch"][0].abs().max().item())
    for p, g, r in zip(params, grads["hip"][1], grads["torch"][1]):
        assert g.shape == r.shape
        assert (g - r).abs().max().item() <= 1e-4 * max(float(r.abs().max()), 1e-3), (tuple(p.shape), float((g - r).abs().max()), float(r.abs().max()))

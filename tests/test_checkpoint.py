"""Reference checkpoint format (nerf/utils.py:1302-1396): round trip through this tree's NeRFNetwork."""
import os

import torch


def test_reference_checkpoint_round_trip(hiplib, tmp_path):
    from radnerf.checkpoint import load_checkpoint, save_checkpoint
    from radnerf.scene import SyntheticScene, default_opt
    a = SyntheticScene(H=8, W=8, n_frames=8, device="cpu", opt=default_opt(), seed=0).model
    a.mean_count, a.mean_density = 1234, 0.5
    full = save_checkpoint(a, os.path.join(tmp_path, "ngp_ep0001.pth"), epoch=1, global_step=10, stats={})
    best = save_checkpoint(a, os.path.join(tmp_path, "ngp.pth"), best=True)
    b = SyntheticScene(H=8, W=8, n_frames=8, device="cpu", opt=default_opt(), seed=7).model
    assert not torch.equal(a.sigma_net.net[0].weight, b.sigma_net.net[0].weight)
    missing, unexpected = load_checkpoint(b, full)
    assert missing == [] and unexpected == []
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    assert b.mean_count == 1234 and b.mean_density == 0.5 and b.mean_density_torso == a.mean_density_torso
    c = SyntheticScene(H=8, W=8, n_frames=8, device="cpu", opt=default_opt(), seed=9).model
    missing, unexpected = load_checkpoint(c, best)
    assert missing == ["density_grid"] and unexpected == []       # "best" checkpoints drop the float grid (:1351)
    assert torch.equal(c.density_bitfield, a.density_bitfield)
    # a bare state_dict is accepted too (:1376-1379)
    d = SyntheticScene(H=8, W=8, n_frames=8, device="cpu", opt=default_opt(), seed=11).model
    load_checkpoint(d, a.state_dict())
    assert torch.equal(d.encoder.embeddings, a.encoder.embeddings)
    # state-dict key names the reference's files use
    keys = set(a.state_dict().keys())
    for k in ("encoder.embeddings", "encoder.offsets", "encoder_ambient.embeddings", "torso_encoder.embeddings",
              "sigma_net.net.0.weight", "color_net.net.1.weight", "ambient_net.net.2.weight", "torso_net.net.2.weight",
              "audio_net.encoder_conv.0.weight", "audio_att_net.attentionConvNet.0.weight", "audio_att_net.attentionNet.0.weight",
              "individual_codes", "individual_codes_torso", "density_bitfield", "density_grid", "density_grid_torso",
              "aabb_train", "aabb_infer", "step_counter"):
        assert k in keys, k

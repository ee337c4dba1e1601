"""Reference checkpoint format (nerf/utils.py:1302-1396): round trip through this tree's NeRFNetwork."""
import os

import pytest
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


def test_state_dict_layout_is_the_reference_models(hiplib):
    """Names, shapes and dtypes of every state-dict entry of the reference's NeRFNetwork (tests/golden/reference_flow.npz, taken
    from the reference class itself): a file written by the reference's Trainer.save_checkpoint has exactly these."""
    import numpy as np
    from radnerf.scene import SyntheticScene, default_opt
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_flow.npz"))
    sd = SyntheticScene(H=8, W=8, n_frames=8, device="cpu", opt=default_opt()).model.state_dict()
    names = sorted(sd.keys())
    assert names == [str(n) for n in gold["param_names"]]
    assert ["x".join(str(d) for d in sd[k].shape) for k in names] == [str(v) for v in gold["param_shapes"]]
    assert [str(sd[k].dtype) for k in names] == [str(v) for v in gold["param_dtypes"]]


def test_full_checkpoint_restores_optimizer_and_meta(hiplib, tmp_path):
    from radnerf.checkpoint import load_checkpoint, save_checkpoint
    from radnerf.scene import SyntheticScene, default_opt
    from radnerf.train import make_optimizer
    opt = default_opt(torso=False)
    a = SyntheticScene(H=8, W=8, n_frames=8, device="cpu", opt=opt, seed=0).model
    oa = make_optimizer(a, fused=False)
    (a.sigma_net.net[0].weight.sum() + a.individual_codes.sum()).backward()
    oa.step()
    path = save_checkpoint(a, os.path.join(tmp_path, "ngp_ep0003.pth"), epoch=3, global_step=77, stats={"loss": [1.0]},
                           optimizer=oa.state_dict())
    b = SyntheticScene(H=8, W=8, n_frames=8, device="cpu", opt=opt, seed=5).model
    ob = make_optimizer(b, fused=False)
    missing, unexpected = load_checkpoint(b, path, optimizer=ob, half_tables=False)
    assert missing == [] and unexpected == []
    assert b.checkpoint_meta["epoch"] == 3 and b.checkpoint_meta["global_step"] == 77 and b.checkpoint_meta["optimizer_loaded"]
    sa, sb = oa.state_dict()["state"], ob.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) for k in sa)


@pytest.mark.gpu
def test_half_tables_checkpoint_feeds_the_fused_engine(hiplib, tmp_path):
    """load_checkpoint(half_tables=True) builds persistent fp16 copies of the three grid tables and the fused
    engine renders from them; frames equal, bit for bit, those rendered from tables cast anew on every call (what the
    reference's -O mode does, gridencoder/grid.py:43-44), the copies are not re-made while the parameters stand still, and
    they follow an in-place update of the parameters."""
    from radnerf.checkpoint import load_checkpoint, save_checkpoint
    from radnerf.scene import SyntheticScene, default_opt

    def scene(seed):
        return SyntheticScene(H=48, W=48, n_frames=8, device="cuda", opt=default_opt(engine="fused", mlp_dtype="f16"), seed=seed)
    src = scene(0)
    path = save_checkpoint(src.model, os.path.join(tmp_path, "ngp.pth"))
    a, b = scene(0), scene(0)                        # same stream (poses, audio); the parameters come from the file
    for s in (a, b):
        with torch.no_grad():
            for p in s.model.parameters():
                p.mul_(0.5)
        missing, unexpected = load_checkpoint(s.model, path, map_location="cuda", half_tables=True)
        assert missing == [] and unexpected == [] and s.model.opt.half_tables
    encs = lambda m: (m.encoder, m.encoder_ambient, m.torso_encoder)  # noqa: E731
    assert all(e._half is not None and e._half[1].dtype == torch.float16 for e in encs(a.model))
    kept = [e._half[1] for e in encs(a.model)]

    def frame(s, i, recast):
        if recast:                                   # the reference's behaviour: a fresh cast of every table for this call
            for e in encs(s.model):
                e._half = None
        with torch.no_grad():
            return s.render(i)["image"].clone()
    for i in (0, 1):
        fa, fb = frame(a, i, False), frame(b, i, True)
        assert torch.equal(fa, fb)
    assert all(e._half[1] is k for e, k in zip(encs(a.model), kept)), "persistent copies were re-made without a parameter change"
    from radnerf import fused
    st = fused._state(a.model)
    assert st.tables[0].dtype == torch.float16 and st.gx.dtype == hiplib.RN_F16 and st.gt.dtype == hiplib.RN_F16
    # an in-place parameter update (the optimizer's) re-makes the copies
    with torch.no_grad():
        for s in (a, b):
            s.model.encoder.embeddings.mul_(1.25)
            s.model.enc_a = None
    b.model.enc_a = None
    f2a, f2b = frame(a, 0, False), frame(b, 0, True)
    assert torch.equal(f2a, f2b)
    assert a.model.encoder._half[1] is not kept[0]

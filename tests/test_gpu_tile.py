"""BASELINE config 4 on one GPU: the bands a TileParallelRenderer rank renders, put back together, are the frame a
single whole-image render gives (every stage of the path is per pixel, nerf/renderer.py:225-311).  Ranks are emulated
one after the other in this process; the gather itself is covered by the gloo tests in test_distributed_cpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("engine", ["fused", "ops"])
@pytest.mark.parametrize("world,size", [(4, 64), (3, 80)])
def test_bands_reassemble_to_the_whole_frame(po, hiplib, engine, world, size):
    from radnerf.parallel import TileParallelRenderer
    from radnerf.scene import SyntheticScene, default_opt
    scene = SyntheticScene(H=size, W=size, n_frames=8, device="cuda", opt=default_opt(engine=engine))
    m = scene.model
    with torch.no_grad():
        m.enc_a = None
        whole = (scene.render(0)["image"].reshape(size, size, 3) * 255).to(torch.uint8)
        frame = torch.zeros_like(whole)
        covered = torch.zeros(size, dtype=torch.int32)
        for r in range(world):
            tpr = TileParallelRenderer(scene, r, world, None, band=8)
            m.enc_a = None                      # each real rank owns its model: frame 0 starts without EMA history
            frame[tpr.rows[r].cuda()] = tpr.render_local(0)
            covered[tpr.rows[r]] += 1
    assert (covered == 1).all()
    diff = (frame.int() - whole.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3      # same pixels, up to a rounding flip

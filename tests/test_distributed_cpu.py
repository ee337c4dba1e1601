"""Frame-parallel sharding (radnerf/parallel.py) on CPU with the gloo backend, world_size 2 and 3: the frames
every rank ends up with, and the lip-smoothing EMA state each rank renders with, must equal a sequential
single-process render of the same stream.  The renderer itself is replaced by a small deterministic stand-in
(no GPU here); what is under test is the sharding / audio-state / gather logic bench.py relies on."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

H = W = 8


class _StandInModel:
    """encode_audio + EMA exactly like NeRFRenderer._audio_code (nerf/renderer.py:188-194)."""
    smooth_lips = True

    def __init__(self):
        self.enc_a = None
        self.last_stats = None

    def encode_audio(self, auds):
        return auds.mean(dim=(0, 2)).reshape(1, -1) * 0.1


class _StandInScene:
    def __init__(self, n_frames):
        from types import SimpleNamespace
        self.H, self.W, self.n_frames = H, W, n_frames
        self.opt = SimpleNamespace(att=2)
        g = torch.Generator().manual_seed(0)
        self.aud_features = torch.randn(n_frames, 6, 16, generator=g)
        self.model = _StandInModel()
        self.log = []

    def render(self, i):
        from radnerf.rays import get_audio_features
        m = self.model
        enc = m.encode_audio(get_audio_features(self.aud_features, 2, i % self.n_frames))
        if m.enc_a is not None:
            enc = 0.35 * m.enc_a + (1 - 0.35) * enc
        m.enc_a = enc
        self.log.append((i, enc.clone()))
        base = torch.sigmoid(enc.sum()) * 0.5 + (i % 7) / 20.0
        img = (base + torch.arange(H * W * 3, dtype=torch.float32).reshape(1, H * W, 3) / (H * W * 3) * 0.3).clamp(0, 1)
        return {"image": img}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sequential(n_total, n_frames):
    scene = _StandInScene(n_frames)
    frames, encs = [], []
    for g in range(n_total):
        out = scene.render(g)
        frames.append((out["image"].reshape(H, W, 3) * 255).to(torch.uint8))
        encs.append(scene.model.enc_a.clone())
    return frames, encs


def _worker(rank, world, port, steps, n_frames, ret, gather_every=8, gather_to="all"):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rad-nerf_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from radnerf.parallel import FrameParallelRenderer, frame_of
    scene = _StandInScene(n_frames)
    fpr = FrameParallelRenderer(scene, rank, world, dist, gather_every=gather_every, gather_to=gather_to)
    for s in range(steps):
        fpr.step(s)
    stacks = fpr.finish()
    ret[rank] = dict(stacks=[t.numpy() for t in stacks],
                     encs={frame_of(s, rank, world): None for s in range(steps)},
                     log=[(i, e.numpy()) for i, e in scene.log])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,steps,gather_every", [(2, 4, 8), (3, 4, 8), (2, 7, 3)])   # (2, 7, 3): two full batches + a partial one
def test_frame_parallel_equals_sequential(hiplib, world, steps, gather_every):
    n_frames = 16
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, steps, n_frames, ret, gather_every), nprocs=world, join=True)
    seq_frames, seq_encs = _sequential(steps * world, n_frames)
    for rank in range(world):
        r = ret[rank]
        assert len(r["stacks"]) == steps
        for s in range(steps):
            stack = r["stacks"][s]  # [world, H, W, 3]: frame s*world + q rendered by rank q
            for q in range(world):
                assert np.array_equal(stack[q], seq_frames[s * world + q].numpy()), (rank, s, q)
        # the EMA state this rank rendered with equals the sequential one at its global frames
        rendered = {i: e for i, e in r["log"]}
        for s in range(steps):
            g = s * world + rank
            np.testing.assert_allclose(rendered[g], seq_encs[g].numpy(), rtol=0, atol=1e-7)


def test_frame_parallel_gather_to_rank0(hiplib):
    """Default collective of the frame-parallel renderer: finished frames go to rank 0 only (world 4, partial last batch)."""
    world, steps, n_frames = 4, 5, 32
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), steps, n_frames, ret, 2, "rank0"), nprocs=world, join=True)
    seq_frames, _ = _sequential(steps * world, n_frames)
    assert len(ret[0]["stacks"]) == steps and all(len(ret[r]["stacks"]) == 0 for r in range(1, world))
    for s in range(steps):
        for q in range(world):
            assert np.array_equal(ret[0]["stacks"][s][q], seq_frames[s * world + q].numpy()), (s, q)


def test_skipped_frames_partition():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rad-nerf_amd"))
    from radnerf.parallel import frame_of, skipped_frames
    for world in (1, 2, 4, 8):
        for rank in range(world):
            seen = []
            for s in range(5):
                seen += skipped_frames(s, rank, world) + [frame_of(s, rank, world)]
            assert seen == list(range(frame_of(4, rank, world) + 1))  # every frame's audio is folded exactly once, in order


# ---- tile-parallel (BASELINE config 4): one frame split in interleaved row bands -------------------------------

class _PixelModel:
    """Per-pixel stand-in for NeRFRenderer.render: every output pixel depends only on its own ray, bg coordinate and
    bg colour -- the property of the real path that tile-parallel rendering relies on (nerf/renderer.py:225-311)."""

    def render(self, rays_o, rays_d, auds, bg_coords, poses, eye=None, index=0, bg_color=None, **kw):
        v = torch.sigmoid(3 * rays_d[..., :1] + bg_coords[..., :1] * 0.7 + poses.sum() * 0.01 + auds.mean())
        return {"image": (v * bg_color * torch.tensor([1.0, 0.8, 0.6])).clamp(0, 1)}


class _TileScene:
    def __init__(self, Hh, Ww):
        from types import SimpleNamespace
        self.H, self.W, self.device = Hh, Ww, torch.device("cpu")
        g = torch.Generator().manual_seed(1)
        self.rays_d = torch.randn(1, Hh * Ww, 3, generator=g)
        self.bg_coords = torch.rand(1, Hh * Ww, 2, generator=g)
        self.bg_color = torch.rand(1, Hh * Ww, 3, generator=g)
        self.model = _PixelModel()

    def frame(self, i):
        return dict(rays_o=torch.zeros_like(self.rays_d), rays_d=self.rays_d + 0.01 * i, auds=torch.full((8, 4, 16), 0.1 * i),
                    bg_coords=self.bg_coords, poses=torch.ones(1, 6), eye=None, index=0, bg_color=self.bg_color)

    def render_kwargs(self):
        return {}


def _tile_worker(rank, world, port, Hh, Ww, band, ret, gather_to="rank0"):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rad-nerf_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from radnerf.parallel import TileParallelRenderer
    tpr = TileParallelRenderer(_TileScene(Hh, Ww), rank, world, dist, band=band, gather_to=gather_to)
    for i in range(2):
        tpr.step(i)
    ret[rank] = [f.numpy() for f in tpr.finish()]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gather_to", ["rank0", "all"])
@pytest.mark.parametrize("world,Hh,band", [(2, 16, 4), (3, 20, 4), (2, 10, 8), (8, 128, 8)])   # even / ragged split, short last band, config 4's world of 8
def test_tile_parallel_equals_whole_frame(hiplib, world, Hh, band, gather_to):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rad-nerf_amd"))
    from radnerf.parallel import TileParallelRenderer, stripe_rows
    Ww = 12
    rows = torch.cat([stripe_rows(Hh, r, world, band) for r in range(world)])
    assert sorted(rows.tolist()) == list(range(Hh))                      # the bands partition the image
    whole = TileParallelRenderer(_TileScene(Hh, Ww), 0, 1, None, band=band)
    for i in range(2):
        whole.step(i)
    expect = [f.numpy() for f in whole.finish()]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_tile_worker, args=(world, _free_port(), Hh, Ww, band, ret, gather_to), nprocs=world, join=True)
    for rank in range(world):
        if gather_to == "rank0" and rank != 0:
            assert ret[rank] == []                      # the rows went to rank 0 only
            continue
        for i in range(2):
            assert np.array_equal(ret[rank][i], expect[i]), (rank, i)


def test_band_schedules_are_checked_against_the_whole_frame_schedule():
    """The host-side verdict of TileParallelRenderer(schedule="verify"): per-iteration live-ray counts of every rank (they ride in
    the frame's gather) -> did band-local policies n_step = max(min(N_r // alive_r, 8), 1) equal the whole frame's?"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rad-nerf_amd"))
    from radnerf.parallel import band_schedule_is_frame_schedule as ok
    N = [1000, 1000]
    assert ok(N, [[1000, 300, 280, 0], [1000, 310, 290, 0]], 16)                    # 2000 // 610 = 3 = 1000 // 300 = 1000 // 310
    assert not ok(N, [[1000, 250, 0, 0], [1000, 340, 0, 0]], 16)                    # rank 0: 4, rank 1: 2, frame: 2000 // 590 = 3
    assert ok(N, [[1000, 0, 0, 0], [1000, 400, 390, 0]], 16) is False               # frame: 2000 // 400 = 5, rank 1: 1000 // 400 = 2
    assert ok([1000, 500], [[1000, 400, 0], [500, 200, 0]], 16)                     # ragged bands, same ratio
    assert ok(N, [[1000, 300, 290, 280], [1000, 300, 290, 280]], 4)                 # step reaches max_steps = 4 after 1 + 3: later counts ignored

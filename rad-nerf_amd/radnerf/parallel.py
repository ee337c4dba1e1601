"""Frame-parallel rendering across the GPUs of one node (BASELINE config 3).

One process per GPU.  Rank r renders global frames r, r + W, r + 2W, ... of the pose/audio stream; the
weights, tables and occupancy bitfield (~16 MB) are replicated.  The only cross-frame state of the path is
the lip-smoothing EMA of the audio code (nerf/renderer.py:190-194); a rank therefore encodes the audio
windows of the frames it skips (one batched AudioNet pass) and folds them into its EMA state before it
renders, which reproduces the sequential result exactly.  The only collective is the gather of finished
frames (uint8, 786 KB at 512^2) over RCCL; it is issued asynchronously so it overlaps the next frame.
"""
import json
import os

import torch

from .rays import get_audio_features


def frame_of(step, rank, world):
    """Global frame index rendered by `rank` at its local step `step`."""
    return step * world + rank


def skipped_frames(step, rank, world):
    """Frames whose audio this rank must fold into its EMA state before rendering frame_of(step)."""
    g = frame_of(step, rank, world)
    first = 0 if step == 0 else g - world + 1
    return list(range(first, g))


class FrameParallelRenderer:
    def __init__(self, scene, rank=0, world=1, dist=None, gather=True):
        self.scene, self.rank, self.world, self.dist = scene, rank, world, dist
        self.gather = gather and dist is not None and world > 1
        self.pending = []
        self.frames_u8 = None

    # -- audio state ------------------------------------------------------------------------------
    def _advance_audio(self, frames):
        """Fold the raw audio codes of `frames` (ascending) into model.enc_a, as rendering them would."""
        m = self.scene.model
        if not frames or not m.smooth_lips:
            return
        n = self.scene.n_frames
        for g in frames:  # tiny networks; kept sequential so the arithmetic equals the reference's per-frame path
            enc = m.encode_audio(get_audio_features(self.scene.aud_features, self.scene.opt.att, g % n))
            m.enc_a = enc if m.enc_a is None else 0.35 * m.enc_a + (1 - 0.35) * enc

    # -- one step = one frame on this rank ----------------------------------------------------------
    def step(self, step):
        self._advance_audio(skipped_frames(step, self.rank, self.world))
        g = frame_of(step, self.rank, self.world)
        out = self.scene.render(g)
        image = out["image"]
        u8 = (image.reshape(self.scene.H, self.scene.W, 3) * 255).to(torch.uint8)
        if self.gather:
            if self.frames_u8 is None:
                self.frames_u8 = torch.empty((self.world,) + tuple(u8.shape), dtype=torch.uint8, device=u8.device)
            # async: RCCL runs the gather on its own stream, overlapping the next frame's kernels
            buf = torch.empty_like(self.frames_u8)
            if self.dist.get_backend() == "nccl":
                work = self.dist.all_gather_into_tensor(buf, u8.contiguous(), async_op=True)
            else:  # gloo (CPU tests)
                work = self.dist.all_gather(list(buf.unbind(0)), u8.contiguous(), async_op=True)
            self.pending.append((work, buf, u8))
        self.last_frame = u8
        return u8

    def finish(self):
        """Wait for every outstanding gather; returns the gathered [world, H, W, 3] uint8 stacks in step order."""
        done = []
        for work, buf, _ in self.pending:
            work.wait()
            done.append(buf)
        if done:
            self.frames_u8 = done[-1]
        self.pending = []
        return done

    # -- bookkeeping for bench.py -------------------------------------------------------------------
    def loop_counters(self):
        """Cumulative (iterations, live samples, sample slots) of the fused engine's device-side loop, or None."""
        try:
            from . import fused
        except ImportError:
            return None
        return fused.loop_counters(self.scene.model)

    def count_samples(self, steps):
        """Untimed replay of `steps` with the per-iteration live-sample count switched on; returns the mean
        number of live samples (deltas[:,0] > 0) and of padded sample slots per frame."""
        m = self.scene.model
        live = slots = 0
        m.count_samples = True
        try:
            for s in steps:
                self.scene.render(frame_of(s, self.rank, self.world))
                live += m.last_stats["live_samples"]
                slots += m.last_stats["sample_slots"]
        finally:
            m.count_samples = False
        return live / max(len(steps), 1), slots / max(len(steps), 1)

    @staticmethod
    def measured_traffic(key):
        """HBM bytes per launch from a committed rocprofv3 --pmc pass (profiles/traffic.json), else None."""
        p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles",
                         "traffic.json")
        if os.path.exists(p):
            try:
                return json.load(open(p)).get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                return None
        return None

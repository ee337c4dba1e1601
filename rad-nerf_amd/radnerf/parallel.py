"""Frame-parallel (BASELINE config 3) and tile-parallel (config 4) rendering across the GPUs of one node
(SURVEY §8(e)).

Frame-parallel:

One process per GPU.  Rank r renders global frames r, r + W, r + 2W, ... of the pose/audio stream; the
weights, tables and occupancy bitfield (~16 MB) are replicated.  The only cross-frame state of the path is
the lip-smoothing EMA of the audio code (nerf/renderer.py:190-194); a rank therefore encodes the audio
windows of the frames it skips (one batched AudioNet pass) and folds them into its EMA state before it
renders, which reproduces the sequential result exactly.  The only collective is the gather of finished
frames (uint8, 786 KB at 512^2) over RCCL; it is issued asynchronously so it overlaps the next frame.

Tile-parallel: ONE frame is split over the ranks in interleaved bands of `band` image rows (the head sits in the
centre of the image, so contiguous bands would leave the outer ranks idle).  Every stage of the path is per pixel
-- head march/composite, torso pass, blend (nerf/renderer.py:225-311) -- so a rank runs the whole path on its own
pixels and the only bulk exchange is the gather of finished uint8 rows; every rank advances the audio EMA itself.
One scalar is shared: the reference's step policy n_step = max(min(N // n_alive, 8), 1) (renderer.py:249) looks at
the live-ray count of the whole call, and because `step += n_step` can overshoot max_steps the schedule shows in
the pixels.  With schedule="frame" (default) the ranks therefore sum their live counts between loop iterations
(a 4-byte all-reduce enqueued on the device, no host read-back), which makes the assembled frame equal to a
single-GPU render of the whole image; schedule="band" skips it.
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


class _Bookkeeping:
    """What bench.py reads from either renderer, and the speculative loop length both can use."""

    def _update_loop_hint(self):
        from . import fused
        m = self.scene.model
        cur = fused.loop_counters(m)                # synchronises; finish() is a synchronisation point anyway
        unfinished = fused.unfinished_frames(m)
        prev, self._counters = self._counters, (cur, unfinished)
        if cur is None or not self._frames_since_finish:
            return
        if prev is not None and unfinished != prev[1]:
            fused.set_loop_hint(m, None)
            self._frames_since_finish = 0
            raise LoopHintTooSmall(f"{unfinished - prev[1]} frame(s) needed more loop iterations than the hint")
        iters = ((cur[0] - (prev[0][0] if prev else 0)) & 0xFFFFFFFF) / self._frames_since_finish
        fused.set_loop_hint(m, int(-(-iters // 1)) + 1)     # one spare iteration: a no-op iteration is 3 launches, ~10 us
        self._frames_since_finish = 0


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
                self._render_for_count(s)
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



class LoopHintTooSmall(RuntimeError):
    """A batch rendered with a speculative loop length contained frames that needed more iterations."""


class FrameParallelRenderer(_Bookkeeping):
    def __init__(self, scene, rank=0, world=1, dist=None, gather=True, speculate_loop=False, gather_every=8, audio_batch=0):
        """speculate_loop (fused engine): after the first finish() the renderer knows how many loop iterations the
        stream's frames take (device counters) and enqueues that many + 2 per frame instead of max_steps, skipping
        the no-op launches behind them; the device flags any frame for which that was not enough and finish() raises
        LoopHintTooSmall so the caller can render the batch again (never observed on a continuous pose stream)."""
        self.scene, self.rank, self.world, self.dist = scene, rank, world, dist
        self.gather = gather and dist is not None and world > 1
        # Finished frames leave in batches of `gather_every` (one collective per batch: a frame takes ~1 ms, and a collective
        # per millisecond would keep an RCCL kernel waiting for CUs that the persistent network kernels occupy entirely).
        # finish() flushes a partial batch, so every rank issues the same sequence of collectives.
        self.gather_every = max(1, int(gather_every))
        self._batch = []
        self.pending = []
        self.frames_u8 = None
        self.speculate_loop = speculate_loop and getattr(scene.opt, "engine", "ops") == "fused"
        self._frames_since_finish = 0
        self._counters = None
        # audio_batch = K > 0 (fused engine, a resident feature stream): the audio codes, their smoothing recurrence and the
        # per-frame bias blocks of this rank's next K frames are computed together -- 4 launches per K frames instead of 4
        # per frame; the frames then start at the ray prologue.  Same numbers: the recurrence runs over every global frame
        # in order (the other ranks' included), exactly as _advance_audio does frame by frame.
        self.audio_batch = max(0, int(audio_batch))
        self._ab = None                  # (first step, smoothed codes [K, dim], bias blocks [K, 192])
        self._ab_next = None             # the step the prepared batch expects next

    # -- audio state ------------------------------------------------------------------------------
    def _advance_audio(self, frames):
        """Fold the raw audio codes of `frames` (ascending) into model.enc_a, as rendering them would."""
        m = self.scene.model
        if not frames or not m.smooth_lips:
            return
        n = self.scene.n_frames
        feats = self.scene.aud_features
        if (getattr(m, "fused_audio_enabled", None) is not None and self.scene.opt.att == 2 and n >= 8 and feats.is_cuda
                and feats.dtype == torch.float32 and m.fused_audio_enabled()):
            # fused engine: the codes of all skipped frames in ONE launch (windows cut on the device), then the recurrence
            from . import audio
            audio.smooth_(m, audio.encode_stream(m, feats, frames[0] % n, len(frames)))
            return
        for g in frames:  # tiny networks; kept sequential so the arithmetic equals the reference's per-frame path
            enc = m.encode_audio(get_audio_features(self.scene.aud_features, self.scene.opt.att, g % n))
            m.enc_a = enc if m.enc_a is None else 0.35 * m.enc_a + (1 - 0.35) * enc

    def _audio_batch_ok(self):
        m, sc = self.scene.model, self.scene
        return (self.audio_batch > 0 and getattr(m, "fused_audio_enabled", None) is not None and m.smooth_lips and sc.opt.att == 2
                and sc.n_frames >= 8 and sc.aud_features.is_cuda and sc.aud_features.dtype == torch.float32 and m.fused_audio_enabled())

    def _prepare_audio(self, step):
        """Codes + bias blocks of this rank's frames at steps step .. step + K - 1 (one stream encode over every global frame
        from the first one not folded yet, one recurrence, one bias launch)."""
        from . import audio, fused
        m, sc, K = self.scene.model, self.scene, self.audio_batch
        mine = [frame_of(step + j, self.rank, self.world) for j in range(K)]
        skipped = skipped_frames(step, self.rank, self.world)
        g0 = skipped[0] if skipped else mine[0]
        codes = audio.encode_stream(m, sc.aud_features, g0 % sc.n_frames, mine[-1] - g0 + 1)
        states = audio.smooth_seq_(m, codes)
        pick = torch.tensor([g - g0 for g in mine], dtype=torch.long, device=states.device)
        states = states.index_select(0, pick) if len(mine) != states.shape[0] else states
        code = m.individual_codes[0] if m.individual_dim > 0 else None
        self._ab = (step, states, fused.frame_bias_batch(m, states, sc.eye, code))
        self._ab_next = step

    # -- one step = one frame on this rank ----------------------------------------------------------
    def step(self, step):
        g = frame_of(step, self.rank, self.world)
        if self._audio_batch_ok():
            if self._ab is None or step != self._ab_next or step - self._ab[0] >= self._ab[1].shape[0]:
                self._prepare_audio(step)
            j = step - self._ab[0]
            self._ab_next = step + 1
            out = self.scene.render(g, want_u8=True, audio_code=(self._ab[1][j:j + 1], self._ab[2][j]))
            return self._after_render(out)
        self._advance_audio(skipped_frames(step, self.rank, self.world))
        try:
            out = self.scene.render(g, want_u8=True)
        except TypeError:                      # a scene object without the want_u8 option
            out = self.scene.render(g)
        return self._after_render(out)

    def _after_render(self, out):
        if "image_u8" in out:                  # quantised by the blend kernel itself
            u8 = out["image_u8"].reshape(self.scene.H, self.scene.W, 3)
        else:
            u8 = (out["image"].reshape(self.scene.H, self.scene.W, 3) * 255).to(torch.uint8)
        if self.gather:
            self._batch.append(u8)
            if len(self._batch) >= self.gather_every:
                self._flush()
        self.last_frame = u8
        self._frames_since_finish += 1
        return u8

    def _flush(self):
        """One collective for the frames collected since the last one: [world, n, H, W, 3] uint8."""
        if not self._batch:
            return
        frames = torch.stack(self._batch, 0)
        self._batch = []
        if self.dist.get_backend() == "nccl":
            # async: RCCL runs the gather on its own stream, overlapping the next frames' kernels
            buf = torch.empty((self.world,) + tuple(frames.shape), dtype=torch.uint8, device=frames.device)
            work = self.dist.all_gather_into_tensor(buf, frames, async_op=True)
        else:  # gloo gathers host tensors (CPU tests, multi-rank rehearsals on a one-GPU box)
            frames = frames.cpu()
            buf = torch.empty((self.world,) + tuple(frames.shape), dtype=torch.uint8)
            work = self.dist.all_gather(list(buf.unbind(0)), frames, async_op=True)
        self.pending.append((work, buf, frames))

    def _render_for_count(self, step):
        self.scene.render(frame_of(step, self.rank, self.world))

    def finish(self):
        """Wait for every outstanding gather; returns the gathered [world, H, W, 3] uint8 stacks in step order."""
        if self.gather:
            self._flush()
        done = []
        for work, buf, _ in self.pending:
            work.wait()
            done.extend(buf[:, i] for i in range(buf.shape[1]))   # per step: [world, H, W, 3]
        if done:
            self.frames_u8 = done[-1]
        self.pending = []
        if self.speculate_loop:
            self._update_loop_hint()
        return done


def stripe_rows(H, rank, world, band=8):
    """Image rows owned by `rank`: bands of `band` consecutive rows dealt round-robin over the ranks."""
    rows = torch.arange(H)
    return rows[(rows // band) % world == rank]


class TileParallelRenderer(_Bookkeeping):
    """BASELINE config 4 (a single 1024^2 frame over 8 GPUs).  `step(i)` renders this rank's rows of global frame i
    and starts the gather; `finish()` returns the assembled [H, W, 3] uint8 frames (identical on every rank)."""

    def __init__(self, scene, rank=0, world=1, dist=None, band=8, schedule=None, speculate_loop=False):
        """schedule="frame": the ranks agree on the whole frame's step schedule (one 4-byte all-reduce per loop
        iteration, enqueued on the device; fused engine only) so the assembled image IS the whole-frame render;
        schedule="band": no collective inside the loop, each band follows the reference's policy for its own rays.
        Default: "frame" with the fused engine, "band" otherwise."""
        self.scene, self.rank, self.world, self.dist, self.band = scene, rank, world, dist, band
        fused_engine = getattr(getattr(scene, "opt", None), "engine", "ops") == "fused"
        if schedule is None:
            schedule = "frame" if fused_engine else "band"
        if schedule == "frame" and dist is not None and world > 1:
            if not fused_engine:
                raise RuntimeError("schedule='frame' needs the fused engine (device-resident loop state)")
            scene.model.shard_schedule = (dist, scene.H * scene.W)
        self.schedule = schedule
        self.speculate_loop = speculate_loop and fused_engine      # as in FrameParallelRenderer
        self._frames_since_finish = 0
        self._counters = None
        H, W = scene.H, scene.W
        self.rows = [stripe_rows(H, r, world, band) for r in range(world)]
        self.n_max = max(len(r) for r in self.rows)
        mine = self.rows[rank].to(scene.device)
        self.pix = (mine[:, None] * W + torch.arange(W, device=scene.device)[None, :]).reshape(-1)   # row-major ray ids
        self.pending = []
        self._static = None     # this rank's slice of the per-pixel inputs that do not change with the frame
        self._rays = {}

    def _inputs(self, i):
        sc, px = self.scene, self.pix
        f = sc.frame(i)
        if self._static is None:
            self._static = (f["bg_coords"][:, px].contiguous(), f["bg_color"][:, px].contiguous())
        key = i % getattr(sc, "n_frames", 1 << 30)
        if key not in self._rays:
            self._rays[key] = (f["rays_o"][:, px].contiguous(), f["rays_d"][:, px].contiguous())
        return f, self._rays[key], self._static

    def render_local(self, i):
        """This rank's pixels of frame i through the scene's model: [n_rows, W, 3] uint8."""
        sc = self.scene
        f, (rays_o, rays_d), (bg_coords, bg_color) = self._inputs(i)
        out = sc.model.render(rays_o, rays_d, f["auds"], bg_coords, f["poses"], eye=f["eye"], index=f["index"],
                              bg_color=bg_color, **sc.render_kwargs())
        return (out["image"].reshape(-1, sc.W, 3) * 255).to(torch.uint8)

    def _render_for_count(self, step):
        self.render_local(step)

    def step(self, i):
        u8 = self.render_local(i)
        self._frames_since_finish += 1
        if self.dist is None or self.world == 1:
            self.pending.append((None, None, u8))
            return u8
        send = u8
        if u8.shape[0] < self.n_max:                     # ragged last band: pad so one fixed-size gather does it
            send = torch.cat([u8, u8.new_zeros((self.n_max - u8.shape[0],) + tuple(u8.shape[1:]))])
        nccl = self.dist.get_backend() == "nccl"
        if not nccl:
            send = send.cpu()                            # gloo gathers host tensors (CPU tests, 1-GPU rehearsals)
        buf = torch.empty((self.world,) + tuple(send.shape), dtype=torch.uint8, device=send.device)
        if nccl:
            work = self.dist.all_gather_into_tensor(buf, send.contiguous(), async_op=True)
        else:
            work = self.dist.all_gather(list(buf.unbind(0)), send.contiguous(), async_op=True)
        self.pending.append((work, buf, send))
        return u8

    def assemble(self, buf):
        sc = self.scene
        frame = torch.empty((sc.H, sc.W, 3), dtype=torch.uint8, device=buf.device)
        for r in range(self.world):
            rows = self.rows[r].to(buf.device)
            frame[rows] = buf[r, :len(rows)]
        return frame

    def finish(self):
        frames = []
        for work, buf, u8 in self.pending:
            if work is None:
                frames.append(self.assemble(u8[None]))
            else:
                work.wait()
                frames.append(self.assemble(buf))
        self.pending = []
        if self.speculate_loop:
            self._update_loop_hint()
        return frames

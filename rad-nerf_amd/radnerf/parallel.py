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
the pixels.  Interleaved bands are statistically alike, so a band's own policy (N_band // alive_band) almost always
picks the whole frame's n_step -- but not provably.  schedule="verify" (default with the fused engine) therefore lets
every rank follow its band's policy with NO collective inside the loop, appends the 17 live-ray counts the loop went
through to the rows it sends into the frame's gather (68 bytes), and checks on every rank, when the gather is
consumed, that the band-local schedules were the whole frame's; a frame for which they were not (rare) is rendered
again with schedule="frame" semantics -- the ranks sum their live counts between loop iterations (a 4-byte
all-reduce enqueued on the device, no host read-back).  Either way the assembled frame equals a single-GPU render of
the whole image and the common case costs ONE collective per frame, the gather.  schedule="frame" always takes the
exact path; schedule="band" never checks.
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
        stalled = fused.stalled_workgroups(m)
        if stalled != getattr(self, "_stalled", 0):     # the in-launch barrier of the one-launch loop step timed out: never expected
            self._stalled = stalled
            m.opt.loop_launch = "split"
            raise RuntimeError(f"{stalled} workgroup(s) gave up at the loop step's grid barrier (RN_LOOP_COOP); the frames since the "
                               "last finish() are invalid -- switched to opt.loop_launch = 'split'")
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
    def measured_traffic(key, field="hbm_bytes_per_launch"):
        """HBM bytes per launch from a committed rocprofv3 --pmc pass (profiles/traffic.json), else None."""
        p = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles",
                         "traffic.json")
        if os.path.exists(p):
            try:
                return json.load(open(p)).get(key, {}).get(field)
            except Exception:
                return None
        return None



class LoopHintTooSmall(RuntimeError):
    """A batch rendered with a speculative loop length contained frames that needed more iterations."""


class FrameParallelRenderer(_Bookkeeping):
    def __init__(self, scene, rank=0, world=1, dist=None, gather=True, speculate_loop=False, gather_every=8, audio_batch=0,
                 gather_to="rank0", streams=1):
        """speculate_loop (fused engine): after the first finish() the renderer knows how many loop iterations the
        stream's frames take (device counters) and enqueues that many + 2 per frame instead of max_steps, skipping
        the no-op launches behind them; the device flags any frame for which that was not enough and finish() raises
        LoopHintTooSmall so the caller can render the batch again (never observed on a continuous pose stream)."""
        self.scene, self.rank, self.world, self.dist = scene, rank, world, dist
        self.gather = gather and dist is not None and world > 1
        # gather_to="rank0" (default): finished frames go to rank 0 only (the process that encodes / shows the video): 1/world of
        # the bytes an all_gather moves over the xGMI links; "all": every rank receives every frame.
        if gather_to not in ("rank0", "all"):
            raise ValueError("gather_to must be 'rank0' or 'all'")
        self.gather_to = gather_to
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
        # streams = 2 (fused engine): consecutive frames alternate between two HIP streams, so one frame's small latency-bound
        # kernels (ray prologue, compositor, compaction + march, torso) run beside the other frame's network kernel instead of
        # leaving most CUs idle; the persistent network kernels themselves still take turns (two of their 96 KB LDS images do not
        # fit one CU).  Frames stay independent: each stream has its own loop state, scratch and ray buffers; the audio codes of
        # a batch are computed on the default stream and both render streams wait for that event.  Throughput mode -- a single
        # frame's latency gets longer.
        self.n_streams = max(1, int(streams)) if getattr(scene.opt, "engine", "ops") == "fused" and scene.device.type == "cuda" else 1
        self._streams = [torch.cuda.Stream(device=scene.device) for _ in range(self.n_streams)] if self.n_streams > 1 else None
        if self.n_streams > 3:       # the one-launch loop step needs its workgroups co-resident: at most three such streams (radnerf_fused.h)
            scene.opt.loop_launch = "split"
        self._audio_ready = None
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
        if self._streams:
            self._audio_ready = torch.cuda.Event()
            self._audio_ready.record()

    # -- one step = one frame on this rank ----------------------------------------------------------
    def render_stream(self, step):
        """The stream step `step`'s frame is enqueued on (the current stream unless streams > 1)."""
        return self._streams[step % self.n_streams] if self._streams else torch.cuda.current_stream(self.scene.device)

    def step(self, step):
        if self._streams:
            if not self._audio_batch_ok():
                raise RuntimeError("streams > 1 needs audio_batch > 0 (the per-frame audio recurrence would serialise the frames)")
            if self._ab is None or step != self._ab_next or step - self._ab[0] >= self._ab[1].shape[0]:
                for st in self._streams:                       # the batch buffers about to be replaced may still be read
                    torch.cuda.current_stream(self.scene.device).wait_stream(st)
                self._prepare_audio(step)
            side = self._streams[step % self.n_streams]
            side.wait_event(self._audio_ready)
            with torch.cuda.stream(side):
                u8 = self._step(step, flush=False)             # only rendering (and the batch append) happens on the side stream
            if self.gather and len(self._batch) >= self.gather_every:
                self._flush_from_default()
            return u8
        return self._step(step)

    def _flush_from_default(self):
        """The collective of a batch whose frames were rendered on side streams: issued on the default stream once EVERY side
        stream has finished its frames of the batch; the frames were allocated on the side streams, so the caching allocator is
        told that the default stream (and through it the collective) reads them."""
        cur = torch.cuda.current_stream(self.scene.device)
        for st in self._streams:
            cur.wait_stream(st)
        for u8 in self._batch:
            u8.record_stream(cur)
        self._flush()

    def _step(self, step, flush=True):
        g = frame_of(step, self.rank, self.world)
        if self._audio_batch_ok():
            if self._ab is None or step != self._ab_next or step - self._ab[0] >= self._ab[1].shape[0]:
                self._prepare_audio(step)
            j = step - self._ab[0]
            self._ab_next = step + 1
            out = self.scene.render(g, want_u8=True, audio_code=(self._ab[1][j:j + 1], self._ab[2][j]))
            return self._after_render(out, flush)
        self._advance_audio(skipped_frames(step, self.rank, self.world))
        try:
            out = self.scene.render(g, want_u8=True)
        except TypeError:                      # a scene object without the want_u8 option
            out = self.scene.render(g)
        return self._after_render(out, flush)

    def _after_render(self, out, flush=True):
        if "image_u8" in out:                  # quantised by the blend kernel itself
            u8 = out["image_u8"].reshape(self.scene.H, self.scene.W, 3)
        else:
            u8 = (out["image"].reshape(self.scene.H, self.scene.W, 3) * 255).to(torch.uint8)
        if self.gather:
            self._batch.append(u8)
            if flush and len(self._batch) >= self.gather_every:
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
        nccl = self.dist.get_backend() == "nccl"
        if not nccl:
            frames = frames.cpu()          # gloo moves host tensors (CPU tests, multi-rank rehearsals on a one-GPU box)
        shape = (self.world,) + tuple(frames.shape)
        if self.gather_to == "rank0":
            buf = torch.empty(shape, dtype=torch.uint8, device=frames.device) if self.rank == 0 else None
            work = self.dist.gather(frames, list(buf.unbind(0)) if self.rank == 0 else None, dst=0, async_op=True)
        else:
            buf = torch.empty(shape, dtype=torch.uint8, device=frames.device)
            if nccl:                       # async: RCCL runs the collective on its own stream, overlapping the next frames' kernels
                work = self.dist.all_gather_into_tensor(buf, frames, async_op=True)
            else:
                work = self.dist.all_gather(list(buf.unbind(0)), frames, async_op=True)
        self.pending.append((work, buf, frames))

    def _render_for_count(self, step):
        self.scene.render(frame_of(step, self.rank, self.world))

    def finish(self):
        """Wait for every outstanding gather; returns the gathered [world, H, W, 3] uint8 stacks in step order (on rank 0
        only when gather_to == "rank0": the other ranks get an empty list)."""
        if self.gather:
            if self._streams and self._batch:
                self._flush_from_default()
            else:
                self._flush()
        done = []
        for work, buf, _ in self.pending:
            work.wait()
            if buf is not None:
                done.extend(buf[:, i] for i in range(buf.shape[1]))   # per step: [world, H, W, 3]
        if done:
            self.frames_u8 = done[-1]
        self.pending = []
        if self._streams:
            for st in self._streams:
                st.synchronize()
        if self.speculate_loop:
            self._update_loop_hint()
        return done


def band_schedule_is_frame_schedule(n_rays, histories, max_steps):
    """Did ranks that each followed n_step = max(min(N_r // alive_r, 8), 1) for their own band walk the whole frame's
    schedule max(min(sum N_r // sum alive_r, 8), 1)?  histories[r][i] = live rays of rank r entering loop iteration i (0
    once its loop is over).  Checked iteration by iteration: while every earlier iteration agreed the counts ARE the whole
    frame's, so their sums are what a single-GPU render would have seen."""
    total_n = sum(n_rays)
    step = 0
    for i in range(len(histories[0])):
        alive = [int(h[i]) for h in histories]
        total = sum(alive)
        if total == 0 or step >= max_steps:
            return True
        want = max(min(total_n // total, 8), 1)
        for n_r, a in zip(n_rays, alive):
            if a > 0 and max(min(n_r // a, 8), 1) != want:
                return False
        step += want
    return True


def stripe_rows(H, rank, world, band=8):
    """Image rows owned by `rank`: bands of `band` consecutive rows dealt round-robin over the ranks."""
    rows = torch.arange(H)
    return rows[(rows // band) % world == rank]


class TileParallelRenderer(_Bookkeeping):
    """BASELINE config 4 (a single 1024^2 frame over 8 GPUs).  `step(i)` renders this rank's rows of global frame i
    and starts the gather; `finish()` returns the assembled [H, W, 3] uint8 frames (on rank 0; on every rank with gather_to="all")."""

    def __init__(self, scene, rank=0, world=1, dist=None, band=8, schedule=None, speculate_loop=False, gather_to="rank0",
                 audio_batch=8):
        """gather_to="rank0" (default): the finished rows go to rank 0 only (the process that shows / encodes the frame) --
        1/world of the bytes an all_gather moves; finish() then returns the assembled frames on rank 0 and an empty list
        elsewhere, and the schedule verdicts of schedule="verify" travel back in ONE small broadcast per finish().  "all":
        every rank assembles every frame (round 2's behaviour).
        audio_batch = K > 0 (fused engine, resident feature stream): the audio codes, their smoothing recurrence and the bias
        blocks of the next K frames in four launches instead of four per frame (every rank renders every frame, so the batch is
        K consecutive global frames).
        schedule="verify" (default with the fused engine): band-local step policy, no collective inside the loop, the
        schedule checked from counts that ride in the frame's gather, the rare mismatching frame rendered again exactly;
        "frame": every frame with the whole frame's schedule (one 4-byte all-reduce per loop iteration, enqueued on the
        device); "band": band-local policy, never checked (each band is then the reference applied to that band's rays).
        Default without the fused engine: "band"."""
        self.scene, self.rank, self.world, self.dist, self.band = scene, rank, world, dist, band
        fused_engine = getattr(getattr(scene, "opt", None), "engine", "ops") == "fused"
        if schedule is None:
            schedule = "verify" if fused_engine else "band"
        if schedule in ("frame", "verify") and dist is not None and world > 1 and not fused_engine:
            raise RuntimeError(f"schedule='{schedule}' needs the fused engine (device-resident loop state)")
        if schedule == "frame" and dist is not None and world > 1:
            scene.model.shard_schedule = (dist, scene.H * scene.W)
        self.schedule = schedule
        self.redone = 0                                            # frames rendered again because the band schedules disagreed
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
        if gather_to not in ("rank0", "all"):
            raise ValueError("gather_to must be 'rank0' or 'all'")
        self.gather_to = gather_to
        self.audio_batch = max(0, int(audio_batch)) if fused_engine else 0
        self._ab = None         # (first frame, smoothed codes [K, dim], bias blocks [K, 192])

    def _audio_batch_ok(self):
        m, sc = self.scene.model, self.scene
        return (self.audio_batch > 0 and getattr(m, "fused_audio_enabled", None) is not None and m.smooth_lips and sc.opt.att == 2
                and getattr(sc, "n_frames", 0) >= 8 and sc.aud_features.is_cuda and sc.aud_features.dtype == torch.float32
                and not m.training and m.fused_audio_enabled())

    def _audio_code(self, i):
        """(smoothed code [1, dim], bias block [192]) of frame i from a batch computed for frames i0 .. i0 + K - 1, or None
        (per-frame audio kernels).  Frames must come in order (a stream); a jump starts a new batch from the current state."""
        if not self._audio_batch_ok():
            return None
        from . import audio, fused
        m, sc = self.scene.model, self.scene
        if self._ab is None or not (self._ab[0] <= i < self._ab[0] + self._ab[1].shape[0]) or i != self._ab[3]:
            codes = audio.encode_stream(m, sc.aud_features, i % sc.n_frames, self.audio_batch)
            states = audio.smooth_seq_(m, codes)
            code = m.individual_codes[0] if m.individual_dim > 0 else None
            self._ab = [i, states, fused.frame_bias_batch(m, states, sc.eye, code), i]
        j = i - self._ab[0]
        self._ab[3] = i + 1
        return self._ab[1][j:j + 1], self._ab[2][j]

    def _inputs(self, i):
        sc, px = self.scene, self.pix
        f = sc.frame(i)
        if self._static is None:
            self._static = (f["bg_coords"][:, px].contiguous(), f["bg_color"][:, px].contiguous())
        key = i % getattr(sc, "n_frames", 1 << 30)
        if key not in self._rays:
            self._rays[key] = (f["rays_o"][:, px].contiguous(), f["rays_d"][:, px].contiguous())
        return f, self._rays[key], self._static

    def render_local(self, i, audio_code=None):
        """This rank's pixels of frame i through the scene's model: [n_rows, W, 3] uint8."""
        sc = self.scene
        f, (rays_o, rays_d), (bg_coords, bg_color) = self._inputs(i)
        kw = dict(sc.render_kwargs())
        if audio_code is None:
            audio_code = self._audio_code(i)
        if audio_code is not None:
            kw["audio_code"] = audio_code
        fused_engine = getattr(getattr(sc, "opt", None), "engine", "ops") == "fused"
        if fused_engine:
            kw["want_u8"] = True                 # quantised by the frame's epilogue kernel, as the frame path does
        out = sc.model.render(rays_o, rays_d, f["auds"], bg_coords, f["poses"], eye=f["eye"], index=f["index"],
                              bg_color=bg_color, **kw)
        self._last_code = audio_code[0] if audio_code is not None else getattr(sc.model, "enc_a", None)
        if "image_u8" in out:
            return out["image_u8"].reshape(-1, sc.W, 3)
        return (out["image"].reshape(-1, sc.W, 3) * 255).to(torch.uint8)

    def _render_for_count(self, step):
        self.render_local(step)

    _HIST = 17          # live-ray counts entering iterations 0 .. 16 (max_steps = 16 in every BASELINE config)

    def _gather(self, payload, to_all=False):
        """One collective: every rank's payload (uint8, same length) to rank 0 (gather_to="rank0") or to every rank.
        Returns (work, buf [world, len] or None on the ranks that receive nothing, payload)."""
        nccl = self.dist.get_backend() == "nccl"
        if not nccl:
            payload = payload.cpu()                      # gloo gathers host tensors (CPU tests, 1-GPU rehearsals)
        payload = payload.contiguous()
        if self.gather_to == "rank0" and not to_all:
            buf = torch.empty((self.world, payload.numel()), dtype=torch.uint8, device=payload.device) if self.rank == 0 else None
            work = self.dist.gather(payload, list(buf.unbind(0)) if self.rank == 0 else None, dst=0, async_op=True)
            return work, buf, payload
        buf = torch.empty((self.world, payload.numel()), dtype=torch.uint8, device=payload.device)
        if nccl:
            work = self.dist.all_gather_into_tensor(buf, payload, async_op=True)
        else:
            work = self.dist.all_gather(list(buf.unbind(0)), payload, async_op=True)
        return work, buf, payload

    def _payload(self, u8, verify):
        rows = u8
        if u8.shape[0] < self.n_max:                     # ragged last band: pad so one fixed-size gather does it
            rows = torch.cat([u8, u8.new_zeros((self.n_max - u8.shape[0],) + tuple(u8.shape[1:]))])
        parts = [rows.reshape(-1)]
        if verify:
            from . import fused
            parts.append(fused.loop_history(self.scene.model, self._HIST).clone().view(torch.uint8))
        return torch.cat(parts) if len(parts) > 1 else parts[0]

    def step(self, i):
        u8 = self.render_local(i)
        self._frames_since_finish += 1
        if self.dist is None or self.world == 1:
            self.pending.append(dict(frame=i, work=None, buf=None, keep=u8))
            return u8
        verify = self.schedule == "verify"
        work, buf, sent = self._gather(self._payload(u8, verify))
        enc_a = getattr(self, "_last_code", None)
        self.pending.append(dict(frame=i, work=work, buf=buf, keep=sent, verify=verify,
                                 enc_a=enc_a.clone() if (verify and torch.is_tensor(enc_a)) else None))
        return u8

    def _rows_of(self, buf):
        n = self.n_max * self.scene.W * 3
        return buf[:, :n].reshape(self.world, self.n_max, self.scene.W, 3)

    def assemble(self, rows):
        sc = self.scene
        frame = torch.empty((sc.H, sc.W, 3), dtype=torch.uint8, device=rows.device)
        for r in range(self.world):
            own = self.rows[r].to(rows.device)
            frame[own] = rows[r, :len(own)]
        return frame

    def _redo_exact(self, p):
        """Frame p["frame"] again with the whole frame's schedule (per-iteration all-reduce), with the audio code it had."""
        m = self.scene.model
        m.shard_schedule = (self.dist, self.scene.H * self.scene.W)
        try:
            u8 = self.render_local(p["frame"], audio_code=(p["enc_a"], None) if p["enc_a"] is not None else None)
            work, buf, _ = self._gather(self._payload(u8, False))
            work.wait()
        finally:
            m.shard_schedule = None
        self.redone += 1
        return self.assemble(self._rows_of(buf)) if buf is not None else None

    def finish(self):
        """Wait for the outstanding gathers; returns the assembled frames in step order (gather_to="rank0": on rank 0, an empty
        list elsewhere)."""
        frames = []
        max_steps = int(getattr(getattr(self.scene, "opt", None), "max_steps", 16))
        n = self.n_max * self.scene.W * 3
        n_rays = [len(r) * self.scene.W for r in self.rows]
        for p in self.pending:
            if p["work"] is not None:
                p["work"].wait()
        # schedule verdicts: a rank that holds every band's loop counts (all ranks with gather_to="all", rank 0 otherwise)
        # decides; with gather_to="rank0" the verdicts of all pending frames go back to the ranks in ONE broadcast
        bad = []
        for p in self.pending:
            ok = True
            if p["work"] is not None and p.get("verify") and p["buf"] is not None:
                hist = p["buf"][:, n:n + 4 * self._HIST].cpu().contiguous().view(torch.int32).reshape(self.world, self._HIST).tolist()
                ok = band_schedule_is_frame_schedule(n_rays, hist, max_steps)
            bad.append(0 if ok else 1)
        if self.gather_to == "rank0" and self.dist is not None and self.world > 1 and any(p.get("verify") for p in self.pending):
            nccl = self.dist.get_backend() == "nccl"
            flags = torch.tensor(bad, dtype=torch.uint8, device=self.scene.device if nccl else "cpu")
            self.dist.broadcast(flags, src=0)
            bad = flags.cpu().tolist()
        for p, redo in zip(self.pending, bad):
            if p["work"] is None:
                frames.append(self.assemble(p["keep"][None]))
            elif redo:
                frame = self._redo_exact(p)                  # every rank takes part (it holds an all-reduce per loop iteration)
                if frame is not None:
                    frames.append(frame)
            elif p["buf"] is not None:
                frames.append(self.assemble(self._rows_of(p["buf"])))
        self.pending = []
        if self.speculate_loop:
            self._update_loop_hint()
        return frames

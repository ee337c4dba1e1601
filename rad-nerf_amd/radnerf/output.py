"""Output side of the render path: hand finished frames to the host and time them.

The reference turns a frame into `(pred * 255).astype(np.uint8)` on the host after a blocking `.cpu()` (nerf/utils.py:955-971)
and its GUI brackets a frame with two CUDA events (nerf/gui.py:170-227).  Here the frame is already uint8 on the device (the
blend kernel writes it), so

  * FrameSink copies it device -> pinned host memory on a SIDE stream, ordered behind the frame by an event, into a ring of
    page-locked slots: the render stream never waits for PCIe (786 KB per 512^2 frame, ~13 us at 63 GB/s), the host picks
    frames up when their copy event has fired;
  * FrameTimer brackets each frame with HIP events on the render stream and reports per-frame latencies (mean / p50 / p95).

PyTorch is plumbing here (streams, events, pinned memory); there is no kernel in this file.
"""
import torch


class FrameSink:
    def __init__(self, height, width, slots=8, device=None):
        self.device = torch.device(device if device is not None else "cuda")
        self.ring = torch.empty(slots, height, width, 3, dtype=torch.uint8).pin_memory()
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._done = [None] * slots            # copy-finished event of each slot
        self._tag = [None] * slots
        self._head = self._tail = 0            # next slot to fill / next slot to hand out
        self.slots = slots

    def push(self, frame_u8, tag=None):
        """Enqueue the copy of a finished uint8 frame [H, W, 3] (device).  Blocks only when every slot holds a frame the host
        has not taken yet (then it waits for the oldest one's consumer: call pop())."""
        if self._head - self._tail >= self.slots:
            raise RuntimeError("FrameSink is full: pop() frames before pushing more")
        slot = self._head % self.slots
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))          # the frame is complete at this point of the render stream
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            self.ring[slot].copy_(frame_u8.reshape(self.ring.shape[1:]), non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        frame_u8.record_stream(self.copy_stream)                      # the allocator must not recycle it before the copy ran
        self._done[slot], self._tag[slot] = done, tag
        self._head += 1

    def ready(self):
        """Frames whose copy has landed (non-blocking)."""
        n = 0
        while self._tail + n < self._head and self._done[(self._tail + n) % self.slots].query():
            n += 1
        return n

    def pop(self, wait=True):
        """Oldest frame as (tag, numpy uint8 [H, W, 3] view of its pinned slot) -- valid until `slots` more frames were pushed --
        or None when nothing is pending (or, with wait=False, not landed yet)."""
        if self._tail == self._head:
            return None
        slot = self._tail % self.slots
        if not self._done[slot].query():
            if not wait:
                return None
            self._done[slot].synchronize()
        self._tail += 1
        return self._tag[slot], self.ring[slot].numpy()


class FrameTimer:
    """Per-frame latency from HIP events on the render stream (the GUI's method, nerf/gui.py:174-202)."""

    def __init__(self):
        self._pairs = []
        self._open = None

    def start(self):
        self._open = torch.cuda.Event(enable_timing=True)
        self._open.record()

    def stop(self):
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        self._pairs.append((self._open, end))
        self._open = None

    def summary(self):
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in self._pairs)
        if not ms:
            return None
        return dict(frames=len(ms), mean=sum(ms) / len(ms), p50=ms[len(ms) // 2], p95=ms[min(len(ms) - 1, int(0.95 * len(ms)))],
                    fps=1e3 * len(ms) / sum(ms))

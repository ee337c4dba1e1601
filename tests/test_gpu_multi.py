"""Multi-process renders on the GPU box (BASELINE configs 3 and 4): the frames two ranks gather equal a
single-process render of the same stream.

  * one GPU (the builder's box): two gloo ranks share cuda:0 -- the protocol (frame striding with the audio EMA folded through
    the skipped frames, interleaved bands with verified step schedules, gather to rank 0, side streams) runs end to end;
  * two or more GPUs (the driver's node): the same checks over RCCL (`nccl` backend), one rank per GPU -- skipped where
    torch.cuda.device_count() < 2, so the first multi-GPU run of this suite exercises the collectives on xGMI.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, nproc, *args):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", tool), *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("streams", [1, 2])
def test_two_gloo_ranks_frame_parallel_equal_the_sequential_stream(hiplib, streams):
    """streams = 2: frames alternate between two HIP streams per rank; the collective of a batch is issued on the default
    stream after every side stream that rendered one of its frames (the gather must not ship unfinished frames)."""
    out = _run("frame_check.py", 2, "--size", "64", "--steps", "5", "--streams", str(streams))
    assert "identical: True" in out and "backend=gloo world=2" in out


nccl = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")


@nccl
@pytest.mark.parametrize("streams", [1, 2])
def test_two_nccl_ranks_frame_parallel_equal_the_sequential_stream(hiplib, streams):
    out = _run("frame_check.py", 2, "--backend", "nccl", "--size", "128", "--steps", "9", "--streams", str(streams), "--gather-every", "4")
    assert "identical: True" in out and "backend=nccl world=2" in out


@nccl
@pytest.mark.parametrize("schedule", ["verify", "frame"])
@pytest.mark.parametrize("gather_to", ["rank0", "all"])
def test_two_nccl_ranks_tile_parallel_equal_the_whole_frame(hiplib, schedule, gather_to):
    out = _run("tile_check.py", 2, "--backend", "nccl", "--size", "128", "--frames", "3", "--schedule", schedule, "--gather-to", gather_to)
    assert "frame 2: max |d|" in out and "backend=nccl world=2" in out

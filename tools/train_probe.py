"""Where a training step's time goes on the host/graph side: pure hipGraph replay (no occupancy refresh, no recapture), the cost
of one capture, and the eager step -- tools/gpu_train_profile.sh has the kernel view.

    python tools/train_probe.py [--grid hash19] [--steps 100]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="hash19")
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    from bench import GRIDS
    from radnerf.scene import SyntheticScene, default_opt
    from radnerf.train import GraphedTrainer, SyntheticTrainStream
    scene = SyntheticScene(H=512, W=512, n_frames=8, device="cuda", opt=default_opt(engine="ops", torso=False, smooth_lips=False, **GRIDS[args.grid]))
    stream = SyntheticTrainStream(scene, n_rays=4096)
    tr = GraphedTrainer(scene.model, scene.opt)
    for _ in range(34):
        tr.step(stream.batch())
    torch.cuda.synchronize()
    tr.update_extra_interval = 0
    out = {}
    for name, K in (("replay_ms", args.steps), ("replay_ms_again", args.steps)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            tr.step(stream.batch())
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) * 1e3 / K
    # GPU time of one replay alone (events around graph.replay())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        a.record()
        tr._graph.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    out["graph_gpu_ms"] = sorted(ts)[len(ts) // 2]
    # batch generation alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        stream.batch()
    torch.cuda.synchronize()
    out["batch_ms"] = (time.perf_counter() - t0) * 1e3 / 50
    caps = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr._capture(stream.batch(), tr._key)
        torch.cuda.synchronize()
        caps.append((time.perf_counter() - t0) * 1e3)
    out["capture_ms"] = caps
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            scene.model.update_extra_state()
        torch.cuda.synchronize()
        out["refresh_ms_wall"] = (time.perf_counter() - t0) * 1e3 / 4
    out["graph_nodes"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()

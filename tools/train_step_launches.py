"""The kernel launches of ONE eager training step, in order (torch.profiler device activities), with their durations.

    python tools/train_step_launches.py [--rays 4096] [--size 512] [--grid hash19]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--grid", default="hash19")
    args = ap.parse_args()
    from bench import GRIDS
    from radnerf.scene import SyntheticScene, default_opt
    from radnerf.train import SyntheticTrainStream, Trainer
    scene = SyntheticScene(H=args.size, W=args.size, n_frames=8, device="cuda", opt=default_opt(engine="ops", torso=False, smooth_lips=False, **GRIDS[args.grid]))
    stream = SyntheticTrainStream(scene, n_rays=args.rays)
    trainer = Trainer(scene.model, scene.opt)
    for _ in range(35):
        trainer.step(stream.batch())
    torch.cuda.synchronize()
    assert trainer.global_step % 16 != 0          # not a refresh step
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True, with_stack=bool(os.environ.get("RN_FILL_STACKS"))) as prof:
        trainer.step(stream.batch())
        torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type is not None and "cuda" in str(e.device_type).lower()]
    ev.sort(key=lambda e: e.time_range.start)
    rows = [dict(name=e.name[:110], us=round(e.time_range.elapsed_us(), 1)) for e in ev]
    kernels = [r for r in rows if not r["name"].startswith("Memcpy") and not r["name"].startswith("Memset")]
    if os.environ.get("RN_FILL_STACKS"):       # which Python line asked for each memset-like kernel
        for e in prof.events():
            if e.name in ("aten::fill_", "aten::zero_") and e.input_shapes:
                where = [fr for fr in (e.stack or []) if "/rad-nerf_amd/" in fr or "bench.py" in fr][:2]
                print("fill", e.input_shapes[0], where, file=sys.stderr)
    print(json.dumps(dict(launches=len(rows), kernels=len(kernels), gpu_us=round(sum(r["us"] for r in rows), 1),
                          samples=int(scene.model.step_counter[(scene.model.local_step - 1) % 16, 0]), sequence=rows), indent=1))


if __name__ == "__main__":
    main()

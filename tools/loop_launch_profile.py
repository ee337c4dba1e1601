"""Per-launch duration of the network kernel inside the inference loop against the work of that launch.

    python tools/loop_launch_profile.py [--frames 6] [--mlp f32] [--grid hash19]
For a few frames of the benchmark stream: live rays entering each loop iteration (state[RN_HEAD_ST_HIST + i]), the iteration's
n_step (renderer.py:245 policy), sample slots = rays x n_step, 32-sample tiles per wave slot (256 CUs x 12 waves), and the
HIP-event duration of that iteration's k_nerf_fused dispatch (rn_prof_*).  Shows what a launch costs beyond its tiles: the
latency of a wave's first tile and the ceil() of tiles per wave."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--mlp", default="f32")
    ap.add_argument("--grid", default="hash19")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    import radnerf_hip as hip
    from bench import GRIDS
    from radnerf import fused
    from radnerf.scene import SyntheticScene, default_opt
    scene = SyntheticScene(H=512, W=512, n_frames=250, device="cuda", opt=default_opt(engine="fused", mlp_dtype=args.mlp, **GRIDS[args.grid]))
    N = 512 * 512
    with torch.no_grad():
        for i in range(20):
            scene.render(i)
        torch.cuda.synchronize()
        rows = []
        for i in range(20, 20 + args.frames):
            hip.prof_enable(True)
            scene.render(i)
            torch.cuda.synchronize()
            hip.prof_collect()
            durs = hip.prof_durations()
            hip.prof_enable(False)
            hist = [int(v) for v in fused.loop_history(scene.model, 17).cpu().tolist()]
            it = 0
            for alive, d in zip([h for h in hist if h > 0], [d for d in durs]):
                n_step = max(min(N // alive, 8), 1)
                slots = alive * n_step
                rows.append(dict(frame=i, iteration=it, rays=alive, n_step=n_step, slots=slots, tiles_per_wave=round(slots / 32 / 3072, 2),
                                 us=round(d * 1e3, 1)))
                it += 1
            print(f"frame {i}: dispatch durations (ms) {[round(d, 4) for d in durs]}  hist {hist}", flush=True)
    for r in rows:
        if r["frame"] == 20 + args.frames - 1:
            print(r)
    if args.out:
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()

"""Isolated timing of the fused training pass of the per-sample network (include/radnerf_train.h) at a training step's size:
HIP events around each C-ABI call (pack, forward, backward, weight gradients, table scatter).

    python tools/bench_train_head.py [--M 62000] [--grid hash19] [--rounds 20]
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
    ap.add_argument("--M", type=int, default=62000)
    ap.add_argument("--grid", default="hash19")
    ap.add_argument("--rounds", type=int, default=20)
    args = ap.parse_args()
    os.environ["RN_TRAIN_OVERLAP"] = "0"           # every call on one stream: durations are the kernels' own
    import radnerf_hip as hip
    from bench import GRIDS
    from radnerf import train_head
    from radnerf.scene import SyntheticScene, default_opt
    scene = SyntheticScene(H=64, W=64, n_frames=8, device="cuda", opt=default_opt(engine="ops", torso=False, smooth_lips=False, **GRIDS[args.grid]))
    m = scene.model
    m.train()
    M = args.M
    g = torch.Generator(device="cuda").manual_seed(0)
    # 16 consecutive samples per ray through the head region, as the marcher orders them
    o = (torch.rand(M // 16 + 1, 1, 3, device="cuda", generator=g) - 0.5) * 0.8
    d = torch.nn.functional.normalize(torch.randn(M // 16 + 1, 1, 3, device="cuda", generator=g), dim=-1)
    xyzs = (o + d * (0.027 * torch.arange(16, device="cuda"))[None, :, None]).reshape(-1, 3)[:M].clamp(-1, 1).contiguous()
    dirs = d.expand(-1, 16, -1).reshape(-1, 3)[:M].contiguous()
    enc_a = torch.randn(1, 64, device="cuda", generator=g) * 0.3
    eye = torch.full((1, 1), 0.25, device="cuda")
    ind = m.individual_codes[0]
    up = [torch.randn(M, device="cuda", generator=g), torch.randn(M, 3, device="cuda", generator=g), torch.randn(M, device="cuda", generator=g) * 0.1]
    timer = hip.KernelTimer(lambda name, a: name if name.startswith(("rn_train_head", "rn_grid_scatter")) else None)

    def step():
        for p in m.parameters():
            p.grad = None
        s, c, a, aa = train_head.head_forward(m, xyzs, dirs, enc_a, ind, eye)
        ((s * up[0]).sum() + (c * up[1]).sum() + (aa * up[2]).sum()).backward()
    for _ in range(3):
        step()
    hip.set_timer(timer)
    for _ in range(args.rounds):
        step()
    hip.set_timer(None)
    res = timer.results()
    out = {k: round(v["avg_ms"] * 1e3, 1) for k, v in sorted(res.items())}
    out["sum_us"] = round(sum(out.values()), 1)
    out.update(M=M, grid=args.grid, fwd_groups=os.environ.get("RN_TRAIN_FWD_GROUPS", "11"), wparts=os.environ.get("RN_TRAIN_WPARTS", "96"),
               scatter=os.environ.get("RN_SCATTER", "lbc"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()

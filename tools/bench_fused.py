"""Time the fused network kernel alone (HIP events), fp32-MFMA or f16-MFMA variant, tiled or hash xyz grid.

    python tools/bench_fused.py [--mlp f32|f16] [--grid hash19|tiled16] [--rounds 20] [--lib path/to/lib.so]
Sample counts: 2^20 (steady state) and 206 000 (an in-loop launch of a 512^2 frame)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mlp", default="f16", choices=["f32", "f32x2", "f16"])
    ap.add_argument("--grid", default="hash19")
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--tag", default="")
    ap.add_argument("--sweep", default="", help="comma-separated sample counts instead of the two defaults")
    args = ap.parse_args()
    import numpy as np
    import torch
    from bench import GRIDS
    from radnerf import fused
    from radnerf.scene import SyntheticScene, default_opt
    from tools.bench_kernels import ray_points, time_ms
    scene = SyntheticScene(H=16, W=16, n_frames=8, device="cuda", opt=default_opt(engine="fused", mlp_dtype=args.mlp, **GRIDS[args.grid]))
    m = scene.model
    rng = np.random.default_rng(0)
    out = {"tag": args.tag, "mlp": args.mlp, "grid": args.grid}
    for M in ([int(v) for v in args.sweep.split(',')] if args.sweep else (1 << 20, 206000)):
        x = torch.from_numpy(ray_points(M - M % 8, rng) * 2 - 1).cuda()
        M = x.shape[0]
        d = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda"), dim=1)
        enc_a = torch.randn(1, 64, device="cuda")
        eye = torch.tensor([[0.25]], device="cuda")
        c = m.individual_codes[0].detach()

        def run():
            fused.network_forward(m, x, d, enc_a, c, eye, want_ambient=False)
        run()
        med, best = time_ms(run, args.rounds)
        out[f"M{M}_ms"] = round(med, 4)
        out[f"M{M}_Gsamples_s"] = round(M / med / 1e6, 3)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

"""Read the per-phase wall-clock ticks a -DRN_PHASE_CLOCK build of k_nerf_fused leaves in `ambient` (tools/gpu_phase_clock.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from bench import GRIDS
    from radnerf import fused
    from radnerf.scene import SyntheticScene, default_opt
    from tools.bench_kernels import ray_points
    for grid in os.environ.get("PHASE_GRIDS", "hash19,tiled16").split(","):
        scene = SyntheticScene(H=16, W=16, n_frames=8, device="cuda", opt=default_opt(engine="fused", mlp_dtype="f32", **GRIDS[grid]))
        m = scene.model
        rng = np.random.default_rng(0)
        for M in (8192, 206016, 1 << 20):
            x = torch.from_numpy(ray_points(M - M % 64, rng) * 2 - 1).cuda()
            M = x.shape[0]
            d = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda"), dim=1)
            enc_a = torch.randn(1, 64, device="cuda")
            eye = torch.tensor([[0.25]], device="cuda")
            c = m.individual_codes[0].detach()
            for _ in range(3):
                _, _, amb = fused.network_forward(m, x, d, enc_a, c, eye, want_ambient=True)
            torch.cuda.synchronize()
            a = amb.view(-1, 32, 2)[: M // 32].cpu().numpy() / 100.0  # us
            ph = [a[:, 0, 0], a[:, 0, 1], a[:, 1, 0], a[:, 1, 1]]
            print(grid, "M", M, "per-tile us  xyz-gather %.1f  ambient-net %.1f  ambient-gather %.1f  sigma+color %.1f   (p90 %s)" % (
                *[float(np.median(v)) for v in ph], " ".join("%.1f" % np.percentile(v, 90) for v in ph)), flush=True)


if __name__ == "__main__":
    main()

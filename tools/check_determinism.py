"""Launch-to-launch determinism of the fused network kernels (all three arithmetic variants): N launches on the same
inputs must agree bit for bit.  Written after the f16 kernels, built on v_mfma_f32_32x32x16_f16, were found to return
slightly different results from launch to launch with two waves per SIMD.

    python tools/check_determinism.py [--launches 16] [--samples 20000]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=16)
    ap.add_argument("--samples", type=int, default=20000)
    args = ap.parse_args()
    import torch
    from radnerf import fused
    from radnerf.scene import SyntheticScene, default_opt
    M = args.samples
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(M, 3, device="cuda", generator=g) * 1.4 - 0.7
    d = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda", generator=g), dim=1)
    enc_a = torch.randn(1, 64, device="cuda", generator=g)
    eye = torch.tensor([[0.25]], device="cuda")
    bad = 0
    for mlp in ("f32", "f32x2", "f16"):
        scene = SyntheticScene(H=16, W=16, n_frames=8, device="cuda", opt=default_opt(engine="fused", mlp_dtype=mlp))
        m = scene.model
        c = m.individual_codes[0].detach()
        with torch.no_grad():
            outs = [[t.clone() for t in fused.network_forward(m, x, d, enc_a, c, eye)] for _ in range(args.launches)]
        diffs = [sum(int((a != b).sum()) for a, b in zip(outs[0], o)) for o in outs[1:]]
        print(f"{mlp:6s} elements differing from launch 0: {diffs}")
        bad += sum(diffs)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

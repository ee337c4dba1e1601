"""Micro-benchmarks of the isolated hot kernels on one MI355X (HIP events, interleaved rounds in one process).

    python tools/bench_kernels.py [--rounds 10] [--out gpurun_out/kernels.json]

* grid lookup (north-star gate: >= 60 % of the HBM roofline on algorithmic bytes, measured at B >= 2^20):
  xyz grid D=3, L=16, C=2 with the shipped tiled T=2^16 table and the hash T=2^19 table of BASELINE config 1,
  fp32 and fp16, both decompositions (level-major [L,B,C] / sample-major [B,L*C]), ray-coherent and uniform points.
* fused per-sample network kernel at M = 2^20.
Algorithmic bytes: 16 levels x 8 corners x row bytes + 12 B in + L*C*sizeof out per sample.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8000.0


def time_ms(fn, rounds):
    evs = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs)
    return t[len(t) // 2], t[0]


def ray_points(B, rng):
    """Ray-ordered samples like the marcher emits them: consecutive samples step along a ray by dt."""
    n_step = 8
    n_rays = B // n_step
    o = np.array([0.0, 3.35, 0.0], np.float32)
    tgt = rng.uniform(-0.4, 0.4, (n_rays, 3)).astype(np.float32)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t0 = 3.0 + rng.uniform(0, 0.3, (n_rays, 1)).astype(np.float32)
    ts = t0 + 0.02706 * np.arange(n_step, dtype=np.float32)[None, :]
    p = o[None, None, :] + ts[..., None] * d[:, None, :]
    p = np.clip(p.reshape(-1, 3), -1, 1)
    return ((p + 1) / 2).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "kernels.json"))
    args = ap.parse_args()
    import radnerf_hip as hip
    from gridencoder.encoder import level_offsets

    dev = "cuda"
    rng = np.random.default_rng(0)
    results = []
    L, C, D = 16, 2, 3
    for log2T, gridtype, tag in ((16, 1, "tiled T=2^16 (shipped model)"), (19, 0, "hash T=2^19 (BASELINE config 1)")):
        pls = np.exp2(np.log2(2048 / 16) / (L - 1))
        offsets = level_offsets(D, L, pls, 16, log2T, False)
        S = float(np.log2(pls))
        rows = int(offsets[-1])
        emb32 = torch.from_numpy(rng.uniform(-0.5, 0.5, (rows, C)).astype(np.float32)).to(dev)
        emb16 = emb32.half()
        toff = torch.from_numpy(offsets).to(dev)
        for B in (1 << 20, 1 << 22):
            pts = {"ray-ordered": torch.from_numpy(ray_points(B, rng)).to(dev),
                   "uniform": torch.rand(B, D, device=dev)}
            for pname, x in pts.items():
                for dt_name, emb, dtype_id, esz in (("f32", emb32, hip.RN_F32, 4), ("f16", emb16, hip.RN_F16, 2)):
                    for layout, lname in ((hip.RN_LAYOUT_LBC, "level-major [L,B,C]"), (hip.RN_LAYOUT_BLC, "sample-major [B,L*C]"),
                                          (hip.RN_LAYOUT_BLC_LEVELMAJOR, "level-major [B,L*C]")):
                        out = torch.empty(L * B * C, device=dev, dtype=emb.dtype)

                        def run():
                            hip.call("rn_grid_encode_forward", hip.ptr(x), hip.ptr(emb), hip.ptr(toff), hip.ptr(out), B, D, C, L, S,
                                     16, None, gridtype, 0, 0, dtype_id, layout, hip.stream())
                        run()
                        med, best = time_ms(run, args.rounds)
                        bytes_per = L * 8 * C * esz + 12 + L * C * esz
                        gbs = B * bytes_per / (med * 1e-3) / 1e9
                        results.append(dict(kernel="grid_encode_forward", table=tag, B=B, points=pname, dtype=dt_name, layout=lname,
                                            median_ms=med, best_ms=best, algorithmic_bytes_per_sample=bytes_per,
                                            achieved_GBps=gbs, frac_of_hbm_peak=gbs / HBM_PEAK, Gsamples_per_s=B / med / 1e6))
                        print(json.dumps(results[-1]), flush=True)

    # fused network kernel, M = 2^20 live samples
    from radnerf import fused
    from radnerf.scene import SyntheticScene, default_opt
    scene = SyntheticScene(H=16, W=16, n_frames=8, device=dev, opt=default_opt(engine="fused"))
    m = scene.model
    M = 1 << 20
    x = torch.from_numpy(ray_points(M, rng) * 2 - 1).to(dev)
    d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=1)
    enc_a = torch.randn(1, 64, device=dev)
    eye = torch.tensor([[0.25]], device=dev)
    c = m.individual_codes[0].detach()

    def run_fused():
        fused.network_forward(m, x, d, enc_a, c, eye, want_ambient=False)
    run_fused()
    med, best = time_ms(run_fused, args.rounds)
    results.append(dict(kernel="nerf_fused_forward (+frame_bias, allocations)", M=M, median_ms=med, best_ms=best,
                        achieved_GBps=M * 1580 / (med * 1e-3) / 1e9, frac_of_hbm_peak=M * 1580 / (med * 1e-3) / 1e9 / HBM_PEAK,
                        achieved_TFLOPs=M * 56704 / (med * 1e-3) / 1e12, frac_of_fp32_mfma_peak=M * 56704 / (med * 1e-3) / 1e12 / 157.3,
                        Gsamples_per_s=M / med / 1e6))
    print(json.dumps(results[-1]), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()

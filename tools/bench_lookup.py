"""The standalone multires grid lookup (rn_grid_encode_forward) on one MI355X: north-star gate ">= 60 % of the HBM
roofline on the hash-grid lookup", measured on algorithmic bytes (16 levels x 8 corners x 8 B gathered
+ 12 B coordinates in + 128 B features out = 1 164 B per sample, fp32, D = 3).

    python tools/bench_lookup.py [--table hash19|tiled16] [--B 4194304] [--points frame|bundle|uniform]
                                 [--layouts lbc,blc,module] [--dtype f32|f16] [--rounds 20] [--per-level] [--out f.json]

points: "frame"   = what the marcher emits for consecutive 512^2 frames of the benchmark's pose stream (ray-major, the
                    order GridEncoder.forward sees them in NeRFNetwork.forward);
        "bundle"  = round 1's synthetic bundle: 8 consecutive steps along rays towards random targets (neighbouring
                    rays unrelated: far less coherent than a frame);
        "uniform" = i.i.d. uniform points (no coherence at all).
layouts: lbc / blc = rn_grid_encode_forward_ws ([L,B,C]: the reference kernel's layout, what compat_backend calls; [B,L*C]:
         what GridEncoder.forward calls); lbc0 / blc0 = round 1's rn_grid_encode_forward (per-level / sample-major kernel);
         module = gridencoder.GridEncoder.forward (the operator surface; since round 3 its input scaling is folded into the lookup).
Timing: HIP events around each call on torch's current stream, median of --rounds; run under
`rocprofv3 --kernel-trace --stats` (tools/gpu_lookup_profile.sh) for the profiler's view of the same launches."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8000.0
L, C, D = 16, 2, 3


def time_ms(fn, rounds):
    fn()
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


def bundle_points(B, rng):
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
    return torch.from_numpy(((p + 1) / 2).astype(np.float32)).cuda()


def frame_points(B, size=512):
    """Samples of consecutive frames of the benchmark stream, in the order the marcher writes them (ray-major, up to 16 per
    ray); coordinates already mapped to [0,1] as GridEncoder.forward does."""
    import raymarching
    from radnerf.scene import SyntheticScene, default_opt
    scene = SyntheticScene(H=size, W=size, n_frames=250, device="cuda", opt=default_opt(engine="ops"))
    m = scene.model
    chunks, have, i = [], 0, 0
    N = size * size
    while have < B:
        f = scene.frame(i)
        rays_o, rays_d = f["rays_o"].reshape(-1, 3), f["rays_d"].reshape(-1, 3)
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, m.aabb_infer, m.min_near)
        alive = torch.arange(N, dtype=torch.int32, device="cuda")
        xyzs, _, deltas = raymarching.march_rays(N, 16, alive, nears.clone(), rays_o, rays_d, m.bound, m.density_bitfield,
                                                 m.cascade, m.grid_size, nears, fars, 128, False, scene.opt.dt_gamma, 16)
        x = xyzs[deltas[:, 0] > 0]
        chunks.append(((x + m.bound) / (2 * m.bound)).contiguous())
        have += x.shape[0]
        i += 1
    return torch.cat(chunks)[:B].contiguous(), i


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table", default="hash19", choices=["hash19", "tiled16"])
    ap.add_argument("--B", type=int, default=1 << 22)
    ap.add_argument("--points", default="frame,bundle")
    ap.add_argument("--layouts", default="lbc0,lbc,blc0,blc,module")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"])
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--per-level", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import radnerf_hip as hip
    from gridencoder import GridEncoder

    log2T, gridtype = (19, "hash") if args.table == "hash19" else (16, "tiled")
    enc = GridEncoder(input_dim=D, num_levels=L, level_dim=C, base_resolution=16, log2_hashmap_size=log2T,
                      desired_resolution=2048, gridtype=gridtype).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    enc.embeddings.data = (torch.rand(enc.embeddings.shape, device="cuda", generator=g) - 0.5)
    table = enc.embeddings.detach() if args.dtype == "f32" else enc.embeddings.detach().half()
    esz = 4 if args.dtype == "f32" else 2
    dtype_id = hip.RN_F32 if args.dtype == "f32" else hip.RN_F16
    S = float(np.log2(enc.per_level_scale))
    offs = enc.offsets
    B = args.B
    rng = np.random.default_rng(0)
    results = []
    for pname in args.points.split(","):
        if pname == "frame":
            x, n_frames = frame_points(B)
            note = f"marcher output of {n_frames} consecutive 512x512 frames of the benchmark stream"
        elif pname == "bundle":
            x, note = bundle_points(B, rng), "8 steps along rays to random targets (round-1 synthetic bundle)"
        else:
            x, note = torch.rand(B, D, device="cuda"), "i.i.d. uniform"
        bytes_per = L * 8 * C * esz + 12 + L * C * esz
        for lay in args.layouts.split(","):
            if lay == "module":
                xin = (x * 2 - 1).contiguous()
                if args.dtype == "f16":
                    def run():
                        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                            enc(xin, bound=1)
                else:
                    def run():
                        with torch.no_grad():
                            enc(xin, bound=1)
                label = "gridencoder.GridEncoder.forward -> [B, L*C]"
            elif lay in ("lbc0", "blc0"):       # round-1 entry point: per-level kernel / sample-major kernel
                layout = hip.RN_LAYOUT_LBC if lay == "lbc0" else hip.RN_LAYOUT_BLC
                out = torch.empty(L * B * C, device="cuda", dtype=table.dtype)

                def run():
                    hip.call("rn_grid_encode_forward", hip.ptr(x), hip.ptr(table), hip.ptr(offs), hip.ptr(out), B, D, C, L, S, 16,
                             None, enc.gridtype_id, 0, 0, dtype_id, layout, hip.stream())
                label = "rn_grid_encode_forward " + ("[L,B,C] (per-level kernel)" if lay == "lbc0" else "[B,L*C] (sample-major kernel)")
            else:                                # planned path: LDS-staged coarse pass + level-major pass (+ transposition)
                layout = hip.RN_LAYOUT_LBC if lay == "lbc" else hip.RN_LAYOUT_BLC
                out = torch.empty(L * B * C, device="cuda", dtype=table.dtype)
                ws = hip.grid_forward_workspace(B, L, C, dtype_id, x.device) if lay == "blc" else None
                oh = hip.host_offsets(offs)

                def run():
                    hip.call("rn_grid_encode_forward_ws", hip.ptr(x), hip.ptr(table), hip.ptr(offs), oh, hip.ptr(out), B, D, C, L, S, 16,
                             None, enc.gridtype_id, 0, 0, dtype_id, layout, hip.ptr(ws), ws.numel() if ws is not None else 0, hip.stream())
                label = "rn_grid_encode_forward_ws " + ("[L,B,C]" if lay == "lbc" else "[B,L*C]")
            med, best = time_ms(run, args.rounds)
            gbs = B * bytes_per / (med * 1e-3) / 1e9
            results.append(dict(kernel=label, table=args.table, dtype=args.dtype, B=B, points=pname, points_note=note, median_ms=med,
                                best_ms=best, algorithmic_bytes_per_sample=bytes_per, achieved_GBps=gbs,
                                frac_of_hbm_peak=gbs / HBM_PEAK, Gsamples_per_s=B / med / 1e6))
            print(json.dumps(results[-1]), flush=True)
        if args.per_level:
            # one level at a time (L = 1 calls on the level's slice; H = that level's resolution, so the lattice differs from the
            # 16-level call by rounding only): where the 16-level launch spends its time
            o = offs.cpu().numpy()
            for l in range(L):
                res = int(np.ceil(16 * enc.per_level_scale ** l))
                sub = table[int(o[l]):int(o[l + 1])].contiguous()
                so = torch.tensor([0, int(o[l + 1] - o[l])], dtype=torch.int32, device="cuda")
                out1 = torch.empty(B * C, device="cuda", dtype=table.dtype)

                def run1():
                    hip.call("rn_grid_encode_forward", hip.ptr(x), hip.ptr(sub), hip.ptr(so), hip.ptr(out1), B, D, C, 1, 0.0, res, None,
                             enc.gridtype_id, 0, 0, dtype_id, hip.RN_LAYOUT_LBC, hip.stream())
                med, _ = time_ms(run1, max(5, args.rounds // 2))
                results.append(dict(kernel="one level [1,B,C]", level=l, resolution=res, rows=int(o[l + 1] - o[l]), points=pname, median_ms=med))
                print(json.dumps(results[-1]), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()

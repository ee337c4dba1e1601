"""Isolated timing of the grid table gradient (rn_grid_encode_backward) at a training step's size.

    python tools/bench_grid_backward.py [--B 58000]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=58000)
    args = ap.parse_args()
    import radnerf_hip as hip
    from radnerf import train_head  # noqa: F401  (declares the argument types of rn_grid_scatter_lbc)
    from gridencoder.encoder import level_offsets
    out = {}
    for name, (D, log2T, gridtype) in {"xyz hash T=2^19": (3, 19, 0), "ambient tiled T=2^16": (2, 16, 1)}.items():
        C, L, B = 2, 16, args.B
        pls = np.exp2(np.log2(2048 / 16) / (L - 1))
        offsets = level_offsets(D, L, pls, 16, log2T, False)
        S = float(np.log2(pls))
        off_d = torch.from_numpy(np.asarray(offsets, np.int32)).cuda()
        emb = torch.zeros(int(offsets[-1]), C, device="cuda")
        res = {}
        for pts in ("uniform", "ray runs"):
            if pts == "uniform":
                x = torch.rand(B, D, device="cuda")
            else:                      # 16 consecutive samples per ray, as the marcher orders them
                o = torch.rand(B // 16 + 1, 1, D, device="cuda") * 0.6 + 0.2
                d = torch.nn.functional.normalize(torch.randn(B // 16 + 1, 1, D, device="cuda"), dim=-1)
                x = (o + d * (0.0135 * torch.arange(16, device="cuda"))[None, :, None]).reshape(-1, D)[:B].clamp(0, 1).contiguous()
            g = torch.randn(B, L * C, device="cuda")
            ge = torch.zeros_like(emb)
            fn = lambda: hip.call("rn_grid_encode_backward", hip.ptr(g), hip.ptr(x), hip.ptr(emb), hip.ptr(off_d), hip.ptr(ge), B, D, C, L, S, 16,
                                  None, None, gridtype, 0, 0, hip.RN_F32, hip.RN_LAYOUT_BLC, hip.stream())
            # round 3: the line-keyed scatter of the fused training pass (level-major gradients, rn_grid_scatter_lbc)
            import ctypes as C_
            from radnerf.fused import GridT
            gd = GridT()
            gd.embeddings, gd.offsets, gd.D, gd.L, gd.H, gd.S, gd.gridtype, gd.dtype = emb.data_ptr(), off_d.data_ptr(), D, L, 16, S, gridtype, hip.RN_F32
            g_lbc = g.view(B, L, C).permute(1, 0, 2).contiguous()
            fn2 = lambda: hip.call("rn_grid_scatter_lbc", hip.ptr(g_lbc), hip.ptr(x), B, None, C_.byref(gd), hip.ptr(ge), hip.stream())
            class _E:          # what train_head.grid_scatter reads of a GridEncoder
                offsets = off_d
            fn3 = lambda: train_head.grid_scatter([(g_lbc, x, _E, gd, ge)], B, None)
            def fn4():
                os.environ["RN_SCATTER"] = "binned"
                try:
                    train_head.grid_scatter([(g_lbc, x, _E, gd, ge)], B, None)
                finally:
                    os.environ["RN_SCATTER"] = "lbc"
            # line-keyed = rn_grid_scatter_lbc (LDS line merge on every level); default = train_head.grid_scatter (hashed levels
            # straight to memory from adjacent lanes, the others line-merged); binned = hashed levels summed by table region
            for label, f in (("", fn), (" line-keyed", fn2), (" default", fn3), (" binned", fn4)):
                for _ in range(3):
                    f()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20):
                    f()
                b.record()
                torch.cuda.synchronize()
                res[pts + label + " us"] = round(a.elapsed_time(b) / 20 * 1e3, 1)
        out[name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()

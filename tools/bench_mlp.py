"""Isolated timings of the training MLP kernels (rn_mlp64_*): forward, backward-data, weight gradients, per shape.

    python tools/bench_mlp.py [--M 58181] [--reps 50]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=58181)
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import radnerf_hip as hip
    from radnerf import mlp_train as mt
    lib = hip._lib
    out = {}
    for name, (din, dout, nl) in {"ambient_net": (96, 2, 3), "sigma_net": (65, 65, 3), "color_net": (84, 3, 2)}.items():
        M = args.M
        pad = (din + 3) & ~3
        x = torch.rand(M, pad, device="cuda")
        x[:, din:] = 0
        dims = [din] + [64] * (nl - 1) + [dout]
        ws = [torch.randn(dims[i + 1], dims[i], device="cuda") / 8 for i in range(nl)]
        image = torch.empty(int(lib.rn_mlp64_image_floats(din, dout, nl)), device="cuda")
        tile = int(lib.rn_mlp64_tile_floats(M))
        h0, h1, dz0, dz1 = (torch.empty(tile, device="cuda") for _ in range(4))
        y = torch.empty(M, dout, device="cuda")
        gy = torch.rand(M, dout, device="cuda")
        gx = torch.empty(M, pad, device="cuda")
        gws = [torch.empty_like(w) for w in ws]
        wsp = torch.empty(int(lib.rn_mlp64_wgrad_workspace(nl)), dtype=torch.uint8, device="cuda")
        s = hip.stream()
        P = hip.ptr
        w1 = P(ws[1]) if nl == 3 else None
        calls = {
            "pack": lambda: hip.call("rn_mlp64_pack", P(ws[0]), din, w1, P(ws[-1]), din, dout, nl, P(image), s),
            "forward": lambda: hip.call("rn_mlp64_forward", P(x), M, P(image), None, din, dout, nl, P(y), P(h0), P(h1) if nl == 3 else None, s),
            "backward": lambda: hip.call("rn_mlp64_backward", P(gy), M, P(image), din, dout, nl, P(h0), P(h1) if nl == 3 else None, P(gx), P(dz0),
                                         P(dz1) if nl == 3 else None, s),
            "weight_grads": lambda: hip.call("rn_mlp64_weight_grads", P(x), P(gy), M, din, dout, nl, P(h0), P(h1) if nl == 3 else None, P(dz0),
                                             P(dz1) if nl == 3 else None, P(gws[0]), din, P(gws[1]) if nl == 3 else None, P(gws[-1]), None, P(wsp), s),
        }
        res = {}
        for k, fn in calls.items():
            for _ in range(3):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            res[k + "_us"] = round(a.elapsed_time(b) * 1e3 / args.reps, 2)
        flops = 2.0 * M * sum(dims[i] * dims[i + 1] for i in range(nl))
        res["forward_TFLOPs"] = round(flops / (res["forward_us"] * 1e-6) / 1e12, 1)
        out[name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()

"""Per-parameter gradient differences between the fused training head and the per-operator path (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))
import numpy as np, torch
from radnerf.scene import SyntheticScene, default_opt

def run(m, xyzs, dirs, enc_a, eye, up, mode):
    os.environ["RN_TRAIN_HEAD"] = mode
    for p in m.parameters(): p.grad = None
    enc_a = enc_a.clone().requires_grad_(True); eye = eye.clone().requires_grad_(True)
    sigma, rgb, amb = m(xyzs, dirs, enc_a, m.individual_codes[3], eye)
    loss = (sigma * up[0]).sum() + (rgb * up[1]).sum() + (amb.abs().sum(-1) * up[2]).sum() + (amb * up[3]).sum()
    loss.backward()
    g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    g["enc_a"], g["eye"] = enc_a.grad.clone(), eye.grad.clone()
    return (sigma.detach(), rgb.detach(), amb.detach()), g

scene = SyntheticScene(H=32, W=32, n_frames=8, device="cuda", opt=default_opt(engine="ops", torso=False, smooth_lips=False))
m = scene.model; m.train()
M = 4099
g = torch.Generator(device="cuda").manual_seed(3)
xyzs = (torch.rand(M, 3, device="cuda", generator=g) * 2 - 1) * 0.98
if os.environ.get("OOB"): xyzs[:7] = 1.25
dirs = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda", generator=g), dim=-1)
enc_a = torch.randn(1, 64, device="cuda", generator=g) * 0.5
eye = torch.full((1, 1), 0.25, device="cuda")
base = [torch.randn(M, device="cuda", generator=g), torch.randn(M, 3, device="cuda", generator=g),
        torch.randn(M, device="cuda", generator=g) * 0.3, torch.randn(M, 2, device="cuda", generator=g) * 0.3]
os.environ["RN_TRAIN_HEAD"] = "ops"
with torch.no_grad(): amb = m(xyzs, dirs, enc_a, m.individual_codes[3], eye)[2]
enc_w = m.encoder_ambient
scales = torch.tensor([2.0 ** (l * float(np.log2(enc_w.per_level_scale))) * enc_w.base_resolution - 1 for l in range(16)], dtype=torch.float64, device="cuda")
pos = ((amb.double() + 1) / 2).unsqueeze(-1) * scales + 0.5
frac = pos - pos.floor()
for eps in (2e-5,):
    margin = eps * scales
    stable = ((frac > margin) & (frac < 1 - margin)).all(-1).all(-1).float()
    print("eps", eps, "stable fraction", float(stable.mean()))
    for label, keep in (("all", [1, 1, 1, 1]), ("sigma only", [1, 0, 0, 0]), ("rgb only", [0, 1, 0, 0]), ("abs only", [0, 0, 1, 0]), ("amb only", [0, 0, 0, 1])):
        up = [u * k * (stable if u.dim() == 1 else stable.unsqueeze(-1)) for u, k in zip(base, keep)]
        o1, g1 = run(m, xyzs, dirs, enc_a, eye, up, "ops")
        o2, g2 = run(m, xyzs, dirs, enc_a, eye, up, "fused")
        print(" ", label, "amb diff", float((o1[2] - o2[2]).abs().max()))
        for n in sorted(g1):
            a, b = g2[n], g1[n]
            sc = float(b.abs().max()) + 1e-20
            print(f"    {n:40s} err {float((a - b).abs().max()) / sc:.2e}  max {sc:.2e}")

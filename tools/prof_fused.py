"""Runs only the fused per-sample kernel (M = 2^20) a few times -- the target of rocprofv3 --pmc passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bench_kernels import ray_points  # noqa: E402
from radnerf import fused  # noqa: E402
from radnerf.scene import SyntheticScene, default_opt  # noqa: E402

dev = "cuda"
rng = np.random.default_rng(0)
scene = SyntheticScene(H=16, W=16, n_frames=8, device=dev, opt=default_opt(engine="fused"))
m = scene.model
M = 1 << 20
x = torch.from_numpy(ray_points(M, rng) * 2 - 1).to(dev)
d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=1)
enc_a = torch.randn(1, 64, device=dev)
eye = torch.tensor([[0.25]], device=dev)
c = m.individual_codes[0].detach()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    fused.network_forward(m, x, d, enc_a, c, eye, want_ambient=False)
torch.cuda.synchronize()
print("done")

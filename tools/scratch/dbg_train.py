import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))
import torch
from radnerf.scene import SyntheticScene, default_opt
from radnerf.train import SyntheticTrainStream, Trainer, train_step
scene = SyntheticScene(H=64, W=64, n_frames=8, device="cuda", opt=default_opt(engine="ops", torso=False, smooth_lips=False))
stream = SyntheticTrainStream(scene, n_rays=2048)
m = scene.model
print("target mean", stream.target.mean().item(), "face frac", stream.face_mask.float().mean().item())
m.train()
b = stream.batch()
pred, rgb, loss = train_step(m, b, scene.opt)
print("loss0", loss.item(), "mse", ((pred - rgb) ** 2).mean().item())
out = m.render(b["rays_o"], b["rays_d"], b["auds"], b["bg_coords"], b["poses"], eye=b["eye"], index=[0], bg_color=b["bg_color"], perturb=True, force_all_rays=False, dt_gamma=scene.opt.dt_gamma, max_steps=16)
print({k: (v.shape, float(v.float().mean())) for k, v in out.items() if torch.is_tensor(v)})
with torch.no_grad():
    m.color_net.net[-1].weight.add_(0.3 * torch.randn_like(m.color_net.net[-1].weight))
pred, rgb, loss = train_step(m, b, scene.opt)
print("loss1", loss.item(), "mse", ((pred - rgb) ** 2).mean().item())
loss.backward()
for n, p in m.named_parameters():
    if p.grad is not None:
        print(n, float(p.grad.abs().max()))

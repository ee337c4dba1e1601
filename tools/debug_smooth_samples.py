import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo/rad-nerf_amd"); sys.path.insert(0, "/root/repo/tests/golden")
import cases
from radnerf.scene import SyntheticScene, default_opt
st = np.load("/root/repo/tests/golden/reference_train_stable.npz")
os.environ["RN_TRAIN_HEAD"] = "ops"
scene = SyntheticScene(H=256, W=256, n_frames=8, device="cuda", opt=default_opt(torso=False, smooth_lips=False, engine="ops"))
m, opt = scene.model, scene.opt
m.train()
cpu = SyntheticScene(H=256, W=256, n_frames=8, device="cpu", opt=default_opt(torso=False, smooth_lips=False, engine="ops"))
fc = cpu.frame(0)
fg = scene.frame(0)
for k in fc:
    if torch.is_tensor(fc[k]):
        print(k, "GPU-built vs CPU-built inputs: max |d| =", float((fg[k].cpu().float() - fc[k].float()).abs().max()))
f = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in fc.items()}
px = torch.from_numpy(st["train_px"]).cuda()
seen = {}
def _enc_a(mod, a, o):
    seen["enc_a"] = a[2].detach().clone()
m.register_forward_hook(_enc_a)
def _enc_x(mod, a, o):
    seen["enc_x"] = o.detach().clone()
m.encoder.register_forward_hook(_enc_x)
def _amb(mod, a, o):
    seen["enc_w"], seen["amb_in"] = o.detach().clone(), a[0].detach().clone()
m.encoder_ambient.register_forward_hook(_amb)
m.mean_count, m.local_step = 49152, 0
m.step_counter.zero_()
res = m.render(f["rays_o"][:, px], f["rays_d"][:, px], f["auds"], f["bg_coords"][:, px], f["poses"], eye=f["eye"], index=[0],
               bg_color=f["bg_color"][:, px], staged=False, perturb=False, force_all_rays=False, dt_gamma=opt.dt_gamma, max_steps=opt.max_steps)
live = int(st["counter"][0])
n = lambda t: t.detach().float().cpu().numpy()
d = np.abs(n(seen["enc_a"]) - st["enc_a"]); print("enc_a max|d|", d.max(), "max|ref|", np.abs(st["enc_a"]).max())
for k in ("enc_x", "enc_w"):
    want = st[f"every256::{k}"][: (live + 255) // 256]; got = n(seen[k][::256])[: want.shape[0]]
    d = np.abs(got - want); print(k, "max|d|", d.max(), "max|ref|", np.abs(want).max(), "per level max|d|", d.reshape(-1, 16, 2).max((0, 2)))
want = st["every8::ambient"][: (live + 7) // 8]; got = n(seen["amb_in"][::8])[: want.shape[0]]
print("ambient max|d|", np.abs(got - want).max(), "mean|d|", np.abs(got - want).mean(), "max|ref|", np.abs(want).max())
# same audio code as the reference -> how much of the ambient deviation is the audio code's?

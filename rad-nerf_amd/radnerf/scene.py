"""Synthetic scene for benchmarks and parity tests: no dataset or checkpoint is needed.

* `default_opt()`       -- the hot-path flags with the defaults of the reference CLI (main.py:28-120,
                           `-O --torso`, esperanto audio features).
* `SyntheticScene`      -- seeded model + ellipsoid head occupancy (128^3 morton bitfield) + 2-D torso
                           grid + a 250-frame OrbitCamera pose stream + random audio features.
Everything here is host-side setup; the per-frame tensors it hands out already live on the device.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch

from .rays import convert_poses, get_audio_features, get_bg_coords, get_rays, intrinsics_from_fovy, orbit_pose


def default_opt(**overrides):
    opt = SimpleNamespace(
        bound=1, min_near=0.05, dt_gamma=1 / 256, max_steps=16, density_thresh=10, density_thresh_torso=0.01,
        torso_shrink=0.8, att=2, ind_dim=4, ind_dim_torso=8, ind_num=10000, exp_eye=True, test_train=False,
        smooth_lips=True, torso=True, cuda_ray=True, emb=False, train_camera=False, fp16=False,
        asr_model="cpierse/wav2vec2-large-xlsr-53-esperanto", radius=3.35, fovy=21.24, engine="ops")
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt


def _part1by2(v):
    """Spread the low 10 bits of v two apart (same bit tricks as raymarching.cu:56-63)."""
    v = v.astype(np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3d_np(x, y, z):
    return _part1by2(x) | (_part1by2(y) << np.uint32(1)) | (_part1by2(z) << np.uint32(2))


def ellipsoid_bitfield(grid_size=128, bound=1.0, semi_axes=(0.40, 0.42, 0.40), centre=(0.0, 0.0, 0.0)):
    """Occupancy bitfield (cascade 0) whose set cells are those with their centre inside the ellipsoid.
    Returns (bitfield uint8 [H^3/8], density_grid float32 [1, H^3]) in morton order."""
    H = grid_size
    idx = np.arange(H, dtype=np.int64)
    c = ((idx + 0.5) / H * 2 - 1) * bound
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    inside = (((X - centre[0]) / semi_axes[0]) ** 2 + ((Y - centre[1]) / semi_axes[1]) ** 2 +
              ((Z - centre[2]) / semi_axes[2]) ** 2) <= 1.0
    I, J, K = np.meshgrid(idx, idx, idx, indexing="ij")
    m = morton3d_np(I.reshape(-1), J.reshape(-1), K.reshape(-1)).astype(np.int64)
    occ = np.zeros(H ** 3, dtype=bool)
    occ[m] = inside.reshape(-1)
    bitfield = np.packbits(occ, bitorder="little")
    return bitfield, occ.astype(np.float32)[None, :]


def torso_grid(grid_size=128, row_from=0.6, col_from=0.1, col_to=0.9):
    """2-D torso occupancy: 1 in the lower-centre ~32 % of the image, flat index [y * G + x].
    grid_sample's x (fast axis) is driven by bg_coords[:, 0], which varies along image ROWS
    (nerf/utils.py:241-244, nerf/renderer.py:282,472)."""
    G = grid_size
    g = np.zeros((G, G), dtype=np.float32)  # [y, x]
    x0 = int(round(row_from * G))
    y0, y1 = int(round(col_from * G)), int(round(col_to * G))
    g[y0:y1, x0:] = 1.0
    return g.reshape(-1)


def init_synthetic_state(model, opt, seed=0, semi_axes=(0.40, 0.42, 0.40), embedding_range=0.5):
    """Put a freshly constructed NeRFNetwork-like module (this tree's or the reference's) into the synthetic
    scene's state: re-drawn grid tables, ellipsoid occupancy bitfield, lower-centre torso grid."""
    # the shipped 1e-4 init gives featureless encodings; re-draw the tables
    g = torch.Generator().manual_seed(seed + 1)
    for enc in (model.encoder, model.encoder_ambient, getattr(model, "torso_encoder", None)):
        if enc is not None:
            enc.embeddings.data = (torch.rand(enc.embeddings.shape, generator=g) * 2 - 1) * embedding_range
    bits, dens = ellipsoid_bitfield(model.grid_size, float(opt.bound), semi_axes)
    with torch.no_grad():
        model.density_bitfield.copy_(torch.from_numpy(bits))
        model.density_grid.copy_(torch.from_numpy(dens))
        model.mean_density = float(dens.mean())
        if opt.torso:
            model.density_grid_torso.copy_(torch.from_numpy(torso_grid(model.grid_size)))
            model.mean_density_torso = 0.29
    return model


class SyntheticScene:
    def __init__(self, H=512, W=512, n_frames=250, device="cuda", seed=0, opt=None, semi_axes=(0.40, 0.42, 0.40),
                 embedding_range=0.5, model=None):
        self.H, self.W, self.n_frames = H, W, n_frames
        self.device = torch.device(device)
        self.opt = opt if opt is not None else default_opt()

        if model is None:
            from .network import NeRFNetwork  # late import: needs the HIP extension to be loadable
            torch.manual_seed(seed)
            model = NeRFNetwork(self.opt)
        init_synthetic_state(model, self.opt, seed, semi_axes, embedding_range)
        self.model = model.to(self.device).eval()
        self.model.ray_order_width = W   # rays handed out below are row-major pixels (hint for the fused engine's loop order)

        # pose stream: OrbitCamera, yaw 8 deg * sin(2 pi t / 4 s), pitch 4 deg * sin(2 pi t / 2.5 s), 25 FPS
        self.intrinsics = intrinsics_from_fovy(H, W, self.opt.fovy)
        poses = []
        for i in range(n_frames):
            t = i / 25.0
            poses.append(orbit_pose(self.opt.radius, 8.0 * math.sin(2 * math.pi * t / 4.0),
                                    4.0 * math.sin(2 * math.pi * t / 2.5)))
        self.poses = torch.from_numpy(np.stack(poses)).to(self.device)  # [T,4,4]
        on_device = self.device.type == "cuda" and getattr(self.opt, "engine", "ops") == "fused" and \
            getattr(self.opt, "ray_engine", "fused") == "fused"
        if on_device:       # the torso pass's other two inputs, one launch each (rn_convert_poses, rn_get_bg_coords)
            from . import fused as _fused
            self.poses6 = _fused.convert_poses(self.poses)
        else:
            self.poses6 = convert_poses(self.poses)                    # [T,6]

        rng = np.random.default_rng(seed)
        feats = (3.0 * rng.standard_normal((n_frames, 16, 44))).astype(np.float32)
        self.aud_features = torch.from_numpy(feats).permute(0, 2, 1).contiguous().to(self.device)  # [T,44,16]
        self.eye = torch.tensor([[0.25]], dtype=torch.float32, device=self.device)
        self.bg_coords = _fused.get_bg_coords(H, W, self.device) if on_device else get_bg_coords(H, W, self.device)
        self.bg_color = torch.ones(1, H * W, 3, dtype=torch.float32, device=self.device)
        self._rays = {}

    def _lazy_rays(self):
        return (self.device.type == "cuda" and getattr(self.opt, "engine", "ops") == "fused"
                and getattr(self.opt, "ray_engine", "fused") == "fused" and getattr(self.opt, "frame_kernels", "merged") == "merged")

    def frame(self, i, lazy_rays=False):
        """Inputs of model.render for frame i (everything resident on the device).  lazy_rays (fused engine): rays_o / rays_d
        are two reused buffers and `ray_source` = (pose, intrinsics, W) tells the frame prologue kernel to fill them -- ray
        generation costs no launch and no cache of per-pose rays is kept."""
        i = i % self.n_frames
        base = dict(auds=get_audio_features(self.aud_features, self.opt.att, i), bg_coords=self.bg_coords, poses=self.poses6[i:i + 1],
                    eye=self.eye, index=0, bg_color=self.bg_color)
        if lazy_rays and self._lazy_rays():
            bufs = self.__dict__.setdefault("_ray_bufs", {})
            key = torch.cuda.current_stream().cuda_stream          # frames in flight on different streams must not share them
            if key not in bufs:
                bufs[key] = (torch.empty(1, self.H * self.W, 3, device=self.device), torch.empty(1, self.H * self.W, 3, device=self.device))
            return dict(base, rays_o=bufs[key][0], rays_d=bufs[key][1], ray_source=(self.poses[i], self.intrinsics, self.W))
        if i not in self._rays:
            if self.device.type == "cuda" and getattr(self.opt, "engine", "ops") == "fused" and getattr(self.opt, "ray_engine", "fused") == "fused":
                from . import fused                                   # one kernel (rn_get_rays) instead of ~12 torch launches
                r = fused.get_rays(self.poses[i], self.intrinsics, self.H, self.W)
            else:
                r = get_rays(self.poses[i:i + 1], self.intrinsics, self.H, self.W, -1)
            self._rays[i] = (r["rays_o"].contiguous(), r["rays_d"].contiguous())
        rays_o, rays_d = self._rays[i]
        return dict(base, rays_o=rays_o, rays_d=rays_d)

    def render_kwargs(self):
        o = self.opt
        return dict(dt_gamma=o.dt_gamma, max_steps=o.max_steps, perturb=False, force_all_rays=True, T_thresh=1e-4)

    def render(self, i, want_u8=False, audio_code=None):
        """want_u8: the fused engine's blend kernel also writes the quantised frame (out["image_u8"], SURVEY f-4).
        audio_code: (smoothed code, bias block) of this frame when the caller computed them ahead (radnerf/parallel.py)."""
        f = self.frame(i, lazy_rays=True)
        kw = self.render_kwargs()
        if want_u8:
            kw["want_u8"] = True
        if "ray_source" in f:
            kw["ray_source"] = f["ray_source"]
        if audio_code is not None:
            kw["audio_code"] = audio_code
        return self.model.render(f["rays_o"], f["rays_d"], f["auds"], f["bg_coords"], f["poses"], eye=f["eye"],
                                 index=f["index"], bg_color=f["bg_color"], **kw)

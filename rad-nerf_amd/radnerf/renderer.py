"""Occupancy-grid renderer -- host-side mirror of the reference's nerf/renderer.py (NeRFRenderer).

Keeps the reference's public contract: constructor reading the same `opt` fields, the same
registered buffers / parameters (aabb_train, aabb_infer, density_grid, density_bitfield,
density_grid_torso, step_counter, individual_codes[_torso]), `render(...)` -> dict(image, depth, ...),
`run_cuda(...)`, `update_extra_state`, `mark_untrained_grid`, `reset_extra_state`.

`run_cuda` has two inference engines selected by `self.engine`:
  * "ops"   -- the reference's loop shape (nerf/renderer.py:225-262): march -> network -> composite ->
               boolean-mask compaction, one host sync per iteration, MLPs as torch layers;
  * "fused" -- the MI355X path (radnerf/fused.py): the whole <=max_steps loop is enqueued without a
               single host read-back; live-ray counts, n_step policy and sample counts stay on the device.
Both produce the reference's results (parity tests in tests/).
"""
import math
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import raymarching

from .rays import convert_poses, get_audio_features


class NeRFRenderer(nn.Module):
    def __init__(self, opt):
        # nerf/renderer.py:63-133
        super().__init__()
        self.opt = opt
        self.bound = opt.bound
        self.cascade = 1 + math.ceil(math.log2(opt.bound))
        self.grid_size = 128
        self.density_scale = 1
        self.min_near = opt.min_near
        self.density_thresh = opt.density_thresh
        self.density_thresh_torso = opt.density_thresh_torso
        self.exp_eye = opt.exp_eye
        self.test_train = opt.test_train
        self.smooth_lips = opt.smooth_lips
        self.torso = opt.torso
        self.cuda_ray = opt.cuda_ray
        self.engine = getattr(opt, "engine", "ops")

        aabb_train = torch.FloatTensor([-opt.bound, -opt.bound / 2, -opt.bound, opt.bound, opt.bound / 2, opt.bound])
        self.register_buffer("aabb_train", aabb_train)
        self.register_buffer("aabb_infer", aabb_train.clone())

        self.individual_num = opt.ind_num
        self.individual_dim = opt.ind_dim
        if self.individual_dim > 0:
            self.individual_codes = nn.Parameter(torch.randn(self.individual_num, self.individual_dim) * 0.1)
        if self.torso:
            self.individual_dim_torso = opt.ind_dim_torso
            if self.individual_dim_torso > 0:
                self.individual_codes_torso = nn.Parameter(torch.randn(self.individual_num, self.individual_dim_torso) * 0.1)

        self.train_camera = self.opt.train_camera
        if self.train_camera:
            self.camera_dR = nn.Parameter(torch.zeros(self.individual_num, 3))
            self.camera_dT = nn.Parameter(torch.zeros(self.individual_num, 3))

        self.register_buffer("density_grid", torch.zeros([self.cascade, self.grid_size ** 3]))
        self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
        self.mean_density = 0
        self.iter_density = 0
        if self.torso:
            self.register_buffer("density_grid_torso", torch.zeros([self.grid_size ** 2]))
        self.mean_density_torso = 0

        self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
        self.mean_count = 0
        self.local_step = 0

        if self.smooth_lips:
            self.enc_a = None
        self.last_stats = None  # filled by the inference engines: iterations / sample slots of the last frame
        self.count_samples = False  # when set, the engines also count live samples (costs a device read-back)

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def reset_extra_state(self):
        # nerf/renderer.py:145-155
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    # ------------------------------------------------------------------------------------------
    def _audio_code(self, auds):
        """encode_audio + the lip-smoothing EMA (nerf/renderer.py:188-194); stateful across frames."""
        if self._fused_audio(auds):
            # one kernel for AudioNet + AudioAttNet, one for the smoothing recurrence (radnerf/audio.py)
            from . import audio
            enc_a = audio.encode_windows(self, auds)
            if self.smooth_lips:
                enc_a = audio.smooth_(self, enc_a)
            return enc_a
        enc_a = self.encode_audio(auds)
        if enc_a is not None and self.smooth_lips:
            if self.enc_a is not None:
                _lambda = 0.35
                enc_a = _lambda * self.enc_a + (1 - _lambda) * enc_a
            self.enc_a = enc_a
        return enc_a

    def fused_audio_enabled(self):
        """The fused engine also takes the audio code off PyTorch (inference only; --emb ids and odd shapes stay torch)."""
        if self.training or self.engine != "fused" or getattr(self.opt, "audio_engine", "fused") != "fused":
            return False
        ok = getattr(self, "_fused_audio_ok", None)
        if ok is None:
            from . import audio
            ok = self._fused_audio_ok = audio.supported(self)
        return ok

    def _fused_audio(self, auds):
        if auds is None or not self.fused_audio_enabled() or not auds.is_cuda or auds.dtype != torch.float32:
            return False
        return tuple(auds.shape) == (8 if self.att > 0 else 1, self.audio_in_dim, 16)

    def _march_loop_ops(self, rays_o, rays_d, nears, fars, enc_a, ind_code, eye, perturb, dt_gamma, max_steps,
                        T_thresh):
        """Inference loop in the reference's shape (nerf/renderer.py:227-262)."""
        N, device = rays_o.shape[0], rays_o.device
        weights_sum = torch.zeros(N, dtype=torch.float32, device=device)
        depth = torch.zeros(N, dtype=torch.float32, device=device)
        image = torch.zeros(N, 3, dtype=torch.float32, device=device)
        rays_alive = torch.arange(N, dtype=torch.int32, device=device)
        rays_t = nears.clone()
        step = iters = slots = 0
        live = torch.zeros((), dtype=torch.int64, device=device) if self.count_samples else None
        while step < max_steps:
            n_alive = rays_alive.shape[0]
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)
            xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound,
                                                        self.density_bitfield, self.cascade, self.grid_size, nears, fars,
                                                        128, perturb if step == 0 else False, dt_gamma, max_steps)
            sigmas, rgbs, ambient = self(xyzs, dirs, enc_a, ind_code, eye)
            sigmas = self.density_scale * sigmas
            raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth,
                                       image, T_thresh)
            rays_alive = rays_alive[rays_alive >= 0]
            step += n_step
            iters += 1
            slots += xyzs.shape[0]
            if live is not None:
                live += (deltas[:, 0] > 0).sum()
        self.last_stats = {"iterations": iters, "sample_slots": slots}
        if live is not None:
            self.last_stats["live_samples"] = int(live.item())
        return weights_sum, depth, image

    def run_cuda(self, rays_o, rays_d, auds, bg_coords, poses, eye=None, index=0, dt_gamma=0, bg_color=None,
                 perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-4, **kwargs):
        # nerf/renderer.py:158-316.  rays_o, rays_d: [1,N,3]; auds: [8,C,16]; bg_coords: [1,N,2]; poses: [1,6]
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        bg_coords = bg_coords.contiguous().view(-1, 2)

        if self.train_camera and (self.training or self.test_train):
            from .rays import euler_angles_to_matrix
            dT = self.camera_dT[index]
            dR = euler_angles_to_matrix(self.camera_dR[index] / 180 * np.pi + 1e-8).squeeze(0)
            rays_o = rays_o + dT
            rays_d = rays_d @ dR

        N, device = rays_o.shape[0], rays_o.device
        results = {}

        if not self.training and self.engine == "fused" and not perturb:
            # MI355X path: the whole frame through the fused C ABI (radnerf/fused.py)
            from . import fused
            enc_a = self._audio_code(auds)
            ind_code = self.individual_codes[0] if self.individual_dim > 0 else None
            ind_code_torso = (self.individual_codes_torso[0] if (self.torso and self.individual_dim_torso > 0) else None)
            out = fused.render_frame(self, rays_o, rays_d, enc_a, ind_code, eye, bg_coords, poses, ind_code_torso, bg_color,
                                     dt_gamma, max_steps, T_thresh, want_u8=kwargs.get("want_u8", False))
            results["image"] = out["image"].view(*prefix, 3)
            results["depth"] = out["depth"].view(*prefix)
            for key in ("torso_alpha", "torso_color", "image_u8"):
                if key in out:
                    results[key] = out[key]
            return results

        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer,
                                                     self.min_near)
        nears, fars = nears.detach(), fars.detach()

        enc_a = self._audio_code(auds)

        if self.individual_dim > 0:
            ind_code = self.individual_codes[index] if self.training else self.individual_codes[0]
        else:
            ind_code = None

        if self.training:
            # nerf/renderer.py:206-223
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield,
                                                                    self.cascade, self.grid_size, nears, fars, counter,
                                                                    self.mean_count, perturb, 128, force_all_rays,
                                                                    dt_gamma, max_steps)
            sigmas, rgbs, ambient = self(xyzs, dirs, enc_a, ind_code, eye)
            sigmas = self.density_scale * sigmas
            weights_sum, ambient_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, ambient.abs().sum(-1),
                                                                                      deltas, rays)
            results["weights_sum"] = weights_sum
            results["ambient"] = ambient_sum
        else:
            weights_sum, depth, image = self._march_loop_ops(rays_o, rays_d, nears, fars, enc_a, ind_code, eye, perturb,
                                                             dt_gamma, max_steps, T_thresh)

        if bg_color is None:
            bg_color = 1

        if self.torso:
            # nerf/renderer.py:269-302: blend the 2-D torso layer over the background first
            if self.individual_dim_torso > 0:
                ind_code_torso = self.individual_codes_torso[index] if self.training else self.individual_codes_torso[0]
            else:
                ind_code_torso = None
            density_thresh_torso = min(self.density_thresh_torso, self.mean_density_torso)
            occupancy = F.grid_sample(self.density_grid_torso.view(1, 1, self.grid_size, self.grid_size),
                                      bg_coords.view(1, -1, 1, 2), align_corners=True).view(-1)
            mask = occupancy > density_thresh_torso
            torso_alpha = torch.zeros([N, 1], device=device)
            torso_color = torch.zeros([N, 3], device=device)
            if mask.any():
                torso_alpha_mask, torso_color_mask, deform = self.forward_torso(bg_coords[mask], poses, enc_a, ind_code_torso)
                torso_alpha[mask] = torso_alpha_mask.float()
                torso_color[mask] = torso_color_mask.float()
                results["deform"] = deform
            bg_color = torso_color * torso_alpha + bg_color * (1 - torso_alpha)
            results["torso_alpha"] = torso_alpha
            results["torso_color"] = bg_color

        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        image = image.view(*prefix, 3).clamp(0, 1)
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        results["depth"] = depth.view(*prefix)
        results["image"] = image
        return results

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        # nerf/renderer.py:318-379: cells no training camera sees get density -1
        if not self.cuda_ray:
            return
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        B = poses.shape[0]
        fx, fy, cx, cy = intrinsic
        dev = self.density_bitfield.device
        axis = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
        count = torch.zeros_like(self.density_grid)
        poses = poses.to(count.device)
        for xs in axis:
            for ys in axis:
                for zs in axis:
                    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                    coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                    indices = raymarching.morton3D(coords).long()
                    world_xyzs = (2 * coords.float() / (self.grid_size - 1) - 1).unsqueeze(0)
                    for cas in range(self.cascade):
                        bound = min(2 ** cas, self.bound)
                        half_grid_size = bound / self.grid_size
                        cas_world_xyzs = world_xyzs * (bound - half_grid_size)
                        head = 0
                        while head < B:
                            tail = min(head + S, B)
                            cam_xyzs = cas_world_xyzs - poses[head:tail, :3, 3].unsqueeze(1)
                            cam_xyzs = cam_xyzs @ poses[head:tail, :3, :3]
                            mask_z = cam_xyzs[:, :, 2] > 0
                            mask_x = torch.abs(cam_xyzs[:, :, 0]) < cx / fx * cam_xyzs[:, :, 2] + half_grid_size * 2
                            mask_y = torch.abs(cam_xyzs[:, :, 1]) < cy / fy * cam_xyzs[:, :, 2] + half_grid_size * 2
                            mask = (mask_z & mask_x & mask_y).sum(0).reshape(-1)
                            count[cas, indices] += mask
                            head += S
        self.density_grid[count == 0] = -1

    def _grid_density(self, xyzs, enc_a, eye):
        """sigma at the occupancy-grid probe points (nerf/renderer.py:438: self.density(...)['sigma']).  On the GPU the
        2.1 M-point query goes through the fused network kernel (one launch instead of 2 grid encodes + 6 GEMMs + their
        glue per chunk; its colour branch runs on a dummy direction and is discarded) unless
        opt.grid_refresh_engine == "torch"."""
        if xyzs.is_cuda and getattr(self.opt, "grid_refresh_engine", "fused") == "fused":
            from . import fused
            if fused.supported(self):
                dirs = torch.zeros_like(xyzs)
                dirs[:, 2] = 1.0
                ind = self.individual_codes[0] if self.individual_dim > 0 else None
                return fused.network_forward(self, xyzs.contiguous(), dirs, enc_a, ind, eye, want_ambient=False)[0]
        return self.density(xyzs, enc_a, eye)["sigma"]

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        # nerf/renderer.py:383-499: refresh the 3-D occupancy grid (head) or the 2-D one (torso)
        if not self.cuda_ray:
            return
        dev = self.density_bitfield.device
        rand_idx = random.randint(0, self.aud_features.shape[0] - 1)
        auds = get_audio_features(self.aud_features, self.att, rand_idx).to(dev)
        enc_a = self.encode_audio(auds)

        if not self.torso:
            tmp_grid = torch.zeros_like(self.density_grid)
            eye = self.eye_area[[rand_idx]].to(dev) if self.exp_eye else None
            axis = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            for xs in axis:
                for ys in axis:
                    for zs in axis:
                        xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                        coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                        indices = raymarching.morton3D(coords).long()
                        xyzs = 2 * coords.float() / (self.grid_size - 1) - 1
                        for cas in range(self.cascade):
                            bound = min(2 ** cas, self.bound)
                            half_grid_size = bound / self.grid_size
                            cas_xyzs = xyzs * (bound - half_grid_size)
                            cas_xyzs += (torch.rand_like(cas_xyzs) * 2 - 1) * half_grid_size
                            sigmas = self._grid_density(cas_xyzs, enc_a, eye).reshape(-1).detach().to(tmp_grid.dtype)
                            sigmas *= self.density_scale
                            tmp_grid[cas, indices] = sigmas
            tmp_grid = raymarching.morton3D_dilation(tmp_grid)
            valid_mask = (self.density_grid >= 0) & (tmp_grid >= 0)
            self.density_grid[valid_mask] = torch.maximum(self.density_grid[valid_mask] * decay, tmp_grid[valid_mask])
            self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
            self.iter_density += 1
            density_thresh = min(self.mean_density, self.density_thresh)
            self.density_bitfield = raymarching.packbits(self.density_grid, density_thresh, self.density_bitfield)

        if self.torso:
            tmp_grid_torso = torch.zeros_like(self.density_grid_torso)
            rand_idx = random.randint(0, self.poses.shape[0] - 1)
            pose = convert_poses(self.poses[[rand_idx]]).to(dev)
            ind_code = self.individual_codes_torso[[rand_idx]] if self.opt.ind_dim_torso > 0 else None
            axis = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            half_grid_size = 1 / self.grid_size
            for xs in axis:
                for ys in axis:
                    xx, yy = torch.meshgrid(xs, ys, indexing="ij")
                    coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1)], dim=-1)
                    indices = (coords[:, 1] * self.grid_size + coords[:, 0]).long()  # xy transposed on purpose (:472)
                    xys = (2 * coords.float() / (self.grid_size - 1) - 1) * (1 - half_grid_size)
                    xys += (torch.rand_like(xys) * 2 - 1) * half_grid_size
                    alphas, _, _ = self.forward_torso(xys, pose, enc_a, ind_code)
                    tmp_grid_torso[indices] = alphas.squeeze(1).float()
            tmp_grid_torso = F.max_pool2d(tmp_grid_torso.view(1, 1, self.grid_size, self.grid_size), kernel_size=5,
                                          stride=1, padding=2).view(-1)
            self.density_grid_torso = torch.maximum(self.density_grid_torso * decay, tmp_grid_torso)
            self.mean_density_torso = torch.mean(self.density_grid_torso).item()

        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    def render(self, rays_o, rays_d, auds, bg_coords, poses, staged=False, max_ray_batch=4096, **kwargs):
        # nerf/renderer.py:504-537: with cuda_ray (always on) the frame is never staged
        return self.run_cuda(rays_o, rays_d, auds, bg_coords, poses, **kwargs)

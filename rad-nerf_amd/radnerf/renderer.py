"""Occupancy-grid renderer: the host side of the render path, with the reference's NeRFRenderer contract
(nerf/renderer.py:63-537) -- the `opt` fields it reads, the registered buffers / parameters and their names
(aabb_train, aabb_infer, density_grid, density_bitfield, density_grid_torso, step_counter, individual_codes[_torso],
camera_dR / camera_dT), `render(...) -> dict(image, depth, ...)`, `run_cuda(...)`, `update_extra_state`,
`mark_untrained_grid`, `reset_extra_state`.

How a frame is produced here:
  * inference, engine "fused" (radnerf/fused.py): one enqueue of the device-resident loop, torso pass and blend -- no host
    read-back anywhere;
  * inference, engine "ops": the reference's loop shape over the drop-in operators (march -> network -> composite -> compact,
    one host sync per iteration), torso layer through the fused torso kernel when the network has its shape;
  * training: march_rays_train -> network -> composite_rays_train with autograd through the HIP backward kernels; the torso
    layer gathers the covered pixels (index list from a HIP kernel), runs the PyTorch layers on them and copies the result
    back by index.
Occupancy-grid maintenance (`update_extra_state`, `mark_untrained_grid`) is radnerf/occupancy.py: kernels over the cells
in morton order, no Python block loops.
"""
import math
import random

import numpy as np
import torch
import torch.nn as nn

import raymarching

from . import occupancy
from .rays import convert_poses, get_audio_features

_LIP_SMOOTHING = 0.35     # weight of the previous frame's audio code (nerf/renderer.py:192)


def _n_step_policy(n_rays, n_alive):
    """Samples per live ray of one loop iteration: the fewer rays are left, the more steps each takes (renderer.py:249)."""
    return max(1, min(8, n_rays // n_alive))


class NeRFRenderer(nn.Module):
    def __init__(self, opt):
        # state and its registration order follow nerf/renderer.py:63-133 (seeded construction reproduces the reference's values)
        super().__init__()
        self.opt = opt
        self.bound, self.min_near = opt.bound, opt.min_near
        self.grid_size, self.density_scale = 128, 1
        self.cascade = 1 + math.ceil(math.log2(opt.bound))
        self.density_thresh, self.density_thresh_torso = opt.density_thresh, opt.density_thresh_torso
        self.exp_eye, self.test_train, self.smooth_lips = opt.exp_eye, opt.test_train, opt.smooth_lips
        self.torso, self.cuda_ray, self.train_camera = opt.torso, opt.cuda_ray, opt.train_camera
        self.engine = getattr(opt, "engine", "ops")

        half = opt.bound / 2        # the head sits in a slab half as high as the cube
        box = torch.tensor([-opt.bound, -half, -opt.bound, opt.bound, half, opt.bound], dtype=torch.float32)
        self.register_buffer("aabb_train", box)
        self.register_buffer("aabb_infer", box.clone())

        self.individual_num, self.individual_dim = opt.ind_num, opt.ind_dim
        if self.individual_dim > 0:
            self.individual_codes = nn.Parameter(torch.randn(self.individual_num, self.individual_dim) * 0.1)
        if self.torso:
            self.individual_dim_torso = opt.ind_dim_torso
            if self.individual_dim_torso > 0:
                self.individual_codes_torso = nn.Parameter(torch.randn(self.individual_num, self.individual_dim_torso) * 0.1)
        if self.train_camera:
            self.camera_dR = nn.Parameter(torch.zeros(self.individual_num, 3))
            self.camera_dT = nn.Parameter(torch.zeros(self.individual_num, 3))

        cells = self.cascade * self.grid_size ** 3
        self.register_buffer("density_grid", torch.zeros(self.cascade, self.grid_size ** 3))
        self.register_buffer("density_bitfield", torch.zeros(cells // 8, dtype=torch.uint8))
        if self.torso:
            self.register_buffer("density_grid_torso", torch.zeros(self.grid_size ** 2))
        self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
        self._mean_density = self._mean_density_torso = 0
        self._mean_density_dev = self._mean_density_torso_dev = None
        self.iter_density = self.mean_count = self.local_step = 0
        if self.smooth_lips:
            self.enc_a = None
        self.last_stats = None      # iterations / sample slots of the last frame (filled by the inference engines)
        self.count_samples = False  # the "ops" loop also counts live samples (costs a read-back)

    # mean densities: refreshed on the device, read by the host only when Python needs the number
    @property
    def mean_density(self):
        if self._mean_density_dev is not None:
            self._mean_density, self._mean_density_dev = float(self._mean_density_dev[0].item()), None
        return self._mean_density

    @mean_density.setter
    def mean_density(self, value):
        self._mean_density, self._mean_density_dev = value, None

    @property
    def mean_density_torso(self):
        if self._mean_density_torso_dev is not None:
            self._mean_density_torso, self._mean_density_torso_dev = float(self._mean_density_torso_dev[0].item()), None
        return self._mean_density_torso

    @mean_density_torso.setter
    def mean_density_torso(self, value):
        self._mean_density_torso, self._mean_density_torso_dev = value, None

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def reset_extra_state(self):
        # nerf/renderer.py:145-155
        if self.cuda_ray:
            self.density_grid.zero_()
            self.step_counter.zero_()
            self.mean_density = 0
            self.iter_density = self.mean_count = self.local_step = 0

    # ---------------------------------------------------------------------------------------------- audio code
    def fused_audio_enabled(self):
        """Inference with the fused engine also takes the audio nets off PyTorch (--emb ids and odd shapes stay there)."""
        if self.training or self.engine != "fused" or getattr(self.opt, "audio_engine", "fused") != "fused":
            return False
        ok = getattr(self, "_fused_audio_ok", None)
        if ok is None:
            from . import audio
            ok = self._fused_audio_ok = audio.supported(self)
        return ok

    def _fused_audio(self, auds):
        if auds is None or not auds.is_cuda or auds.dtype != torch.float32 or not self.fused_audio_enabled():
            return False
        return tuple(auds.shape) == (8 if self.att > 0 else 1, self.audio_in_dim, 16)

    def _audio_code(self, auds):
        """encode_audio, then the running blend of consecutive frames' codes when lips are smoothed (renderer.py:188-194)."""
        if self._fused_audio(auds):
            from . import audio
            code = audio.encode_windows(self, auds)
            return audio.smooth_(self, code) if self.smooth_lips else code
        code = self.encode_audio(auds)
        if code is None or not self.smooth_lips:
            return code
        if self.enc_a is not None:
            code = _LIP_SMOOTHING * self.enc_a + (1 - _LIP_SMOOTHING) * code
        self.enc_a = code
        return code

    # ---------------------------------------------------------------------------------------------- head
    def _head_inference_ops(self, rays_o, rays_d, nears, fars, enc_a, ind_code, eye, perturb, dt_gamma, max_steps, T_thresh):
        """The early-terminating loop in the reference's shape (renderer.py:227-262) over the drop-in operators."""
        n_rays, dev = rays_o.shape[0], rays_o.device
        acc = dict(weights_sum=torch.zeros(n_rays, device=dev), depth=torch.zeros(n_rays, device=dev),
                   image=torch.zeros(n_rays, 3, device=dev))
        alive = torch.arange(n_rays, dtype=torch.int32, device=dev)
        t = nears.clone()
        done = iterations = slots = 0
        live = torch.zeros((), dtype=torch.int64, device=dev) if self.count_samples else None
        while done < max_steps and alive.numel() > 0:
            n_alive = alive.numel()
            n_step = _n_step_policy(n_rays, n_alive)
            xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, alive, t, rays_o, rays_d, self.bound, self.density_bitfield,
                                                        self.cascade, self.grid_size, nears, fars, 128, perturb and done == 0,
                                                        dt_gamma, max_steps)
            sigmas, rgbs, _ = self(xyzs, dirs, enc_a, ind_code, eye)
            raymarching.composite_rays(n_alive, n_step, alive, t, self.density_scale * sigmas, rgbs, deltas, acc["weights_sum"],
                                       acc["depth"], acc["image"], T_thresh)
            alive = alive[alive >= 0]          # finished rays were marked -1 by the compositor
            done += n_step
            iterations += 1
            slots += xyzs.shape[0]
            if live is not None:
                live += (deltas[:, 0] > 0).sum()
        self.last_stats = {"iterations": iterations, "sample_slots": slots}
        if live is not None:
            self.last_stats["live_samples"] = int(live.item())
        return acc

    def _step_marcher_ok(self, rays_o, force_all_rays):
        """The one-launch marcher of a budgeted training step: a device-side budget is set, the rays fit one workgroup per CU."""
        import os
        if getattr(self, "_sample_budget", None) is None or force_all_rays or self.mean_count <= 0 or not rays_o.is_cuda:
            return False
        if os.environ.get("RN_TRAIN_MARCH", "step") != "step" or not torch.is_grad_enabled() or torch.is_autocast_enabled():
            return False
        from raymarching.ops import step_marcher_supported
        return step_marcher_supported(rays_o.shape[0], rays_o.device)

    def _fused_head_expected(self, rays_o, auds):
        """Will _head_network take the fused training kernels for this call?  (What train_head.usable() will say once the samples
        exist: they inherit device and dtype from the rays.)"""
        from .network import _train_head
        th = _train_head()
        return th is not None and auds is not None and th.usable(self, rays_o, auds)

    def _head_training(self, rays_o, rays_d, nears, fars, enc_a, ind_code, eye, perturb, force_all_rays, dt_gamma, max_steps, wait_for=None,
                       box=None, ind_index=None):
        """Train branch (renderer.py:206-223): every sample of every ray, packed; one sample counter per step (ring of 16)."""
        counter = getattr(self, "_static_counter", None)           # a captured training step counts into a fixed pair
        if counter is None:
            counter = self.step_counter[self.local_step % 16]
        budget = getattr(self, "_sample_budget", None)             # (device int32 budget, row capacity): see radnerf/train.py
        from .network import _train_glue, _train_head
        th = _train_head()
        if nears is None:
            # one launch: near / far, count, slices, samples, counters (set, not added to).  The fused network pass stops at the
            # counter, below which the launch has written every row -- no memset of the sample buffers for it.
            from raymarching.ops import march_rays_train_step
            self.local_step += 1
            fused_head = th is not None and th.usable(self, rays_o, enc_a)
            import os
            jitter = "hash" if (perturb and os.environ.get("RN_TRAIN_NOISE", "hash") == "hash") else perturb
            nears, fars, xyzs, dirs, deltas, rays = march_rays_train_step(rays_o, rays_d, box, self.min_near, self.bound, self.density_bitfield,
                                                                          self.cascade, self.grid_size, counter, budget[0], budget[1], jitter,
                                                                          dt_gamma, max_steps, not fused_head)
            return self._head_network(xyzs, dirs, deltas, rays, nears, fars, enc_a, ind_code, eye, counter, wait_for, ind_index)
        counter.zero_()
        self.local_step += 1
        if budget is not None and not force_all_rays and self.mean_count > 0:
            from raymarching.ops import march_rays_train_budget
            xyzs, dirs, deltas, rays = march_rays_train_budget(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                               self.grid_size, nears, fars, counter, budget[0], budget[1], perturb,
                                                               dt_gamma, max_steps)
        else:
            xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade,
                                                                    self.grid_size, nears, fars, counter, self.mean_count, perturb, 128,
                                                                    force_all_rays, dt_gamma, max_steps)
        return self._head_network(xyzs, dirs, deltas, rays, nears, fars, enc_a, ind_code, eye, counter, wait_for, ind_index)

    def _head_network(self, xyzs, dirs, deltas, rays, nears, fars, enc_a, ind_code, eye, counter, wait_for, ind_index=None):
        """The network over a step's samples + the training compositor (renderer.py:213-223)."""
        if wait_for is not None:                 # the audio code was computed on a side stream (run_cuda)
            main = torch.cuda.current_stream(xyzs.device)
            main.wait_stream(wait_for)
            if torch.is_tensor(enc_a):
                enc_a.record_stream(main)
        from .network import _train_glue, _train_head
        th = _train_head()
        if th is not None and th.usable(self, xyzs, enc_a):
            # one forward kernel for the network (+ |ambient| sum); the marcher's counter bounds the rows it visits
            sigmas, rgbs, ambient, ambient_abs = th.head_forward(self, xyzs, dirs, enc_a, ind_code, eye, m_dev=counter, ind_index=ind_index)
        else:
            if ind_index is not None:       # expected the fused kernels, got the operator chain after all: pick the row here
                ind_code = torch.index_select(self.individual_codes, 0, ind_index)
            sigmas, rgbs, ambient = self(xyzs, dirs, enc_a, ind_code, eye)
            glue = _train_glue()
            if glue is not None and ambient.dim() == 2 and ambient.shape[1] == 2 and glue.enabled(ambient):
                ambient_abs = glue.abs_sum2(ambient)
            else:
                ambient_abs = ambient.abs().sum(-1)
        if self.density_scale != 1:
            sigmas = self.density_scale * sigmas
        weights_sum, ambient_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, ambient_abs, deltas, rays)
        return dict(weights_sum=weights_sum, ambient=ambient_sum, depth=depth, image=image, nears=nears, fars=fars)

    # ---------------------------------------------------------------------------------------------- torso layer
    def _torso_layer(self, bg_coords, poses, enc_a, index, background, results):
        """Background with the 2-D torso layer blended over it (renderer.py:269-302); fills results[torso_*]."""
        n_px, dev = bg_coords.shape[0], bg_coords.device
        code = None
        if self.individual_dim_torso > 0:
            code = self.individual_codes_torso[index if self.training else 0]
        thresh = min(self.density_thresh_torso, self.mean_density_torso)
        if not torch.is_tensor(background):
            background = torch.full((n_px, 3), float(background), dtype=torch.float32, device=dev)
        background = background.reshape(-1, 3).float()
        if background.shape[0] != n_px:
            background = background.expand(n_px, 3)

        if not torch.is_grad_enabled() and bg_coords.is_cuda:
            from . import fused
            if fused.supported(self):
                out, alpha = fused.torso_forward(self, bg_coords, poses, code, thresh, bg_in=background.contiguous())
                results["torso_alpha"], results["torso_color"] = alpha, out
                return out
        # differentiable formulation: gather the covered pixels, PyTorch layers, copy back by index
        covered = occupancy.torso_pixels(self, bg_coords, thresh)
        alpha = torch.zeros(n_px, 1, dtype=torch.float32, device=dev)
        color = torch.zeros(n_px, 3, dtype=torch.float32, device=dev)
        if covered.numel() > 0:
            a, c, deform = self.forward_torso(bg_coords.index_select(0, covered), poses, enc_a, code)
            alpha = alpha.index_copy(0, covered, a.float())
            color = color.index_copy(0, covered, c.float())
            results["deform"] = deform
        out = color * alpha + background * (1 - alpha)
        results["torso_alpha"], results["torso_color"] = alpha, out
        return out

    # ---------------------------------------------------------------------------------------------- frame
    def run_cuda(self, rays_o, rays_d, auds, bg_coords, poses, eye=None, index=0, dt_gamma=0, bg_color=None,
                 perturb=False, force_all_rays=False, max_steps=1024, T_thresh=1e-4, **kwargs):
        # nerf/renderer.py:158-316.  rays_o, rays_d: [1,N,3]; auds: [8,C,16]; bg_coords: [1,N,2]; poses: [1,6]
        lead = rays_o.shape[:-1]
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        bg_coords = bg_coords.contiguous().view(-1, 2)
        if self.train_camera and (self.training or self.test_train):
            from .rays import euler_angles_to_matrix
            rays_o = rays_o + self.camera_dT[index]
            rays_d = rays_d @ euler_angles_to_matrix(self.camera_dR[index] / 180 * np.pi + 1e-8).squeeze(0)

        if not self.training and self.engine == "fused" and not perturb:
            from . import fused
            code = self.individual_codes[0] if self.individual_dim > 0 else None
            code_torso = self.individual_codes_torso[0] if (self.torso and self.individual_dim_torso > 0) else None
            # a caller that knows the coming frames' audio hands in this frame's smoothed code and bias block
            # (kwargs["audio_code"] = (enc_a [1,64], bias [192]); radnerf/parallel.py), else the per-frame kernels run
            given = kwargs.get("audio_code")
            enc_a, frame_bias = given if given is not None else (self._audio_code(auds), None)
            out = fused.render_frame(self, rays_o, rays_d, enc_a, code, eye, bg_coords, poses, code_torso, bg_color,
                                     dt_gamma, max_steps, T_thresh, want_u8=kwargs.get("want_u8", False),
                                     ray_source=kwargs.get("ray_source"), frame_bias=frame_bias)
            results = {k: out[k] for k in ("torso_alpha", "torso_color", "image_u8") if k in out}
            results["image"], results["depth"] = out["image"].view(*lead, 3), out["depth"].view(*lead)
            return results

        box = self.aabb_train if self.training else self.aabb_infer
        # training on the GPU: the audio nets (four latency-bound launches) run on a side stream beside near/far and the marcher,
        # which need nothing of them; the stream that owns the step waits just before the network kernel reads the code
        audio_side = None
        if self.training and rays_o.is_cuda and torch.is_grad_enabled() and auds is not None and auds.is_cuda:
            from .network import _train_head
            th = _train_head()
            if th is not None and th.overlap_enabled():
                audio_side = th.side_stream(rays_o.device, 1)
        # a step with a device-side sample budget (radnerf/train.py) marches in ONE launch that also intersects the rays with the
        # box and sets the step's counters (raymarching.ops.march_rays_train_step): no near/far launch here then
        one_launch = self.training and self._step_marcher_ok(rays_o, force_all_rays)
        if audio_side is not None:
            main = torch.cuda.current_stream(rays_o.device)
            audio_side.wait_stream(main)
            with torch.cuda.stream(audio_side):
                enc_a = self._audio_code(auds)
            nears, fars = (None, None) if one_launch else (
                t.detach() for t in raymarching.near_far_from_aabb(rays_o, rays_d, box, self.min_near))
        else:
            nears, fars = (None, None) if one_launch else (
                t.detach() for t in raymarching.near_far_from_aabb(rays_o, rays_d, box, self.min_near))
            enc_a = self._audio_code(auds)
        ind_code = ind_index = None
        if self.individual_dim > 0:
            if self.training and not isinstance(index, int):
                # index_select = the same rows as individual_codes[index] (nerf/renderer.py:199); its backward is one index_add
                # instead of index_put's sort + segmented scatter (5 launches for a one-element index)
                if torch.is_tensor(index):
                    idx = index
                else:       # the loader's Python list: uploaded once per distinct value, not once per step
                    key = tuple(int(i) for i in index)
                    cache = self.__dict__.setdefault("_index_cache", {})
                    idx = cache.get(key)
                    if idx is None or idx.device != self.individual_codes.device:
                        if len(cache) > 4096:
                            cache.clear()
                        idx = cache[key] = torch.as_tensor(key, dtype=torch.long, device=self.individual_codes.device)
                idx = idx.reshape(-1)
                if idx.numel() == 1 and idx.dtype == torch.int64 and idx.is_cuda and self._fused_head_expected(rays_o, auds):
                    ind_index = idx                 # the fused training kernels pick the row themselves (train_head.head_forward)
                else:
                    ind_code = torch.index_select(self.individual_codes, 0, idx.long())
            else:
                ind_code = self.individual_codes[index if self.training else 0]

        results = {}
        if self.training:
            head = self._head_training(rays_o, rays_d, nears, fars, enc_a, ind_code, eye, perturb, force_all_rays, dt_gamma, max_steps,
                                       wait_for=audio_side, box=box, ind_index=ind_index)
            nears, fars = head["nears"], head["fars"]
            results["weights_sum"], results["ambient"] = head["weights_sum"], head["ambient"]
        else:
            head = self._head_inference_ops(rays_o, rays_d, nears, fars, enc_a, ind_code, eye, perturb, dt_gamma, max_steps, T_thresh)

        background = 1 if bg_color is None else bg_color
        if self.training and kwargs.get("defer_blend") and not self.torso and torch.is_tensor(background):
            # the caller blends, clamps and takes the loss in one kernel (radnerf/train.py: train_head.head_loss)
            results["head_image"], results["background"] = head["image"], background
            return results
        if self.torso:
            background = self._torso_layer(bg_coords, poses, enc_a, index, background, results)
        elif torch.is_tensor(background):
            background = background.reshape(-1, 3)
        transmittance = (1 - head["weights_sum"]).unsqueeze(-1)
        results["image"] = (head["image"] + transmittance * background).view(*lead, 3).clamp(0, 1)
        results["depth"] = (torch.clamp(head["depth"] - nears, min=0) / (fars - nears)).view(*lead)
        return results

    def render(self, rays_o, rays_d, auds, bg_coords, poses, staged=False, max_ray_batch=4096, **kwargs):
        # nerf/renderer.py:504-537: with cuda_ray (always on in this fork) a frame is never staged
        return self.run_cuda(rays_o, rays_d, auds, bg_coords, poses, **kwargs)

    # ---------------------------------------------------------------------------------------------- occupancy grid
    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """Cells no training camera sees get density -1 (nerf/renderer.py:318-379).  `S` (the reference's block size) is
        accepted and unused: one kernel covers every cell, cascade and camera."""
        if self.cuda_ray:
            occupancy.mark_untrained(self, poses, intrinsic)

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, noise=None):
        """Refresh the occupancy grid the marchers skip through (nerf/renderer.py:383-499): the 3-D grid + bitfield when the
        head is trained, the 2-D grid when the torso is.  A random audio window (and eye value / head pose) conditions the
        query, as in the reference.  `noise`: optional uniform [0,1) numbers for the jitter of the probe points
        ([C*H^3, 3] head / [H^2, 2] torso) instead of the kernels' own hash -- parity tests pin it."""
        if not self.cuda_ray:
            return
        dev = self.density_bitfield.device
        pick = random.randint(0, self.aud_features.shape[0] - 1)
        enc_a = self.encode_audio(get_audio_features(self.aud_features, self.att, pick).to(dev))
        if self.torso:
            pick = random.randint(0, self.poses.shape[0] - 1)
            pose6 = convert_poses(self.poses[[pick]]).to(dev)
            code = self.individual_codes_torso[[pick]] if self.opt.ind_dim_torso > 0 else None
            self._mean_density_torso_dev = occupancy.refresh_torso(self, enc_a, pose6, code, decay, noise)
        else:
            eye = self.eye_area[[pick]].to(dev) if self.exp_eye else None
            self._mean_density_dev = occupancy.refresh_head(self, enc_a, eye, decay, noise)
            self.iter_density += 1
        # sample budget of the next steps: mean of the counters of the steps since the last refresh (renderer.py:496-499)
        steps = min(16, self.local_step)
        if steps > 0:
            self.mean_count = int(self.step_counter[:steps, 0].sum().item() / steps)
        self.local_step = 0

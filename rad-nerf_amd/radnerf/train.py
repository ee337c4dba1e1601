"""One training step of the render path (BASELINE config 2): the train branch of run_cuda
(march_rays_train -> NeRFNetwork.forward -> composite_rays_train, all over the HIP operators and their
backward kernels), the reference's loss and its Adam parameter groups.

Reference: Trainer.train_step (nerf/utils.py:718-806) for the loss, Trainer.train_one_epoch (:1003-1040) for
the order zero_grad -> step -> backward -> optimizer and the `update_extra_state` cadence
(`--update_extra_interval 16`, main.py:31), main.py:204 for Adam(betas=(0.9, 0.99), eps=1e-15) over
NeRFNetwork.get_params(lr, lr_net) (nerf/network.py:328-357).  The data loader, LPIPS, EMA, GradScaler and the
learning-rate schedule are outside the path (out of scope) and are not rebuilt.
"""
import torch


def make_optimizer(model, lr=5e-3, lr_net=5e-4, fused=None, capturable=False, kernel=None):
    """main.py:204; lr for the grid tables, lr_net for the MLPs / audio nets / individual codes.  On the GPU the update runs
    as ONE kernel over all tensors (HipAdam; `kernel="torch"` or RN_ADAM=torch: torch's fused Adam, one launch per parameter
    group) -- the 49 MB table is read and written once per step either way; same arithmetic as the default implementation."""
    import os
    on_gpu = next(model.parameters()).is_cuda
    if kernel is None:
        kernel = os.environ.get("RN_ADAM", "hip") if on_gpu else "torch"
    if kernel == "hip" and on_gpu:
        return HipAdam(model.get_params(lr, lr_net))
    if fused is None:
        fused = on_gpu
    return torch.optim.Adam(model.get_params(lr, lr_net), betas=(0.9, 0.99), eps=1e-15, fused=bool(fused),
                            capturable=bool(capturable))


class HipAdam:
    """torch.optim.Adam(betas, eps, weight_decay=0) over the model's parameter groups with ONE update kernel for all
    tensors (C ABI rn_adam_step, csrc/rn_train.hip) -- the table, its moments and its gradient stream through HBM once.
    Same interface as far as Trainer needs it (param_groups, zero_grad, step, state_dict / load_state_dict in
    torch.optim.Adam's format, so reference checkpoints carry over); the step counter lives on the device, so a step is
    capturable in a hipGraph."""

    # what torch.optim.Adam keeps in every param_group (torch/optim/adam.py): a state dict written here loads into the
    # reference's torch.optim.Adam (nerf/utils.py:1302-1426 saves / restores optimizer.state_dict())
    _TORCH_ADAM_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                                fused=None, decoupled_weight_decay=False)

    def __init__(self, params, betas=(0.9, 0.99), eps=1e-15):
        import ctypes as C
        import radnerf_hip as hip
        self._C, self._hip = C, hip

        class AdamTensorT(C.Structure):
            _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                        ("numel", C.c_uint32), ("lr", C.c_float)]
        self._T = AdamTensorT
        fn = hip._lib.rn_adam_step_lr
        fn.argtypes = [C.POINTER(AdamTensorT), C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        fn.restype = C.c_int
        self.param_groups = []
        for g in params:
            g = dict(g)
            g["params"] = [g["params"]] if torch.is_tensor(g["params"]) else list(g["params"])    # a bare tensor is one parameter
            if g.get("weight_decay", 0):
                raise ValueError("HipAdam: weight decay is not part of the reference's configuration (main.py:204) and not implemented")
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            for k, v in self._TORCH_ADAM_DEFAULTS.items():
                g.setdefault(k, v)
            g.setdefault("initial_lr", g["lr"])      # what torch's LR schedulers look for (LambdaLR, main.py:219)
            self.param_groups.append(g)
        self.betas, self.eps = betas, eps
        self.defaults = dict(lr=self.param_groups[0]["lr"], betas=betas, eps=eps, **self._TORCH_ADAM_DEFAULTS)
        dev = self.param_groups[0]["params"][0].device
        self.state = {p: {"exp_avg": torch.zeros_like(p, memory_format=torch.contiguous_format),
                          "exp_avg_sq": torch.zeros_like(p, memory_format=torch.contiguous_format)}
                      for g in self.param_groups for p in g["params"]}
        self._step = torch.zeros(1, dtype=torch.int32, device=dev)
        self._corr = torch.zeros(2, dtype=torch.float32, device=dev)
        n = sum(len(g["params"]) for g in self.param_groups)
        # learning rates live on the device, one per updated tensor in the order of the last step(): a captured step reads
        # them at run time, so a schedule that rewrites param_groups[i]["lr"] is followed by every replay (refresh_lr())
        self._lr_dev = torch.zeros(max(n, 1), dtype=torch.float32, device=dev)
        self._lr_groups, self._lr_sent, self._written = [], None, []

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def refresh_lr(self):
        """Upload the groups' current learning rates when they changed since the last upload (call outside a capture; a
        GraphedTrainer does so before every replay)."""
        lrs = [float(self.param_groups[gi]["lr"]) for gi in self._lr_groups]
        if lrs and lrs != self._lr_sent:
            self._lr_dev[:len(lrs)].copy_(torch.tensor(lrs, dtype=torch.float32))
            self._lr_sent = lrs

    @torch.no_grad()
    def step(self):
        import sys
        th = sys.modules.get("radnerf.train_head")
        if th is not None:                       # table-gradient scatters still running on the side stream (train_head.deferred_join)
            pending = th.take_pending_events()
            if pending:
                held = {p.grad.data_ptr() for g in self.param_groups for p in g["params"] if p.grad is not None}
                for ev, *ptrs in pending:
                    torch.cuda.current_stream().wait_event(ev)
                    if not all(q in held for q in ptrs):
                        # autograd copied a table gradient instead of keeping the buffer the scatter writes (a parameter that
                        # already had a .grad): that copy raced with the side stream
                        raise RuntimeError("HipAdam: a table gradient was copied before its scatter had finished; use "
                                           "zero_grad(set_to_none=True) or RN_TRAIN_OVERLAP=0")
        entries, keep, written, groups = [], [], [], []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is None:
                    continue
                written.append(p)
                groups.append(gi)
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous() or p.dtype != torch.float32 or grad.dtype != torch.float32:
                    raise RuntimeError("HipAdam: parameters and gradients must be contiguous fp32")
                st = self.state[p]
                keep.append(grad)
                entries.append((p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), float(g["lr"])))
        arr = (self._T * max(len(entries), 1))()
        for i, e in enumerate(entries):
            arr[i].param, arr[i].grad, arr[i].exp_avg, arr[i].exp_avg_sq, arr[i].numel, arr[i].lr = e
        if groups != self._lr_groups:
            self._lr_groups, self._lr_sent = groups, None
        if not torch.cuda.is_current_stream_capturing():
            self.refresh_lr()
        elif self._lr_sent is None:
            raise RuntimeError("HipAdam: take one eager step (or call refresh_lr()) before capturing a step in a graph")
        self._hip.call("rn_adam_step_lr", arr, len(entries), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                       self._hip.ptr(self._step), self._hip.ptr(self._corr), self._hip.ptr(self._lr_dev), self._hip.stream())
        self._written = written
        self._hip.mark_written(written)      # the kernel wrote through raw pointers: caches keyed on tensor versions must see it

    def state_dict(self):
        """torch.optim.Adam's layout: state by running parameter index, one `step` per tensor."""
        step = self._step.to(torch.float32).reshape(()).clone()
        state, groups, at = {}, [], 0
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                state[at] = {"step": step.clone(), "exp_avg": self.state[p]["exp_avg"], "exp_avg_sq": self.state[p]["exp_avg_sq"]}
                ids.append(at)
                at += 1
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        params = [p for g in self.param_groups for p in g["params"]]
        steps = []
        for i, p in enumerate(params):
            st = sd["state"].get(i)
            if st is None:
                continue
            self.state[p]["exp_avg"].copy_(st["exp_avg"])
            self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
            steps.append(int(st["step"]))
        if steps:
            self._step.fill_(max(steps))
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            g["lr"] = sg.get("lr", g["lr"])


def entropy_of(alphas):
    """nerf/utils.py:785-794: binary entropy that pushes an opacity towards 0 or 1."""
    a = alphas.clamp(1e-5, 1 - 1e-5)
    return -a * torch.log2(a) - (1 - a) * torch.log2(1 - a)


def train_step(model, data, opt, global_step=0, iters=200000, lambda_amb=0.1, amb_weight=None):
    """-> (pred_rgb, target_rgb, loss); `data` has the keys of the reference's loader batch
    (nerf/provider.py:588-690): rays_o, rays_d [B,N,3], bg_coords [1,N,2], poses [B,6], face_mask [B,N],
    eye [B,1], auds, index, bg_color [B,N,3], images (head) or bg_torso_color (torso) [B,N,3]."""
    torso = bool(opt.torso)
    rgb = data["bg_torso_color"] if torso else data["images"]
    import os
    fused_loss = (not torso and rgb.is_cuda and rgb.dtype == torch.float32 and os.environ.get("RN_TRAIN_LOSS", "fused") == "fused"
                  and torch.is_tensor(data.get("bg_color")) and data["bg_color"].dtype == torch.float32)
    out = model.render(data["rays_o"], data["rays_d"], data["auds"], data["bg_coords"], data["poses"], eye=data["eye"],
                       index=data["index"], staged=False, bg_color=data["bg_color"], perturb=True, force_all_rays=False,
                       dt_gamma=opt.dt_gamma, max_steps=opt.max_steps, defer_blend=fused_loss)
    if "head_image" in out:
        # blend over the background (nerf/renderer.py:306), clamp and the loss below with their gradients: ONE kernel
        from . import train_head
        face = data["face_mask"]
        face = face if face.dtype == torch.float32 else face.float()
        w = amb_weight if amb_weight is not None else torch.full((1,), min(global_step / iters, 1.0) * lambda_amb, device=rgb.device)
        loss, pred = train_head.head_loss(out["head_image"], out["weights_sum"], out["ambient"], out["background"], rgb, face, w)
        return pred.view(rgb.shape), rgb, loss
    pred = out["torso_color"] if torso else out["image"]
    if not torso and pred.is_cuda:
        from . import train_glue
        face = data["face_mask"].reshape(-1)
        if train_glue.enabled(pred, rgb, out["weights_sum"], out["ambient"]):
            w = amb_weight if amb_weight is not None else torch.full((1,), min(global_step / iters, 1.0) * lambda_amb, device=pred.device)
            loss = train_glue.train_loss(pred, rgb, out["weights_sum"], out["ambient"], face if face.dtype == torch.float32 else face.float(), w)
            return pred, rgb, loss
    loss = torch.nn.functional.mse_loss(pred, rgb, reduction="none").mean(-1).mean()
    if torso:
        loss = loss + 1e-4 * entropy_of(out["torso_alpha"]).mean()
    else:
        loss = loss + 1e-4 * entropy_of(out["weights_sum"]).mean()
        # ambient coordinates should stay put outside the face (nerf/utils.py:796-803), weight ramped over `iters`
        face = data["face_mask"].reshape(-1)
        loss_amb = (out["ambient"] * ((~face) if face.dtype == torch.bool else (1.0 - face))).mean()   # a 0/1 float mask is the same product
        # amb_weight: the same factor as a device scalar (a captured step reads it instead of baking the Python float in)
        loss = loss + (amb_weight if amb_weight is not None else min(global_step / iters, 1.0) * lambda_amb) * loss_amb
    return pred, rgb, loss


def _backward(loss):
    """loss.backward(); a loss of the fused head-loss kernel hands its ready-made input gradients to autograd directly."""
    if getattr(loss, "_rn_direct", None) is not None:
        from . import train_head
        train_head.backward(loss)
    else:
        loss.backward()


class SyntheticTrainStream:
    """Batches of `n_rays` random pixels of a SyntheticScene frame with the scene's own frozen render as target
    (SURVEY 8(d) config 2); everything stays on the device."""

    def __init__(self, scene, n_rays=4096, frame=0, seed=0):
        self.scene, self.n_rays, self.frame = scene, n_rays, frame
        self.gen = torch.Generator(device=scene.device).manual_seed(seed)
        m = scene.model
        was_training = m.training
        m.eval()
        with torch.no_grad():
            out = scene.render(frame)
        m.train(was_training)
        self.f = scene.frame(frame)
        self.target = out["image"].reshape(1, -1, 3).clamp(0, 1).detach().clone()
        # "face" = pixels the head layer changed (the loader's face_mask comes from a parsing net, provider.py:199-215)
        self.face_mask = (self.target - self.f["bg_color"].reshape(1, -1, 3)).abs().sum(-1) > 1e-3
        # what update_extra_state samples from (main.py:183-186 hands the loader's arrays to the model)
        m.aud_features, m.poses = scene.aud_features, scene.poses
        m.eye_area = torch.full((scene.n_frames, 1), 0.25, device=scene.device)
        self._table = None

    def batch(self):
        """One gather for the whole batch: the per-pixel arrays are kept side by side in one [n_px, 15] table, the batch's
        tensors are views of the gathered rows (key `_packed`: GraphedTrainer copies that one tensor into its static input)."""
        f, n_px = self.f, self.target.shape[1]
        if self._table is None:
            cols = [f["rays_o"][0], f["rays_d"][0], f["bg_coords"][0], f["bg_color"][0], self.target[0], self.face_mask[0].float().unsqueeze(-1)]
            self._table = torch.cat([c.reshape(n_px, -1).float() for c in cols], dim=1).contiguous()
        idx = torch.randint(0, n_px, (self.n_rays,), device=self.target.device, generator=self.gen)
        import os
        if self._table.is_cuda and os.environ.get("RN_TRAIN_PACKED", "1") != "0":
            # one kernel: the picked rows, every column section written as its own contiguous array of one flat buffer
            from . import train_head
            flat, _ = train_head.batch_gather(self._table, idx, self._WIDTHS)
            return self.unpack(flat)
        rows = self._table.index_select(0, idx)
        flat = torch.cat([rows[:, a:a + w].reshape(-1) for a, w in zip(self._COL0, self._WIDTHS)])
        out = self.unpack(flat)
        if os.environ.get("RN_TRAIN_PACKED", "1") == "0":       # experiment switch: separate tensors, as a generic loader would hand over
            out = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items() if not k.startswith("_")}
        return out

    _WIDTHS = (3, 3, 2, 3, 3, 1)          # rays_o | rays_d | bg_coords | bg_color | target | face: the table's column sections
    _COL0 = (0, 3, 6, 8, 11, 14)

    def unpack(self, flat):
        """Batch dict over the sections of `flat` ([n, 3] rays_o | [n, 3] rays_d | [n, 2] bg_coords | [n, 3] bg_color | [n, 3] target |
        [n] face, each contiguous): every per-ray entry is a VIEW (face_mask stays 0/1 floats), so refreshing `flat` in place
        refreshes the batch."""
        n = flat.numel() // sum(self._WIDTHS)
        sec, at = [], 0
        for w in self._WIDTHS:
            sec.append(flat[at:at + n * w].view(1, n, w))
            at += n * w
        f = self.f
        return dict(rays_o=sec[0], rays_d=sec[1], bg_coords=sec[2], poses=f["poses"], face_mask=sec[5].view(1, n),
                    eye=f["eye"], auds=f["auds"], index=[self.frame], bg_color=sec[3], images=sec[4],
                    bg_torso_color=sec[4], _packed=flat, _unpack=self.unpack)


class Trainer:
    """zero_grad -> train_step -> backward -> Adam, with the occupancy grid refreshed every
    `update_extra_interval` steps as the reference's loop does under --cuda_ray (nerf/utils.py:1015-1018)."""

    def __init__(self, model, opt, lr=5e-3, lr_net=5e-4, update_extra_interval=16, iters=200000, lambda_amb=0.1,
                 prefer_rocblas=True, capturable=False):
        # The weight gradients of the 64-wide MLPs are [64 x ~40 000] x [~40 000 x 96] products: all reduction, almost
        # no output.  On this stack hipBLASLt runs them without a K split (132 us each, measured), rocBLAS in 37 us;
        # eight of them per step make that the largest single item of the step.
        if prefer_rocblas and torch.cuda.is_available() and getattr(torch.version, "hip", None):
            torch.backends.cuda.preferred_blas_library("cublas")      # "cublas" selects rocBLAS on ROCm builds
        self.model, self.opt = model, opt
        self.optimizer = make_optimizer(model, lr, lr_net, capturable=capturable)
        self.update_extra_interval, self.iters, self.lambda_amb = update_extra_interval, iters, lambda_amb
        self.global_step = 0

    def step(self, data):
        m = self.model
        m.train()
        if self.update_extra_interval and self.global_step % self.update_extra_interval == 0:
            with torch.no_grad():
                m.update_extra_state()
        self.global_step += 1
        self.optimizer.zero_grad(set_to_none=True)
        m._sample_budget = self._device_budget(m)
        try:
            with _join_in_optimizer(self.optimizer):
                _, _, loss = train_step(m, data, self.opt, self.global_step, self.iters, self.lambda_amb)
                _backward(loss)
                self.optimizer.step()
        finally:
            m._sample_budget = None
        return loss.detach()

    def _device_budget(self, m):
        """(device int32 budget, row capacity) for the renderer's one-launch marcher, or None (first window, CPU).  The budget is
        raymarching.py:226-229's aligned running average; the device copy is rewritten only when it moves (every 16 steps)."""
        dev = next(m.parameters()).device
        if m.mean_count <= 0 or dev.type != "cuda":
            return None
        budget = int(m.mean_count)
        budget += 128 - budget % 128
        held = getattr(self, "_budget_held", None)
        if held is None or held[0] != budget or held[1].device != dev:
            t = held[1] if held is not None and held[1].device == dev else torch.zeros(1, dtype=torch.int32, device=dev)
            t.fill_(budget)
            self._budget_held = held = (budget, t)
        return held[1], budget


class _join_in_optimizer:
    """HipAdam.step waits for the table-gradient scatter the backward pass left on a side stream (train_head.deferred_join); any
    other optimizer gets gradients that are complete when backward() returns."""

    def __init__(self, optimizer):
        self.ctx = None
        if isinstance(optimizer, HipAdam) and torch.cuda.is_available():
            from . import train_head
            self.ctx = train_head.deferred_join()

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


class GraphedTrainer(Trainer):
    """Trainer.step with the steady-state step replayed from a hipGraph (torch.cuda.CUDAGraph): forward, loss, backward and the
    fused Adam update are ~320 launches whose enqueue costs the host more than the GPU needs to run them; replaying them as
    one graph makes the step GPU-bound and its duration repeatable.

    What stays outside the graph: the occupancy refresh every `update_extra_interval` steps, the batch (copied into the
    graph's static input buffers), the ring of per-step sample counters (the captured step counts into one static pair,
    copied to the ring afterwards) and the first window, whose marcher reads back its sample count (raymarching.py:249-255)
    and therefore runs eagerly.  The refresh changes `mean_count`, the marcher's sample budget M (raymarching.py:226-229): the
    graph is captured for a row CAPACITY (the budget rounded up to a multiple of `capacity_step`, kept while the budget stays
    within it) and the budget itself is a device scalar the marcher reads (rn_march_rays_train_budget), so a refresh changes a
    number, not the graph; rows between budget and capacity stay zero and belong to no ray.  A new graph is captured only
    when the budget leaves the capacity window for a capacity not seen before (the last `max_graphs` graphs are kept: the
    budget of a model whose density depends on the audio swings by +-20 % between refreshes).  Same arithmetic as Trainer.step."""

    def __init__(self, model, opt, capacity_step=4096, max_graphs=8, **kw):
        super().__init__(model, opt, capturable=True, **kw)
        self.capacity_step = int(capacity_step)
        self.max_graphs = int(max_graphs)
        self._capacity = 0
        self._graphs = {}              # key -> (graph, loss): a budget that comes back to an earlier capacity replays that graph
        self._graph = self._static = self._loss = self._key = None
        self._static_index = None
        dev = next(model.parameters()).device
        self._amb_weight = torch.zeros((), dtype=torch.float32, device=dev)
        self._counter = torch.zeros(2, dtype=torch.int32, device=dev)
        self._budget = torch.zeros(1, dtype=torch.int32, device=dev)
        self.captures = self.replays = 0
        self.capture_log = []          # (step, budget, capacity) of every capture

    def _capture(self, data, key):
        m = self.model
        self._loss = None
        if self._static is None:
            if "_packed" in data:        # a batch that is one table of rows: one static tensor, the step's inputs are its views
                self._static = dict(data["_unpack"](data["_packed"].clone()))
            else:
                self._static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
            if isinstance(self._static.get("index"), (list, tuple)):     # a Python list would be uploaded inside the capture
                self._static_index = list(self._static["index"])
                self._static["index"] = torch.tensor(self._static["index"], dtype=torch.long, device=self._counter.device)
        from raymarching.ops import step_marcher_prepare, step_marcher_supported
        n_rays = int(self._static["rays_o"].reshape(-1, 3).shape[0])
        if step_marcher_supported(n_rays, self._counter.device):
            step_marcher_prepare(n_rays, self._counter.device)       # persistent state: must not be born inside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        m._static_counter = self._counter                     # renderer._head_training counts into this pair while captured
        m._sample_budget = (self._budget, self._capacity)
        try:
            with torch.cuda.graph(g), _join_in_optimizer(self.optimizer):   # a private memory pool per graph: cached graphs never alias each other
                _, _, loss = train_step(m, self._static, self.opt, amb_weight=self._amb_weight)
                _backward(loss)
                self.optimizer.step()
        finally:
            m._static_counter = None
            m._sample_budget = None
        m.local_step -= 1                                     # the capture pass went through the Python bookkeeping once
        self._graph, self._loss, self._key = g, loss, key
        self._graphs[key] = (g, loss)
        while len(self._graphs) > self.max_graphs:           # drop the oldest (dicts keep insertion order); never the newest
            del self._graphs[next(iter(self._graphs))]
        self.captures += 1
        self.capture_log.append((self.global_step, int(self._budget.item()), self._capacity))

    def step(self, data):
        m = self.model
        m.train()
        if self.update_extra_interval and self.global_step % self.update_extra_interval == 0:
            with torch.no_grad():
                m.update_extra_state()
        self.global_step += 1
        if m.mean_count <= 0 or not next(m.parameters()).is_cuda:        # first window / CPU: the eager step
            self.optimizer.zero_grad(set_to_none=True)
            with _join_in_optimizer(self.optimizer):
                _, _, loss = train_step(m, data, self.opt, self.global_step, self.iters, self.lambda_amb)
                _backward(loss)
                self.optimizer.step()
            return loss.detach()
        budget = int(m.mean_count)
        budget += 128 - budget % 128                                     # raymarching.py:226-229 (align = 128)
        step = self.capacity_step
        if not (budget <= self._capacity <= budget + 2 * step):         # keep the capacity while the budget stays inside its window
            self._capacity = -(-(budget + step // 4) // step) * step
        if budget != getattr(self, "_budget_value", None):               # the running average moves every 16 steps, not every step
            self._budget.fill_(budget)
            self._budget_value = budget
        key = (self._capacity, tuple(data["rays_o"].shape))
        if key != self._key:
            hit = self._graphs.pop(key, None)
            if hit is not None:                                  # seen before: replay that graph (and mark it most recent)
                self._graphs[key] = hit
                self._graph, self._loss, self._key = hit[0], hit[1], key
            else:
                self._capture(data, key)
        if "_packed" in data and "_packed" in self._static:
            self._static["_packed"].copy_(data["_packed"])
            for k in ("poses", "eye", "auds"):
                if data[k] is not self._static[k]:
                    self._static[k].copy_(data[k])
        else:
            for k, v in data.items():
                if torch.is_tensor(v) and k != "_packed":
                    self._static[k].copy_(v)
        v = data.get("index")
        if isinstance(v, (list, tuple)) and list(v) != self._static_index:
            self._static["index"].copy_(torch.tensor(v, dtype=torch.long))
            self._static_index = list(v)
        w_amb = min(self.global_step / self.iters, 1.0) * self.lambda_amb
        if w_amb != getattr(self, "_amb_value", None):                   # constant once the ramp is over
            self._amb_weight.fill_(w_amb)
            self._amb_value = w_amb
        if isinstance(self.optimizer, HipAdam):
            self.optimizer.refresh_lr()             # a schedule may have rewritten param_groups[i]["lr"] since the last step
        self._graph.replay()
        if isinstance(self.optimizer, HipAdam):     # the replayed update went through raw pointers (see HipAdam.step)
            self.optimizer._hip.mark_written(self.optimizer._written)
        m.step_counter[m.local_step % 16].copy_(self._counter)
        m.local_step += 1
        self.replays += 1
        return self._loss.detach()

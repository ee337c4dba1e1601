"""GPU parity of the fused training pass of the per-sample network (csrc/rn_train_head.hip, include/radnerf_train.h) against the
per-operator path it replaces: NeRFNetwork.forward (nerf/network.py:222-283) over grid_encode / MLP / activation operators with
torch.autograd, which tests/test_gpu_mlp_train.py, tests/test_gpu_ops.py and the reference-generated gradients of
tests/test_golden_frames.py pin.  Also: the line-keyed table scatter against the operator's scatter, the one-kernel head loss
against the PyTorch expression of nerf/utils.py:772-803, and the launch count of a whole training step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(size=32, **kw):
    from radnerf.scene import SyntheticScene, default_opt
    return SyntheticScene(H=size, W=size, n_frames=8, device="cuda", opt=default_opt(engine="ops", torso=False, smooth_lips=False, **kw))


def _samples(M, seed, oob=7):
    g = torch.Generator(device="cuda").manual_seed(seed)
    xyzs = (torch.rand(M, 3, device="cuda", generator=g) * 2 - 1) * 0.98
    xyzs[:oob] = 1.25                     # outside [-bound, bound]: zero features, no table gradient (gridencoder.cu:110-135)
    dirs = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda", generator=g), dim=-1)
    enc_a = torch.randn(1, 64, device="cuda", generator=g) * 0.5
    eye = torch.full((1, 1), 0.25, device="cuda")
    return xyzs, dirs, enc_a, eye, g


def _stable_samples(m, xyzs, dirs, enc_a, eye, ind, monkeypatch, cell_margin=2e-5, relu_margin=2e-5):
    """Samples at which NeRFNetwork.forward is smooth in its parameters: no ambient coordinate within `cell_margin` (normalised
    units) of a cell boundary of any level of the 2-D grid, no hidden pre-activation within `relu_margin` of 0 (recomputed here
    with plain torch matmuls over the module's encoders)."""
    monkeypatch.setenv("RN_TRAIN_HEAD", "ops")
    with torch.no_grad():
        n = xyzs.shape[0]
        enc_x = m.encoder(xyzs, bound=m.bound)
        pre = []

        def mlp(net, x):
            for i, layer in enumerate(net.net):
                x = x @ layer.weight.t()
                if i != len(net.net) - 1:
                    pre.append(x)
                    x = torch.relu(x)
            return x
        amb = torch.tanh(mlp(m.ambient_net, torch.cat([enc_x, enc_a.repeat(n, 1)], -1)))
        enc_w = m.encoder_ambient(amb, bound=1)
        h = mlp(m.sigma_net, torch.cat([enc_x, enc_w, eye.repeat(n, 1)], -1))
        mlp(m.color_net, torch.cat([m.encoder_dir(dirs), h[:, 1:], ind.reshape(1, -1).repeat(n, 1)], -1))
        relu_ok = torch.stack([(z.abs() > relu_margin).all(-1) for z in pre]).all(0)
        enc = m.encoder_ambient
        scales = torch.tensor([2.0 ** (l * float(np.log2(enc.per_level_scale))) * enc.base_resolution - 1 for l in range(16)],
                              dtype=torch.float64, device=xyzs.device)
        pos = ((amb.double() + 1) / 2).unsqueeze(-1) * scales + 0.5              # [M, 2, 16]
        frac = pos - pos.floor()
        margin = cell_margin * scales
        cell_ok = ((frac > margin) & (frac < 1 - margin)).all(-1).all(-1)
    return relu_ok & cell_ok


def _run(m, xyzs, dirs, enc_a, eye, index, up, mode, monkeypatch):
    """forward + backward of NeRFNetwork.forward (+ |ambient| sum) under upstream gradients `up`; returns outputs and gradients."""
    monkeypatch.setenv("RN_TRAIN_HEAD", mode)
    for p in m.parameters():
        p.grad = None
    enc_a = enc_a.clone().requires_grad_(True)
    eye = eye.clone().requires_grad_(True)
    ind = m.individual_codes[index]
    sigma, rgb, amb = m(xyzs, dirs, enc_a, ind, eye)
    amb_abs = amb.abs().sum(-1)
    loss = (sigma * up[0]).sum() + (rgb * up[1]).sum() + (amb_abs * up[2]).sum() + (amb * up[3]).sum()
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    grads["enc_a"], grads["eye"] = enc_a.grad.clone(), eye.grad.clone()
    return (sigma.detach(), rgb.detach(), amb.detach()), grads


@pytest.mark.parametrize("ref_mlp", ["hip", "torch"])
@pytest.mark.parametrize("grid,M", [("tiledgrid16", 4099), ("hashgrid19", 12345)])
def test_fused_head_matches_the_operator_path(hiplib, monkeypatch, grid, M, ref_mlp):
    """ref_mlp = "torch": the comparison path is the reference's own formulation of the network -- nn.Linear / F.relu / tanh /
    trunc_exp / sigmoid / cat / repeat under torch.autograd (nerf/network.py:222-283) over the grid and SH operators -- with
    none of this tree's MLP or glue kernels in it; "hip": the round-2 path (rn_mlp64_* + glue kernels)."""
    if ref_mlp == "torch":
        monkeypatch.setenv("RN_MLP_TRAIN", "torch")
        monkeypatch.setenv("RN_TRAIN_GLUE", "torch")
    kw = dict(xyz_grid="hashgrid", xyz_log2_hashmap_size=19) if grid == "hashgrid19" else {}
    scene = _scene(32, **kw)
    m = scene.model
    m.train()
    from radnerf import train_head
    assert train_head.supported(m)
    xyzs, dirs, enc_a, eye, g = _samples(M, 3)
    up = [torch.randn(M, device="cuda", generator=g), torch.randn(M, 3, device="cuda", generator=g),
          torch.randn(M, device="cuda", generator=g) * 0.3, torch.randn(M, 2, device="cuda", generator=g) * 0.3]
    # Two things make a per-sample gradient discontinuous: the derivative of the 2-D grid with respect to the ambient coordinate
    # is piecewise constant (a coordinate within rounding distance of a cell boundary of some level may take the neighbouring
    # cell's derivative in one of the two paths, whose ambient outputs differ by ~1e-6), and so is ReLU's (a hidden unit whose
    # pre-activation is within rounding distance of 0).  Upstream of the 2-D grid a contribution carries that grid's derivative
    # (~2047 x table differences) and the contributions largely cancel in the sum, so ONE such sample moves a gradient by ~1 % of its
    # maximum.  Those samples (~45 %: 16 levels x 2 dimensions of cell boundaries) get zero upstream gradient in BOTH runs; every
    # remaining contribution is smooth and all gradients are compared at the tight tolerance.
    stable = _stable_samples(m, xyzs, dirs, enc_a, eye, m.individual_codes[3], monkeypatch)
    assert 0.4 < float(stable.float().mean()) < 1.0
    up = [u * (stable.float() if u.dim() == 1 else stable.float().unsqueeze(-1)) for u in up]
    out_ops, g_ops = _run(m, xyzs, dirs, enc_a, eye, 3, up, "ops", monkeypatch)
    out_fused, g_fused = _run(m, xyzs, dirs, enc_a, eye, 3, up, "fused", monkeypatch)
    # forward: the inference kernel's arithmetic (sigma rel 2e-4, rgb / ambient abs 2e-5: DESIGN 3)
    assert torch.allclose(out_fused[0], out_ops[0], rtol=2e-4, atol=1e-6)
    assert torch.allclose(out_fused[1], out_ops[1], rtol=0, atol=2e-5)
    assert torch.allclose(out_fused[2], out_ops[2], rtol=0, atol=2e-5)
    assert set(g_fused) == set(g_ops)
    for name in sorted(g_ops):
        a, b = g_fused[name], g_ops[name]
        assert a.shape == b.shape, name
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max()) / scale
        cos = float(torch.nn.functional.cosine_similarity(a.reshape(1, -1).double(), b.reshape(1, -1).double()))
        assert err < 2e-3 and cos > 0.99999, (name, err, cos)


def test_fused_head_live_count_bounds_the_rows(hiplib, monkeypatch):
    """Rows past the device-side live count get no output and contribute no gradient (the zero rows behind the marcher's
    counter, raymarching/raymarching.py:231-257)."""
    monkeypatch.setenv("RN_TRAIN_HEAD_ZERO", "1")          # zero-filled output block, so that the untouched rows can be told
    scene = _scene(32)
    m = scene.model
    m.train()
    from radnerf import train_head
    M, live = 4096, 1500
    xyzs, dirs, enc_a, eye, g = _samples(M, 5)
    ind = m.individual_codes[0]

    def run(x, d, m_dev):
        for p in m.parameters():
            p.grad = None
        s, c, a, aa = train_head.head_forward(m, x, d, enc_a, ind, eye, m_dev=m_dev)
        n = live
        ((s[:n] ** 2).sum() + (c[:n] ** 2).sum() + aa[:n].sum()).backward()
        return (s.detach().clone(), c.detach().clone(), aa.detach().clone()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    cnt = torch.tensor([live, 0], dtype=torch.int32, device="cuda")
    out_a, g_a = run(xyzs, dirs, cnt)
    out_b, g_b = run(xyzs[:live].contiguous(), dirs[:live].contiguous(), None)
    assert torch.equal(out_a[0][:live], out_b[0]) and torch.equal(out_a[1][:live], out_b[1])
    assert float(out_a[0][live:].abs().max()) == 0.0 and float(out_a[1][live:].abs().max()) == 0.0
    for name in g_b:
        scale = float(g_b[name].abs().max()) + 1e-12
        assert float((g_a[name] - g_b[name]).abs().max()) / scale < 1e-4, name


def test_individual_code_row_picked_on_the_device(hiplib):
    """head_forward(ind_index=...) == head_forward(ind_code=individual_codes[index]): same outputs, same gradients, and the gradient of
    individual_codes comes back whole -- the picked row filled, zeros everywhere else -- from the constants' launch (what
    index_select's backward builds with a memset and an index_add)."""
    scene = _scene(32)
    m = scene.model
    m.train()
    from radnerf import train_head
    M = 3000
    xyzs, dirs, enc_a, eye, gen = _samples(M, 9)
    g = [torch.randn(M, device="cuda", generator=gen), torch.randn(M, 3, device="cuda", generator=gen), torch.randn(M, device="cuda", generator=gen)]
    with torch.no_grad():
        m.individual_codes.normal_(0, 0.3)
    row = 5
    idx = torch.tensor([row], dtype=torch.int64, device="cuda")

    def run(**how):
        for p in m.parameters():
            p.grad = None
        s, c, a, aa = train_head.head_forward(m, xyzs, dirs, enc_a, how.get("ind_code"), eye, ind_index=how.get("ind_index"))
        ((s * g[0]).sum() + (c * g[1]).sum() + (aa * g[2]).sum()).backward()
        return (s.detach().clone(), c.detach().clone(), aa.detach().clone()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    out_a, g_a = run(ind_index=idx)
    out_b, g_b = run(ind_code=torch.index_select(m.individual_codes, 0, idx))
    for a, b in zip(out_a, out_b):
        assert torch.equal(a, b)
    ga, gb = g_a["individual_codes"], g_b["individual_codes"]
    assert ga.shape == m.individual_codes.shape and torch.equal(ga, gb) and float(ga[row].abs().max()) > 0
    assert float(ga.abs().sum()) == float(ga[row].abs().sum())            # every other row is exactly zero
    assert g_a.keys() == g_b.keys()
    for name in g_b:
        if "embeddings" in name:                 # the tables' scatter adds in another order from launch to launch
            scale = float(g_b[name].abs().max()) + 1e-12
            assert float((g_a[name] - g_b[name]).abs().max()) / scale < 1e-5, name
        else:
            assert torch.equal(g_a[name], g_b[name]), name


@pytest.mark.parametrize("points", ["rays", "coincident"])
@pytest.mark.parametrize("D", [2, 3])
def test_line_keyed_scatter_matches_the_operator_scatter(hiplib, monkeypatch, D, points):
    """rn_grid_scatter_lbc == rn_grid_encode_backward's table gradient (kernel_grid_backward, gridencoder.cu:247-339)."""
    import ctypes as C
    import radnerf_hip as hip
    from gridencoder import GridEncoder
    from radnerf import train_head
    from radnerf.fused import _grid_desc
    enc = (GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048, gridtype="hash")
           if D == 3 else GridEncoder(input_dim=2, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048,
                                      gridtype="tiled")).cuda()
    g = torch.Generator(device="cuda").manual_seed(11)
    B = 20000
    # ray-like clusters (runs of samples in the same coarse cell) + a few rows outside [0, 1]
    base = torch.rand(B // 16, 1, D, device="cuda", generator=g)
    x = (base + 0.02 * torch.arange(16, device="cuda").view(1, 16, 1) * torch.randn(B // 16, 1, D, device="cuda", generator=g)).reshape(-1, D)
    x = x.clamp(0, 1)
    if points == "coincident":             # every sample in the same few cells: the buckets of those rows overflow into the spill list
        x = x[:7].repeat(B // 7 + 1, 1)[:B]
    x[:5] = 1.5
    x = x.contiguous()
    grad_lbc = torch.randn(16, B, 2, device="cuda", generator=g)
    live = B - 37
    cnt = torch.tensor([live], dtype=torch.int32, device="cuda")
    got = torch.zeros_like(enc.embeddings)
    gd = _grid_desc(enc, enc.embeddings)
    hip.call("rn_grid_scatter_lbc", hip.ptr(grad_lbc), hip.ptr(x), B, hip.ptr(cnt), C.byref(gd), hip.ptr(got), hip.stream())
    want = torch.zeros_like(enc.embeddings)
    xl, gl = x[:live].contiguous(), grad_lbc[:, :live].contiguous()
    hip.call("rn_grid_encode_backward", hip.ptr(gl), hip.ptr(xl), hip.ptr(enc.embeddings.detach()), hip.ptr(enc.offsets, torch.int32), hip.ptr(want),
             live, D, 2, 16, float(np.log2(enc.per_level_scale)), 16, None, None, enc.gridtype_id, 0, 0, hip.RN_F32, hip.RN_LAYOUT_LBC, hip.stream())
    scale = float(want.abs().max())
    tol = 1e-5 if points == "rays" else 2e-4        # thousands of terms per row in another order
    assert float((got - want).abs().max()) / scale < tol
    assert int((got != 0).sum()) == int((want != 0).sum())
    # the opt-in table-region sum (RN_SCATTER=binned): hashed levels binned + one workgroup per bucket (two launches) where the
    # grid has such levels, called twice in a row -- the bucket cursors must come back to zero
    monkeypatch.setenv("RN_SCATTER", "binned")
    for _ in range(2):
        got2 = torch.zeros_like(enc.embeddings)
        train_head.grid_scatter([(grad_lbc, x, enc, gd, got2)], B, cnt)
        assert float((got2 - want).abs().max()) / scale < tol
        assert int((got2 != 0).sum()) == int((want != 0).sum())
    if D == 3:
        need = int(train_head._lib.rn_grid_scatter_workspace(B, C.byref(gd), hip.host_offsets(enc.offsets)))
        assert need > 1 << 20                      # the T = 2^19 hash table has binned levels


def test_head_loss_kernel_matches_the_pytorch_expression(hiplib):
    from radnerf import train_head
    from radnerf.train import entropy_of
    g = torch.Generator(device="cuda").manual_seed(2)
    N = 4096
    table = torch.rand(N, 15, device="cuda", generator=g)                # bg at columns 8:11, target 11:14, face 14 (SyntheticTrainStream)
    table[:, 14] = (table[:, 14] > 0.5).float()
    bg, target, face = table[:, 8:11], table[:, 11:14], table[:, 14]
    image = (torch.rand(N, 3, device="cuda", generator=g) * 1.2 - 0.1).requires_grad_(True)     # some blends leave [0, 1]
    ws = torch.rand(N, device="cuda", generator=g).requires_grad_(True)
    with torch.no_grad():
        ws[:4] = torch.tensor([0.0, 1.0, 1e-6, 1 - 1e-7], device="cuda")
    amb = torch.rand(N, device="cuda", generator=g).requires_grad_(True)
    w_amb = torch.tensor(0.037, device="cuda")
    loss, pred = train_head.head_loss(image, ws, amb, bg, target, face, w_amb)
    (loss * 1.7).backward()
    got = (float(loss), pred.clone(), image.grad.clone(), ws.grad.clone(), amb.grad.clone())
    image.grad = ws.grad = amb.grad = None
    p2 = (image + (1 - ws).unsqueeze(-1) * bg).clamp(0, 1)
    l2 = torch.nn.functional.mse_loss(p2, target, reduction="none").mean(-1).mean() + 1e-4 * entropy_of(ws).mean() + w_amb * (amb * (1 - face)).mean()
    (l2 * 1.7).backward()
    assert abs(got[0] - float(l2)) < 1e-6 * max(1.0, abs(float(l2)))
    assert torch.allclose(got[1], p2, atol=1e-7)
    assert torch.allclose(got[2], image.grad, rtol=1e-5, atol=1e-10)
    assert torch.allclose(got[3], ws.grad, rtol=2e-5, atol=1e-9)
    assert torch.allclose(got[4], amb.grad, rtol=1e-5, atol=1e-10)
    # train_head.backward(loss) == loss.backward(): the kernel's own input gradients handed to autograd as they are
    image.grad = ws.grad = amb.grad = None
    loss, _ = train_head.head_loss(image, ws, amb, bg, target, face, w_amb)
    loss.backward()
    ref = (image.grad.clone(), ws.grad.clone(), amb.grad.clone())
    image.grad = ws.grad = amb.grad = None
    loss, _ = train_head.head_loss(image, ws, amb, bg, target, face, w_amb)
    train_head.backward(loss)
    assert torch.equal(image.grad, ref[0]) and torch.equal(ws.grad, ref[1]) and torch.equal(amb.grad, ref[2])


def _train_losses(monkeypatch, head, steps=6):
    from radnerf.train import SyntheticTrainStream, Trainer
    monkeypatch.setenv("RN_TRAIN_HEAD", head)
    monkeypatch.setenv("RN_TRAIN_LOSS", "fused" if head == "fused" else "torch")
    torch.manual_seed(0)
    scene = _scene(64)
    stream = SyntheticTrainStream(scene, n_rays=1024, seed=4)
    trainer = Trainer(scene.model, scene.opt, update_extra_interval=0)
    scene.model.mean_count = 0
    import random
    random.seed(0)
    losses = [float(trainer.step(stream.batch())) for _ in range(steps)]
    return losses, {n: p.detach().clone() for n, p in scene.model.named_parameters()}


def test_training_steps_equal_the_operator_path(hiplib, monkeypatch):
    """Six optimizer steps of Trainer (march -> network -> composite -> loss -> backward -> Adam) through the fused head and loss
    kernels follow the per-operator steps."""
    l_ops, p_ops = _train_losses(monkeypatch, "ops")
    l_fused, p_fused = _train_losses(monkeypatch, "fused")
    assert np.allclose(l_fused, l_ops, rtol=2e-4, atol=1e-7), (l_fused, l_ops)
    # (Parameters are not compared entry by entry: Adam with eps = 1e-15 turns any difference in a near-zero gradient into a
    # full +-lr step, so two trajectories that agree in their losses to 2e-4 still differ in individual entries.)
    for name in p_ops:
        assert torch.isfinite(p_fused[name]).all(), name
        assert float((p_fused[name] - p_ops[name]).abs().max()) <= 2 * 6 * 5e-3 + 1e-6, name


def test_training_step_launch_count(hiplib, monkeypatch):
    """An eager training step is <= 32 kernel launches in the product configuration (VERDICT r2 item 1 asked for <= 40; round 2:
    133): counted with the profiler's kernel events, on a step with the running-average budget (the one-launch marcher)."""
    from radnerf.train import SyntheticTrainStream, Trainer
    monkeypatch.setenv("RN_TRAIN_HEAD", "fused")
    monkeypatch.setenv("RN_TRAIN_NOISE", "hash")
    scene = _scene(64)
    stream = SyntheticTrainStream(scene, n_rays=1024, seed=4)
    trainer = Trainer(scene.model, scene.opt, update_extra_interval=0)
    for _ in range(3):
        trainer.step(stream.batch())
    scene.model.mean_count = 12000                           # as after the first 16 steps: the budgeted step
    trainer.step(stream.batch())
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        batch = stream.batch()
        trainer.step(batch)
        torch.cuda.synchronize()
    kernels = [e for e in prof.events() if e.device_type is not None and "cuda" in str(e.device_type).lower()]
    names = [e.name for e in kernels]
    print(len(names), "device activities:", sorted(set(names)))
    assert any("k_march_train_step" in n for n in names), names
    assert 0 < len(names) <= 32, names


def test_batch_gather_equals_indexing(hiplib):
    from radnerf import train_head
    g = torch.Generator(device="cuda").manual_seed(9)
    table = torch.rand(5000, 15, device="cuda", generator=g)
    idx = torch.randint(0, 5000, (4096,), device="cuda", generator=g)
    widths = (3, 3, 2, 3, 3, 1)
    flat, secs = train_head.batch_gather(table, idx, widths)
    rows, at = table[idx], 0
    for sec, w in zip(secs, widths):
        assert sec.is_contiguous() and torch.equal(sec, rows[:, at:at + w])
        at += w
    assert flat.numel() == 4096 * 15

"""GPU parity of the training-side rows of SURVEY 8(a): the train branch of run_cuda (a2: march_rays_train ->
network -> composite_rays_train, forward and backward) and the occupancy-grid maintenance (a3: update_extra_state,
mark_untrained_grid) of this tree's renderer mirror over the HIP operators."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(size=64, **kw):
    from radnerf.scene import SyntheticScene, default_opt
    return SyntheticScene(H=size, W=size, n_frames=8, device="cuda", opt=default_opt(engine="ops", **kw))


def test_train_branch_forward_matches_oracle(po, hiplib):
    scene = _scene(64)
    m = scene.model
    m.train()
    f = scene.frame(0)
    torch.manual_seed(0)
    n_rays = 4096
    idx = torch.randperm(64 * 64, device="cuda")[:n_rays]
    rays_o, rays_d = f["rays_o"][:, idx], f["rays_d"][:, idx]
    bg_coords = f["bg_coords"][:, idx]
    out = m.render(rays_o, rays_d, f["auds"], bg_coords, f["poses"], eye=f["eye"], index=[3], bg_color=f["bg_color"][:, idx],
                   perturb=False, force_all_rays=False, dt_gamma=scene.opt.dt_gamma, max_steps=scene.opt.max_steps)
    # oracle: the same three stages (raymarching.cu:352-698 + network.py:222-283)
    o, d = rays_o.reshape(-1, 3).cpu().numpy(), rays_d.reshape(-1, 3).cpu().numpy()
    nears, fars = po.near_far_from_aabb(o, d, m.aabb_train.cpu().numpy(), m.min_near)
    bits = m.density_bitfield.cpu().numpy()
    M = n_rays * scene.opt.max_steps
    xyzs, dirs, deltas, rays, cnt = po.march_rays_train(o, d, bits, 1.0, scene.opt.dt_gamma, scene.opt.max_steps, 1, 128, M,
                                                        nears, fars, np.zeros(n_rays, np.float32))
    assert int(m.step_counter[0, 0].item()) == int(cnt[0]) and int(m.step_counter[0, 1].item()) == n_rays
    total = int(cnt[0])
    om = po.model_from_module(m)
    ind = m.individual_codes[3].detach().cpu().numpy()
    sig, rgb, amb = po.nerf_forward(om, xyzs[:total], dirs[:total], m.enc_a.detach().cpu().numpy(), ind, f["eye"].cpu().numpy())
    ws, am, dp, im = po.composite_rays_train_forward(sig, rgb, np.abs(amb).sum(-1), deltas[:total], rays, 1e-4)
    np.testing.assert_allclose(out["weights_sum"].detach().cpu().numpy(), ws, rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["ambient"].detach().cpu().numpy(), am, rtol=1e-4, atol=1e-4)
    # the train branch also blends the torso layer and clamps (renderer.py:269-308): check the head part via weights_sum
    # and the full image against the oracle's blend of its own pieces
    thresh = min(m.density_thresh_torso, m.mean_density_torso)
    import torch.nn.functional as F
    occ = F.grid_sample(m.density_grid_torso.view(1, 1, 128, 128), bg_coords.view(1, -1, 1, 2), align_corners=True).view(-1)
    mask = (occ > thresh).cpu().numpy()
    bg = f["bg_color"][:, idx].reshape(-1, 3).cpu().numpy().copy()
    if mask.any():
        ta, tc, _ = po.torso_forward(om, bg_coords.reshape(-1, 2).cpu().numpy()[mask], f["poses"].cpu().numpy(),
                                     m.individual_codes_torso[3].detach().cpu().numpy())
        bg[mask] = tc * ta + bg[mask] * (1 - ta)
    expect = np.clip(im + (1 - ws)[:, None] * bg, 0, 1)
    np.testing.assert_allclose(out["image"].detach().reshape(-1, 3).cpu().numpy(), expect, rtol=0, atol=5e-5)


def test_train_step_backward_is_the_gradient_of_the_loss(hiplib):
    """Directional finite differences through the whole chain: composite_rays_train backward -> torch MLP backward ->
    grid_encode backward (atomics) / input backward for the ambient grid."""
    scene = _scene(48, torso=False)
    m = scene.model
    m.train()
    f = scene.frame(0)
    torch.manual_seed(1)
    idx = torch.randperm(48 * 48, device="cuda")[:1024]
    args = dict(eye=f["eye"], index=[0], bg_color=f["bg_color"][:, idx], perturb=False, force_all_rays=True,
                dt_gamma=scene.opt.dt_gamma, max_steps=scene.opt.max_steps)
    w = torch.randn(1024, 3, device="cuda")

    def loss_fn():
        m.enc_a = None
        out = m.render(f["rays_o"][:, idx], f["rays_d"][:, idx], f["auds"], f["bg_coords"][:, idx], f["poses"], **args)
        return (out["image"].reshape(-1, 3) * w).sum() + 0.1 * out["ambient"].sum() + 0.3 * out["weights_sum"].sum()

    for p in m.parameters():
        p.grad = None
    loss = loss_fn()
    loss.backward()
    checks = {"encoder.embeddings": m.encoder.embeddings, "encoder_ambient.embeddings": m.encoder_ambient.embeddings,
              "sigma_net.net.0.weight": m.sigma_net.net[0].weight, "ambient_net.net.2.weight": m.ambient_net.net[2].weight}
    g = torch.Generator(device="cuda").manual_seed(5)
    for name, p in checks.items():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        v = torch.randn(p.shape, device="cuda", generator=g) * (p.grad != 0)  # direction inside the touched entries
        v = v / (v.norm() + 1e-12)
        analytic = float((p.grad * v).sum())
        eps = 2e-2 if "embeddings" in name else 5e-3
        with torch.no_grad():
            p.add_(eps * v)
            lp = float(loss_fn())
            p.sub_(2 * eps * v)
            lm = float(loss_fn())
            p.add_(eps * v)
        numeric = (lp - lm) / (2 * eps)
        assert abs(numeric - analytic) <= 0.08 * max(abs(numeric), abs(analytic)) + 2e-3, (name, numeric, analytic)


def test_update_extra_state_head(po, hiplib, monkeypatch):
    """a3: morton coords -> density query -> morton dilation -> EMA max -> packbits, with the jitter pinned to zero."""
    scene = _scene(16, torso=False)
    m = scene.model
    m.aud_features = scene.aud_features
    m.eye_area = torch.full((scene.n_frames, 1), 0.25, device="cuda")
    centred = torch.full((128 ** 3, 3), 0.5, device="cuda")                            # (u*2-1) = 0: probe points at the cell centres
    import random
    monkeypatch.setattr(random, "randint", lambda a, b: 2)
    m.density_grid.zero_()
    m.local_step = 3
    m.step_counter.zero_()
    m.step_counter[:3, 0] = torch.tensor([100, 200, 330], dtype=torch.int32)
    with torch.no_grad():
        m.update_extra_state(noise=centred)
    assert m.mean_count == 210 and m.local_step == 0
    # oracle recomputation
    H = 128
    ii = np.arange(H, dtype=np.int32)
    coords = np.stack(np.meshgrid(ii, ii, ii, indexing="ij"), -1).reshape(-1, 3)
    mort = po.morton3D(coords)
    xyz = ((2 * coords.astype(np.float32) / (H - 1) - 1) * np.float32(1 - 1 / H)).astype(np.float32)
    from radnerf.rays import get_audio_features
    with torch.no_grad():
        enc_a = m.encode_audio(get_audio_features(scene.aud_features, 2, 2)).cpu().numpy()
    sig = po.nerf_density(po.model_from_module(m), xyz, enc_a, np.array([[0.25]], np.float32))
    tmp = np.zeros((1, H ** 3), np.float32)
    tmp[0, mort] = sig
    dil = po.morton3D_dilation(tmp)
    got = m.density_grid.cpu().numpy()
    np.testing.assert_allclose(got, np.maximum(0 * 0.95, dil), rtol=2e-3, atol=1e-5)
    assert abs(m.mean_density - float(np.clip(got, 0, None).mean())) < 1e-5
    assert np.array_equal(m.density_bitfield.cpu().numpy(), po.packbits(got, min(m.mean_density, m.density_thresh)))


def test_update_extra_state_torso_and_mark_untrained(hiplib, monkeypatch):
    scene = _scene(16)
    m = scene.model
    m.aud_features = scene.aud_features
    m.poses = scene.poses
    m.eye_area = torch.full((scene.n_frames, 1), 0.25, device="cuda")
    before = m.density_grid_torso.clone()
    with torch.no_grad():
        m.update_extra_state()
    assert m.density_grid_torso.shape == before.shape and torch.isfinite(m.density_grid_torso).all()
    assert (m.density_grid_torso >= before * 0.95 - 1e-6).all()          # EMA max never drops below the decayed value
    assert abs(m.mean_density_torso - float(m.density_grid_torso.mean())) < 1e-6
    # mark_untrained_grid: cells outside every training frustum become -1 (renderer.py:318-379)
    m.density_grid.zero_()
    m.mark_untrained_grid(scene.poses[:4], scene.intrinsics)
    g = m.density_grid[0].cpu().numpy()
    from radnerf.scene import morton3d_np
    centre = int(morton3d_np(np.array([64]), np.array([64]), np.array([64]))[0])
    corner = int(morton3d_np(np.array([0]), np.array([127]), np.array([0]))[0])
    assert g[centre] == 0 and g[corner] == -1 and 0.02 < (g == -1).mean() < 0.98


def test_trainer_steps_reduce_the_loss(hiplib):
    """BASELINE config 2 as a loop: Trainer.step = update_extra_state cadence + train_step + backward + Adam
    (nerf/utils.py:718-806, 1003-1040; main.py:204).  Target = the scene's own frozen render, so a perturbed copy
    of the model must move back towards it."""
    from radnerf.train import SyntheticTrainStream, Trainer
    scene = _scene(64, torso=False, smooth_lips=False)       # --smooth_lips is a test-time flag (main.py:49)
    stream = SyntheticTrainStream(scene, n_rays=2048)
    m = scene.model
    with torch.no_grad():                       # knock the colour head off target
        m.color_net.net[-1].weight.add_(0.5 * torch.randn_like(m.color_net.net[-1].weight))
    trainer = Trainer(m, scene.opt, lr_net=5e-3, update_extra_interval=0)   # keep the ellipsoid occupancy (a3 has its own tests)
    from radnerf.train import train_step
    probe = stream.batch()

    def mse():
        m.train()
        with torch.no_grad():
            pred, rgb, _ = train_step(m, probe, scene.opt)
        return float(((pred - rgb) ** 2).mean())
    before = mse()
    losses = [float(trainer.step(stream.batch())) for _ in range(40)]
    after = mse()
    assert all(np.isfinite(losses))
    assert after < 0.5 * before, (before, after)
    assert int(m.step_counter[:, 0].max()) > 0 and m.local_step == 42


def test_trainer_refreshes_occupancy_every_interval(hiplib):
    from radnerf.train import SyntheticTrainStream, Trainer
    scene = _scene(32, torso=False, smooth_lips=False)
    stream = SyntheticTrainStream(scene, n_rays=512)
    m = scene.model
    trainer = Trainer(m, scene.opt, update_extra_interval=4)
    before = m.density_bitfield.clone()
    for _ in range(5):
        loss = trainer.step(stream.batch())
    assert torch.isfinite(loss)
    assert m.local_step == 1                    # reset by the refresh before step 5 (renderer.py:499), then one step
    assert m.mean_count > 0 and not torch.equal(before, m.density_bitfield)


def test_graphed_trainer_replays_the_step_and_learns(hiplib):
    """GraphedTrainer: the steady-state step captured in a hipGraph (recaptured after every occupancy refresh) trains like the
    eager Trainer: loss falls, sample counters and mean_count keep moving, one capture per refresh window."""
    from radnerf.train import GraphedTrainer, SyntheticTrainStream, train_step
    scene = _scene(64, torso=False, smooth_lips=False)
    stream = SyntheticTrainStream(scene, n_rays=2048)
    m = scene.model
    with torch.no_grad():
        m.color_net.net[-1].weight.add_(0.5 * torch.randn_like(m.color_net.net[-1].weight))
    trainer = GraphedTrainer(m, scene.opt, lr_net=5e-3, update_extra_interval=0)   # keep the ellipsoid occupancy, as the eager test does
    probe = stream.batch()

    def mse():
        m.train()
        with torch.no_grad():
            pred, rgb, _ = train_step(m, probe, scene.opt)
        return float(((pred - rgb) ** 2).mean())
    before = mse()
    losses = [float(trainer.step(stream.batch())) for _ in range(4)]          # no sample budget yet: eager steps
    assert trainer.captures == 0
    m.mean_count = int(m.step_counter[:4, 0].float().mean().item() * 1.2)     # what update_extra_state would derive (+ margin)
    losses += [float(trainer.step(stream.batch())) for _ in range(36)]
    after = mse()
    assert all(np.isfinite(losses)) and after < 0.5 * before, (before, after)
    assert trainer.replays == 36 and trainer.captures == 1
    assert int(m.step_counter[:, 0].max()) > 0 and m.local_step == 4 + 36 + 2   # +2: the two probes


@pytest.mark.parametrize("budget_frac", [1.3, 0.6])
def test_device_budget_marcher_equals_the_host_budget_marcher(hiplib, budget_frac):
    """rn_march_rays_train_budget (budget = device scalar, capacity = buffer rows) against march_rays_train(mean_count =
    budget): identical sample rows, counters and compositor outputs -- with a budget every ray fits in, and with one that
    drops the tail rays (raymarching.cu:446-457)."""
    import raymarching
    from raymarching.ops import march_rays_train_budget
    scene = _scene(64)
    m, f = scene.model, scene.frame(0)
    o, d = f["rays_o"].reshape(-1, 3).contiguous(), f["rays_d"].reshape(-1, 3).contiguous()
    nears, fars = raymarching.near_far_from_aabb(o, d, m.aabb_train, m.min_near)
    args = (o, d, m.bound, m.density_bitfield, m.cascade, m.grid_size, nears, fars)
    c0 = torch.zeros(2, dtype=torch.int32, device="cuda")
    raymarching.march_rays_train(*args, c0, -1, False, 128, True, scene.opt.dt_gamma, scene.opt.max_steps)
    total = int(c0[0].item())
    budget = int(total * budget_frac)
    budget += 128 - budget % 128
    c1, c2 = torch.zeros_like(c0), torch.zeros_like(c0)
    x1, d1, dl1, r1 = raymarching.march_rays_train(*args, c1, budget - 128, False, 128, False, scene.opt.dt_gamma, scene.opt.max_steps)
    assert x1.shape[0] == budget
    cap = budget + 3000
    x2, d2, dl2, r2 = march_rays_train_budget(*args, c2, torch.tensor([budget], dtype=torch.int32, device="cuda"), cap, False,
                                              scene.opt.dt_gamma, scene.opt.max_steps)
    assert x2.shape[0] == cap and torch.equal(c1, c2)
    assert torch.equal(x1, x2[:budget]) and torch.equal(d1, d2[:budget]) and torch.equal(dl1, dl2[:budget])
    assert not x2[budget:].any() and not dl2[budget:].any()
    dropped = (r1[:, 1] + r1[:, 2]) > budget
    assert bool(dropped.any()) == (budget_frac < 1)
    assert torch.equal(r1[~dropped], r2[~dropped]) and torch.equal(r1[dropped][:, :2], r2[dropped][:, :2]) and not r2[dropped][:, 2].any()
    g = torch.Generator(device="cuda").manual_seed(3)
    sig1, rgb1, amb1 = (torch.rand(budget, device="cuda", generator=g) * 4, torch.rand(budget, 3, device="cuda", generator=g),
                        torch.rand(budget, device="cuda", generator=g))
    pad = lambda t: torch.cat([t, torch.rand(cap - budget, *t.shape[1:], device="cuda", generator=g)])   # rows no ray owns: anything
    out1 = raymarching.composite_rays_train(sig1, rgb1, amb1, dl1, r1)
    out2 = raymarching.composite_rays_train(pad(sig1), pad(rgb1), pad(amb1), dl2, r2)
    for a, b in zip(out1, out2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("budget_frac", [1.3, 0.6])
@pytest.mark.parametrize("perturb", [False, True])
@pytest.mark.parametrize("max_steps", [16, 48])
def test_one_launch_step_marcher_equals_the_five_launch_chain(hiplib, budget_frac, perturb, max_steps):
    """rn_march_rays_train_step (near / far + count + ordered slices + samples + counters in ONE launch, counts exchanged between
    workgroups inside it) against near_far_from_aabb + zeroed counters + rn_march_rays_train_budget: identical nears / fars, rays,
    counters and sample rows; on buffers that were NOT zeroed (NaN-filled here) every row below min(counter[0], capacity) is
    defined -- samples, or zeros where the budget cut a ray -- and nothing beyond is touched.  Run several times over: the launch
    epoch in the persistent state words must carry from one launch to the next, and no launch may time out.  max_steps 16: the
    launch records its samples' t in LDS and rebuilds them; 48: it walks twice."""
    import raymarching
    from raymarching import ops
    scene = _scene(64)
    m, f = scene.model, scene.frame(0)
    o, d = f["rays_o"].reshape(-1, 3).contiguous(), f["rays_d"].reshape(-1, 3).contiguous()
    N = o.shape[0]
    assert ops.step_marcher_supported(N, o.device) and (N + 255) // 256 > 4      # several workgroups exchange counts
    nears, fars = raymarching.near_far_from_aabb(o, d, m.aabb_train, m.min_near)
    c0 = torch.zeros(2, dtype=torch.int32, device="cuda")
    raymarching.march_rays_train(o, d, m.bound, m.density_bitfield, m.cascade, m.grid_size, nears, fars, c0, -1, False, 128, True,
                                 scene.opt.dt_gamma, max_steps)
    total = int(c0[0].item())
    budget = int(total * budget_frac)
    budget += 128 - budget % 128
    cap = budget + 3000
    bt = torch.tensor([budget], dtype=torch.int32, device="cuda")
    state = ops._step_state(N, o.device)
    for rep in range(3):
        seed = 11 + rep
        torch.manual_seed(seed)
        c1 = torch.zeros(2, dtype=torch.int32, device="cuda")
        x1, d1, dl1, r1 = ops.march_rays_train_budget(o, d, m.bound, m.density_bitfield, m.cascade, m.grid_size, nears, fars, c1, bt, cap,
                                                      perturb, scene.opt.dt_gamma, max_steps)
        torch.manual_seed(seed)                                   # the same jitter draw
        c2 = torch.full((2,), 12345, dtype=torch.int32, device="cuda")     # SET by the launch, whatever it held
        epoch = int(state[0].item())
        n2, f2, x2, d2, dl2, r2 = ops.march_rays_train_step(o, d, m.aabb_train, m.min_near, m.bound, m.density_bitfield, m.cascade,
                                                            m.grid_size, c2, bt, cap, perturb, scene.opt.dt_gamma, max_steps, True)
        assert int(state[0].item()) == epoch + 1 and int(state[1].item()) == 0
        assert torch.equal(n2, nears) and torch.equal(f2, fars)
        assert torch.equal(c1, c2) and torch.equal(r1, r2)
        assert torch.equal(x1, x2) and torch.equal(d1, d2) and torch.equal(dl1, dl2)
    # buffers the caller did not zero
    real_empty = torch.empty

    def nan_empty(*a, **k):
        t = real_empty(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t
    torch.manual_seed(11)
    c3 = torch.zeros(2, dtype=torch.int32, device="cuda")
    try:
        torch.empty = nan_empty
        n3, f3, x3, d3, dl3, r3 = ops.march_rays_train_step(o, d, m.aabb_train, m.min_near, m.bound, m.density_bitfield, m.cascade,
                                                            m.grid_size, c3, bt, cap, perturb, scene.opt.dt_gamma, max_steps, False)
    finally:
        torch.empty = real_empty
    torch.manual_seed(11)
    c1 = torch.zeros(2, dtype=torch.int32, device="cuda")
    x1, d1, dl1, r1 = ops.march_rays_train_budget(o, d, m.bound, m.density_bitfield, m.cascade, m.grid_size, nears, fars, c1, bt, cap, perturb,
                                                  scene.opt.dt_gamma, max_steps)
    live = min(int(c3[0].item()), cap)
    assert torch.equal(c3, c1) and torch.equal(r3, r1)
    assert torch.equal(x3[:live], x1[:live]) and torch.equal(d3[:live], d1[:live]) and torch.equal(dl3[:live], dl1[:live])
    assert bool(torch.isnan(x3[live:]).all()) and bool(torch.isnan(dl3[live:]).all())


def test_step_marcher_hash_jitter_is_a_new_uniform_draw_per_launch(hiplib):
    """perturb = "hash": the first sample of a ray is moved by u * dt with u from the launch's own counter-based hash (no
    torch.rand launch).  With every cell occupied the first sample sits at near + u * dt, so u can be read back: uniform on
    [0, 1), and a new draw per launch (the launch epoch is part of the hash)."""
    from raymarching import ops
    scene = _scene(64)
    m, f = scene.model, scene.frame(0)
    o, d = f["rays_o"].reshape(-1, 3).contiguous(), f["rays_d"].reshape(-1, 3).contiguous()
    N = o.shape[0]
    cap = N * scene.opt.max_steps
    bt = torch.tensor([cap], dtype=torch.int32, device="cuda")
    full = torch.full_like(m.density_bitfield, 255)

    def run(perturb):
        c = torch.zeros(2, dtype=torch.int32, device="cuda")
        nears, _, x, dd, dl, r = ops.march_rays_train_step(o, d, m.aabb_train, m.min_near, m.bound, full, m.cascade, m.grid_size, c, bt, cap,
                                                           perturb, scene.opt.dt_gamma, scene.opt.max_steps, True)
        assert int(c[0]) == int(r[:, 2].sum()) and int(c[1]) == N
        hit = r[:, 2] > 0
        first = r[hit, 1].long()
        return hit, (dl[first, 1] - dl[first, 0] - nears[hit]) / dl[first, 0]      # (t of the first sample - near) / dt
    hit0, u0 = run(False)
    assert int(hit0.sum()) > N // 2 and float(u0.abs().max()) < 1e-4
    hit1, u1 = run("hash")
    hit2, u2 = run("hash")
    assert torch.equal(hit0, hit1) and torch.equal(hit0, hit2)
    assert not torch.equal(u1, u2)                                                 # a new draw per launch
    for u in (u1, u2):
        assert float(u.min()) > -1e-4 and float(u.max()) < 1 + 1e-4
        assert abs(float(u.mean()) - 0.5) < 0.02 and abs(float(u.var()) - 1 / 12) < 0.01
        assert abs(float(torch.corrcoef(torch.stack([u[:-1], u[1:]]))[0, 1])) < 0.05   # neighbouring rays: unrelated draws


def test_replayed_steps_draw_new_jitter(hiplib, monkeypatch):
    """Product configuration (RN_TRAIN_NOISE=hash) under GraphedTrainer: the marcher's launch epoch lives OUTSIDE the captured graph
    (allocated before the capture), so every replay advances it and draws new jitter -- a state born inside the capture would be
    re-zeroed by every replay and the same jitter would come back each step."""
    from raymarching import ops
    from radnerf.train import GraphedTrainer, SyntheticTrainStream
    monkeypatch.setenv("RN_TRAIN_NOISE", "hash")
    scene = _scene(64, torso=False, smooth_lips=False)
    stream = SyntheticTrainStream(scene, n_rays=2048, seed=3)
    m = scene.model
    trainer = GraphedTrainer(m, scene.opt, update_extra_interval=0)
    for _ in range(3):
        trainer.step(stream.batch())
    m.mean_count = 30000
    batch = stream.batch()
    epochs, counts = [], []
    for _ in range(6):
        trainer.step(batch)                                        # the SAME rays every time: only the jitter can move the count
        epochs.append(int(ops._STEP_STATE[(m.density_bitfield.device.index, 2048)][0]))
        counts.append(int(m.step_counter[(m.local_step - 1) % 16, 0]))
    assert trainer.captures == 1 and trainer.replays == 6
    assert epochs == list(range(epochs[0], epochs[0] + 6)), epochs
    assert int(ops._STEP_STATE[(m.density_bitfield.device.index, 2048)][1]) == 0      # no exchange timed out
    assert len(set(counts)) > 1, counts


def test_graphed_trainer_keeps_its_graph_when_the_budget_moves(hiplib):
    """The sample budget is a device scalar of the captured step: mean_count moving inside the capacity window costs no capture,
    leaving it costs one."""
    from radnerf.train import GraphedTrainer, SyntheticTrainStream
    scene = _scene(64, torso=False, smooth_lips=False)
    stream = SyntheticTrainStream(scene, n_rays=2048)
    m = scene.model
    trainer = GraphedTrainer(m, scene.opt, update_extra_interval=0, capacity_step=2048)
    for _ in range(3):
        trainer.step(stream.batch())
    base = int(m.step_counter[:3, 0].float().mean().item())
    m.mean_count = base
    trainer.step(stream.batch())
    cap = trainer._capacity
    assert trainer.captures == 1 and cap % 2048 == 0 and cap >= base
    for mc in (base + 200, base - 300, base + 100):
        m.mean_count = mc
        loss = trainer.step(stream.batch())
        assert int(trainer._budget.item()) == mc + 128 - mc % 128 and int(m.step_counter[(m.local_step - 1) % 16, 0]) > 0
    assert trainer.captures == 1 and trainer._capacity == cap and np.isfinite(float(loss))
    m.mean_count = cap + 500                    # leaves the window
    trainer.step(stream.batch())
    assert trainer.captures == 2 and trainer._capacity > cap


def test_hip_adam_matches_torch_adam(hiplib):
    """radnerf.train.HipAdam (rn_adam_step: one kernel for all tensors, step counter on the device) against torch.optim.Adam with
    the reference's settings (main.py:204) over 6 steps with two learning rates: parameters and both moments within 2e-6 relative;
    the state dict is torch.optim.Adam's (a torch optimizer loads it and continues identically)."""
    from radnerf.train import HipAdam
    gen = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(1000003,), (64, 96), (7,), (10000, 4), (1,), (33, 3)]
    pa = [torch.nn.Parameter(torch.randn(*s, device="cuda", generator=gen)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    groups = lambda ps: [{"params": ps[:2], "lr": 5e-3}, {"params": ps[2:], "lr": 5e-4, "weight_decay": 0}]
    oa = HipAdam(groups(pa), betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam(groups(pb), betas=(0.9, 0.99), eps=1e-15)
    for step in range(6):
        for a, b in zip(pa, pb):
            g = torch.randn(a.shape, device="cuda", generator=gen) * (10.0 ** (step - 3))
            if step == 2 and a.numel() > 100:
                g[::3] = 0                                         # rows without gradient still move (dense Adam)
            a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    def close(x, y):
        return float((x - y).abs().max()) <= 2e-6 * float(y.abs().max()) + 1e-30
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert close(a.detach(), b.detach()), i
        assert close(oa.state[a]["exp_avg"], ob.state[b]["exp_avg"]) and close(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"]), i
    assert int(oa._step.item()) == 6
    oc = torch.optim.Adam(groups([torch.nn.Parameter(p.detach().clone()) for p in pa]), betas=(0.9, 0.99), eps=1e-15)
    oc.load_state_dict(oa.state_dict())
    assert all(float(s["step"]) == 6 for s in oc.state_dict()["state"].values())
    od = HipAdam(groups(pa), betas=(0.9, 0.99), eps=1e-15)
    od.load_state_dict(ob.state_dict())
    assert int(od._step.item()) == 6 and close(od.state[pa[0]]["exp_avg"], ob.state[pb[0]]["exp_avg"])
    with pytest.raises(ValueError):
        HipAdam([{"params": pa[:1], "lr": 1e-3, "weight_decay": 0.1}])


@pytest.mark.parametrize("graphed", [False, True])
def test_packed_batches_train_like_separate_tensors(hiplib, monkeypatch, graphed):
    """SyntheticTrainStream hands a batch over as views of ONE gathered table (GraphedTrainer then refreshes one static tensor per
    step); with RN_TRAIN_PACKED=0 it hands over separate contiguous tensors as a generic loader would.  Same seeds -> same
    samples, same noise: the loss curves agree step by step (atomics reorder sums: 2e-3 relative)."""
    from radnerf.train import GraphedTrainer, SyntheticTrainStream, Trainer
    curves = {}
    for packed in ("1", "0"):
        monkeypatch.setenv("RN_TRAIN_PACKED", packed)
        torch.manual_seed(11)
        scene = _scene(64, torso=False, smooth_lips=False)
        stream = SyntheticTrainStream(scene, n_rays=2048, seed=3)
        m = scene.model
        trainer = (GraphedTrainer if graphed else Trainer)(m, scene.opt, update_extra_interval=0)
        losses = [float(trainer.step(stream.batch())) for _ in range(3)]
        m.mean_count = 40000
        torch.manual_seed(12)
        losses += [float(trainer.step(stream.batch())) for _ in range(12)]
        curves[packed] = np.array(losses)
        if graphed:
            assert trainer.captures == 1 and trainer.replays == 12
    np.testing.assert_allclose(curves["1"], curves["0"], rtol=2e-3, atol=1e-7)


def test_glue_kernels_match_the_torch_expressions(hiplib, monkeypatch):
    """radnerf.train_glue (rn_head_mid_*, rn_abs_sum2_*, rn_train_loss) against the PyTorch expressions they replace
    (nerf/network.py:266-276, nerf/renderer.py:216, nerf/utils.py:772-803): values and gradients, element by element."""
    from activation import trunc_exp
    from radnerf import train_glue as tg
    from radnerf.train import entropy_of
    g = torch.Generator(device="cuda").manual_seed(2)
    M, N = 5003, 4096
    h = (torch.randn(M, 65, device="cuda", generator=g) * 3).requires_grad_(True)
    h.data[::17, 0] = 20.0                                     # beyond the clamp of trunc_exp's backward
    enc_d = torch.randn(M, 16, device="cuda", generator=g)
    sig, xc = tg.head_mid(h, enc_d)
    hr = h.detach().clone().requires_grad_(True)
    sig_r, xc_r = trunc_exp(hr[:, 0]), torch.cat([enc_d, hr[:, 1:]], -1)
    gs, gx = torch.randn(M, device="cuda", generator=g), torch.randn(M, 80, device="cuda", generator=g)
    (ga,), (gb,) = torch.autograd.grad([sig, xc], [h], [gs, gx]), torch.autograd.grad([sig_r, xc_r], [hr], [gs, gx])
    assert torch.allclose(sig, sig_r, rtol=2e-6, atol=0) and torch.equal(xc, xc_r) and torch.allclose(ga, gb, rtol=2e-6, atol=0)
    a = torch.randn(M, 2, device="cuda", generator=g).requires_grad_(True)
    a.data[3] = 0.0
    ar = a.detach().clone().requires_grad_(True)
    go = torch.randn(M, device="cuda", generator=g)
    assert torch.equal(tg.abs_sum2(a), ar.abs().sum(-1))
    assert torch.equal(torch.autograd.grad(tg.abs_sum2(a), a, go)[0], torch.autograd.grad(ar.abs().sum(-1), ar, go)[0])
    # the loss: weights_sum incl. values outside the entropy clamp
    pred = torch.rand(1, N, 3, device="cuda", generator=g).requires_grad_(True)
    tgt = torch.rand(1, N, 3, device="cuda", generator=g)
    ws = torch.rand(N, device="cuda", generator=g)
    ws[:64] = 0.0; ws[64:128] = 1.0; ws[128] = 1e-5
    ws.requires_grad_(True)
    amb = torch.rand(N, device="cuda", generator=g).requires_grad_(True)
    face = (torch.rand(1, N, device="cuda", generator=g) > 0.5)
    w_amb = torch.tensor([0.037], device="cuda")
    loss = tg.train_loss(pred, tgt, ws, amb, face.float(), w_amb)
    ref = torch.nn.functional.mse_loss(pred, tgt, reduction="none").mean(-1).mean() + 1e-4 * entropy_of(ws).mean() + \
        w_amb[0] * (amb * (~face.view(-1))).mean()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    ga = torch.autograd.grad(loss, [pred, ws, amb])
    gb = torch.autograd.grad(ref, [pred, ws, amb])
    for x, y in zip(ga, gb):
        assert torch.allclose(x, y, rtol=2e-5, atol=1e-12), float((x - y).abs().max())


def test_train_step_with_glue_kernels_equals_the_torch_step(hiplib, monkeypatch):
    """One training step with RN_TRAIN_GLUE=hip against RN_TRAIN_GLUE=torch on identical models and batch: same loss and
    parameter gradients (summation order of the scatter-adds aside)."""
    from radnerf.train import SyntheticTrainStream, train_step
    res = {}
    for mode in ("hip", "torch"):
        monkeypatch.setenv("RN_TRAIN_GLUE", mode)
        torch.manual_seed(4)
        scene = _scene(64, torso=False, smooth_lips=False)
        stream = SyntheticTrainStream(scene, n_rays=2048, seed=1)
        m = scene.model
        m.train()
        torch.manual_seed(9)
        _, _, loss = train_step(m, stream.batch(), scene.opt, global_step=1000)
        params = [p for n, p in m.named_parameters() if not n.startswith("torso") and "individual_codes_torso" not in n and p.requires_grad]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        res[mode] = (float(loss), [None if gr is None else gr.clone() for gr in grads])
    assert abs(res["hip"][0] - res["torch"][0]) <= 1e-5 * abs(res["torch"][0])
    for a, b in zip(res["hip"][1], res["torch"][1]):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()) + 1e-12


@pytest.mark.parametrize("graphed", [False, True])
def test_fused_engine_sees_weights_updated_by_the_optimizer(hiplib, graphed):
    """HipAdam and a replayed hipGraph write the parameters through raw pointers.  The fused engine's packed weight images (and
    GridEncoder.half_table) are caches keyed on tensor versions, so the update must bump them: after N training steps the
    occupancy refresh's density query (fused.density_forward, what update_extra_state runs) must use the CURRENT weights."""
    from radnerf import fused
    from radnerf.train import GraphedTrainer, SyntheticTrainStream, Trainer
    scene = _scene(64, torso=False, smooth_lips=False)
    stream = SyntheticTrainStream(scene, n_rays=2048)
    m = scene.model
    f = scene.frame(0)
    enc_a = m.encode_audio(f["auds"]).detach()
    pts = (torch.rand(4096, 3, device="cuda") - 0.5) * 0.8
    m.eval()
    with torch.no_grad():
        first = fused.density_forward(m, pts, enc_a, f["eye"]).clone()        # packs the initial weights
        half0 = m.encoder.half_table().clone()
    trainer = (GraphedTrainer if graphed else Trainer)(m, scene.opt, lr=5e-2, lr_net=5e-2, update_extra_interval=0)
    for _ in range(4):
        trainer.step(stream.batch())
    if graphed:
        m.mean_count = int(m.step_counter[:4, 0].float().mean().item() * 1.2)
        for _ in range(4):
            trainer.step(stream.batch())
        assert trainer.replays == 4
    m.eval()
    with torch.no_grad():
        now = fused.density_forward(m, pts, enc_a, f["eye"])
        ref = m.density(pts, enc_a, f["eye"])["sigma"]
        half1 = m.encoder.half_table()
    assert float((now - first).abs().max()) > 1e-3 * float(first.abs().max())      # the weights did move
    np.testing.assert_allclose(now.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=1e-6)
    assert torch.equal(half1, m.encoder.embeddings.detach().half()) and not torch.equal(half0, half1)


def test_hip_adam_captured_step_follows_lr_changes(hiplib):
    """The Adam launch reads its learning rates from device memory (rn_adam_step_lr): a step captured in a hipGraph follows
    changes of optimizer.param_groups[i]['lr'] between replays (the reference decays it with LambdaLR, main.py:219), like
    torch.optim.Adam stepping eagerly; the state dict carries torch.optim.Adam's keys."""
    from radnerf.train import HipAdam
    gen = torch.Generator(device="cuda").manual_seed(9)
    shapes = [(4099,), (64, 96), (5,)]
    pa = [torch.nn.Parameter(torch.randn(*s, device="cuda", generator=gen)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    groups = lambda ps: [{"params": ps[:1], "lr": 5e-3}, {"params": ps[1:], "lr": 5e-4, "weight_decay": 0}]
    oa = HipAdam(groups(pa), betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam(groups(pb), betas=(0.9, 0.99), eps=1e-15)
    for a, b in zip(pa, pb):
        a.grad = torch.randn(a.shape, device="cuda", generator=gen)
        b.grad = a.grad.clone()
    oa.step(), ob.step()                                    # eager: uploads the learning rates
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        oa.step()
    ob.step()                                               # the capture pass is not a step; replay it once to stay in sync
    graph.replay()
    for i in range(5):
        for a, b in zip(pa, pb):
            a.grad.copy_(torch.randn(a.shape, device="cuda", generator=gen))
            b.grad.copy_(a.grad)
        for o in (oa, ob):
            for g in o.param_groups:
                g["lr"] = g["lr"] * 0.5
        oa.refresh_lr()
        graph.replay()
        ob.step()
    for a, b in zip(pa, pb):
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * float(b.detach().abs().max())
    sd = oa.state_dict()
    assert all(k in g for g in sd["param_groups"] for k in ("weight_decay", "amsgrad", "maximize", "betas", "eps", "lr"))
    fresh = torch.optim.Adam(groups([torch.nn.Parameter(p.detach().clone()) for p in pa]), betas=(0.9, 0.99), eps=1e-15)
    fresh.load_state_dict(sd)
    assert fresh.param_groups[0]["lr"] == oa.param_groups[0]["lr"]


def test_graphed_trainer_follows_a_learning_rate_schedule(hiplib):
    """GraphedTrainer refreshes the device-side learning rates before every replay: with a decaying schedule its loss curve is
    the eager Trainer's (2e-3: atomics reorder sums), and it is NOT the curve of a constant learning rate."""
    from radnerf.train import GraphedTrainer, SyntheticTrainStream, Trainer
    curves = {}
    for kind in ("eager", "graph", "graph_const"):
        torch.manual_seed(21)
        scene = _scene(64, torso=False, smooth_lips=False)
        stream = SyntheticTrainStream(scene, n_rays=2048, seed=4)
        m = scene.model
        with torch.no_grad():
            m.color_net.net[-1].weight.add_(0.5 * torch.randn_like(m.color_net.net[-1].weight))
        trainer = (Trainer if kind == "eager" else GraphedTrainer)(m, scene.opt, lr_net=5e-3, update_extra_interval=0)
        losses = []
        for i in range(15):
            if i == 3:
                m.mean_count = 40000
                torch.manual_seed(22)
            if kind != "graph_const":
                for g in trainer.optimizer.param_groups:
                    g["lr"] = g["initial_lr"] * 0.1 ** (i / 5.0)
            losses.append(float(trainer.step(stream.batch())))
        curves[kind] = np.array(losses)
        if kind != "eager":
            assert trainer.captures == 1 and trainer.replays == 12
    print("lr schedule curves", {k: v.tolist() for k, v in curves.items()})
    np.testing.assert_allclose(curves["graph"], curves["eager"], rtol=1e-2, atol=1e-7)
    gap = np.abs(curves["graph_const"] - curves["graph"]) / curves["graph"]
    assert gap.max() > 5 * (np.abs(curves["graph"] - curves["eager"]) / curves["eager"]).max() and gap.max() > 3e-2, gap

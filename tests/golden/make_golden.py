"""Generates the golden fixtures under tests/golden/ -- run ONLY in the build container (needs /root/reference).

    python tests/golden/make_golden.py            # reference_{flow,ops,frames,mixed,train_stable}.npz

What runs is the reference's OWN Python, imported unmodified from /root/reference:
  * the operator wrappers raymarching/raymarching.py, gridencoder/grid.py, shencoder/sphere_harmonics.py,
    freqencoder/freq.py (their padding `M += 128 - M % 128`, the `.item()` trim, the [L,B,C] -> [B,L*C] permute, the
    autocast cast of the table, the zero-initialised gradients, ...), plus encoding.py and activation.py;
  * the control flow nerf/network.py (AudioNet, AudioAttNet, MLP, NeRFNetwork.forward / forward_torso / density /
    encode_audio), nerf/renderer.py (NeRFRenderer.run_cuda, both branches) and nerf/utils.py (get_rays, get_bg_coords,
    convert_poses, get_audio_features).
Underneath, the four native modules the wrappers import (`_raymarching_face`, `_gridencoder`, `_shencoder`,
`_freqencoder`) are objects with the pybind11 signatures computing with the CPU oracle (tests/golden/oracle_backends.py):
the reference's CUDA sources cannot be built here (no nvcc / NVIDIA GPU).  The wrappers' `.cuda()` hops are stubbed to the
identity so everything stays on CPU tensors.  Third-party packages the reference imports at module level but never uses
on this path (tensorboardX, cv2, trimesh, mcubes, torch_ema, imageio, lpips) are empty stub modules.

So these fixtures pin, with real reference code: the wrapper rules, the loop policy / compaction order / torso mask /
blend / EMA, the MLP and audio arithmetic and the ray utilities.  The kernel arithmetic inside the vectors is the
oracle's own (a self-comparison for the oracle; for the HIP path it is the oracle check at the reference's call sites).
Only data is committed: inputs, expected outputs, parameter checksums -- no reference source text.
"""
import hashlib
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
for p in (HERE, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "rad-nerf_amd")):
    sys.path.insert(0, p)

import cases  # noqa: E402
import oracle_backends as ob  # noqa: E402

warnings.filterwarnings("ignore", category=FutureWarning)


def t2n(x):
    return x.detach().cpu().numpy()


def sha(x):
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()


def install_reference():
    """Make `import raymarching / gridencoder / shencoder / freqencoder / encoding / activation / nerf.*` resolve to the
    reference's files, on top of the oracle-backed native modules."""
    for name in ("trimesh", "tensorboardX", "cv2", "mcubes", "imageio", "lpips", "torch_ema"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torch_ema"].ExponentialMovingAverage = object
    sys.modules["_raymarching_face"] = ob.raymarching_backend()
    sys.modules["_gridencoder"] = ob.gridencoder_backend()
    sys.modules["_shencoder"] = ob.shencoder_backend()
    sys.modules["_freqencoder"] = ob.freqencoder_backend()
    torch.Tensor.cuda = lambda self, *a, **k: self          # raymarching.py:34-35, freq.py:22, ...: stay on the CPU
    sys.path.insert(0, REF)
    import raymarching
    import gridencoder
    import shencoder
    import freqencoder
    import encoding
    import activation
    import nerf.network as ref_network
    import nerf.utils as ref_utils
    for mod in (raymarching, gridencoder, shencoder, freqencoder, encoding, activation, ref_network, ref_utils):
        assert mod.__file__.startswith(REF), mod.__file__
    # scene setup + inputs come from this tree (pure torch on CPU).  Imported AFTER the reference's packages: this tree's
    # radnerf/__init__ does `import raymarching`, `from encoding import ...`, which now resolve to the reference's modules
    # already in sys.modules (its mirror classes are not used here; the scene is handed the reference model).
    from radnerf.scene import SyntheticScene, default_opt
    return SyntheticScene, default_opt, raymarching, gridencoder, shencoder, freqencoder, ref_network, ref_utils


# ------------------------------------------------------------------------------------------------ reference_flow.npz
def make_flow(env):
    SyntheticScene, default_opt, _, _, _, _, ref_network, ref_utils = env
    out = {}
    H = W = 32
    opt = default_opt()
    torch.manual_seed(0)
    ref_model = ref_network.NeRFNetwork(opt)
    scene = SyntheticScene(H=H, W=W, n_frames=8, device="cpu", opt=opt, model=ref_model)
    m = scene.model

    # parameter checksums: the mirror built with the same seed must reproduce these bit for bit
    sd = m.state_dict()
    out["param_names"] = np.array(sorted(sd.keys()))
    out["param_sha256"] = np.array([sha(t2n(sd[k])) for k in sorted(sd.keys())])
    out["param_shapes"] = np.array(["x".join(str(d) for d in sd[k].shape) for k in sorted(sd.keys())])
    out["param_dtypes"] = np.array([str(sd[k].dtype) for k in sorted(sd.keys())])

    # ---- inputs of the path, from the reference's own utilities
    pose = scene.poses[3:4]
    r = ref_utils.get_rays(pose, scene.intrinsics, H, W, -1)
    out["pose"] = t2n(pose)
    out["intrinsics"] = np.asarray(scene.intrinsics)
    out["rays_o"], out["rays_d"] = t2n(r["rays_o"]), t2n(r["rays_d"])
    out["bg_coords"] = t2n(ref_utils.get_bg_coords(H, W, "cpu"))
    out["poses6"] = t2n(ref_utils.convert_poses(scene.poses))
    out["poses_all"] = t2n(scene.poses)
    for idx in (0, 2, 5, 7):
        out[f"aud_window_{idx}"] = t2n(ref_utils.get_audio_features(scene.aud_features, 2, idx))
    out["aud_features"] = t2n(scene.aud_features)
    # dataset-side pose utilities (nerf/provider.py:19-45): axis convention of the loader, trajectory smoothing
    import nerf.provider as ref_provider
    raw = t2n(scene.poses).astype(np.float64)
    out["ngp_pose_in"] = raw[5]
    out["ngp_pose_out"] = ref_provider.nerf_matrix_to_ngp(raw[5], scale=4, offset=[0.1, -0.2, 0.3])
    out["smooth_path"] = ref_provider.smooth_camera_path(t2n(scene.poses).copy(), 5)

    # ---- per-sample network (network.py:222-325), torso (188-219), audio (170-185)
    rng = np.random.default_rng(11)
    M = 256
    x = rng.uniform(-0.7, 0.7, (M, 3)).astype(np.float32)
    d = rng.standard_normal((M, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    xy = rng.uniform(-1, 1, (M, 2)).astype(np.float32)
    with torch.no_grad():
        enc_a0 = m.encode_audio(scene.frame(0)["auds"])
        c, ct = m.individual_codes[0], m.individual_codes_torso[0]
        sigma, color, amb = m(torch.from_numpy(x), torch.from_numpy(d), enc_a0, c, scene.eye)
        dens = m.density(torch.from_numpy(x), enc_a0, scene.eye)["sigma"]
        ta, tc, tdx = m.forward_torso(torch.from_numpy(xy), scene.poses6[0:1], enc_a0, ct)
    out.update(net_x=x, net_d=d, net_enc_a=t2n(enc_a0), net_sigma=t2n(sigma), net_color=t2n(color), net_ambient=t2n(amb),
               net_density=t2n(dens), torso_xy=xy, torso_alpha=t2n(ta), torso_color=t2n(tc), torso_dx=t2n(tdx))

    # ---- frames through NeRFRenderer.render -> run_cuda (renderer.py:158-316), 2 frames (EMA state)
    m.enc_a = None
    for i in (0, 1):
        f = scene.frame(i)
        with torch.no_grad():
            res = m.render(f["rays_o"], f["rays_d"], f["auds"], f["bg_coords"], f["poses"], eye=f["eye"], index=0,
                           bg_color=f["bg_color"], staged=True, perturb=False, **{"dt_gamma": opt.dt_gamma, "max_steps": opt.max_steps})
        out[f"frame{i}_image"] = t2n(res["image"]).reshape(-1, 3)
        out[f"frame{i}_depth"] = t2n(res["depth"]).reshape(-1)
        out[f"frame{i}_enc_a"] = t2n(m.enc_a)
        out[f"frame{i}_torso_alpha"] = t2n(res["torso_alpha"]).reshape(-1)
    return out


# ------------------------------------------------------------------------------------------------ reference_ops.npz
def make_ops(env):
    """The reference's operator wrappers called one by one (the names a caller of the packages sees)."""
    SyntheticScene, default_opt, rm, gridencoder, shencoder, freqencoder, ref_network, ref_utils = env
    out = {}
    opt = default_opt()
    torch.manual_seed(0)
    scene = SyntheticScene(H=32, W=32, n_frames=8, device="cpu", opt=opt, model=ref_network.NeRFNetwork(opt))
    m = scene.model
    f = scene.frame(3)
    rays_o, rays_d = f["rays_o"].reshape(-1, 3), f["rays_d"].reshape(-1, 3)
    N = rays_o.shape[0]
    bits = m.density_bitfield
    draw = cases.rm_inputs()

    # -- near/far (raymarching.py:19-49)
    nears, fars = rm.near_far_from_aabb(rays_o, rays_d, m.aabb_infer, m.min_near)
    out["nears"], out["fars"] = t2n(nears), t2n(fars)

    # -- inference marcher: the padding rule M += align - M % align, also when M is already aligned (raymarching.py:380-383)
    alive = torch.arange(N, dtype=torch.int32)
    x1, d1, dl1 = rm.march_rays(N, 1, alive, nears.clone(), rays_o, rays_d, m.bound, bits, m.cascade, m.grid_size, nears, fars, 128,
                                False, opt.dt_gamma, opt.max_steps)
    assert x1.shape[0] == N + 128
    out["march1_xyzs"], out["march1_dirs"], out["march1_deltas"] = t2n(x1), t2n(d1), t2n(dl1)
    hit = torch.nonzero(dl1[:N, 0] > 0).reshape(-1).int()
    n_alive = min(300, hit.numel())
    alive3 = hit[:n_alive].contiguous()
    out["march3_alive"] = t2n(alive3)
    rays_t = nears.clone()
    x3, d3, dl3 = rm.march_rays(n_alive, 3, alive3, rays_t, rays_o, rays_d, m.bound, bits, m.cascade, m.grid_size, nears, fars, 128,
                                False, opt.dt_gamma, opt.max_steps)
    out["march3_xyzs"], out["march3_dirs"], out["march3_deltas"] = t2n(x3), t2n(d3), t2n(dl3)
    xn, _, _ = rm.march_rays(n_alive, 3, alive3, rays_t, rays_o, rays_d, m.bound, bits, m.cascade, m.grid_size, nears, fars, -1,
                             False, opt.dt_gamma, opt.max_steps)
    assert xn.shape[0] == n_alive * 3
    # -- composite (in place; raymarching.py:415-437) with T_thresh = 1e-4 as run_cuda passes it
    M3 = x3.shape[0]
    sig, rgb = draw(M3, lo=0.0, hi=60.0), draw(M3, 3)
    out["comp_sigmas"], out["comp_rgbs"] = t2n(sig), t2n(rgb)
    ws, dp, im = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
    alive_c, t_c = alive3.clone(), rays_t.clone()
    rm.composite_rays(n_alive, 3, alive_c, t_c, sig, rgb, dl3, ws, dp, im, 1e-4)
    out["comp_weights_sum"], out["comp_depth"], out["comp_image"] = t2n(ws), t2n(dp), t2n(im)
    out["comp_rays_alive"], out["comp_rays_t"] = t2n(alive_c), t2n(t_c)

    # -- training marcher (raymarching.py:187-281): first-epoch path (.item() trim + pad) and the mean_count path
    xt, dt_, dlt, rt = rm.march_rays_train(rays_o, rays_d, m.bound, bits, m.cascade, m.grid_size, nears, fars, None, -1, False, 128,
                                           False, opt.dt_gamma, opt.max_steps)
    out["train_xyzs"], out["train_dirs"], out["train_deltas"], out["train_rays"] = t2n(xt), t2n(dt_), t2n(dlt), t2n(rt)
    counter = torch.zeros(2, dtype=torch.int32)
    mean_count = max(128, int(xt.shape[0] * 0.6))              # too small on purpose: rays at the end are dropped (cu:457)
    xm, dm, dlm, rmn = rm.march_rays_train(rays_o, rays_d, m.bound, bits, m.cascade, m.grid_size, nears, fars, counter, mean_count,
                                           False, 128, False, opt.dt_gamma, opt.max_steps)
    out["trainm_mean_count"] = np.array(mean_count)
    out["trainm_xyzs"], out["trainm_deltas"], out["trainm_rays"], out["trainm_counter"] = t2n(xm), t2n(dlm), t2n(rmn), t2n(counter)
    # -- training compositor forward + backward (raymarching.py:284-342)
    Mt = xt.shape[0]
    sg = draw(Mt, lo=0.0, hi=40.0).requires_grad_(True)
    rg = draw(Mt, 3).requires_grad_(True)
    am = draw(Mt).requires_grad_(True)
    out["ctrain_sigmas"], out["ctrain_rgbs"], out["ctrain_ambient"] = t2n(sg), t2n(rg), t2n(am)
    wsum, asum, dep, img = rm.composite_rays_train(sg, rg, am, dlt, rt, 1e-4)
    g_ws, g_as, g_im = draw(N, lo=-1, hi=1), draw(N, lo=-1, hi=1), draw(N, 3, lo=-1, hi=1)
    out["ctrain_g_ws"], out["ctrain_g_as"], out["ctrain_g_im"] = t2n(g_ws), t2n(g_as), t2n(g_im)
    ((wsum * g_ws).sum() + (asum * g_as).sum() + (img * g_im).sum()).backward()
    out.update(ctrain_weights_sum=t2n(wsum), ctrain_ambient_sum=t2n(asum), ctrain_depth=t2n(dep), ctrain_image=t2n(img),
               ctrain_grad_sigmas=t2n(sg.grad), ctrain_grad_rgbs=t2n(rg.grad), ctrain_grad_ambient=t2n(am.grad))
    # -- marcher backward (--train_camera; raymarching.py:264-279)
    ro, rd = rays_o.clone().requires_grad_(True), rays_d.clone().requires_grad_(True)
    xb, db, _, _ = rm.march_rays_train(ro, rd, m.bound, bits, m.cascade, m.grid_size, nears, fars, None, -1, False, 128, False,
                                       opt.dt_gamma, opt.max_steps)
    gx, gd = draw(xb.shape[0], 3, lo=-1, hi=1), draw(xb.shape[0], 3, lo=-1, hi=1)
    out["trainb_gx"], out["trainb_gd"] = t2n(gx), t2n(gd)
    ((xb * gx).sum() + (db * gd).sum()).backward()
    out["trainb_grad_rays_o"], out["trainb_grad_rays_d"] = t2n(ro.grad), t2n(rd.grad)

    # -- occupancy bookkeeping (raymarching.py:83-181)
    rng = np.random.default_rng(9)
    coords = torch.from_numpy(rng.integers(0, 128, (2000, 3)).astype(np.int32))
    idx = rm.morton3D(coords)
    out["morton_coords"], out["morton_indices"] = t2n(coords), t2n(idx)
    out["morton_invert"] = t2n(rm.morton3D_invert(idx))
    grid = torch.from_numpy(rng.uniform(-1, 3, (2, 16 ** 3)).astype(np.float32))
    out["occ_grid"] = t2n(grid)
    out["occ_bits"] = t2n(rm.packbits(grid, 1.0))
    out["occ_dilated"] = t2n(rm.morton3D_dilation(grid))

    # -- grid encoder (gridencoder/grid.py:24-161)
    for name in ("hash3d", "tiled2d"):
        kw, x, grad = cases.grid_case(name)
        enc = cases.redraw_table(gridencoder.GridEncoder(**kw), 100 + len(name))
        xin = x.clone().requires_grad_(name == "tiled2d")          # requires_grad -> dy_dx -> grad_inputs (grid.py:156)
        y = enc(xin, bound=1)
        (y * grad).sum().backward()
        out[f"grid_{name}_offsets"] = t2n(enc.offsets)
        out[f"grid_{name}_out"] = t2n(y)
        out[f"grid_{name}_grad_table"] = t2n(enc.embeddings.grad)
        if xin.grad is not None:
            out[f"grid_{name}_grad_inputs"] = t2n(xin.grad)
    # autocast rule (grid.py:41-44): half table, half outputs, float32 inputs
    kw, x, grad = cases.grid_case("half3d")
    enc = cases.redraw_table(gridencoder.GridEncoder(**kw), 123)
    torch.set_autocast_enabled(True)
    try:
        assert torch.is_autocast_enabled()
        xin = x.clone().requires_grad_(True)
        y = enc(xin, bound=1)
        assert y.dtype == torch.float16
        (y.float() * grad).sum().backward()
    finally:
        torch.set_autocast_enabled(False)
    out["grid_half3d_out"] = t2n(y.float())
    out["grid_half3d_grad_table"] = t2n(enc.embeddings.grad)
    out["grid_half3d_grad_inputs"] = t2n(xin.grad)
    assert enc.embeddings.grad.dtype == torch.float32 and xin.grad.dtype == torch.float32
    # total-variation gradient (grid.py:163-184)
    kw, x, grad = cases.grid_case("tv3d")
    enc = cases.redraw_table(gridencoder.GridEncoder(**kw), 77)
    enc.embeddings.grad = torch.zeros_like(enc.embeddings)
    enc.grad_total_variation(weight=1e-3, inputs=x, bound=1)
    out["grid_tv3d_grad"] = t2n(enc.embeddings.grad)

    # -- SH (sphere_harmonics.py:14-86) and frequency (freq.py:15-76) encoders
    d, g = cases.dir_case()
    din = d.clone().requires_grad_(True)
    y = shencoder.SHEncoder(degree=4)(din)
    (y * g).sum().backward()
    out["sh_out"], out["sh_grad_inputs"] = t2n(y), t2n(din.grad)
    for D, deg in ((2, 10), (6, 4)):
        x, g = cases.freq_case(D, deg)
        xin = x.clone().requires_grad_(True)
        y = freqencoder.FreqEncoder(input_dim=D, degree=deg)(xin)
        (y * g).sum().backward()
        out[f"freq{D}_out"], out[f"freq{D}_grad_inputs"] = t2n(y), t2n(xin.grad)
    return out


# --------------------------------------------------------------------------------------------- reference_frames.npz
def make_frames(env):
    SyntheticScene, default_opt, rm, gridencoder, _, _, ref_network, ref_utils = env
    out = {}
    kw = lambda opt: {"dt_gamma": opt.dt_gamma, "max_steps": opt.max_steps}  # noqa: E731

    # ---- BASELINE config[1] in small: 64 x 64, xyz grid = hash, T = 2^19, two frames (EMA)
    opt = default_opt()
    torch.manual_seed(0)
    scene = cases.swap_in_hash19(SyntheticScene, opt, gridencoder.GridEncoder, 64, 64, model=ref_network.NeRFNetwork(opt))
    m = scene.model
    assert m.encoder.gridtype == "hash" and m.encoder.embeddings.shape[0] == 6119864
    out["hash19_table_sha256"] = np.array(sha(t2n(m.encoder.embeddings)))
    m.enc_a = None
    for i in (0, 1):
        f = scene.frame(i)
        with torch.no_grad():
            res = m.render(f["rays_o"], f["rays_d"], f["auds"], f["bg_coords"], f["poses"], eye=f["eye"], index=0,
                           bg_color=f["bg_color"], staged=True, perturb=False, **kw(opt))
        out[f"hash19_frame{i}_image"] = t2n(res["image"]).reshape(-1, 3)
        out[f"hash19_frame{i}_depth"] = t2n(res["depth"]).reshape(-1)
        out[f"hash19_frame{i}_enc_a"] = t2n(m.enc_a)
    del scene, m

    # ---- BASELINE config[0]: ONE 256 x 256 frame of the shipped model, everything on the CPU
    opt = default_opt()
    torch.manual_seed(0)
    scene = SyntheticScene(H=256, W=256, n_frames=8, device="cpu", opt=opt, model=ref_network.NeRFNetwork(opt))
    m = scene.model
    m.enc_a = None
    f = scene.frame(0)
    with torch.no_grad():
        res = m.render(f["rays_o"], f["rays_d"], f["auds"], f["bg_coords"], f["poses"], eye=f["eye"], index=0,
                       bg_color=f["bg_color"], staged=True, perturb=False, **kw(opt))
    out["config0_image"] = t2n(res["image"]).reshape(-1, 3)
    out["config0_depth"] = t2n(res["depth"]).reshape(-1)
    out["config0_enc_a"] = t2n(m.enc_a)

    # ---- BASELINE config[2] call shape: the TRAIN branch of run_cuda on 4096 rays of that frame (renderer.py:206-223),
    # head model (torso off, as the head is trained), then backward of a seeded scalar
    opt = default_opt(torso=False, smooth_lips=False)
    torch.manual_seed(0)
    scene = SyntheticScene(H=256, W=256, n_frames=8, device="cpu", opt=opt, model=ref_network.NeRFNetwork(opt))
    m = scene.model
    m.train()
    f = scene.frame(0)
    px = cases.train_pixels(256 * 256)
    out["train_px"] = t2n(px)
    for tag, mean_count in (("first", 0), ("steady", 49152)):
        m.zero_grad(set_to_none=True)
        m.mean_count, m.local_step = mean_count, 0
        m.step_counter.zero_()
        res = m.render(f["rays_o"][:, px], f["rays_d"][:, px], f["auds"], f["bg_coords"][:, px], f["poses"], eye=f["eye"], index=[0],
                       bg_color=f["bg_color"][:, px], staged=False, perturb=False, force_all_rays=False, **kw(opt))
        g = cases.rm_inputs(17)
        loss = (res["image"].reshape(-1, 3) * g(4096, 3, lo=-1, hi=1)).sum() + (res["weights_sum"] * g(4096, lo=-1, hi=1)).sum() \
            + (res["ambient"] * g(4096, lo=-1, hi=1)).sum()
        loss.backward()
        out[f"train_{tag}_image"] = t2n(res["image"]).reshape(-1, 3)
        out[f"train_{tag}_depth"] = t2n(res["depth"]).reshape(-1)
        out[f"train_{tag}_weights_sum"] = t2n(res["weights_sum"])
        out[f"train_{tag}_ambient"] = t2n(res["ambient"])
        out[f"train_{tag}_counter"] = t2n(m.step_counter[0])
        out[f"train_{tag}_loss"] = t2n(loss)
        for name in ("sigma_net.net.2.weight", "color_net.net.0.weight", "ambient_net.net.0.weight", "audio_net.encoder_fc1.2.weight",
                     "individual_codes"):
            p = dict(m.named_parameters())[name]
            gr = p.grad if name != "individual_codes" else p.grad[:1]
            out[f"train_{tag}_grad::{name}"] = t2n(gr)
        for name in ("encoder", "encoder_ambient"):
            gt = getattr(m, name).embeddings.grad
            nz = torch.nonzero(gt.abs().sum(1)).reshape(-1)
            out[f"train_{tag}_gradrows::{name}"] = t2n(nz[::97].int())        # a fixed 1-in-97 sample of the touched rows
            out[f"train_{tag}_gradvals::{name}"] = t2n(gt[nz[::97]])
            out[f"train_{tag}_gradsum::{name}"] = np.array([float(gt.double().sum()), float(gt.double().abs().sum()), float(nz.numel())])
    return out


# ------------------------------------------------------------------------------------------ reference_train_stable.npz
STABLE_PARAMS = ("ambient_net.net.0.weight", "ambient_net.net.1.weight", "ambient_net.net.2.weight", "audio_net.encoder_conv.0.weight",
                 "audio_net.encoder_fc1.2.weight", "audio_att_net.attentionConvNet.0.weight", "sigma_net.net.0.weight",
                 "sigma_net.net.2.weight", "color_net.net.0.weight", "individual_codes")


def make_stable(env):
    """The config-2 call of make_frames ("steady" budget) once more, with the gradient restricted to the samples at which
    NeRFNetwork.forward is smooth in its parameters: no ambient coordinate within 2e-5 (normalised units) of a cell boundary of
    any level of the 2-D grid (d(grid)/d(coordinate) is piecewise constant: across a boundary a sample's term jumps), no hidden
    pre-activation of the three MLPs within 1e-4 of zero (ReLU; pre-activations of the two implementations differ by up to ~1e-4).  The restriction is made OUTSIDE the reference's code: forward
    hooks on the unmodified model read the ambient coordinates and pre-activations, and gradient hooks on forward()'s three
    outputs zero the rows of the other samples.  With it, the gradients of the parameters upstream of the ambient grid
    (ambient_net, the audio nets) can be compared at 5e-3 instead of the 3e-2 the unrestricted sum allows."""
    SyntheticScene, default_opt, rm, gridencoder, _, _, ref_network, ref_utils = env
    opt = default_opt(torso=False, smooth_lips=False)
    torch.manual_seed(0)
    scene = SyntheticScene(H=256, W=256, n_frames=8, device="cpu", opt=opt, model=ref_network.NeRFNetwork(opt))
    m = scene.model
    m.train()
    f = scene.frame(0)
    px = cases.train_pixels(256 * 256)
    cell_margin, relu_margin = 2e-5, 1e-4
    seen = {"relu": [], "cell": None, "mask": None}

    def relu_hook(mod, args, out):                         # before the MLP's (in-place) ReLU touches `out`
        seen["relu"].append((out.detach().abs() > relu_margin).all(-1))

    def cell_hook(mod, args):
        enc = m.encoder_ambient
        amb = args[0].detach().double().reshape(-1, 2)
        scales = torch.tensor([2.0 ** (l * float(np.log2(enc.per_level_scale))) * enc.base_resolution - 1 for l in range(enc.num_levels)],
                              dtype=torch.float64)
        pos = ((amb + 1) / 2).unsqueeze(-1) * scales + 0.5
        frac = pos - pos.floor()
        margin = cell_margin * scales
        seen["cell"] = ((frac > margin) & (frac < 1 - margin)).all(-1).all(-1)

    def out_hook(mod, args, outs):
        mask = torch.stack(seen["relu"]).all(0) & seen["cell"]
        seen["mask"] = mask
        seen["outs"] = [t.detach().clone() for t in outs]
        seen["enc_a"] = args[2].detach().clone()
        for t in outs:
            t.register_hook(lambda g, k=mask: g * k.to(g.dtype).reshape(-1, *([1] * (g.dim() - 1))))

    hooks = [layer.register_forward_hook(relu_hook) for net in (m.ambient_net, m.sigma_net, m.color_net) for layer in list(net.net)[:-1]]
    hooks.append(m.encoder_ambient.register_forward_pre_hook(cell_hook))
    hooks.append(m.register_forward_hook(out_hook))
    hooks.append(m.encoder.register_forward_hook(lambda mod, a, o: seen.__setitem__("enc_x", o.detach().clone())))
    hooks.append(m.encoder_ambient.register_forward_hook(lambda mod, a, o: seen.__setitem__("enc_w", o.detach().clone())))
    m.zero_grad(set_to_none=True)
    m.mean_count, m.local_step = 49152, 0
    m.step_counter.zero_()
    res = m.render(f["rays_o"][:, px], f["rays_d"][:, px], f["auds"], f["bg_coords"][:, px], f["poses"], eye=f["eye"], index=[0],
                   bg_color=f["bg_color"][:, px], staged=False, perturb=False, force_all_rays=False, dt_gamma=opt.dt_gamma,
                   max_steps=opt.max_steps)
    g = cases.rm_inputs(17)
    loss = (res["image"].reshape(-1, 3) * g(4096, 3, lo=-1, hi=1)).sum() + (res["weights_sum"] * g(4096, lo=-1, hi=1)).sum() \
        + (res["ambient"] * g(4096, lo=-1, hi=1)).sum()
    loss.backward()
    for h in hooks:
        h.remove()
    mask = seen["mask"]
    out = {"train_px": t2n(px), "mask": t2n(mask).astype(np.uint8), "counter": t2n(m.step_counter[0]), "loss": t2n(loss),
           "margins": np.array([cell_margin, relu_margin])}
    for name, t in zip(("sigma", "color", "ambient"), seen["outs"]):     # every 8th sample's outputs: the per-sample check
        out[f"every8::{name}"] = t2n(t[::8])
    out["enc_a"] = t2n(seen["enc_a"])
    out["every256::enc_x"], out["every256::enc_w"] = t2n(seen["enc_x"][::256]), t2n(seen["enc_w"][::256])
    print(f"stable samples: {int(mask.sum())} of {mask.numel()}")
    params = dict(m.named_parameters())
    for name in STABLE_PARAMS:
        gr = params[name].grad
        out[f"grad::{name}"] = t2n(gr if name != "individual_codes" else gr[:1])
    for name in ("encoder", "encoder_ambient"):
        gt = getattr(m, name).embeddings.grad
        nz = torch.nonzero(gt.abs().sum(1)).reshape(-1)
        out[f"gradrows::{name}"] = t2n(nz[::97].int())
        out[f"gradvals::{name}"] = t2n(gt[nz[::97]])
        out[f"gradsum::{name}"] = np.array([float(gt.double().sum()), float(gt.double().abs().sum()), float(nz.numel())])
    return out


# ---------------------------------------------------------------------------------------------- reference_mixed.npz
class reference_autocast:
    """The reference's `-O` mode (main.py:111-120: fp16 + cuda_ray) as Trainer.test runs it: the whole test_step under
    `torch.cuda.amp.autocast(enabled=True)` (nerf/utils.py:944).  On this CPU-only box that context disables itself, so the
    same two switches are set by hand: (1) the legacy CUDA autocast flag, which is what the reference's own code looks at --
    `torch.is_autocast_enabled()` in gridencoder/grid.py:41-44 (half table, half outputs) and the
    `@custom_fwd(cast_inputs=torch.float32)` of raymarching / shencoder / freqencoder / activation.trunc_exp (fp32 inputs,
    autocast off inside); (2) torch's CPU autocast in float16, which makes nn.Linear / nn.Conv1d of the unmodified
    nerf/network.py run in fp16 with fp32-promoting `cat`, exactly the op classes CUDA autocast lowers / promotes."""

    def __enter__(self):
        import torch.amp.autocast_mode as am
        self.cpu = torch.autocast("cpu", dtype=torch.float16)
        self.cpu.__enter__()
        torch.set_autocast_enabled(True)
        assert torch.is_autocast_enabled() and torch.is_autocast_enabled("cpu")
        # custom_fwd(cast_inputs=float32) casts the floating-point tensors that live on ITS device ("cuda"); here every tensor
        # stands in for a CUDA tensor (the `.cuda()` hops are the identity), so CPU tensors are cast too
        self.am, self.cast = am, am._cast
        am._cast = lambda value, device_type, dtype: self.cast(value, "cpu" if device_type == "cuda" else device_type, dtype)
        return self

    def __exit__(self, *exc):
        self.am._cast = self.cast
        torch.set_autocast_enabled(False)
        return self.cpu.__exit__(*exc)


def make_mixed(env):
    """NeRFNetwork.forward / forward_torso / density and whole frames of the UNMODIFIED reference under fp16 autocast."""
    SyntheticScene, default_opt, rm, gridencoder, _, _, ref_network, ref_utils = env
    out = {}
    flow = np.load(os.path.join(HERE, "reference_flow.npz"))
    opt = default_opt()
    torch.manual_seed(0)
    scene = SyntheticScene(H=32, W=32, n_frames=8, device="cpu", opt=opt, model=ref_network.NeRFNetwork(opt))
    m = scene.model
    x, d, xy = (torch.from_numpy(flow[k]) for k in ("net_x", "net_d", "torso_xy"))
    f = scene.frame(0)            # the loader's work (rays, windows) happens outside the autocast region (nerf/utils.py:934-946)
    with torch.no_grad(), reference_autocast():
        enc_a0 = m.encode_audio(f["auds"])
        c, ct = m.individual_codes[0], m.individual_codes_torso[0]
        sigma, color, amb = m(x, d, enc_a0, c, scene.eye)
        dens = m.density(x, enc_a0, scene.eye)["sigma"]
        ta, tc, tdx = m.forward_torso(xy, scene.poses6[0:1], enc_a0, ct)
    out["dtypes"] = np.array([f"{n}:{t.dtype}" for n, t in (("enc_a", enc_a0), ("sigma", sigma), ("color", color), ("ambient", amb),
                                                            ("torso_alpha", ta), ("torso_color", tc), ("torso_dx", tdx))])
    out.update(net_enc_a=t2n(enc_a0.float()), net_sigma=t2n(sigma.float()), net_color=t2n(color.float()), net_ambient=t2n(amb.float()),
               net_density=t2n(dens.float()), torso_alpha=t2n(ta.float()), torso_color=t2n(tc.float()), torso_dx=t2n(tdx.float()))
    # the same calls with autocast off but the SAME audio code: isolates what the 16-bit layers change
    with torch.no_grad():
        s32, c32, a32 = m(x, d, enc_a0.float(), c, scene.eye)
    out.update(net_sigma_fp32=t2n(s32), net_color_fp32=t2n(c32), net_ambient_fp32=t2n(a32))

    # ---- a 32 x 32 frame of the shipped (tiled) model and a 64 x 64 frame of the hash T = 2^19 model through run_cuda
    kw = {"dt_gamma": opt.dt_gamma, "max_steps": opt.max_steps}
    m.enc_a = None
    f = scene.frame(0)
    with torch.no_grad(), reference_autocast():
        res = m.render(f["rays_o"], f["rays_d"], f["auds"], f["bg_coords"], f["poses"], eye=f["eye"], index=0, bg_color=f["bg_color"],
                       staged=True, perturb=False, **kw)
    out["tiled16_frame0_image"] = t2n(res["image"].float()).reshape(-1, 3)
    out["tiled16_frame0_depth"] = t2n(res["depth"].float()).reshape(-1)
    out["tiled16_frame0_enc_a"] = t2n(m.enc_a.float())
    del scene, m
    opt = default_opt()
    torch.manual_seed(0)
    scene = cases.swap_in_hash19(SyntheticScene, opt, gridencoder.GridEncoder, 64, 64, model=ref_network.NeRFNetwork(opt))
    m = scene.model
    m.enc_a = None
    f = scene.frame(0)
    with torch.no_grad(), reference_autocast():
        res = m.render(f["rays_o"], f["rays_d"], f["auds"], f["bg_coords"], f["poses"], eye=f["eye"], index=0, bg_color=f["bg_color"],
                       staged=True, perturb=False, **kw)
    out["hash19_frame0_image"] = t2n(res["image"].float()).reshape(-1, 3)
    out["hash19_frame0_depth"] = t2n(res["depth"].float()).reshape(-1)
    out["hash19_frame0_enc_a"] = t2n(m.enc_a.float())
    return out


def main():
    env = install_reference()
    todo = sys.argv[1:] or ["flow", "ops", "frames", "mixed", "train_stable"]
    for name, fn in (("flow", make_flow), ("ops", make_ops), ("frames", make_frames), ("mixed", make_mixed),
                     ("train_stable", make_stable)):
        if name in todo:
            out = fn(env)
            path = os.path.join(HERE, f"reference_{name}.npz")
            np.savez_compressed(path, **out)
            print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()

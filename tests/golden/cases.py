"""Seeded INPUTS of the golden cases, shared by the generator (tests/golden/make_golden.py, which feeds them to the
reference's own wrappers) and by the tests (which feed them to the oracle on CPU and to the HIP operators on the GPU box).
Only numpy / torch CPU code here; nothing reads /root/reference."""
import numpy as np
import torch

HASH19 = dict(gridtype="hash", log2_hashmap_size=19)


def swap_in_hash19(scene_cls, opt, grid_encoder_cls, H, W, model=None, n_frames=8):
    """The synthetic scene with the xyz encoder replaced by an instant-ngp hash grid, T = 2^19 (BASELINE config[1]).
    The model is first built exactly like the shipped one (tiled T = 2^16: same seeded nn.Linear / Conv1d values as the
    reference), then `model.encoder` is exchanged and the three tables are re-drawn from the scene's own generator -- so
    the reference model and this tree's mirror end up with identical parameters without sharing any file."""
    from radnerf.scene import init_synthetic_state
    scene = scene_cls(H=H, W=W, n_frames=n_frames, device="cpu", opt=opt, model=model)
    m = scene.model
    m.encoder = grid_encoder_cls(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                                 desired_resolution=2048 * m.bound, gridtype="hash")
    init_synthetic_state(m, opt, 0)
    return scene


def rm_inputs(seed=3):
    """Random material for the raymarching-operator cases (sigmas / rgbs / ambient / incoming gradients)."""
    rng = np.random.default_rng(seed)

    def draw(*shape, lo=0.0, hi=1.0):
        return torch.from_numpy(rng.uniform(lo, hi, shape).astype(np.float32))
    return draw


def grid_case(name):
    """(constructor kwargs, inputs in [-1,1] (a few outside), incoming gradient seed) of the GridEncoder cases."""
    rng = np.random.default_rng({"hash3d": 21, "tiled2d": 22, "half3d": 23, "tv3d": 24}[name])
    if name == "tiled2d":
        kw = dict(input_dim=2, num_levels=8, level_dim=2, base_resolution=16, log2_hashmap_size=10, desired_resolution=512,
                  gridtype="tiled")
        B = 700
    elif name == "half3d":
        kw = dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=12, desired_resolution=2048,
                  gridtype="tiled")
        B = 900
    else:
        kw = dict(input_dim=3, num_levels=8, level_dim=2, base_resolution=16, log2_hashmap_size=12, desired_resolution=512,
                  gridtype="hash")
        B = 600
    x = rng.uniform(-1.0, 1.0, (B, kw["input_dim"])).astype(np.float32)
    x[5] = 1.25       # outside [-bound, bound]: zero features (gridencoder.cu:110-135)
    x[17, 0] = -1.5
    x[1] = 1.0        # exactly on the upper face
    x[2] = -1.0
    grad = rng.standard_normal((B, kw["num_levels"] * kw["level_dim"])).astype(np.float32)
    return kw, torch.from_numpy(x), torch.from_numpy(grad)


def redraw_table(enc, seed):
    g = torch.Generator().manual_seed(seed)
    enc.embeddings.data = (torch.rand(enc.embeddings.shape, generator=g) * 2 - 1) * 0.5
    return enc


def dir_case(seed=31, B=500):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((B, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return torch.from_numpy(d), torch.from_numpy(rng.standard_normal((B, 16)).astype(np.float32))


def freq_case(D, deg, seed=41, B=300):
    rng = np.random.default_rng(seed + D)
    x = rng.uniform(-1, 1, (B, D)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(rng.standard_normal((B, D + 2 * D * deg)).astype(np.float32))


def train_pixels(n_px, n_rays=4096, seed=5):
    """Pixel subset of the train-branch case (config[2]: 4096 rays of a frame)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n_px, (n_rays,), generator=g)

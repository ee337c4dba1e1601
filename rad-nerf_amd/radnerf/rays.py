"""Inputs of the render path: rays, background coordinates, head-pose 6-vectors, audio windows.

Restates the pieces of the reference's nerf/utils.py that feed NeRFRenderer.render:
get_rays (:249-333), get_bg_coords (:240-245), convert_poses (:231-237) with its euler extraction
(:87-169), get_audio_features (:42-72); plus the OrbitCamera pose of nerf/gui.py:12-34 used by the
synthetic pose stream.  Pure torch, any device.
"""
import math

import numpy as np
import torch


def get_audio_features(features, att_mode, index):
    """8-frame attention window around `index`, zero padded at the ends (nerf/utils.py:42-72)."""
    if att_mode == 0:
        return features[[index]]
    if att_mode == 1:
        left = index - 8
        pad_left = 0
        if left < 0:
            pad_left, left = -left, 0
        auds = features[left:index]
        if pad_left > 0:
            auds = torch.cat([torch.zeros(pad_left, *auds.shape[1:], device=auds.device, dtype=auds.dtype), auds], dim=0)
        return auds
    if att_mode == 2:
        left, right = index - 4, index + 4
        pad_left = pad_right = 0
        if left < 0:
            pad_left, left = -left, 0
        if right > features.shape[0]:
            pad_right, right = right - features.shape[0], features.shape[0]
        auds = features[left:right]
        if pad_left > 0:
            auds = torch.cat([torch.zeros_like(auds[:pad_left]), auds], dim=0)
        if pad_right > 0:
            auds = torch.cat([auds, torch.zeros_like(auds[:pad_right])], dim=0)
        return auds
    raise NotImplementedError(f"wrong att_mode: {att_mode}")


def matrix_to_euler_xyz(matrix):
    """XYZ Tait-Bryan angles of rotation matrices [...,3,3] (nerf/utils.py:130-169 with convention 'XYZ')."""
    central = torch.asin(matrix[..., 0, 2])
    first = torch.atan2(-matrix[..., 1, 2], matrix[..., 2, 2])
    third = torch.atan2(-matrix[..., 0, 1], matrix[..., 0, 0])
    return torch.stack((first, central, third), -1)


def euler_angles_to_matrix(euler_angles):
    """XYZ euler angles [...,3] (radians) -> rotation matrices [...,3,3] (nerf/utils.py:172-227)."""
    def axis_rot(axis, angle):
        c, s = torch.cos(angle), torch.sin(angle)
        one, zero = torch.ones_like(angle), torch.zeros_like(angle)
        flat = {"X": (one, zero, zero, zero, c, -s, zero, s, c),
                "Y": (c, zero, s, zero, one, zero, -s, zero, c),
                "Z": (c, -s, zero, s, c, zero, zero, zero, one)}[axis]
        return torch.stack(flat, -1).reshape(angle.shape + (3, 3))

    ex, ey, ez = torch.unbind(euler_angles, -1)
    return torch.matmul(torch.matmul(axis_rot("X", ex), axis_rot("Y", ey)), axis_rot("Z", ez))


def convert_poses(poses):
    """[B,4,4] cam2world -> [B,6] = (euler xyz, translation) (nerf/utils.py:231-237)."""
    out = torch.empty(poses.shape[0], 6, dtype=torch.float32, device=poses.device)
    out[:, :3] = matrix_to_euler_xyz(poses[:, :3, :3].float())
    out[:, 3:] = poses[:, :3, 3]
    return out


def get_bg_coords(H, W, device):
    """[1, H*W, 2] in [-1,1]; component 0 varies along rows (nerf/utils.py:240-245)."""
    X = torch.arange(H, device=device) / (H - 1) * 2 - 1
    Y = torch.arange(W, device=device) / (W - 1) * 2 - 1
    xs, ys = torch.meshgrid(X, Y, indexing="ij")
    return torch.cat([xs.reshape(-1, 1), ys.reshape(-1, 1)], dim=-1).unsqueeze(0)


def get_rays(poses, intrinsics, H, W, N=-1):
    """Full-image (N <= 0) or N random rays; row-major pixels, centres at +0.5 (nerf/utils.py:249-333)."""
    device = poses.device
    B = poses.shape[0]
    fx, fy, cx, cy = intrinsics
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device),
                          indexing="ij")
    i = i.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
    j = j.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
    results = {}
    if N > 0:
        N = min(N, H * W)
        inds = torch.randint(0, H * W, size=[N], device=device).expand([B, N])
        i = torch.gather(i, -1, inds)
        j = torch.gather(j, -1, inds)
    else:
        inds = torch.arange(H * W, device=device).expand([B, H * W])
    results["i"], results["j"], results["inds"] = i, j, inds
    zs = torch.ones_like(i)
    xs = (i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    directions = torch.stack((xs, ys, zs), dim=-1)
    directions = directions / torch.norm(directions, dim=-1, keepdim=True)
    rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    results["rays_o"], results["rays_d"] = rays_o, rays_d
    return results


def orbit_pose(radius, yaw_deg=0.0, pitch_deg=0.0):
    """cam2world of the GUI's OrbitCamera (nerf/gui.py:12-34) after orbiting by yaw (about its up axis
    (1,0,0)) and pitch (about its side axis); returns a float32 [4,4] numpy array."""
    def rot(axis, deg):
        a = np.asarray(axis, dtype=np.float64)
        a = a / np.linalg.norm(a)
        t = math.radians(deg)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + math.sin(t) * K + (1 - math.cos(t)) * (K @ K)

    R0 = np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0]], dtype=np.float64)  # gui.py:19
    R = rot((1, 0, 0), yaw_deg) @ R0
    side = R[:, 0]
    R = rot(side, pitch_deg) @ R
    res = np.eye(4)
    res[2, 3] -= radius
    rotm = np.eye(4)
    rotm[:3, :3] = R
    return (rotm @ res).astype(np.float32)


def intrinsics_from_fovy(H, W, fovy_deg):
    """(fx, fy, cx, cy) as OrbitCamera.intrinsics computes them (nerf/gui.py:52-55)."""
    focal = H / (2 * math.tan(math.radians(fovy_deg) / 2))
    return np.array([focal, focal, W // 2, H // 2], dtype=np.float32)


def nerf_matrix_to_ngp(pose, scale=0.33, offset=(0.0, 0.0, 0.0)):
    """Dataset pose ([..., 4, 4] or [..., 3, 4], torch or numpy) -> the renderer's convention (nerf/provider.py:19-26): rows
    (y, z, x) of the input, second and third column negated, translation scaled and offset.  Any leading batch shape, on
    whatever device the input lives."""
    is_np = isinstance(pose, np.ndarray)
    p = torch.as_tensor(pose, dtype=torch.float32)
    out = torch.zeros(p.shape[:-2] + (4, 4), dtype=torch.float32, device=p.device)
    src = p[..., [1, 2, 0], :]                                  # new rows 0, 1, 2 <- old rows 1, 2, 0
    out[..., :3, 0] = src[..., 0]
    out[..., :3, 1] = -src[..., 1]
    out[..., :3, 2] = -src[..., 2]
    out[..., :3, 3] = src[..., 3] * scale + torch.as_tensor(offset, dtype=torch.float32, device=p.device)
    out[..., 3, 3] = 1.0
    return out.numpy() if is_np else out


def _rotation_mean(rots):
    """Chordal L2 mean of rotation matrices [..., n, 3, 3] -> [..., 3, 3]: the unit quaternion that is the dominant
    eigenvector of sum_j q_j q_j^T (what scipy's Rotation.mean() computes)."""
    m = rots
    # quaternion (x, y, z, w) of each matrix, numerically safe branch per element
    t = m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2]
    q = torch.stack([m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1], 1.0 + t], -1)
    alt = []
    for i, (j, k) in enumerate(((1, 2), (2, 0), (0, 1))):
        v = [None] * 4
        v[i] = 1.0 + m[..., i, i] - m[..., j, j] - m[..., k, k]
        v[j] = m[..., j, i] + m[..., i, j]
        v[k] = m[..., k, i] + m[..., i, k]
        v[3] = m[..., k, j] - m[..., j, k]
        alt.append(torch.stack(v, -1))
    cand = torch.stack([q] + alt, -2)                                      # [..., n, 4 candidates, 4]
    best = cand.norm(dim=-1).argmax(-1)                                    # the best conditioned candidate
    q = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)
    q = q / q.norm(dim=-1, keepdim=True)
    K = torch.einsum("...ni,...nj->...ij", q, q)
    _, vec = torch.linalg.eigh(K.double())
    x, y, z, w = vec[..., :, -1].float().unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(x.shape + (3, 3))


def smooth_camera_path(poses, kernel_size=5):
    """Moving average of a camera trajectory [N, 4, 4] (nerf/provider.py:29-45): translations averaged over the window
    [i - K, i + K] clipped to the stream, rotations replaced by the window's rotation mean.  Returns a new array / tensor of
    the input's kind (the reference overwrites its argument in place)."""
    is_np = isinstance(poses, np.ndarray)
    p = torch.as_tensor(poses, dtype=torch.float32).clone()
    N, K = p.shape[0], kernel_size // 2
    trans, rots = p[:, :3, 3].clone(), p[:, :3, :3].clone()
    for i in range(N):                     # windows differ in length only at the two ends; N is the clip length (host-side, once)
        lo, hi = max(0, i - K), min(N, i + K + 1)
        p[i, :3, 3] = trans[lo:hi].mean(0)
        p[i, :3, :3] = _rotation_mean(rots[lo:hi])
    return p.numpy() if is_np else p

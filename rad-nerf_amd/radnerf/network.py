"""Audio-conditioned NeRF network -- host-side mirror of the reference's nerf/network.py.

Same class names, constructor options, attribute / parameter names (so a reference checkpoint's
state_dict loads unchanged), and the same forward / forward_torso / density / encode_audio /
get_params contracts.  The encoders come from the drop-in packages of this tree (HIP kernels);
the per-sample MLP stack can run either as torch Linear layers (the reference's formulation) or,
for inference, through the fused gfx950 kernel (radnerf.fused).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from activation import trunc_exp
from encoding import get_encoder

from .renderer import NeRFRenderer


class AudioAttNet(nn.Module):
    """Attention over the 8-frame audio window (nerf/network.py:10-37)."""

    def __init__(self, dim_aud=64, seq_len=8):
        super().__init__()
        self.seq_len = seq_len
        self.dim_aud = dim_aud
        chans = [dim_aud, 16, 8, 4, 2, 1]
        layers = []
        for cin, cout in zip(chans[:-1], chans[1:]):
            layers += [nn.Conv1d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.attentionConvNet = nn.Sequential(*layers)
        self.attentionNet = nn.Sequential(nn.Linear(seq_len, seq_len, bias=True), nn.Softmax(dim=1))

    def forward(self, x):
        # x: [1, seq_len, dim_aud] -> [1, dim_aud]
        y = self.attentionConvNet(x.permute(0, 2, 1))
        y = self.attentionNet(y.view(1, self.seq_len)).view(1, self.seq_len, 1)
        return torch.sum(y * x, dim=1)


class AudioNet(nn.Module):
    """Per-frame audio feature extractor (nerf/network.py:41-67)."""

    def __init__(self, dim_in=29, dim_aud=64, win_size=16):
        super().__init__()
        self.win_size = win_size
        self.dim_aud = dim_aud
        chans = [dim_in, 32, 32, 64, 64]
        layers = []
        for cin, cout in zip(chans[:-1], chans[1:]):
            layers += [nn.Conv1d(cin, cout, kernel_size=3, stride=2, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.encoder_conv = nn.Sequential(*layers)
        self.encoder_fc1 = nn.Sequential(nn.Linear(64, 64), nn.LeakyReLU(0.02, True), nn.Linear(64, dim_aud))

    def forward(self, x):
        half_w = int(self.win_size / 2)
        x = x[:, :, 8 - half_w:8 + half_w]
        x = self.encoder_conv(x).squeeze(-1)
        return self.encoder_fc1(x)


class _SkinnyLinear(torch.autograd.Function):
    """y = x W^T for x [M, in] with M >> in, out (the per-sample MLP layers: M ~ 65 k, in / out <= 96).

    Forward and grad_x are ordinary GEMMs.  The weight gradient gy^T x is all reduction and almost no output ([out, in] from a
    65 k-long contraction): rocBLAS runs it as ONE workgroup walking the whole K (1.6 ms per layer, measured), so it is
    computed here as a batched product over row chunks -- [S, out, rows] x [S, rows, in] -- followed by a sum over the
    chunks: every CU gets a slice of the reduction.  Same values up to fp32 summation order."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return F.linear(x, w)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gy @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            M = x.shape[0]
            rows = 2048
            while rows > 1 and M % rows:
                rows >>= 1
            if rows >= 64 and M // rows > 1:
                gw = torch.bmm(gy.reshape(M // rows, rows, -1).transpose(1, 2), x.reshape(M // rows, rows, -1)).sum(0)
            else:
                gw = gy.t() @ x
        return gx, gw


_MLP_KERNELS = []
_AUDIO_KERNELS = []
_TRAIN_GLUE = []
_TRAIN_HEAD = []


def _train_head():
    """radnerf.train_head (the whole per-sample network as one forward + one backward kernel; needs the HIP library and a GPU),
    or None on the CPU."""
    if not torch.cuda.is_available():
        return None
    if not _TRAIN_HEAD:
        from . import train_head
        _TRAIN_HEAD.append(train_head)
    return _TRAIN_HEAD[0]


def _train_glue():
    """radnerf.train_glue (needs the HIP library and a GPU), or None on the CPU."""
    if not torch.cuda.is_available():
        return None
    if not _TRAIN_GLUE:
        from . import train_glue
        _TRAIN_GLUE.append(train_glue)
    return _TRAIN_GLUE[0]


def _audio_kernels(model, need_grad=True):
    """radnerf.audio when its kernels cover the model's audio nets, or None (RN_AUDIO_TRAIN=torch: the nn.Module path)."""
    import os
    if os.environ.get("RN_AUDIO_TRAIN", "hip") == "torch":
        return None
    if not _AUDIO_KERNELS:
        from . import audio
        _AUDIO_KERNELS.append(audio)
    a = _AUDIO_KERNELS[0]
    return a if a.supported(model) and (not need_grad or any(p.requires_grad for p in model.audio_net.parameters())) else None


def _mlp_kernels():
    """radnerf.mlp_train (needs the HIP library), or None when RN_MLP_TRAIN=torch selects the nn.Linear path."""
    import os
    if os.environ.get("RN_MLP_TRAIN", "hip") == "torch":
        return None
    if not _MLP_KERNELS:
        from . import mlp_train
        _MLP_KERNELS.append(mlp_train)
    return _MLP_KERNELS[0]


class MLP(nn.Module):
    """Bias-free Linear stack with ReLU between layers (nerf/network.py:69-88)."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.net = nn.ModuleList([
            nn.Linear(dim_in if l == 0 else dim_hidden, dim_out if l == num_layers - 1 else dim_hidden, bias=False)
            for l in range(num_layers)])

    def forward_split(self, x, constants):
        """forward(cat[x, constant.repeat(rows) for constant in constants]) -- the form every call of the path has
        (nerf/network.py:236, 262, 274: the audio code / eye value / individual code repeated for every sample).  The training
        kernels never materialise the repeats: the constants enter the first layer as a bias."""
        consts = [c.reshape(1, -1) for c in constants if c is not None]
        if not consts:
            return self.forward(x)
        training_shape = x.is_cuda and torch.is_grad_enabled() and x.dim() == 2 and x.dtype == torch.float32 and \
            not torch.is_autocast_enabled()
        if training_shape and x.shape[0] >= 1024 and _mlp_kernels() is not None and x.shape[1] <= 92 and \
                _mlp_kernels().supported(x.shape[1], self.dim_out, self.dim_hidden, self.num_layers):
            return _mlp_kernels().fused_mlp(x, [layer.weight for layer in self.net], torch.cat(consts, dim=1))
        return self.forward(torch.cat([x] + [c.to(x.dtype).repeat(x.shape[0], 1) for c in consts], dim=-1))

    def forward(self, x):
        training_shape = x.is_cuda and torch.is_grad_enabled() and x.dim() == 2 and x.dtype == torch.float32 and \
            not torch.is_autocast_enabled()
        # training on the GPU: hand-written forward / backward kernels for the whole stack (radnerf/mlp_train.py), unless
        # RN_MLP_TRAIN=torch asks for the nn.Linear path (the parity tests compare the two)
        if training_shape and x.shape[0] >= 1024 and _mlp_kernels() is not None and \
                _mlp_kernels().supported(self.dim_in, self.dim_out, self.dim_hidden, self.num_layers):
            return _mlp_kernels().fused_mlp(x, [layer.weight for layer in self.net])
        skinny = training_shape and x.shape[0] >= 4096
        for l, layer in enumerate(self.net):
            x = _SkinnyLinear.apply(x, layer.weight) if skinny and not torch.is_autocast_enabled() else layer(x)
            if l != self.num_layers - 1:
                x = F.relu(x, inplace=True)
        return x


class NeRFNetwork(NeRFRenderer):
    # nerf/network.py:91-167
    def __init__(self, opt, num_layers=3, hidden_dim=64, geo_feat_dim=64, num_layers_color=2, hidden_dim_color=64,
                 audio_dim=64, num_layers_ambient=3, hidden_dim_ambient=64, ambient_dim=2):
        super().__init__(opt)

        self.emb = self.opt.emb
        if "esperanto" in self.opt.asr_model:
            self.audio_in_dim = 44
        elif "deepspeech" in self.opt.asr_model:
            self.audio_in_dim = 29
        else:
            self.audio_in_dim = 32
        if self.emb:
            self.embedding = nn.Embedding(self.audio_in_dim, self.audio_in_dim)

        self.audio_dim = audio_dim
        self.audio_net = AudioNet(self.audio_in_dim, self.audio_dim)
        self.att = self.opt.att
        if self.att > 0:
            self.audio_att_net = AudioAttNet(self.audio_dim)

        grid = dict(num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16, interpolation="linear")
        # The reference hard-codes tiledgrid / T=2^16 for the xyz grid (nerf/network.py:70); `opt.xyz_grid` /
        # `opt.xyz_log2_hashmap_size` select the instant-ngp hash variant (BASELINE config 1: hash, T=2^19) instead.
        xyz_grid = dict(grid, log2_hashmap_size=int(getattr(opt, "xyz_log2_hashmap_size", 16)))
        self.encoder, self.in_dim = get_encoder(getattr(opt, "xyz_grid", "tiledgrid"), input_dim=3,
                                                desired_resolution=2048 * self.bound, **xyz_grid)
        self.encoder_ambient, self.in_dim_ambient = get_encoder("tiledgrid", input_dim=ambient_dim,
                                                                desired_resolution=2048, **grid)
        self.num_layers_ambient, self.hidden_dim_ambient, self.ambient_dim = num_layers_ambient, hidden_dim_ambient, ambient_dim
        self.ambient_net = MLP(self.in_dim + self.audio_dim, self.ambient_dim, self.hidden_dim_ambient, self.num_layers_ambient)

        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        self.eye_dim = 1 if self.exp_eye else 0
        self.sigma_net = MLP(self.in_dim + self.in_dim_ambient + self.eye_dim, 1 + self.geo_feat_dim, self.hidden_dim, self.num_layers)

        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.encoder_dir, self.in_dim_dir = get_encoder("spherical_harmonics")
        self.color_net = MLP(self.in_dim_dir + self.geo_feat_dim + self.individual_dim, 3, self.hidden_dim_color, self.num_layers_color)

        if self.torso:
            self.torso_deform_encoder, self.torso_deform_in_dim = get_encoder("frequency", input_dim=2, multires=10)
            self.pose_encoder, self.pose_in_dim = get_encoder("frequency", input_dim=6, multires=4)
            self.torso_deform_net = MLP(self.torso_deform_in_dim + self.pose_in_dim + self.individual_dim_torso, 2, 64, 3)
            self.torso_encoder, self.torso_in_dim = get_encoder("tiledgrid", input_dim=2, desired_resolution=2048, **grid)
            self.torso_net = MLP(self.torso_in_dim + self.torso_deform_in_dim + self.pose_in_dim + self.individual_dim_torso, 4, 32, 3)

    def encode_audio(self, a):
        # nerf/network.py:170-185; a: [8, audio_in_dim, 16] (or [8, 16] ids with --emb) -> [1, audio_dim]
        if a is None:
            return None
        if self.emb:
            a = self.embedding(a).transpose(-1, -2).contiguous()
        # training on the GPU: forward and backward of both audio nets as two kernels each (radnerf/audio.py) instead of
        # ~150 launches of tiny convolutions; RN_AUDIO_TRAIN=torch keeps the nn.Module path (the parity test compares them)
        if a.is_cuda and not self.emb and a.dtype == torch.float32 and not torch.is_autocast_enabled():
            if torch.is_grad_enabled():
                if _audio_kernels(self) is not None:
                    return _audio_kernels(self).encode_windows_train(self, a)
            elif _audio_kernels(self, need_grad=False) is not None:
                # no gradient wanted (the occupancy refresh draws a random window every 16 training steps): the forward kernels
                return _audio_kernels(self, need_grad=False).encode_windows(self, a)
        enc_a = self.audio_net(a)
        if self.att > 0:
            enc_a = self.audio_att_net(enc_a.unsqueeze(0))
        return enc_a

    def forward_torso(self, x, poses, enc_a, c=None):
        # nerf/network.py:188-219; x: [N,2] in [-1,1], poses: [1,6], c: [ind_dim_torso]
        x = x * self.opt.torso_shrink
        enc_pose = self.pose_encoder(poses)
        enc_x = self.torso_deform_encoder(x)
        # the pose encoding and the individual code are the same for every pixel of a call (the reference repeats and
        # concatenates them, network.py:198-201): MLP.forward_split takes them as constants -- columns in the same order,
        # [enc_x | enc_pose | c] for the deformation net and [grid | enc_x | enc_pose | c] for the torso net
        consts = [enc_pose, c]
        dx = self.torso_deform_net.forward_split(enc_x, consts)
        x = (x + dx).clamp(-1, 1)
        x = self.torso_encoder(x, bound=1)
        h = self.torso_net.forward_split(torch.cat([x, enc_x], dim=-1), consts)
        return torch.sigmoid(h[..., :1]), torch.sigmoid(h[..., 1:]), dx

    def _geometry(self, x, enc_a, e):
        """Shared head of forward() and density(): returns (sigma_net output, ambient)."""
        if enc_a is None:
            ambient = torch.zeros_like(x[:, :self.ambient_dim])
            enc_x = self.encoder(x, bound=self.bound)
            enc_w = self.encoder_ambient(ambient, bound=1)
        else:
            enc_x = self.encoder(x, bound=self.bound)
            ambient = self.ambient_net.forward_split(enc_x, [enc_a]).float()
            ambient = torch.tanh(ambient)
            enc_w = self.encoder_ambient(ambient, bound=1)
        return self.sigma_net.forward_split(torch.cat([enc_x, enc_w], dim=-1), [e]), ambient

    def forward(self, x, d, enc_a, c, e=None):
        # nerf/network.py:222-283; x: [N,3] in [-bound,bound], d: [N,3], enc_a: [1,64], c: [ind_dim], e: [1,1]
        th = _train_head()
        if th is not None and th.usable(self, x, enc_a):
            # training on the GPU: the whole network as one forward and one backward kernel (radnerf/train_head.py);
            # RN_TRAIN_HEAD=ops keeps the per-operator path below (the parity tests compare the two)
            sigma, color, ambient, _ = th.head_forward(self, x, d, enc_a, c, e)
            return sigma, color, ambient
        h, ambient = self._geometry(x, enc_a, e)
        enc_d = self.encoder_dir(d)
        glue = _train_glue()
        if glue is not None and h.dim() == 2 and h.shape[1] == 65 and glue.enabled(h, enc_d) and not enc_d.requires_grad:
            sigma, color_in = glue.head_mid(h, enc_d)           # one kernel: trunc_exp(h[:, 0]) and cat[enc_d, h[:, 1:]]
        else:
            sigma = trunc_exp(h[..., 0])
            color_in = torch.cat([enc_d, h[..., 1:]], dim=-1)
        color = torch.sigmoid(self.color_net.forward_split(color_in, [c]))
        return sigma, color, ambient

    def density(self, x, enc_a, e=None):
        # nerf/network.py:286-325
        h, _ = self._geometry(x, enc_a, e)
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def get_params(self, lr, lr_net, wd=0):
        # nerf/network.py:329-362
        if self.torso:
            params = [
                {"params": self.torso_encoder.parameters(), "lr": lr},
                {"params": self.torso_net.parameters(), "lr": lr_net, "weight_decay": wd},
                {"params": self.torso_deform_net.parameters(), "lr": lr_net, "weight_decay": wd},
            ]
            if self.individual_dim_torso > 0:
                params.append({"params": self.individual_codes_torso, "lr": lr_net, "weight_decay": wd})
            return params
        params = [
            {"params": self.audio_net.parameters(), "lr": lr_net, "weight_decay": wd},
            {"params": self.encoder.parameters(), "lr": lr},
            {"params": self.encoder_ambient.parameters(), "lr": lr},
            {"params": self.ambient_net.parameters(), "lr": lr_net, "weight_decay": wd},
            {"params": self.sigma_net.parameters(), "lr": lr_net, "weight_decay": wd},
            {"params": self.color_net.parameters(), "lr": lr_net, "weight_decay": wd},
        ]
        if self.att > 0:
            params.append({"params": self.audio_att_net.parameters(), "lr": lr_net * 5, "weight_decay": wd})
        if self.emb:
            params.append({"params": self.embedding.parameters(), "lr": lr})
        if self.individual_dim > 0:
            params.append({"params": self.individual_codes, "lr": lr_net, "weight_decay": wd})
        if self.train_camera:
            params.append({"params": self.camera_dT, "lr": 1e-5, "weight_decay": 0})
            params.append({"params": self.camera_dR, "lr": 1e-5, "weight_decay": 0})
        return params

"""ORACLE -- test infrastructure, NOT the product.

A plain-PyTorch (CPU, fp32) restatement of the reference's `--model registration` hot path
(heyblackC/DFMIR).  Every function cites the reference file:line it follows.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module; nothing under
`dfmir_amd/` does.

Pinning: the reference ships no tests / golden vectors, so this restatement is
pinned against outputs of the reference itself, imported in the build container from
/root/reference by `tests/golden/make_golden.py` and committed as `tests/golden/*.npz`
(`tests/test_oracle_golden.py` replays them).  All arithmetic is torch ATen CPU fp32.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# translation generator  (models/networks.py)
# ------------------------------------------------------------------------------------------------
def binomial_filter(n):
    """models/networks.py:15-34 get_filter."""
    row = np.array([math.comb(n - 1, k) for k in range(n)], dtype=np.float64)
    f = torch.tensor(row[:, None] * row[None, :], dtype=torch.float32)
    return f / f.sum()


class BlurDown(nn.Module):
    """models/networks.py:37-60 Downsample(filt_size=3, stride=2, reflect)."""

    def __init__(self, c):
        super().__init__()
        self.register_buffer('filt', binomial_filter(3)[None, None].repeat(c, 1, 1, 1))

    def forward(self, x):
        return F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), self.filt, stride=2, groups=x.shape[1])


class BlurUp(nn.Module):
    """models/networks.py:73-93 Upsample(filt_size=4, stride=2, replicate)."""

    def __init__(self, c):
        super().__init__()
        self.register_buffer('filt', (binomial_filter(4) * 4)[None, None].repeat(c, 1, 1, 1))

    def forward(self, x):
        y = F.conv_transpose2d(F.pad(x, (1, 1, 1, 1), mode='replicate'), self.filt, stride=2, padding=2,
                               groups=x.shape[1])
        return y[:, :, 1:, 1:][:, :, :-1, :-1]


class ResBlock(nn.Module):
    """models/networks.py:1164-1221 (reflect pad, instance norm, bias, no dropout)."""

    def __init__(self, c):
        super().__init__()
        inorm = lambda: nn.InstanceNorm2d(c, affine=False, track_running_stats=False)
        self.conv_block = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(c, c, 3), inorm(), nn.ReLU(True),
                                        nn.ReflectionPad2d(1), nn.Conv2d(c, c, 3), inorm())

    def forward(self, x):
        return x + self.conv_block(x)


class Generator(nn.Module):
    """models/networks.py:956-1051 ResnetGenerator with the defaults of the registration model
    (instance norm, antialiased down/up, reflect padding).  Module indices == the reference's."""

    def __init__(self, input_nc=1, output_nc=1, ngf=64, n_blocks=9):
        super().__init__()
        inorm = lambda c: nn.InstanceNorm2d(c, affine=False, track_running_stats=False)
        m = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, 7), inorm(ngf), nn.ReLU(True)]
        for i in range(2):
            c = ngf * 2 ** i
            m += [nn.Conv2d(c, 2 * c, 3, padding=1), inorm(2 * c), nn.ReLU(True), BlurDown(2 * c)]
        m += [ResBlock(4 * ngf) for _ in range(n_blocks)]
        for i in range(2):
            c = ngf * 2 ** (2 - i)
            m += [BlurUp(c), nn.Conv2d(c, c // 2, 3, padding=1), inorm(c // 2), nn.ReLU(True)]
        m += [nn.ReflectionPad2d(3), nn.Conv2d(ngf, output_nc, 7), nn.Tanh()]
        self.model = nn.Sequential(*m)

    def forward(self, x, layers=(), encode_only=False):
        """models/networks.py:1028-1051."""
        layers = list(layers)
        if not layers:
            return self.model(x)
        feats = []
        for i, layer in enumerate(self.model):
            x = layer(x)
            if i in layers:
                feats.append(x)
            if encode_only and i == layers[-1]:
                return feats
        return x, feats


def init_weights_xavier(net, gain=0.02):
    """models/networks.py:163-195 with init_type='xavier' (the option default)."""
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.Linear)):
            nn.init.xavier_normal_(m.weight.data, gain=gain)
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)


# ------------------------------------------------------------------------------------------------
# PatchNCE  (models/networks.py:493-624, models/patchnce.py)
# ------------------------------------------------------------------------------------------------
def l2_normalize(x):
    """models/networks.py:499-502 Normalize(2)."""
    return x / (x.pow(2).sum(1, keepdim=True).pow(0.5) + 1e-7)


class PatchSampler(nn.Module):
    """models/networks.py:575-624 PatchSampleF(use_mlp=True)."""

    def __init__(self, nc=256, use_mlp=True):
        super().__init__()
        self.nc, self.use_mlp, self.mlp_init = nc, use_mlp, False

    def create_mlp(self, feats):
        for i, f in enumerate(feats):
            setattr(self, 'mlp_%d' % i, nn.Sequential(nn.Linear(f.shape[1], self.nc), nn.ReLU(), nn.Linear(self.nc, self.nc)))
        init_weights_xavier(self)
        self.mlp_init = True

    def forward(self, feats, num_patches=64, patch_ids=None):
        if self.use_mlp and not self.mlp_init:
            self.create_mlp(feats)
        out, ids = [], []
        for i, f in enumerate(feats):
            S = f.shape[2] * f.shape[3]
            pid = patch_ids[i] if patch_ids is not None else torch.randperm(S)[:min(num_patches, S)]
            # == f.permute(0,2,3,1).flatten(1,2)[:, pid, :].flatten(0,1) of the reference, written so that
            # the gradient w.r.t. f is CONTIGUOUS: torch 2.10's CPU instance_norm backward returns wrong
            # values for a channels-last grad_output at N == 1, which the reference's
            # formulation produces at batch_size 1.  Values and (correct) gradients are identical.
            x = f.flatten(2)[:, :, pid].permute(0, 2, 1).flatten(0, 1)       # [B*P, C]
            if self.use_mlp:
                x = getattr(self, 'mlp_%d' % i)(x)
            ids.append(pid)
            out.append(l2_normalize(x))
        return out, ids


def patchnce_loss(feat_q, feat_k, batch_size, T=0.07, all_negatives=False):
    """models/patchnce.py:14-55 (negatives from the same image; reduction 'none').  all_negatives =
    opt.nce_includes_all_negatives_from_minibatch (:32-38): the whole minibatch as ONE group of negatives."""
    n, dim = feat_q.shape
    if all_negatives:
        batch_size = 1
    feat_k = feat_k.detach()
    l_pos = (feat_q * feat_k).sum(1, keepdim=True)
    q = feat_q.view(batch_size, -1, dim)
    k = feat_k.view(batch_size, -1, dim)
    npatch = q.shape[1]
    l_neg = torch.bmm(q, k.transpose(2, 1))
    l_neg = l_neg.masked_fill(torch.eye(npatch, dtype=torch.bool)[None], -10.0).view(-1, npatch)
    out = torch.cat((l_pos, l_neg), dim=1) / T
    return F.cross_entropy(out, torch.zeros(n, dtype=torch.long), reduction='none')


# ------------------------------------------------------------------------------------------------
# VoxelMorph  (models/voxelmorph/torchvoxelmorph/{layers,networks}.py)
# ------------------------------------------------------------------------------------------------
def spatial_transform(src, flow, mode='bilinear'):
    """layers.py:30-48 SpatialTransformer.forward."""
    shape = flow.shape[2:]
    grid = torch.stack(torch.meshgrid([torch.arange(0, s) for s in shape], indexing='ij')).unsqueeze(0).float()
    loc = grid + flow
    loc = torch.stack([2 * (loc[:, i] / (shape[i] - 1) - 0.5) for i in range(len(shape))], dim=1)
    if len(shape) == 2:
        loc = loc.permute(0, 2, 3, 1)[..., [1, 0]]
    else:
        loc = loc.permute(0, 2, 3, 4, 1)[..., [2, 1, 0]]
    return F.grid_sample(src, loc, align_corners=True, mode=mode)


def vec_int(vec, nsteps):
    """layers.py:64-68 VecInt.forward."""
    vec = vec * (1.0 / 2 ** nsteps)
    for _ in range(nsteps):
        vec = vec + spatial_transform(vec, vec)
    return vec


def resize_transform(x, vel_resize):
    """layers.py:71-97 ResizeTransform(vel_resize, ndims).forward."""
    factor = 1.0 / vel_resize
    mode = 'bilinear' if x.dim() == 4 else 'trilinear'
    if factor < 1:
        return factor * F.interpolate(x, align_corners=True, scale_factor=factor, mode=mode)
    if factor > 1:
        return F.interpolate(factor * x, align_corners=True, scale_factor=factor, mode=mode)
    return x


class VxmConvBlock(nn.Module):
    """networks.py:1506-1521."""

    def __init__(self, nd, cin, cout, stride=1):
        super().__init__()
        self.main = (nn.Conv2d if nd == 2 else nn.Conv3d)(cin, cout, 3, stride, 1)

    def forward(self, x):
        return F.leaky_relu(self.main(x), 0.2)


class VxmUnet(nn.Module):
    """networks.py:16-106."""

    def __init__(self, nd, enc_nf, dec_nf):
        super().__init__()
        self.enc_nf, self.dec_nf = enc_nf, dec_nf
        prev = 2
        self.downarm = nn.ModuleList()
        for nf in enc_nf:
            self.downarm.append(VxmConvBlock(nd, prev, nf, 2))
            prev = nf
        hist = list(reversed(enc_nf))
        self.uparm = nn.ModuleList()
        for i, nf in enumerate(dec_nf[:len(enc_nf)]):
            self.uparm.append(VxmConvBlock(nd, prev + hist[i] if i > 0 else prev, nf, 1))
            prev = nf
        prev += 2
        self.extras = nn.ModuleList()
        for nf in dec_nf[len(enc_nf):]:
            self.extras.append(VxmConvBlock(nd, prev, nf, 1))
            prev = nf

    def forward(self, x):
        enc = [x]
        for layer in self.downarm:
            enc.append(layer(enc[-1]))
        x = enc.pop()
        for layer in self.uparm:
            x = F.interpolate(layer(x), scale_factor=2, mode='nearest')
            x = torch.cat([x, enc.pop()], dim=1)
        for layer in self.extras:
            x = layer(x)
        return x


PLUGIN_UNET_FEATURES = [[16, 32, 32, 64, 64, 64], [64, 64, 64, 32, 32, 32, 16]]  # registration_model.py:93-96
DEFAULT_UNET_FEATURES = [[16, 32, 32, 32], [32, 32, 32, 32, 32, 16, 16]]         # networks.py:9-14


class VxmDense(nn.Module):
    """networks.py:1028-1145 with int_downsize=2.  state_dict keys match the reference's except the
    `grid` buffers, which the oracle does not carry."""

    def __init__(self, inshape, features=None, int_steps=7, bidir=False):
        super().__init__()
        nd = len(inshape)
        enc, dec = features if features is not None else DEFAULT_UNET_FEATURES
        self.unet_model = VxmUnet(nd, enc, dec)
        self.flow = (nn.Conv2d if nd == 2 else nn.Conv3d)(dec[-1], nd, 3, padding=1)
        self.flow.weight = nn.Parameter(torch.randn(self.flow.weight.shape) * 1e-5)
        self.flow.bias = nn.Parameter(torch.zeros(self.flow.bias.shape))
        self.int_steps, self.bidir = int_steps, bidir

    def forward(self, source, target, registration=False):
        x = self.unet_model(torch.cat([source, target], dim=1))
        pos = self.flow(x)
        if self.int_steps > 0:
            pos = resize_transform(pos, 2)
        preint = pos
        neg = -pos if self.bidir else None
        if self.int_steps > 0:
            pos = resize_transform(vec_int(pos, self.int_steps), 0.5)
            neg = resize_transform(vec_int(neg, self.int_steps), 0.5) if self.bidir else None
        y_source = spatial_transform(source, pos)
        y_target = spatial_transform(target, neg) if self.bidir else None
        if registration:
            return y_source, pos
        return (y_source, y_target, pos) if self.bidir else (y_source, preint)


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
def smoothing_loss(flow):
    """models/registration_model.py:25-32 smooothing_loss."""
    dy = flow[:, :, 1:, :] - flow[:, :, :-1, :]
    dx = flow[:, :, :, 1:] - flow[:, :, :, :-1]
    return ((dx * dx).mean() + (dy * dy).mean()) / 2.0


def grad_loss(flow, penalty='l2', mask=None, loss_mult=None):
    """util/losses.py:92-130 Grad_Loss._grad2d/_grad3d + forward: |forward differences| (squared for 'l2'), mean per axis,
    mean over axes; `mask` multiplies the field first (:119-121), `loss_mult` the result (:127-128).  The vxm `Grad.loss`
    (models/voxelmorph/torchvoxelmorph/losses.py:102-117) is the same arithmetic on a 3-D field."""
    if mask is not None:
        flow = flow * mask
    nd = flow.dim() - 2
    tot = 0.0
    for ax in range(2, 2 + nd):
        d = torch.abs(flow.narrow(ax, 1, flow.shape[ax] - 1) - flow.narrow(ax, 0, flow.shape[ax] - 1))
        tot = tot + ((d * d) if penalty == 'l2' else d).mean()
    tot = tot / float(nd)
    return tot if loss_mult is None else tot * loss_mult


def grad_loss_l2(flow):
    """util/losses.py:92-115 Grad_Loss._grad2d/_grad3d, penalty 'l2'."""
    return grad_loss(flow, 'l2')


def masked_l1(src, tgt, mask):
    """models/registration_model.py:255-263 calculate_L1_loss."""
    diff = torch.abs(src - tgt)
    if mask is None:
        return diff.mean()
    if mask.sum() == 0:
        return torch.tensor(0.0)
    return (1 / mask.sum()) * (diff * mask).sum()


def ncc_map(pred, target, win=9, eps=1e-5):
    """cc of util/losses.py:183-246 (NCC_Loss._compute_local_sums + ncc, 'mean' kernel) = the same map in
    models/voxelmorph/torchvoxelmorph/losses.py:15-65."""
    nd = pred.dim() - 2
    filt = torch.ones([1, 1] + [win] * nd)
    conv = F.conv2d if nd == 2 else F.conv3d
    pad = win // 2
    I, J = pred, target
    Is, Js = conv(I, filt, padding=pad), conv(J, filt, padding=pad)
    I2s, J2s, IJs = conv(I * I, filt, padding=pad), conv(J * J, filt, padding=pad), conv(I * J, filt, padding=pad)
    wn = filt.sum()
    uI, uJ = Is / wn, Js / wn
    cross = IJs - uJ * Is - uI * Js + uI * uJ * wn
    Iv = I2s - 2 * uI * Is + uI * uI * wn
    Jv = J2s - 2 * uJ * Js + uJ * uJ * wn
    return cross * cross / (Iv * Jv + eps)


def ncc_loss(pred, target, win=9, eps=1e-5, mask=None):
    """util/losses.py:248-261 NCC_Loss(kernel_type='mean').forward: -sqrt(mean(cc)); with a mask
    -sqrt(sum(cc * mask) / sum(mask)), and 0 when the mask is empty."""
    cc = ncc_map(pred, target, win, eps)
    if mask is None:
        return -1.0 * torch.sqrt(cc.mean())
    if torch.sum(mask) == 0:
        return torch.tensor(0)
    return -1.0 * torch.sqrt((1 / torch.sum(mask)) * torch.sum(cc * mask))


def vxm_ncc_loss(y_true, y_pred, win=9):
    """models/voxelmorph/torchvoxelmorph/losses.py:7-67 NCC(win).loss = -mean(cc), eps 1e-5."""
    return -torch.mean(ncc_map(y_true, y_pred, win, 1e-5))


# ------------------------------------------------------------------------------------------------
# whole train step  (models/registration_model.py:73-263)
# ------------------------------------------------------------------------------------------------
class RegistrationStep(object):
    """The reference's REGISTRATIONModel training step with the default CUT options
    (nce_idt=True, lambda_NCE=0.25, nce_layers 0,4,8,12,16, num_patches 256, nce_T 0.07, lr 2e-4,
    betas (0.5, 0.999)).  `ids_hook(call_index, n_layers_sizes) -> list of id tensors` lets tests
    pin the patch ids; otherwise torch.randperm is drawn like the reference."""

    def __init__(self, size, batch_size, ngf=64, n_blocks=9, lr=2e-4, betas=(0.5, 0.999), num_patches=256,
                 nce_T=0.07, lambda_NCE=0.25, nce_layers=(0, 4, 8, 12, 16), netF_nc=256, nce_idt=True,
                 flip_equivariance=False, nce_all_negatives=False):
        self.bs, self.size = batch_size, size
        self.nce_all_negatives = nce_all_negatives         # opt.nce_includes_all_negatives_from_minibatch (patchnce.py:32-38)
        # FastCUT (registration_model.py:63-67): nce_idt False, lambda_NCE 10, flip_equivariance True.  `flip_draw()` ->
        # bool stands in for `np.random.random() < 0.5` (registration_model.py:189) so that tests can force the flip.
        self.nce_idt, self.flip_equivariance, self.flipped = nce_idt, flip_equivariance, False
        self.flip_draw = lambda: bool(__import__("numpy").random.random() < 0.5)
        self.netG = Generator(1, 1, ngf, n_blocks)
        init_weights_xavier(self.netG)
        self.netF = PatchSampler(netF_nc, True)
        self.netR = VxmDense((size, size), PLUGIN_UNET_FEATURES, 7, True)
        self.lr, self.betas = lr, betas
        self.num_patches, self.nce_T, self.lambda_NCE, self.nce_layers = num_patches, nce_T, lambda_NCE, list(nce_layers)
        self.opt_G = torch.optim.Adam(self.netG.parameters(), lr=lr, betas=betas)
        self.opt_R = torch.optim.Adam(self.netR.parameters(), lr=lr, betas=betas)
        self.opt_F = None
        self.ids_hook = None
        self.ids_log = []
        self._nce_calls = 0
        self.losses = {}

    def forward(self, A, B):
        self.real_A, self.real_B = A, B
        real = torch.cat((A, B), dim=0)
        if self.flip_equivariance:                         # registration_model.py:188-191
            self.flipped = self.flip_draw()
            if self.flipped:
                real = torch.flip(real, [3])
        fake = self.netG(real)
        self.fake_B, self.idt_B = fake[:A.shape[0]], fake[A.shape[0]:]

    def nce(self, src, tgt):
        """registration_model.py:237-253."""
        fq = self.netG(tgt, self.nce_layers, encode_only=True)
        if self.flip_equivariance and self.flipped:        # registration_model.py:241-242
            fq = [torch.flip(f, [3]) for f in fq]
        fk = self.netG(src, self.nce_layers, encode_only=True)
        ids = self.ids_hook(self._nce_calls, fk) if self.ids_hook is not None else None
        self._nce_calls += 1
        fk_pool, ids = self.netF(fk, self.num_patches, ids)
        self.ids_log.append([i.clone() for i in ids])
        fq_pool, _ = self.netF(fq, self.num_patches, ids)
        tot = 0.0
        for q, k in zip(fq_pool, fk_pool):
            tot = tot + (patchnce_loss(q, k, self.bs, self.nce_T, self.nce_all_negatives) * self.lambda_NCE).mean()
        return tot / len(self.nce_layers)

    def g_loss(self):
        """registration_model.py:213-235 with nce_idt."""
        self.loss_NCE = self.nce(self.real_A, self.fake_B)
        if not self.nce_idt:                               # registration_model.py:228-232
            self.loss_NCE_Y = 0.0
            return self.loss_NCE
        self.loss_NCE_Y = self.nce(self.real_B, self.idt_B)
        return (self.loss_NCE + self.loss_NCE_Y) * 0.5

    def data_dependent_initialize(self, A, B):
        """registration_model.py:119-136."""
        self.forward(A, B)
        self.g_loss().backward()
        self.opt_F = torch.optim.Adam(self.netF.parameters(), lr=self.lr, betas=self.betas)

    def step(self, A, B):
        """registration_model.py:138-171."""
        self.forward(A, B)
        ys, yt, flow = self.netR(A, B)
        registered = spatial_transform(self.fake_B, flow)
        self.registered, self.regA, self.flow = registered, ys, flow
        for o in (self.opt_G, self.opt_R, self.opt_F):
            o.zero_grad()
        loss_G = self.g_loss()
        mask = (B > -0.95) + (registered > -0.95)
        mask2 = (self.idt_B > -0.95) + (registered > -0.95)
        loss_local = self.nce(B, ys) * 0.25
        loss_R = masked_l1(registered, B, mask) + masked_l1(self.idt_B, registered, mask2) + loss_local
        loss_smooth = smoothing_loss(flow) * 0.20
        (loss_R + loss_G + loss_smooth).backward()
        for o in (self.opt_G, self.opt_R, self.opt_F):
            o.step()
        self.losses = dict(G=float(loss_G), NCE=float(self.loss_NCE), R=float(loss_R), smooth=float(loss_smooth),
                           local=float(loss_local), NCE_Y=float(self.loss_NCE_Y))
        return self.losses


class Registration3DStep(object):
    """The build-defined 3-D step (the reference has no 3-D entry point):
    VxmDense(ndims=3, bidir=True) + NCC_Loss[9,9,9] + lambda * Grad_Loss(l2), Adam(2e-4, (0.5, .999))."""

    def __init__(self, shape, features=None, lam=1.0, lr=2e-4, betas=(0.5, 0.999), win=9):
        self.netR = VxmDense(shape, features, 7, True)
        self.opt = torch.optim.Adam(self.netR.parameters(), lr=lr, betas=betas)
        self.lam, self.win = lam, win

    def step(self, A, B):
        ys, yt, flow = self.netR(A, B)
        self.opt.zero_grad()
        l_sim = ncc_loss(ys, B, self.win)
        l_reg = grad_loss_l2(flow)
        (l_sim + self.lam * l_reg).backward()
        self.opt.step()
        self.ys, self.flow = ys, flow
        return dict(ncc=float(l_sim), grad=float(l_reg))

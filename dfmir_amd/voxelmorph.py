"""Host-side mirror of the reference's VoxelMorph pieces on the hot path
(models/voxelmorph/torchvoxelmorph/layers.py:6-97, networks.py:9-106, 1028-1165, 1506-1521,
modelio.py:7-35): SpatialTransformer, VecInt, ResizeTransform, ConvBlock, Unet, VxmDense.

Same constructor signatures, forward semantics and state_dict keys; compute = libdfmir_hip.so.
2-D and 3-D share every kernel (a 2-D tensor is the D == 1 case).
"""
import torch
import torch.nn as nn
from torch.distributions.normal import Normal

from . import ops
from .networks import Conv2d, Conv3d


class SpatialTransformer(nn.Module):
    """N-D spatial transformer: out = grid_sample(src, grid + flow, align_corners=True, zeros).
    `flow` is a displacement in voxels, channel d along axis d (layers.py:30-48).  The `grid`
    buffer is kept only for state_dict compatibility -- the kernel adds the identity itself."""

    def __init__(self, size, mode='bilinear'):
        super().__init__()
        self.mode = mode
        vectors = [torch.arange(0, s) for s in size]
        grids = torch.meshgrid(vectors, indexing='ij')
        grid = torch.stack(grids).unsqueeze(0).type(torch.FloatTensor)
        self.register_buffer('grid', grid)

    def forward(self, src, flow):
        if tuple(flow.shape[2:]) != tuple(self.grid.shape[2:]):
            raise RuntimeError("SpatialTransformer built for %s got flow %s" %
                               (tuple(self.grid.shape[2:]), tuple(flow.shape[2:])))
        return ops.warp(src, flow, self.mode)


class VecInt(nn.Module):
    """Scaling and squaring (layers.py:51-68): vec/2^n, then n fused `v + warp(v, v)` steps."""

    def __init__(self, inshape, nsteps):
        super().__init__()
        assert nsteps >= 0, 'nsteps should be >= 0, found: %d' % nsteps
        self.nsteps = nsteps
        self.scale = 1.0 / (2 ** self.nsteps)
        self.transformer = SpatialTransformer(inshape)

    def forward(self, vec, scale_folded=False):
        if not scale_folded:
            vec = ops.scale(vec, self.scale)
        for _ in range(self.nsteps):
            vec = ops.vecint_step(vec)
        return vec


class ResizeTransform(nn.Module):
    """Resize + rescale a vector field (layers.py:71-97); the scalar factor is fused."""

    def __init__(self, vel_resize, ndims):
        super().__init__()
        self.factor = 1.0 / vel_resize
        self.mode = {2: 'bilinear', 3: 'trilinear'}.get(ndims, 'linear')

    def forward(self, x, extra_mult=1.0):
        if self.factor == 1 and extra_mult == 1.0:
            return x
        out_sp = [int(s * self.factor) for s in x.shape[2:]]  # floor(size*scale), as F.interpolate
        return ops.resize_linear(x, out_sp, self.factor * extra_mult)


def default_unet_features():
    """Encoder / decoder widths of the stock VoxelMorph U-Net (networks.py:9-14)."""
    return [[16, 32, 32, 32], [32, 32, 32, 32, 32, 16, 16]]


class ConvBlock(nn.Module):
    """Conv{2,3}d(3, stride, pad 1) + LeakyReLU(0.2) in one launch (networks.py:1506-1521)."""

    def __init__(self, ndims, in_channels, out_channels, stride=1):
        super().__init__()
        self.main = (Conv2d, Conv3d)[ndims - 2](in_channels, out_channels, 3, stride, 1)
        self.activation = nn.LeakyReLU(0.2)  # parameter-free marker; fused into the conv epilogue

    def forward(self, x, sole=False):
        # sole=True: the result feeds exactly one consumer (the next conv): see ops.conv
        if isinstance(x, tuple):        # (a, b): this block consumes cat([nearest_up2(a), b], 1) -- see Unet.forward
            return self.main.forward_upcat(x[0], x[1], act=1, slope=0.2, sole=sole)
        return self.main(x, act=1, slope=0.2, sole=sole)


def unet_channel_plan(enc_nf, dec_nf, in_channels=2):
    """(cin, cout) of every conv of the U-Net, grouped as the state_dict groups them (networks.py:60-86):
    `downarm` = one stride-2 conv per encoder width; `uparm` = the first len(enc) decoder widths, each but the first
    fed with the previous decoder output concatenated with the mirrored encoder level; `extras` = the remaining
    decoder widths at full resolution, the first of which also sees the `in_channels` input planes."""
    down, c = [], in_channels
    for nf in enc_nf:
        down.append((c, nf))
        c = nf
    skips = enc_nf[::-1]                   # deepest level first; level i of the decoder meets skips[i]
    up = []
    for i, nf in enumerate(dec_nf[:len(enc_nf)]):
        up.append((c + (skips[i] if i else 0), nf))
        c = nf
    extras, c = [], c + in_channels
    for nf in dec_nf[len(enc_nf):]:
        extras.append((c, nf))
        c = nf
    return down, up, extras


class Unet(nn.Module):
    """networks.py:16-106 for explicit feature lists (the only form the path passes)."""

    def __init__(self, inshape, nb_features=None, nb_levels=None, feat_mult=1):
        super().__init__()
        nd = len(inshape)
        if nd not in (2, 3):
            raise ValueError('ndims should be 2 or 3 on this path. found: %d' % nd)
        if isinstance(nb_features, int):
            raise NotImplementedError("integer nb_features is not on the path (lists are always passed)")
        if nb_levels is not None:
            raise ValueError('cannot use nb_levels if nb_features is not an integer')
        self.enc_nf, self.dec_nf = default_unet_features() if nb_features is None else nb_features
        down, up, extras = unet_channel_plan(list(self.enc_nf), list(self.dec_nf))
        self.upsample = nn.Upsample(scale_factor=2, mode='nearest')  # marker; fused with the concat
        self.downarm = nn.ModuleList(ConvBlock(nd, ci, co, stride=2) for ci, co in down)
        self.uparm = nn.ModuleList(ConvBlock(nd, ci, co) for ci, co in up)
        self.extras = nn.ModuleList(ConvBlock(nd, ci, co) for ci, co in extras)

    def forward(self, x):
        x_enc = [x]
        for layer in self.downarm:
            x_enc.append(layer(x_enc[-1]))
        x = x_enc.pop()
        for layer in self.uparm:
            x = layer(x, sole=True)     # feeds only the next block (through the pair below)
            # nn.Upsample(nearest, x2) + torch.cat with the mirrored encoder level (networks.py:97-100) are consumed by the
            # NEXT ConvBlock: handed over as the pair (a, b), so that the 3-D layers never build the concatenation
            x = (x, x_enc.pop())
        for i, layer in enumerate(self.extras):
            x = layer(x, sole=True)     # a chain: each output feeds the next ConvBlock (the last one the flow head) only
        if isinstance(x, tuple):        # no extras (not a configuration of the path): the concatenation itself
            x = ops.upcat(x[0], x[1])
        return x


class VxmDense(nn.Module):
    """networks.py:1028-1145 (this fork returns the INTEGRATED full-resolution pos_flow when
    bidir=True, networks.py:1143).  `config` holds the constructor arguments, as the reference's
    LoadableModel decorator records them (modelio.py:7-35)."""

    def __init__(self, inshape, nb_unet_features=None, nb_unet_levels=None, unet_feat_mult=1, int_steps=7,
                 int_downsize=2, bidir=False, use_probs=False):
        super().__init__()
        self.config = dict(inshape=inshape, nb_unet_features=nb_unet_features, nb_unet_levels=nb_unet_levels,
                           unet_feat_mult=unet_feat_mult, int_steps=int_steps, int_downsize=int_downsize,
                           bidir=bidir, use_probs=use_probs)
        if use_probs:
            raise NotImplementedError('Flow variance has not been implemented in pytorch - set use_probs to False')
        nd = len(inshape)
        self.training = True
        self.bidir = bidir
        self.unet_model = Unet(inshape, nb_unet_features, nb_unet_levels, unet_feat_mult)
        # flow head: N(0, 1e-5) weights, zero bias (networks.py:1077-1081)
        self.flow = (Conv2d, Conv3d)[nd - 2](self.unet_model.dec_nf[-1], nd, 3, padding=1)
        with torch.no_grad():
            self.flow.weight.copy_(Normal(0, 1e-5).sample(self.flow.weight.shape))
            self.flow.bias.zero_()
        # integration happens at 1/int_downsize resolution
        half = int_steps > 0 and int_downsize > 1
        self.resize = ResizeTransform(int_downsize, nd) if half else None
        self.fullsize = ResizeTransform(1 / int_downsize, nd) if half else None
        self.integrate = VecInt([int(d / int_downsize) for d in inshape], int_steps) if int_steps > 0 else None
        self.transformer = SpatialTransformer(inshape)
        # build-defined: the bidir branch's warp(target, neg_flow) (networks.py:1125-1139) is computed by the reference on
        # every step and read by nobody in REGISTRATIONModel (registration_model.py:143-147 use [0] and [2]).
        # A model that never reads it sets this and gets None in its place: 7 VecInt warps, a resize and a warp less.
        self.skip_unused_target = False

    def forward(self, source, target, registration=False):
        x = ops.upcat_channels(source, target)
        x = self.unet_model(x)
        flow_field = self.flow(x)
        pos_flow = flow_field
        if self.resize:
            pos_flow = self.resize(pos_flow)
        preint_flow = pos_flow
        neg_flow = None
        if self.integrate:
            sc = self.integrate.scale
            # negate + 2^-n scale folded into one launch per direction
            want_neg = self.bidir and not (self.skip_unused_target and not registration)
            if want_neg:
                neg_flow = ops.scale(pos_flow, -sc)
                neg_flow = self.integrate(neg_flow, scale_folded=True)
            pos_flow = ops.scale(pos_flow, sc)
            pos_flow = self.integrate(pos_flow, scale_folded=True)
            if self.fullsize:
                pos_flow = self.fullsize(pos_flow)
                neg_flow = self.fullsize(neg_flow) if neg_flow is not None else None
        elif self.bidir and not (self.skip_unused_target and not registration):
            neg_flow = ops.scale(pos_flow, -1.0)
        y_source = self.transformer(source, pos_flow)
        y_target = self.transformer(target, neg_flow) if (self.bidir and neg_flow is not None) else None
        if not registration:
            return (y_source, y_target, pos_flow) if self.bidir else (y_source, preint_flow)
        return y_source, pos_flow


class _Namespace(object):
    pass


# `vxm.networks.VxmDense`, `vxm.layers.SpatialTransformer` as in the reference's imports
# (models/registration_model.py:7-8,98).
networks = _Namespace()
networks.VxmDense = VxmDense
networks.Unet = Unet
networks.ConvBlock = ConvBlock
networks.default_unet_features = default_unet_features
layers = _Namespace()
layers.SpatialTransformer = SpatialTransformer
layers.VecInt = VecInt
layers.ResizeTransform = ResizeTransform


def _losses_namespace():
    from . import losses as _l               # (losses imports ops only: no cycle)
    ns = _Namespace()
    ns.NCC, ns.Grad = _l.NCC, _l.Grad        # torchvoxelmorph/losses.py:7-67,93-117
    return ns


losses = _losses_namespace()

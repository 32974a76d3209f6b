"""Host-side mirror of the reference's `models/networks.py` for the `--model registration` path:
`define_G` / `ResnetGenerator`, `define_F` / `PatchSampleF`, `get_scheduler`, `init_net`.

Same names, signatures, state_dict keys and module indices as the reference
(models/networks.py:218-289, 575-624, 956-1051, 1164-1221), but every forward/backward runs on
the hand-written gfx950 kernels of libdfmir_hip.so (dfmir_amd.ops).  Modules that the reference
chains as separate torch ops (ReflectionPad2d -> Conv2d, InstanceNorm2d -> ReLU, Conv2d -> Tanh,
x + conv_block(x)) are executed as fused launches by the sequential walker in
`ResnetGenerator.forward`.
"""
import functools
import os
import math

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init
from torch.optim import lr_scheduler

from . import ops


# ------------------------------------------------------------------------------------------------
# leaf modules
# ------------------------------------------------------------------------------------------------
class _ConvNd(nn.Module):
    """N-D convolution parameters in the reference layout [Cout, Cin, *k] plus a cache of the
    tap-major packings the kernels consume (refreshed when the weights change)."""

    def __init__(self, nd, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.nd = nd
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty((out_channels, in_channels) + (kernel_size,) * nd))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self._packs = {}
        self.reset_parameters()

    def reset_parameters(self):
        # torch.nn.modules.conv._ConvNd default initialisation
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size ** self.nd
            bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
            init.uniform_(self.bias, -bound, bound)

    def packed(self, mode):
        return ops.packed_weight(self, self.weight.detach(), mode)

    def forward(self, x, act=0, slope=0.0, reflect=False, sole=False):
        return ops.conv(x, self.weight, self.bias, self, self.stride, self.padding,
                        1 if reflect else 0, act, slope, sole=sole)

    def forward_upcat(self, a, b, act=0, slope=0.0, sole=False):
        """This conv applied to cat([nearest_up2(a), b], 1) (torchvoxelmorph/networks.py:97-100 + the next ConvBlock):
        3-D 3x3x3 layers take the form that never builds the concatenation (ops.upcat_conv3d), the rest materialise it."""
        if (self.nd == 3 and self.kernel_size == 3 and self.stride == 1 and self.padding == 1
                and ops.upcat_conv3d_ok(a, b, self.weight)):
            return ops.upcat_conv3d(a, b, self.weight, self.bias, self, act, slope, sole=sole)
        return self.forward(ops.upcat(a, b), act, slope, sole=sole)

    def extra_repr(self):
        return "%d, %d, kernel_size=%d, stride=%d, padding=%d" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding)


class Conv2d(_ConvNd):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, bias)


class Conv3d(_ConvNd):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, bias)


class Linear(nn.Module):
    """nn.Linear parameters ([out, in]); applied to channel-major rows [in, rows] as a 1x1 conv."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        self._packs = {}
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_features)
        init.uniform_(self.bias, -bound, bound)

    def packed(self, mode):
        return ops.packed_weight(self, self.weight.detach().view(self.out_features, self.in_features, 1), mode)

    def forward(self, x_cr, relu=False):
        """x_cr: [in_features, rows] -> [out_features, rows]."""
        C, rows = x_cr.shape
        y = ops.conv(x_cr.view(1, C, 1, rows), self.weight.view(self.out_features, C, 1, 1), self.bias,
                     self, 1, 0, 0, 1 if relu else 0, 0.0)
        return y.view(self.out_features, rows)


class ReflectionPad2d(nn.Module):
    def __init__(self, padding):
        super().__init__()
        self.padding = padding

    def forward(self, x):
        return ops.reflect_pad2d(x, self.padding)


class InstanceNorm2d(nn.Module):
    """InstanceNorm2d(affine=False, track_running_stats=False) (models/networks.py:125)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps

    def forward(self, x, relu=False, res=None):
        return ops.instance_norm(x, res, relu, self.eps)


class ReLU(nn.Module):
    """Marker: always executed fused into the preceding InstanceNorm2d / Linear."""

    def __init__(self, inplace=True):
        super().__init__()

    def forward(self, x):
        raise NotImplementedError("ReLU is fused into the producing kernel on this path")


class Tanh(nn.Module):
    """Marker: executed as the epilogue of the 7x7 output conv."""

    def forward(self, x):
        raise NotImplementedError("Tanh is fused into the conv epilogue on this path")


def get_filter(filt_size=3):
    """Binomial blur filter (models/networks.py:15-34)."""
    rows = {1: [1.], 2: [1., 1.], 3: [1., 2., 1.], 4: [1., 3., 3., 1.], 5: [1., 4., 6., 4., 1.],
            6: [1., 5., 10., 10., 5., 1.], 7: [1., 6., 15., 20., 15., 6., 1.]}
    a = np.array(rows[filt_size])
    filt = torch.Tensor(a[:, None] * a[None, :])
    return filt / torch.sum(filt)


class Downsample(nn.Module):
    """Anti-aliased stride-2 blur pool (models/networks.py:37-60); `filt` buffer kept for
    checkpoint compatibility, the kernel hard-wires [1 2 1]^2/16 + reflect pad."""

    def __init__(self, channels, pad_type='reflect', filt_size=3, stride=2, pad_off=0):
        super().__init__()
        if filt_size != 3 or stride != 2 or pad_off != 0 or pad_type not in ('refl', 'reflect'):
            raise NotImplementedError("only the filt_size=3, stride=2, reflect Downsample is on the path")
        self.channels = channels
        self.register_buffer('filt', get_filter(3)[None, None].repeat(channels, 1, 1, 1))

    def forward(self, x):
        return ops.blur_down(x)


class Upsample(nn.Module):
    """Anti-aliased x2 up-sampling (models/networks.py:73-93)."""

    def __init__(self, channels, pad_type='repl', filt_size=4, stride=2):
        super().__init__()
        if filt_size != 4 or stride != 2 or pad_type not in ('repl', 'replicate'):
            raise NotImplementedError("only the filt_size=4, stride=2, replicate Upsample is on the path")
        self.channels = channels
        self.register_buffer('filt', (get_filter(4) * 4)[None, None].repeat(channels, 1, 1, 1))

    def forward(self, x):
        return ops.blur_up(x)


def get_norm_layer(norm_type='instance'):
    if norm_type == 'instance':
        return functools.partial(InstanceNorm2d)
    raise NotImplementedError('normalization layer [%s] is not on the MI355X hot path (instance only)' % norm_type)


def get_scheduler(optimizer, opt):
    """LR schedule (models/networks.py:134-160)."""
    if opt.lr_policy == 'linear':
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + opt.epoch_count - opt.n_epochs) / float(opt.n_epochs_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if opt.lr_policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == 'cosine':
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.n_epochs, eta_min=0)
    raise NotImplementedError('learning rate policy [%s] is not implemented' % opt.lr_policy)


def init_weights(net, init_type='normal', init_gain=0.02, debug=False):
    """models/networks.py:163-195 restricted to the module types on the path."""
    def init_func(m):
        if isinstance(m, (_ConvNd, Linear)):
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            if m.bias is not None:
                init.constant_(m.bias.data, 0.0)
    net.apply(init_func)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[], debug=False, initialize_weights=True):
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(gpu_ids[0])
    if initialize_weights:
        init_weights(net, init_type, init_gain=init_gain, debug=debug)
    return net


# ------------------------------------------------------------------------------------------------
# generator
# ------------------------------------------------------------------------------------------------
_NO_SHARE_TAP = ops._env_on("DFMIR_NO_SHARE_TAP")   # A/B switch: dense gradients of the sampled features
_NO_SKIP_FOLD = ops._env_on("DFMIR_NO_SKIP_FOLD")   # A/B switch: skip gradient summed by autograd's add


class ResnetBlock(nn.Module):
    """x + [pad,conv,IN,ReLU,pad,conv,IN](x) (models/networks.py:1164-1221) as 4 fused launches."""

    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias):
        super().__init__()
        if padding_type != 'reflect' or use_dropout:
            raise NotImplementedError("ResnetBlock: reflect padding, no dropout only")
        self.conv_block = nn.Sequential(
            ReflectionPad2d(1), Conv2d(dim, dim, 3, padding=0, bias=use_bias), norm_layer(dim), ReLU(True),
            ReflectionPad2d(1), Conv2d(dim, dim, 3, padding=0, bias=use_bias), norm_layer(dim))

    def forward(self, x):
        cb = self.conv_block
        if torch.is_grad_enabled() and x.requires_grad and x.is_contiguous() and not _NO_SKIP_FOLD:
            # the skip branch leaves through the first conv's autograd node, whose backward sums its gradient
            # inside the halo-fold kernel (one pass instead of fold + add)
            h, xs = ops.conv(x, cb[1].weight, cb[1].bias, cb[1], cb[1].stride, cb[0].padding, 1, 0, 0.0, skip=True)
        else:
            h, xs = _reflect_conv(cb[0], cb[1], x), x
        h = cb[2](h, relu=True)
        h = _reflect_conv(cb[4], cb[5], h)
        return cb[6](h, relu=False, res=xs)


def _reflect_conv(pad, conv, x, act=0):
    """conv(reflection_pad(x)) with the halo resolved inside the conv's gather.  The 7x7 ends
    (Cin == 1 / Cout == 1) run as tap-stack / tap-sum + 1x1 GEMM so that they reach the matrix cores."""
    if conv.kernel_size >= 5 and conv.stride == 1 and (conv.in_channels == 1 or conv.out_channels <= 4):
        return ops.conv_taps(x, conv.weight, conv.bias, pad.padding, 1, act, 0.0)
    return ops.conv(x, conv.weight, conv.bias, conv, conv.stride, pad.padding, 1, act, 0.0)


class ResnetGenerator(nn.Module):
    """models/networks.py:956-1051.  `self.model` has the reference's module indices so that
    `layers=[0,4,8,12,16]` selects the same features and checkpoints interchange."""

    def __init__(self, input_nc, output_nc, ngf=64, norm_layer=InstanceNorm2d, use_dropout=False, n_blocks=6,
                 padding_type='reflect', no_antialias=False, no_antialias_up=False, opt=None):
        assert n_blocks >= 0
        super().__init__()
        self.opt = opt
        use_bias = True  # InstanceNorm => biased convs (models/networks.py:977-980)
        if no_antialias_up:
            raise NotImplementedError("no_antialias_up (ConvTranspose2d) is not on the path")
        model = [ReflectionPad2d(3), Conv2d(input_nc, ngf, 7, padding=0, bias=use_bias), norm_layer(ngf), ReLU(True)]
        n_down = 2
        for i in range(n_down):
            mult = 2 ** i
            if no_antialias:
                model += [Conv2d(ngf * mult, ngf * mult * 2, 3, stride=2, padding=1, bias=use_bias),
                          norm_layer(ngf * mult * 2), ReLU(True)]
            else:
                model += [Conv2d(ngf * mult, ngf * mult * 2, 3, stride=1, padding=1, bias=use_bias),
                          norm_layer(ngf * mult * 2), ReLU(True), Downsample(ngf * mult * 2)]
        mult = 2 ** n_down
        for i in range(n_blocks):
            model += [ResnetBlock(ngf * mult, padding_type=padding_type, norm_layer=norm_layer,
                                  use_dropout=use_dropout, use_bias=use_bias)]
        for i in range(n_down):
            mult = 2 ** (n_down - i)
            model += [Upsample(ngf * mult), Conv2d(ngf * mult, int(ngf * mult / 2), 3, stride=1, padding=1, bias=use_bias),
                      norm_layer(int(ngf * mult / 2)), ReLU(True)]
        model += [ReflectionPad2d(3), Conv2d(ngf, output_nc, 7, padding=0), Tanh()]
        self.model = nn.Sequential(*model)

    def _run(self, x, layers, encode_only):
        """Walk self.model fusing (pad,conv[,tanh]) and (IN,ReLU); an index listed in `layers` is
        always materialised so that the returned features equal the reference's."""
        mods = list(self.model)
        n = len(mods)
        feats = []
        last = layers[-1] if layers else None
        want = set(layers)
        feat = x
        i = 0

        def emit(idx, t, shared=True):
            nonlocal feat
            stop = encode_only and idx == last
            if idx in want:
                if shared and not stop and not _NO_SHARE_TAP and t is feat:
                    feat, t = ops.fork_tap(t)     # t also feeds the next layer: see ops.TapForkFn
                feats.append(t)
            return stop

        while i < n:
            m = mods[i]
            gb = getattr(self, 'grad_boundary', None)
            if gb is not None and i == gb[0] and feat.requires_grad and torch.is_grad_enabled():
                # build-defined hook (data-parallel gradient buckets): fires when backward has produced the gradient of
                # module i's INPUT, i.e. when every weight gradient of modules i.. has been enqueued
                feat.register_hook(lambda g_, cb=gb[1]: (cb(), None)[1])
            if isinstance(m, ReflectionPad2d) and i + 1 < n and isinstance(mods[i + 1], Conv2d):
                if i in want:
                    if emit(i, m(feat), shared=False):
                        return feat, feats, True
                conv = mods[i + 1]
                tanh = i + 2 < n and isinstance(mods[i + 2], Tanh) and (i + 1) not in want
                feat = _reflect_conv(m, conv, feat, act=2 if tanh else 0)
                if emit(i + 1, feat):
                    return feat, feats, True
                i += 2
                if tanh:
                    if emit(i, feat):
                        return feat, feats, True
                    i += 1
                continue
            if isinstance(m, Conv2d):
                tanh = i + 1 < n and isinstance(mods[i + 1], Tanh) and i not in want
                feat = m(feat, act=2 if tanh else 0)
                if emit(i, feat):
                    return feat, feats, True
                i += 1
                if tanh:
                    if emit(i, feat):
                        return feat, feats, True
                    i += 1
                continue
            if (isinstance(m, InstanceNorm2d) and i + 2 < n and isinstance(mods[i + 1], ReLU)
                    and isinstance(mods[i + 2], Downsample) and not ({i, i + 1, i + 2} & want)
                    and ops.in_relu_blurdown_ok(feat)):
                # InstanceNorm + ReLU + blur-pool in one pass: the full-resolution normalised tensor feeds nothing else
                feat = ops.instance_norm_relu_blur_down(feat, m.eps)
                i += 3
                continue
            if isinstance(m, InstanceNorm2d):
                relu = i + 1 < n and isinstance(mods[i + 1], ReLU)
                feat = m(feat, relu=relu)
                # the reference's ReLU is in-place, so the IN output it appended aliases the
                # post-ReLU tensor: emitting the fused result for both indices is exact.
                if emit(i, feat):
                    return feat, feats, True
                i += 1
                if relu:
                    if emit(i, feat):
                        return feat, feats, True
                    i += 1
                continue
            feat = m(feat)
            if emit(i, feat):
                return feat, feats, True
            i += 1
        return feat, feats, False

    def forward(self, input, layers=[], encode_only=False):
        if -1 in layers:
            layers.append(len(self.model))
        if len(layers) > 0:
            feat, feats, stopped = self._run(input, list(layers), encode_only)
            if stopped:
                return feats
            return feat, feats
        feat, _, _ = self._run(input, [], False)
        return feat


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal',
             init_gain=0.02, no_antialias=False, no_antialias_up=False, gpu_ids=[], opt=None):
    """models/networks.py:218-268 (resnet generators with instance norm are the hot path)."""
    norm_layer = get_norm_layer(norm_type=norm)
    blocks = {'resnet_9blocks': 9, 'resnet_6blocks': 6, 'resnet_4blocks': 4}
    if netG not in blocks:
        raise NotImplementedError('Generator model name [%s] is not recognized' % netG)
    net = ResnetGenerator(input_nc, output_nc, ngf, norm_layer=norm_layer, use_dropout=use_dropout,
                          no_antialias=no_antialias, no_antialias_up=no_antialias_up, n_blocks=blocks[netG], opt=opt)
    return init_net(net, init_type, init_gain, gpu_ids, initialize_weights=True)


# ------------------------------------------------------------------------------------------------
# PatchNCE feature head
# ------------------------------------------------------------------------------------------------
class Normalize(nn.Module):
    """x / (||x||_2 + 1e-7) along dim 1 of [rows, C] (models/networks.py:493-502)."""

    def __init__(self, power=2):
        super().__init__()
        if power != 2:
            raise NotImplementedError("only the L2 Normalize is on the path")
        self.power = power

    def forward(self, x_rows_c):
        return ops.l2norm_rows(_as_channel_major(x_rows_c)).t()


def _as_channel_major(x_rows_c):
    """[rows, C] (normally the transposed view of a [C, rows] kernel buffer) -> contiguous [C, rows]."""
    t = x_rows_c.t()
    return t if t.is_contiguous() else t.contiguous()


class PatchSampleF(nn.Module):
    """models/networks.py:575-624.  Returns, per layer, the [B*P, nc] L2-normalised patch features
    (as the transposed view of the kernels' channel-major buffer) and the patch ids."""

    def __init__(self, use_mlp=False, init_type='normal', init_gain=0.02, nc=256, gpu_ids=[]):
        super().__init__()
        self.l2norm = Normalize(2)
        self.use_mlp = use_mlp
        self.nc = nc
        self.mlp_init = False
        self.init_type = init_type
        self.init_gain = init_gain
        self.gpu_ids = gpu_ids

    def create_mlp(self, feats):
        for mlp_id, feat in enumerate(feats):
            input_nc = feat.shape[1]
            mlp = nn.Sequential(Linear(input_nc, self.nc), ReLU(), Linear(self.nc, self.nc))
            mlp.to(feat.device)
            setattr(self, 'mlp_%d' % mlp_id, mlp)
        init_net(self, self.init_type, self.init_gain, self.gpu_ids)
        self.mlp_init = True

    def forward(self, feats, num_patches=64, patch_ids=None):
        """Reference signature (models/networks.py:602).  Build-defined extension carried by the ids themselves: a
        2-D patch_ids[i] of shape [G, P] means `feats` stack the query images of G NCE terms along the batch and each
        G-th of the batch is sampled at its own row of positions; the MLP then runs once over all rows."""
        return_ids, return_feats = [], []
        if self.use_mlp and not self.mlp_init:
            self.create_mlp(feats)
        for feat_id, feat in enumerate(feats):
            if num_patches <= 0:
                raise NotImplementedError("num_patches=0 (dense features) is not on the path")
            S = feat.shape[2] * feat.shape[3]
            if patch_ids is not None:
                patch_id = patch_ids[feat_id]
            else:
                patch_id = torch.randperm(S, device=feats[0].device)
                patch_id = ops.mark_distinct(patch_id[:int(min(num_patches, patch_id.shape[0]))])
            groups = patch_id.shape[0] if patch_id.dim() == 2 else 1
            x = ops.patch_gather(feat, patch_id, groups)      # [C, B*P]
            return_ids.append(patch_id)
            return_feats.append(self.project(feat_id, x).t())  # [B*P, nc] view
        return return_feats, return_ids

    def sample_project(self, feat_id, feat, ids, groups=1):
        """Layer `feat_id`: sample `ids` ([G, P]: image b of the batch uses row b // (B / G)) from feat [B, C, H, W] and
        project -> L2-normalised [nc, B*P] (channel-major; models/networks.py:604-619).  One fused launch where the
        library has one (MLP head, nc = 256, C <= 256), else gather + project."""
        if ops.nce_head_ok(feat.shape[1], self.nc, self.use_mlp):
            mlp = getattr(self, 'mlp_%d' % feat_id)
            return ops.nce_head(feat, ids, mlp[0], mlp[2])
        return self.project(feat_id, ops.patch_gather(feat, ids, groups))

    def sample_project_multi(self, feat_id, srcs, ids):
        """The same for G source tensors (no gradient: the detached key side), group g sampled at ids[g]."""
        if ops.nce_head_ok(srcs[0].shape[1], self.nc, self.use_mlp):
            mlp = getattr(self, 'mlp_%d' % feat_id)
            return ops.nce_head_multi(srcs, ids, mlp[0], mlp[2])
        return self.project(feat_id, ops.patch_gather_multi(srcs, ids))

    def project(self, feat_id, x_cm):
        """Sampled rows of layer `feat_id`, channel-major [C, rows] -> MLP (Linear, ReLU, Linear) -> L2-normalised
        [nc, rows] (models/networks.py:613-619); the transposed view of the result is what forward() returns."""
        if self.use_mlp:
            mlp = getattr(self, 'mlp_%d' % feat_id)
            x_cm = mlp[2](mlp[0](x_cm, relu=True))
        return ops.l2norm_rows(x_cm)


def define_F(input_nc, netF, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02,
             no_antialias=False, gpu_ids=[], opt=None):
    """models/networks.py:276-289."""
    if netF == 'sample':
        net = PatchSampleF(use_mlp=False, init_type=init_type, init_gain=init_gain, gpu_ids=gpu_ids, nc=opt.netF_nc)
    elif netF == 'mlp_sample':
        net = PatchSampleF(use_mlp=True, init_type=init_type, init_gain=init_gain, gpu_ids=gpu_ids, nc=opt.netF_nc)
    else:
        raise NotImplementedError('projection model name [%s] is not recognized' % netF)
    return init_net(net, init_type, init_gain, gpu_ids)


class GANLoss(nn.Module):
    """Constructed but never evaluated on this path (lambda_GAN = 0, registration_model.py:41,217-221)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.gan_mode = gan_mode

    def forward(self, *a, **k):
        raise NotImplementedError("the registration model is discriminator-free (lambda_GAN = 0)")

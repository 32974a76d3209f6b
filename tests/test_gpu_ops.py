"""GPU parity, op by op: every C-ABI kernel (through dfmir_amd.ops) against the CPU oracle /
plain torch fp32 ops on the same seeded inputs, forward and backward.  Bar: 1e-4 relative to the
tensor's max magnitude (fp32; summation order differs), 1e-3 on long-reduction gradients.
Run on the MI355X box:  python -m pytest tests -m gpu -q
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.golden import common as C

pytestmark = pytest.mark.gpu

DEV = "cuda"


ELEMENTWISE = []      # (what, norm-wise rel, element-wise max / 99.9 % / median rel) of every close(..., elem_rtol=...) call
MARGINS = []          # (test id, what, measured rel. error, bound) of every close() call: tests/conftest.py writes them to
#                       $DFMIR_MARGINS_OUT at session end (profiles/rNN_parity_margins.txt is such a file)


def close(got, ref, rtol=1e-4, atol=1e-6, what="", elem_rtol=None, elem_floor=1e-6):
    """Norm-wise bound: max |got - ref| <= atol + rtol * max |ref|.  elem_rtol (the outputs north_star names: translated
    image, deformation field, warped image) adds an ELEMENT-wise one on rel_i = |got_i - ref_i| / max(|ref_i|, elem_floor *
    max |ref|): 99.9 % of the elements within elem_rtol and every element within 1 (an element 1e-6 below the tensor's
    maximum carries an absolute error of ~1e-6 of that maximum on BOTH sides -- fp32 sums over O(1) terms -- so its
    relative error is O(1) by construction; the quantile is the informative figure).  The measured max / 99.9 % / median are
    recorded in ELEMENTWISE and printed by the full-size tests (profiles/r05_elementwise_error.txt)."""
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    ref = ref.detach().float().cpu().numpy() if torch.is_tensor(ref) else np.asarray(ref)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), "%s: non-finite values" % what
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(got - ref).max())
    import os
    MARGINS.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], what, err / scale, rtol + atol / scale))
    assert err <= atol + rtol * scale, "%s: max abs err %.3e vs scale %.3e (rel %.2e)" % (what, err, scale, err / scale)
    if elem_rtol is not None:
        rel = np.abs(got.astype(np.float64) - ref) / np.maximum(np.abs(ref), elem_floor * scale)
        ELEMENTWISE.append((what, err / scale, float(rel.max()), float(np.quantile(rel, 0.999)), float(np.median(rel))))
        assert ELEMENTWISE[-1][3] <= elem_rtol and ELEMENTWISE[-1][2] <= 1.0, \
            "%s: element-wise rel err 99.9 %% %.3e / max %.3e (floor %.0e of the max) vs %.1e / 1" % (
                what, ELEMENTWISE[-1][3], ELEMENTWISE[-1][2], elem_floor, elem_rtol)


def near(got, ref, rtol, what="", floor=1e-6):
    """Scalar comparison |got - ref| <= rtol * max(|ref|, floor), recorded in MARGINS like close()."""
    import os
    err, scale = abs(float(got) - float(ref)), max(abs(float(ref)), floor)
    MARGINS.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], what, err / scale, rtol))
    assert err <= rtol * scale, "%s: %r vs %r (rel %.2e > %.1e)" % (what, float(got), float(ref), err / scale, rtol)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from dfmir_amd import ops as _ops
    return _ops


# ------------------------------------------------------------------------------------------ conv
CONV2D = [
    # Cin, Cout, K, stride, pad, reflect, act(0 none,1 leaky .2,2 tanh), N, H, W
    (16, 16, 3, 1, 1, True, 0, 2, 12, 20),
    (8, 40, 3, 1, 1, False, 0, 2, 9, 13),
    (24, 72, 3, 1, 1, False, 0, 1, 17, 15),
    (256, 256, 3, 1, 1, True, 0, 1, 16, 16),
    (130, 136, 3, 1, 1, False, 0, 1, 8, 8),
    (1, 64, 7, 1, 3, True, 0, 2, 20, 24),
    (64, 1, 7, 1, 3, True, 2, 2, 20, 24),
    (2, 16, 3, 2, 1, False, 1, 2, 16, 24),
    (16, 32, 3, 2, 1, False, 1, 2, 15, 17),
    (34, 16, 3, 1, 1, False, 1, 1, 16, 16),
    (16, 2, 3, 1, 1, False, 0, 2, 10, 14),
    (48, 32, 3, 1, 1, False, 1, 1, 11, 7),
    (33, 3, 1, 1, 0, False, 0, 1, 6, 50),
    (256, 256, 1, 1, 0, False, 1, 1, 1, 512),
    # 1x1 convolutions (weight gradients on the streaming NT-GEMM kernel conv1x1_wgrad_k): ragged channel tiles, a last chunk of
    # 32 of 64 voxels, three images (split-K across image boundaries), exactly two chunks, tanh epilogue
    (72, 49, 1, 1, 0, False, 0, 3, 20, 24),
    (8, 136, 1, 1, 0, False, 1, 1, 8, 16),
    (40, 72, 1, 1, 0, False, 2, 2, 12, 20),
    (40, 49, 1, 1, 0, False, 2, 3, 20, 24),
    (64, 64, 1, 1, 0, False, 1, 2, 16, 32),
    # geometries that take the LDS-resident 3x3 kernels (conv3x3.hip): aligned / unaligned pixel runs,
    # padded-frame dgrad of reflect convs, W >= tile, partial channel chunks, both wgrad tilings
    (64, 128, 3, 1, 1, False, 0, 2, 32, 32),
    (128, 128, 3, 1, 1, True, 0, 2, 64, 64),
    (32, 64, 3, 1, 1, False, 1, 1, 48, 40),
    (128, 64, 3, 1, 1, False, 0, 2, 64, 64),
    (160, 64, 3, 1, 1, False, 0, 1, 34, 48),      # swapped-role weight gradient (64 couts under > 64 cins), ragged cin tile
    (64, 128, 3, 1, 1, False, 0, 1, 16, 256),
    (36, 132, 3, 1, 1, True, 0, 1, 24, 72),
    (256, 256, 3, 1, 1, True, 0, 2, 64, 64),
    (64, 32, 3, 1, 1, False, 1, 1, 64, 256),
    # split-bf16 kernels (conv3x3s.hip) with ragged channel tiles / image rows shorter than a pixel run /
    # a last pixel tile that leaves the second wave group empty; wgrad runs of 2 rows x 16 px
    (72, 136, 3, 1, 1, True, 1, 2, 10, 48),
    (64, 160, 3, 1, 1, False, 0, 1, 6, 16),
    (200, 129, 3, 1, 1, True, 0, 1, 18, 32),
    # 64-cout form of the shared-tile kernel (16 x 32 tiles): reflect halo + activation + 3 channel chunks;
    # ragged everything (Cin 20, 40 of 64 couts, last tile row 14 of 16 rows, 60 of 64 columns)
    (48, 64, 3, 1, 1, True, 1, 2, 32, 64),
    (20, 40, 3, 1, 1, False, 0, 1, 30, 60),
    # width not a multiple of 4: the shared-tile kernel's epilogue leaves its float4 form (scalar stores, ragged last quad)
    (64, 128, 3, 1, 1, True, 0, 1, 16, 62),
    (32, 64, 3, 1, 1, False, 1, 2, 31, 61),
    # small-channel wgrad kernel (16x16x4 MFMA, all input channels' patch in LDS): ragged row blocks (34*9 = 306 rows),
    # 2 output channels, 48 input channels with reflect padding, ragged tiles
    (34, 16, 3, 1, 1, False, 0, 2, 64, 96),
    (16, 2, 3, 1, 1, False, 0, 4, 40, 70),
    (48, 12, 3, 1, 1, True, 1, 2, 64, 64),
]


def torch_conv(x, w, b, stride, pad, reflect, act, nd):
    conv = F.conv2d if nd == 2 else F.conv3d
    if reflect:
        x = F.pad(x, (pad,) * (2 * nd), mode='reflect')
        y = conv(x, w, b, stride=stride)
    else:
        y = conv(x, w, b, stride=stride, padding=pad)
    if act == 1:
        y = F.leaky_relu(y, 0.2)
    elif act == 2:
        y = torch.tanh(y)
    return y


@pytest.mark.parametrize("cfg", CONV2D, ids=[str(c) for c in CONV2D])
def test_conv2d(ops, cfg):
    Cin, Cout, K, stride, pad, reflect, act, N, H, W = cfg
    x = C.randn(1, N, Cin, H, W)
    w = C.randn(2, Cout, Cin, K, K) / (Cin * K * K) ** 0.5
    b = C.randn(3, Cout) * 0.1
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = torch_conv(xr, wr, br, stride, pad, reflect, act, 2)
    cot = C.randn(4, *yr.shape)
    (yr * cot).sum().backward()
    xg, wg, bg = (t.clone().to(DEV).requires_grad_() for t in (x, w, b))
    yg = ops.conv(xg, wg, bg, None, stride, pad, 1 if reflect else 0, act, 0.2)
    (yg * cot.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    close(yg, yr, what="y")
    close(xg.grad, xr.grad, what="dx")
    close(wg.grad, wr.grad, rtol=3e-4, what="dw")
    close(bg.grad, br.grad, rtol=3e-4, what="db")


@pytest.mark.parametrize("cfg", [(256, 256, True, 2, 64, 64), (64, 128, False, 2, 32, 32), (128, 256, True, 1, 16, 64)])
def test_conv3x3_one_wave_per_simd_form(ops, cfg):
    """DFMIR_CONV_W1=1: conv3x3_split_w1_k (one wave per SIMD, 16 x 32 x 128 tiles, weights by buffer_load...lds; an experiment
    kept behind the switch) computes what the shared-tile kernel computes -- forward and input gradient (the zero-padded dgrad
    with the skip gradient and the reflect ring in its epilogue) against torch, and against the default path to round-off."""
    from dfmir_amd import _lib
    Cin, Cout, reflect, N, H, W = cfg
    x = C.randn(201, N, Cin, H, W)
    w = C.randn(202, Cout, Cin, 3, 3) / (Cin * 9) ** 0.5
    b = C.randn(203, Cout) * 0.1
    c1 = C.randn(204, N, Cout, H, W)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = torch_conv(xr, wr, b, 1, 1, reflect, 0, 2)
    (yr * c1).sum().backward()

    def run():
        xg, wg = x.clone().to(DEV).requires_grad_(), w.clone().to(DEV).requires_grad_()
        yg = ops.conv(xg, wg, b.to(DEV), None, 1, 1, 1 if reflect else 0, 0, 0.0)
        (yg * c1.to(DEV)).sum().backward()
        return yg.detach(), xg.grad.detach()

    y0, dx0 = run()
    _lib.set_option("DFMIR_CONV_W1", "1")
    try:
        y1, dx1 = run()
    finally:
        _lib.set_option("DFMIR_CONV_W1", None)
    close(y1, yr, what="y (w1)"); close(dx1, xr.grad, rtol=2e-4, what="dx (w1)")
    close(y1, y0, rtol=2e-6, what="w1 vs default y"); close(dx1, dx0, rtol=2e-6, what="w1 vs default dx")


@pytest.mark.parametrize("xs,ws,gs", [(1e-12, 1.0, 1e-14), (3e9, 1e-6, 7e5), (1.0, 40.0, 1e-30), (0.0, 1.0, 1.0)],
                         ids=["tiny", "huge", "denormal_grads", "zero_input"])
def test_conv3x3_split_dynamic_range(ops, xs, ws, gs):
    """The fp16x2 split rescales every tensor by a power of two taken from its own maximum: inputs, weights and
    gradients of any magnitude (down to denormal gradients, all-zero inputs) must come out at fp32 accuracy --
    checked against an fp64 convolution, relative to each result's own scale."""
    Cin, Cout, N, H, W = 64, 128, 2, 16, 32
    x = (C.randn(71, N, Cin, H, W) * xs).to(DEV).requires_grad_()
    w = (C.randn(72, Cout, Cin, 3, 3) * ws / 24.0).to(DEV).requires_grad_()
    cot = (C.randn(73, N, Cout, H, W) * gs).to(DEV)
    y = ops.conv(x, w, None, None, 1, 1, 1, 0, 0.0)
    (y * cot).sum().backward()
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    yd = F.conv2d(F.pad(xd, (1, 1, 1, 1), mode="reflect"), wd)
    (yd * cot.double()).sum().backward()
    for got, ref, what in ((y, yd, "y"), (x.grad, xd.grad, "dx"), (w.grad, wd.grad, "dw")):
        assert torch.isfinite(got).all(), what
        scale = float(ref.abs().max())
        err = float((got.double() - ref.detach()).abs().max())
        assert err <= 2e-5 * scale + 0.0 * scale, "%s: max err %.3e vs scale %.3e" % (what, err, scale)


def test_conv3x3_split_channel_magnitude_spread(ops):
    """Output-gradient channels 1e6 apart.  With the per-plane maxima of dY (what InstanceNorm's backward leaves) the
    split weight-gradient kernel scales dY per output channel, so EVERY channel's weight gradient is accurate relative to
    its own magnitude (< 1e-5, fp32 accumulation level); with one scale for the tensor the small channels sit 2^-20
    below the maximum and degrade to ~1e-4 (the documented behaviour of the per-tensor form, kept as the A/B)."""
    Cin, Cout, N, H, W = 64, 128, 2, 16, 32
    x = C.randn(81, N, Cin, H, W).to(DEV)
    mag = torch.logspace(0, -6, Cout).view(1, Cout, 1, 1)
    dy = (C.randn(82, N, Cout, H, W) * mag).to(DEV)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), padding=1)

    def per_channel_error(pmax):
        dwt = ops.conv_wgrad_raw(x.unsqueeze(2), dy.unsqueeze(2), (1, 3, 3), 1, (0, 1, 1), 0,
                                 x_amax=ops.absmax(x), dy_amax=ops.absmax(dy), dy_pmax=pmax)
        dw = ops.weight_unpack(dwt, (Cout, Cin, 3, 3)).double()
        return (dw - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1)

    e_ch = per_channel_error(dy.abs().amax(dim=(2, 3)).contiguous())          # [N, Cout] plane maxima
    assert float(e_ch.max()) < 1e-5, e_ch.max()
    e_t = per_channel_error(None)
    assert float(e_t.max()) < 2e-4 and float(e_t[:64].max()) < 2e-6, (e_t.max(), e_t[:64].max())
    assert float(e_ch.max()) < 0.2 * float(e_t.max())


def test_instnorm_backward_plane_maxima_feed_the_wgrad(ops):
    """InstanceNorm backward leaves max |dx| per (n, c) plane; the conv in front of it picks them up for its weight
    gradient (tag on the gradient tensor), and the result matches fp64."""
    from dfmir_amd import networks as N_
    conv = N_.Conv2d(64, 128, 3, padding=1).to(DEV)
    inn = N_.InstanceNorm2d(128)
    x = C.randn(85, 2, 64, 64, 64).to(DEV)
    cot = (C.randn(86, 2, 128, 64, 64) * torch.logspace(0, -5, 128).view(1, 128, 1, 1)).to(DEV)
    y = inn(conv(x), relu=True)
    (y * cot).sum().backward()
    xr = x.double().cpu()
    wr = conv.weight.detach().double().cpu().requires_grad_()
    br = conv.bias.detach().double().cpu().requires_grad_()
    yr = torch.relu(torch.nn.functional.instance_norm(torch.nn.functional.conv2d(xr, wr, br, padding=1), eps=1e-5))
    (yr * cot.double().cpu()).sum().backward()
    e = (conv.weight.grad.double().cpu() - wr.grad).flatten(1).norm(dim=1) / wr.grad.flatten(1).norm(dim=1)
    assert float(e.max()) < 2e-5, e.max()


@pytest.mark.parametrize("cfg", [(1, 64, 7, 3, True, 0, 2, 20, 24), (64, 1, 7, 3, True, 2, 2, 20, 24),
                                 (1, 8, 5, 2, False, 1, 1, 9, 11), (6, 2, 5, 2, True, 0, 1, 12, 10)],
                         ids=["1to64_k7_refl", "64to1_k7_refl_tanh", "1to8_k5_zero_leaky", "6to2_k5_refl"])
def test_conv_taps(ops, cfg):
    """7x7 / 5x5 convs with Cin == 1 or Cout <= 4 as tap-stack / tap-sum + 1x1 GEMM (csrc/taps.hip)."""
    Cin, Cout, K, pad, reflect, act, N, H, W = cfg
    x = C.randn(1, N, Cin, H, W)
    w = C.randn(2, Cout, Cin, K, K) / (Cin * K * K) ** 0.5
    b = C.randn(3, Cout) * 0.1
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = torch_conv(xr, wr, br, 1, pad, reflect, act, 2)
    cot = C.randn(4, *yr.shape)
    (yr * cot).sum().backward()
    xg, wg, bg = (t.clone().to(DEV).requires_grad_() for t in (x, w, b))
    yg = ops.conv_taps(xg, wg, bg, pad, 1 if reflect else 0, act, 0.2)
    (yg * cot.to(DEV)).sum().backward()
    close(yg, yr, what="y")
    close(xg.grad, xr.grad, what="dx")
    close(wg.grad, wr.grad, rtol=3e-4, what="dw")
    close(bg.grad, br.grad, rtol=3e-4, what="db")


CONV3D = [
    (2, 16, 3, 2, 1, 1, 12, 10, 14),
    (16, 32, 3, 2, 1, 1, 9, 11, 13),
    (34, 32, 3, 1, 1, 1, 6, 10, 12),
    (32, 16, 3, 1, 1, 1, 8, 8, 8),
    (16, 3, 3, 1, 1, 1, 7, 9, 11),
    (64, 32, 3, 1, 1, 2, 4, 6, 5),
    # stride-1 shapes that take the 16x16x4-MFMA halo kernels (conv3d.hip): whole + partial patches,
    # channel groups (Cin > 32), the 34-channel concat layer and its dgrad (Cout' = 34 -> 3 row tiles)
    (34, 32, 3, 1, 1, 1, 6, 16, 32),
    (32, 16, 3, 1, 1, 1, 5, 10, 20),
    # widths that are a multiple of 32: the 16-row kernel's 32-voxel-wide tile (ragged in z and y, two tiles along x,
    # 12 of 16 output channels, batch 2)
    (32, 16, 3, 1, 1, 1, 5, 10, 32),
    (16, 12, 3, 1, 1, 2, 6, 9, 64),
    (16, 16, 3, 1, 1, 2, 4, 8, 16),
    (48, 32, 3, 1, 1, 1, 4, 12, 24),
    (64, 32, 3, 1, 1, 1, 3, 8, 16),
    (32, 34, 3, 1, 1, 1, 4, 8, 12),
    # <= 16 output channels: the plane-pair form of conv3d_split_k (odd depth: the second plane of the last pair is outside)
    (16, 16, 3, 1, 1, 1, 7, 9, 19),
    (40, 12, 3, 1, 1, 1, 6, 7, 17),
    (8, 9, 3, 1, 1, 2, 1 + 2, 8, 16),
    # multi-tile plane-pair form (three y-stacked tiles per workgroup share the weights): 4 tile rows = one full group +
    # one with two phantom tiles; the flow-head shape (Cout 3) with 5 tile rows
    (16, 16, 3, 1, 1, 1, 5, 26, 16),
    (16, 3, 3, 1, 1, 1, 6, 40, 8),
    # the flow head on the fp32-FMA kernel (conv3dt.hip; W % 4 == 0): ragged tiles in every axis with batch 2, several x tiles
    (16, 3, 3, 1, 1, 2, 5, 9, 36),
    (16, 3, 3, 1, 1, 1, 9, 17, 68),
    # the flow head's weight gradient on conv3d_flow_wgrad_k ((co, dx) pairs as MFMA columns, z-marching): several z segments,
    # four output channels (the 12-column instance), one 8 x 32 column
    (16, 3, 3, 1, 1, 1, 40, 24, 64),
    (16, 4, 3, 1, 1, 1, 24, 16, 32),
    (16, 2, 3, 1, 1, 2, 16, 8, 32),
    # the first encoder level (2 -> 16, stride 2) from an LDS-staged patch, weight gradient on fp32 MFMA (conv3dt.hip): one
    # tile, ragged tiles in every axis with batch 2, odd input sizes
    (2, 16, 3, 2, 1, 1, 12, 10, 16),
    (2, 16, 3, 2, 1, 2, 9, 17, 72),
    (2, 16, 3, 2, 1, 1, 7, 33, 136),
    # the z-marching kernel of the full-resolution layers (conv3dm.hip: 32 -> 16, 16 -> 16, and 16 -> 32 as their input
    # gradients and as a forward): several 16 x 32 columns in x and y with ragged last ones, batch 2, odd depths
    (32, 16, 3, 1, 1, 2, 9, 37, 72),
    (16, 16, 3, 1, 1, 1, 13, 20, 40),
    (16, 32, 3, 1, 1, 1, 6, 18, 36),
]


@pytest.mark.parametrize("cfg", CONV3D, ids=[str(c) for c in CONV3D])
def test_conv3d(ops, cfg):
    Cin, Cout, K, stride, pad, N, D, H, W = cfg
    x = C.randn(1, N, Cin, D, H, W)
    w = C.randn(2, Cout, Cin, K, K, K) / (Cin * K ** 3) ** 0.5
    b = C.randn(3, Cout) * 0.1
    act = 1 if Cout > 3 else 0
    # fp64 reference (torch's fp32 CPU convolution backward is itself 7e-4 off on the larger shapes)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    yr = torch_conv(xr, wr, br, stride, pad, False, act, 3)
    cot = C.randn(4, *yr.shape)
    (yr * cot.double()).sum().backward()
    xg, wg, bg = (t.clone().to(DEV).requires_grad_() for t in (x, w, b))
    yg = ops.conv(xg, wg, bg, None, stride, pad, 0, act, 0.2)
    (yg * cot.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    close(yg, yr.float(), what="y")
    close(xg.grad, xr.grad.float(), what="dx")
    close(wg.grad, wr.grad.float(), rtol=3e-4, what="dw")
    close(bg.grad, br.grad.float(), rtol=3e-4, what="db")


def test_round6_weight_gradient_kernels_on_seeded_random_geometries(ops):
    """conv1x1_wgrad_k (1x1 weight gradients as a streaming NT GEMM) and conv3d_flow_wgrad_k (the flow head's weight + bias
    gradient, (co, dx) pairs as MFMA columns) on a dozen seeded random geometries each -- ragged channel tiles, chunks and
    columns that end inside the volume, batches, z segments -- against an fp64 contraction; the A/B kernels they replace
    (DFMIR_NO_1X1_WGRAD / DFMIR_CONV3D_NO_FLOW_WGRAD) must agree with them to fp32 round-off of the same sums."""
    import random
    from dfmir_amd._lib import set_option
    rnd = random.Random(606)
    for case in range(12):
        n, cin, cout = rnd.choice([1, 2, 3]), rnd.choice([8, 17, 49, 64, 72, 130]), rnd.choice([8, 33, 49, 64, 100])
        h, w = rnd.choice([4, 9, 16, 31]), 4 * rnd.choice([4, 5, 16, 33])
        x = C.randn(700 + case, n, cin, 1, h, w).to(DEV)
        dy = (C.randn(720 + case, n, cout, 1, h, w) * 1e-2).to(DEV)
        ref = torch.einsum("ncp,ndp->cd", x.flatten(2).double(), dy.flatten(2).double())
        got = ops.conv_wgrad_raw(x, dy, (1, 1, 1), 1, (0, 0, 0), 0).reshape(cin, cout)
        close(got, ref.float(), rtol=2e-5, what="1x1 dW, case %d (%d x %d -> %d @%dx%d)" % (case, n, cin, cout, h, w))
        set_option("DFMIR_NO_1X1_WGRAD", "1")
        try:
            old = ops.conv_wgrad_raw(x, dy, (1, 1, 1), 1, (0, 0, 0), 0).reshape(cin, cout)
        finally:
            set_option("DFMIR_NO_1X1_WGRAD", None)
        close(got, old, rtol=2e-5, what="1x1 dW vs the generic kernel, case %d" % case)
        # the same geometry forward with bias + LeakyReLU, and its data gradient
        wt = (C.randn(780 + case, cout, cin, 1, 1) / cin ** 0.5)
        bt = C.randn(790 + case, cout) * 0.1
        x4 = x[:, :, 0].cpu()
        xr = x4.double().requires_grad_()
        yr = F.leaky_relu(F.conv2d(xr, wt.double(), bt.double()), 0.2)
        cot = C.randn(795 + case, *yr.shape)
        (yr * cot.double()).sum().backward()
        xg = x4.clone().to(DEV).requires_grad_()
        yg = ops.conv(xg, wt.to(DEV), bt.to(DEV), None, 1, 0, 0, 1, 0.2)
        (yg * cot.to(DEV)).sum().backward()
        close(yg, yr.float(), rtol=2e-5, what="1x1 y, case %d" % case)
        close(xg.grad, xr.grad.float(), rtol=2e-5, what="1x1 dx, case %d" % case)
    for case in range(12):
        n, cout = rnd.choice([1, 2]), rnd.choice([1, 2, 3, 4])
        d, h, w = rnd.choice([2, 5, 9, 24, 41]), rnd.choice([3, 8, 13, 20]), 4 * rnd.choice([2, 8, 9, 17])
        if d * h * w < 1024:
            d = 1024 // (h * w) + 2
        x = C.randn(740 + case, n, 16, d, h, w).to(DEV)
        dy = (C.randn(760 + case, n, cout, d, h, w) * 1e-2).to(DEV)
        xr = x.double().cpu()
        wr = torch.zeros(cout, 16, 3, 3, 3, dtype=torch.float64, requires_grad=True)
        br = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        (F.conv3d(xr, wr, br, padding=1) * dy.double().cpu()).sum().backward()
        xa, da = ops.absmax(x), ops.absmax(dy)
        res = []
        for off in (None, "1"):
            set_option("DFMIR_CONV3D_NO_FLOW_WGRAD", off)
            try:
                db = ops.zeros(cout, DEV)
                dwt = ops.conv_wgrad_raw(x, dy, (3, 3, 3), 1, (1, 1, 1), 0, x_amax=xa, dy_amax=da, db=db)
                res.append((ops.weight_unpack(dwt, (cout, 16, 3, 3, 3)), db))
            finally:
                set_option("DFMIR_CONV3D_NO_FLOW_WGRAD", None)
        tag = "case %d (%d x 16 -> %d @%dx%dx%d)" % (case, n, cout, d, h, w)
        close(res[0][0], wr.grad.float(), rtol=3e-5, what="flow dW, " + tag)
        close(res[0][1], br.grad.float(), rtol=3e-5, what="flow db, " + tag)
        close(res[0][0], res[1][0], rtol=3e-5, what="flow dW vs the tiled kernel, " + tag)


@pytest.mark.parametrize("cfg", [(32, 16, 1, 11, 24, 64), (16, 16, 2, 7, 16, 32), (16, 32, 1, 10, 33, 48)],
                         ids=["32to16", "16to16-batch2", "16to32"])
@pytest.mark.parametrize("nseg", [1, 3])
def test_conv3d_march_kernel(ops, cfg, nseg):
    """conv3d_march_k (csrc/conv3dm.hip): forward with LeakyReLU and the input gradient with the LeakyReLU derivative of
    the producing block in its epilogue, against torch fp64 and against the tiled kernels (DFMIR_CONV3D_NO_MARCH=1) -- with
    one z segment per column and with three (every segment re-stages one halo plane at each end; the open accumulators of
    a segment's first and last planes are discarded)."""
    from dfmir_amd import _lib
    Cin, Cout, N, D, H, W = cfg
    x = C.randn(301, N, Cin, D, H, W)
    w = C.randn(302, Cout, Cin, 3, 3, 3) / (Cin * 27) ** 0.5
    b = C.randn(303, Cout) * 0.1
    cot = C.randn(304, N, Cout, D, H, W)
    src = C.randn(305, N, Cin, D, H, W)                      # the activation source of the folded LeakyReLU backward
    xr, wr = x.double().requires_grad_(), w.double()
    yr = F.leaky_relu(F.conv3d(xr, wr, b.double(), padding=1), 0.2)
    (yr * cot.double()).sum().backward()
    dxr = xr.grad * torch.where(src.double() > 0, 1.0, 0.2)

    def run():
        g = ops.DfConvGeom
        xg, wg, bg = x.to(DEV), w.to(DEV), b.to(DEV)
        wt = ops.weight_pack(wg, 0)
        y = ops.conv_raw(xg.contiguous(), wt, bg, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 1, 0.2, (D, H, W), x_amax=ops.absmax(xg))
        # (the LeakyReLU mask of the REFERENCE: an output within round-off of zero must not decide the comparison)
        dy = (cot * torch.where(yr.detach() > 0, 1.0, 0.2).float()).to(DEV).contiguous()
        wtd = ops.weight_pack(wg, 1)                          # dgrad packing: taps flipped, channel roles swapped
        dx = ops.conv_raw(dy, wtd, None, Cin, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, (D, H, W), x_amax=ops.absmax(dy),
                          act_src=src.to(DEV).contiguous(), act_slope=0.2)
        assert ops._LAST_ACTGRAD[0]
        probe = float(ops.amax_of(dx).max())
        torch.cuda.synchronize()
        return y, dx, probe

    _lib.set_option("DFMIR_MARCH_NSEG", str(nseg))
    try:
        assert ops.lib().dfmir_conv3d_march_ok is not None
        y1, dx1, probe = run()
    finally:
        _lib.set_option("DFMIR_MARCH_NSEG", None)
    _lib.set_option("DFMIR_CONV3D_NO_MARCH", "1")
    try:
        y0, dx0, _ = run()
    finally:
        _lib.set_option("DFMIR_CONV3D_NO_MARCH", None)
    close(y1, yr.detach().float(), rtol=2e-5, what="y vs fp64")
    close(dx1, dxr.float(), rtol=2e-5, what="dx vs fp64")
    close(y1, y0, rtol=2e-6, what="march vs tiled y")
    close(dx1, dx0, rtol=2e-6, what="march vs tiled dx")
    true = float(dx1.abs().max())
    assert probe >= true and probe <= true * (1 + 1e-6), (probe, true)     # the epilogue's range probe of its output


@pytest.mark.parametrize("cfg", [(16, 32, 1, 12, 20, 36), (32, 32, 2, 17, 19, 22), (32, 64, 1, 16, 16, 18), (64, 64, 1, 15, 18, 17),
                                 (16, 32, 1, 20, 48, 56), (8, 16, 2, 12, 18, 33)],
                         ids=["16to32", "32to32-odd-batch2", "32to64", "64to64-odd", "16to32-many-patches", "8to16-odd-width"])
def test_conv3d_stride2_kernels(ops, cfg):
    """ConvBlock(stride 2) of the U-Net encoder (torchvoxelmorph/networks.py:66-71,1506-1521) on csrc/conv3ds2.hip: forward
    (+ LeakyReLU, range probe), data gradient in parity classes, weight / bias gradient -- against torch fp64 and against the
    generic gather kernels (DFMIR_CONV3D_NO_S2); even and odd extents (Do = ceil(Di / 2)), batch 2."""
    Cin, Cout, N, D, H, W = cfg
    x = C.randn(331, N, Cin, D, H, W)
    w = C.randn(332, Cout, Cin, 3, 3, 3) / (Cin * 27) ** 0.5
    b = C.randn(333, Cout) * 0.1
    xr, wr, br = (t.double().clone().requires_grad_() for t in (x, w, b))
    yr = F.leaky_relu(F.conv3d(xr, wr, br, stride=2, padding=1), 0.2)
    cot = C.randn(334, *yr.shape)
    (yr * cot.double()).sum().backward()
    g = ops.DfConvGeom(N, Cin, Cout, D, H, W, yr.shape[2], yr.shape[3], yr.shape[4], 3, 3, 3, 2, 1, 1, 1, 1, 0, 1, 0.2)
    assert ops.lib().dfmir_conv3d_s2_ok(ctypes.byref(g))
    g_small = ops.DfConvGeom(N, Cin, Cout, 8, 8, 8, 4, 4, 4, 3, 3, 3, 2, 1, 1, 1, 1, 0, 1, 0.2)
    assert not ops.lib().dfmir_conv3d_s2_ok(ctypes.byref(g_small))     # the deepest levels stay on conv_tinyvol_k
    res = []
    for off in (False, True):
        keep = ops._NO_S2
        ops._NO_S2 = off
        try:
            xg, wg, bg = (t.clone().to(DEV).requires_grad_() for t in (x, w, b))
            yg = ops.conv(xg, wg, bg, None, 2, 1, 0, 1, 0.2)
            probe = float(ops.amax_of(yg).max())
            yg.backward(cot.to(DEV))
            torch.cuda.synchronize()
            res.append((yg.detach(), xg.grad, wg.grad, bg.grad, probe))
        finally:
            ops._NO_S2 = keep
    y1, dx1, dw1, db1, probe = res[0]
    close(y1, yr.detach().float(), rtol=2e-5, what="y vs fp64")
    close(dx1, xr.grad.float(), rtol=1e-4, what="dx vs fp64")
    close(dw1, wr.grad.float(), rtol=1e-4, what="dw vs fp64")
    close(db1, br.grad.float(), rtol=1e-4, what="db vs fp64")
    close(y1, res[1][0], rtol=2e-5, what="y vs the generic kernel")
    close(dx1, res[1][1], rtol=1e-4, what="dx vs the generic kernel")
    close(dw1, res[1][2], rtol=1e-4, what="dw vs the generic kernel")
    true = float(y1.abs().max())
    assert probe >= true and probe <= true * (1 + 1e-6), (probe, true)


@pytest.mark.parametrize("cfg", [(1, 11, 24, 64), (2, 7, 16, 32), (1, 10, 33, 48), (1, 6, 40, 100)],
                         ids=["11x24x64", "batch2-7x16x32", "ragged-10x33x48", "ragged-6x40x100"])
@pytest.mark.parametrize("nseg", [1, 3])
def test_conv3d_flow_head_forward_on_the_march_kernel(ops, cfg, nseg):
    """The flow head Conv3d(16, 3, 3, padding=1) (torchvoxelmorph/networks.py:1076-1080) on conv3d_march_k's FLOW form
    (columns = (open plane, channel), one accumulator set, weights pre-rotated per phase of the march): against torch
    fp64 and against the fp32-FMA kernel it replaces (csrc/conv3dt.hip); range probe of the result."""
    from dfmir_amd import _lib
    N, D, H, W = cfg
    x = C.randn(321, N, 16, D, H, W)
    w = C.randn(322, 3, 16, 3, 3, 3) / (16 * 27) ** 0.5
    b = C.randn(323, 3) * 0.1
    yr = F.conv3d(x.double(), w.double(), b.double(), padding=1)
    xg, wg, bg = x.to(DEV).contiguous(), w.to(DEV), b.to(DEV)
    wt = ops.weight_pack(wg, 0)

    def run():
        y = ops.conv_raw(xg, wt, bg, 3, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, (D, H, W), x_amax=ops.absmax(xg))
        probe = float(ops.amax_of(y).max())
        torch.cuda.synchronize()
        return y, probe

    g = ops.DfConvGeom(N, 16, 3, D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 0, 0, 0.0)
    assert ops.lib().dfmir_conv3d_march_ok(ctypes.byref(g))
    _lib.set_option("DFMIR_MARCH_NSEG", str(nseg))
    try:
        y1, probe = run()
    finally:
        _lib.set_option("DFMIR_MARCH_NSEG", None)
    keep = ops._NO_FLOW_MARCH
    ops._NO_FLOW_MARCH = True
    try:
        y0, _ = run()
    finally:
        ops._NO_FLOW_MARCH = keep
    close(y1, yr.float(), rtol=2e-5, what="flow head vs fp64")
    close(y1, y0, rtol=2e-6, what="march FLOW form vs the fp32-FMA kernel")
    true = float(y1.abs().max())
    assert probe >= true and probe <= true * (1 + 1e-6), (probe, true)


@pytest.mark.parametrize("cfg", [(32, 16), (16, 16), (16, 32), (16, 3)], ids=["32to16", "16to16", "16to32", "16to3-flow"])
def test_conv3d_march_writes_every_voxel_and_is_reproducible(ops, cfg):
    """150 launches of conv3d_march_k into NaN-filled outputs (batch 2, 20 x 80 x 96: 30 columns, three z segments, with and
    without the folded LeakyReLU derivative): every voxel written, every launch bit-identical to the first.  This loop is
    what exposed the gfx950 store hazard (a 16-byte buffer store with an SGPR soffset followed directly by a VALU write of
    its data registers: DESIGN section 4, hardware fact 7) -- one wrong dword in ~10^4 launches."""
    import ctypes
    from dfmir_amd.ops import DfConvGeom, _p, _st, check, lib
    Cin, Cout = cfg
    N, D, H, W = 2, 20, 80, 96
    x = C.randn(311, N, Cin, D, H, W).to(DEV)
    w = (C.randn(312, Cout, Cin, 3, 3, 3) / (Cin * 27) ** 0.5).to(DEV)
    src = C.randn(313, N, Cout, D, H, W).to(DEV)
    wt = ops.weight_pack(w, 0)
    xa = ops.absmax(x)
    g = DfConvGeom(N, Cin, Cout, D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 0, 0, 0.0)
    assert lib().dfmir_conv3d_march_ok(ctypes.byref(g))
    for actg in ((False,) if Cout == 3 else (False, True)):
        ref = None
        for rep in range(75):
            y = torch.full((N, Cout, D, H, W), float("nan"), device=DEV)
            slot = ops.amax_slot(x.device, ops.PROBE_SLOTS)
            check(lib().dfmir_conv3d_march_fwd(ctypes.byref(g), _p(x), _p(xa), 1, _p(wt), None, _p(y), _p(slot),
                                               _p(src) if actg else None, 0.2, _st()))
            if ref is None:
                ref = y
                assert not bool(torch.isnan(y).any()), "unwritten output voxels"
                continue
            assert torch.equal(y, ref), "launch %d differs from the first (actg=%s)" % (rep, actg)


@pytest.mark.parametrize("cfg", [(16, 2, 32, 1, 6, 10, 12, 1), (32, 16, 32, 1, 5, 9, 20, 1), (8, 3, 16, 2, 4, 8, 8, 0),
                                 (64, 64, 64, 1, 3, 5, 8, 1), (32, 2, 32, 1, 9, 17, 36, 1), (32, 32, 16, 2, 4, 6, 20, 0),
                                 (32, 8, 24, 1, 7, 5, 12, 1)],
                         ids=["16+2->32", "32+16->32", "8+3->16-pair-batch2", "64+64->64", "32+2->32-ragged",
                              "32+32->16-batch2", "32+8->24"])
def test_upcat_conv3d_parity_class_form(ops, cfg):
    """ConvBlock over cat([nearest_up2(a), b], 1) (torchvoxelmorph/networks.py:64,97-100,1506-1521) in the form that
    never builds the concatenation -- up-sampled channels as eight 8-tap parity-class convolutions of `a` with summed
    weights + skip channels with the partial sum added in the epilogue -- against torch fp64 on the materialised tensor:
    output, d(a), d(b), dW, db; and against the materialising path of this library."""
    Ca, Cb, Cout, N, D, H, W, act = cfg
    a = C.randn(201, N, Ca, D, H, W)
    b = C.randn(202, N, Cb, 2 * D, 2 * H, 2 * W)
    w = C.randn(203, Cout, Ca + Cb, 3, 3, 3) / ((Ca + Cb) * 27) ** 0.5
    bias = C.randn(204, Cout) * 0.1
    cot = C.randn(205, N, Cout, 2 * D, 2 * H, 2 * W)
    ar, br, wr, biasr = (t.double().clone().requires_grad_() for t in (a, b, w, bias))
    xr = torch.cat([F.interpolate(ar, scale_factor=2, mode="nearest"), br], 1)
    yr = F.conv3d(xr, wr, biasr, padding=1)
    if act:
        yr = F.leaky_relu(yr, 0.2)
    (yr * cot.double()).sum().backward()
    assert ops.upcat_conv3d_ok(a.to(DEV), b.to(DEV), w.to(DEV))
    ag, bg, wg, biasg = (t.clone().to(DEV).requires_grad_() for t in (a, b, w, bias))
    yg = ops.upcat_conv3d(ag, bg, wg, biasg, None, act, 0.2)
    (yg * cot.to(DEV)).sum().backward()
    close(yg, yr, what="y")
    close(ag.grad, ar.grad, rtol=3e-4, what="da")
    close(bg.grad, br.grad, rtol=3e-4, what="db")
    close(wg.grad, wr.grad, rtol=1e-3, what="dw")
    close(biasg.grad, biasr.grad, rtol=1e-3, what="dbias")
    # b without a gradient (the network's input images at the top level): d(a) comes from the parity-class dgrad kernel
    # (4x4x4 stride-2 conv of dy at low resolution) instead of full-resolution dgrad + 2x2x2 sum pool
    if Cout % 8 == 0:
        a3, w3, bias3 = (t.clone().to(DEV).requires_grad_() for t in (a, w, bias))
        y3 = ops.upcat_conv3d(a3, b.clone().to(DEV), w3, bias3, None, act, 0.2)
        (y3 * cot.to(DEV)).sum().backward()
        close(y3, yr, what="y (b no grad)")
        close(a3.grad, ar.grad, rtol=3e-4, what="da (parity-class dgrad)")
        close(w3.grad, wr.grad, rtol=1e-3, what="dw (b no grad)")
    # the materialising path (one conv over the concatenation): same numbers up to the order of the weight sums
    a2, b2 = a.clone().to(DEV), b.clone().to(DEV)
    y2 = ops.conv(ops.upcat(a2, b2), wg.detach(), biasg.detach(), None, 1, 1, 0, act, 0.2)
    close(yg, y2, rtol=2e-5, what="y vs materialised")


@pytest.mark.parametrize("cfg", [(2, 32, 1, 9, 17, 36), (2, 32, 2, 5, 6, 16), (2, 16, 1, 13, 8, 20), (2, 24, 1, 4, 9, 12),
                                 (16, 32, 1, 6, 10, 20), (8, 16, 2, 4, 5, 16)],
                         ids=["+2->32-ragged", "+2->32-batch2", "+2->16-z-segments", "+2->24", "+16->32", "+8->16-batch2"])
@pytest.mark.parametrize("form", ["4wave", "8wave"])
def test_upwgrad_parity_class_kernel(ops, cfg, form, monkeypatch):
    """dfmir_conv3d_upwgrad (csrc/conv3duw.hip): the weight gradient of Conv3d(3, padding 1) over cat([nearest_up2(a), b], 1)
    (torchvoxelmorph/networks.py:64,97-100) with the 32 up-sampled channels in parity classes -- 64 products per
    LOW-resolution voxel -- against torch fp64 on the materialised tensor: dW and db.  Two skip channels ride in the same
    launch (rows = (tap, channel) from the x-/y-paired images of b, the bias gradient as the ones row); more skip channels
    go through the direct kernel.  Shapes with ragged columns (H/2 % 4, W/2 % 16 != 0), several z segments, a batch, fewer
    than 32 output channels; `8wave` = the two-waves-per-SIMD form of the up-sampled share (DFMIR_UPWGRAD_8WAVE)."""
    import ctypes
    from dfmir_amd._lib import set_option
    from dfmir_amd.ops import DfConvGeom, lib
    Cb, Cout, N, D, H, W = cfg
    if form == "8wave" and Cb == 2:
        pytest.skip("two skip channels are always fused into the one-wave-per-SIMD kernel")
    monkeypatch.setattr(ops, "_UPWGRAD_MIN_VOX", 0)
    a = C.randn(211, N, 32, D, H, W)
    b = C.randn(212, N, Cb, 2 * D, 2 * H, 2 * W)
    dy = C.randn(213, N, Cout, 2 * D, 2 * H, 2 * W) * 1e-3
    xr = torch.cat([F.interpolate(a.double(), scale_factor=2, mode="nearest"), b.double()], 1)
    wr = torch.zeros(Cout, 32 + Cb, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(xr, wr, None, padding=1).backward(dy.double())
    ref = wr.grad.permute(2, 3, 4, 1, 0).reshape(27, 32 + Cb, Cout)          # tap-major packing [27][Cin][Cout]
    ag, bg, dyg = a.to(DEV), b.to(DEV), dy.to(DEV)
    xa = ops.absmax(torch.cat([ag.flatten(), bg.flatten()])).clone()
    da = ops.absmax(dyg).clone()
    g = DfConvGeom(N, 32 + Cb, Cout, 2 * D, 2 * H, 2 * W, 2 * D, 2 * H, 2 * W, 3, 3, 3, 1, 1, 1, 1, 1, 0, 0, 0.0)
    assert lib().dfmir_conv3d_upwgrad_ok(ctypes.byref(g), 32)
    set_option("DFMIR_UPWGRAD_8WAVE", "1" if form == "8wave" else None)
    try:
        for nseg in (None, 3):                                            # the launcher's choice, then forced z segments
            set_option("DFMIR_UPWGRAD_NSEG", nseg)
            db = torch.zeros(Cout, device=DEV)
            dw = ops.conv_wgrad_raw(None, dyg, (3, 3, 3), 1, (1, 1, 1), 0, x_amax=xa, dy_amax=da, db=db, parts=(ag, bg))
            close(dw, ref, rtol=1e-4, atol=0, what="dW nseg=%s" % nseg)
            close(dw[:, :32], ref[:, :32], rtol=1e-4, atol=0, what="dW up-sampled rows nseg=%s" % nseg)
            close(dw[:, 32:], ref[:, 32:], rtol=1e-4, atol=0, what="dW skip rows nseg=%s" % nseg)
            close(db, dy.double().sum((0, 2, 3, 4)), rtol=1e-4, atol=0, what="db nseg=%s" % nseg)
            dw2 = ops.conv_wgrad_raw(None, dyg, (3, 3, 3), 1, (1, 1, 1), 0, out=dw.clone(), x_amax=xa, dy_amax=da, db=db,
                                     parts=(ag, bg))                     # accumulates into `out` and `db`
            close(dw2, 2 * ref, rtol=1e-4, atol=0, what="dW accumulated")
            close(db, 2 * dy.double().sum((0, 2, 3, 4)), rtol=1e-4, atol=0, what="db accumulated")
    finally:
        set_option("DFMIR_UPWGRAD_8WAVE", None)
        set_option("DFMIR_UPWGRAD_NSEG", None)


@pytest.mark.parametrize("cfg", [(1, 9, 37, 72), (2, 5, 13, 40), (1, 12, 8, 32), (1, 3, 20, 68)],
                         ids=["ragged", "batch2", "z-segments", "three-planes"])
def test_conv3d_wgrad_march_kernel(ops, cfg):
    """conv3d_wgrad_march_k (csrc/conv3dwm.hip): weight and bias gradient of the full-resolution 32 -> 16 3x3x3 layer
    (torchvoxelmorph/networks.py:73-86) with all 27 tap matrices resident in one wave's accumulators, z-marching -- against
    torch fp64 and against the tiled kernel it replaces (DFMIR_CONV3D_NO_WGRAD_MARCH=1); ragged columns (H % 8, W % 32 != 0),
    forced z segments, a batch, accumulation into an existing gradient."""
    from dfmir_amd._lib import set_option
    N, D, H, W = cfg
    x = C.randn(221, N, 32, D, H, W)
    dy = C.randn(222, N, 16, D, H, W) * 1e-3
    wr = torch.zeros(16, 32, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv3d(x.double(), wr, None, padding=1).backward(dy.double())
    ref = wr.grad.permute(2, 3, 4, 1, 0).reshape(27, 32, 16)               # tap-major packing [27][Cin][Cout]
    dbref = dy.double().sum((0, 2, 3, 4))
    xg, dyg = x.to(DEV), dy.to(DEV)
    xa, da = ops.absmax(xg).clone(), ops.absmax(dyg).clone()
    got = {}
    try:
        for tag, off, nseg in (("march", None, None), ("march-3seg", None, 3), ("tiled", "1", None)):
            set_option("DFMIR_CONV3D_NO_WGRAD_MARCH", off)
            set_option("DFMIR_WGRAD_MARCH_NSEG", nseg)
            db = torch.zeros(16, device=DEV)
            dw = ops.conv_wgrad_raw(xg, dyg, (3, 3, 3), 1, (1, 1, 1), 0, x_amax=xa, dy_amax=da, db=db)
            close(dw, ref, rtol=1e-4, atol=0, what="dW " + tag)
            close(db, dbref, rtol=1e-4, atol=0, what="db " + tag)
            dw2 = ops.conv_wgrad_raw(xg, dyg, (3, 3, 3), 1, (1, 1, 1), 0, out=dw.clone(), x_amax=xa, dy_amax=da, db=db)
            close(dw2, 2 * ref, rtol=1e-4, atol=0, what="dW accumulated " + tag)
            close(db, 2 * dbref, rtol=1e-4, atol=0, what="db accumulated " + tag)
            got[tag] = dw
    finally:
        set_option("DFMIR_CONV3D_NO_WGRAD_MARCH", None)
        set_option("DFMIR_WGRAD_MARCH_NSEG", None)
    close(got["march"], got["tiled"], rtol=2e-5, atol=0, what="march vs tiled")


@pytest.mark.parametrize("cfg", [(32, 32, 4, 1, 1), (32, 32, 2, 1, 1), (32, 48, 8, 1, 1), (16, 32, 8, 2, 1), (32, 16, 4, 1, 2), (64, 20, 4, 1, 1)],
                         ids=["4^3", "2^3", "8^3-48out", "8^3-stride2", "4^3-dgrad-of-stride2", "4^3-20out"])
def test_conv_tiny_volume_kernel(ops, cfg):
    """conv_tinyvol_k (csrc/conv.hip): the convolutions of the deepest U-Net levels (<= 512 output voxels; the 6-level 3-D
    plugin network at 128^3: torchvoxelmorph/networks.py:66-86) -- one workgroup per output voxel, taps dealt over the threads --
    against torch fp64 through the autograd Function (forward, dx incl. the zero-dilated dgrad of a stride-2 conv, dW, db) and
    against the generic MFMA kernel (DFMIR_NO_TINYVOL=1)."""
    from dfmir_amd._lib import set_option
    Cin, Cout, S, stride, batch = cfg
    x = C.randn(231, batch, Cin, S, S, S)
    w = C.randn(232, Cout, Cin, 3, 3, 3) / (Cin * 27) ** 0.5
    b = C.randn(233, Cout) * 0.1
    xr, wr, br = (t.double().clone().requires_grad_() for t in (x, w, b))
    yr = F.leaky_relu(F.conv3d(xr, wr, br, stride=stride, padding=1), 0.2)
    cot = C.randn(234, *yr.shape)
    (yr * cot.double()).sum().backward()
    outs = []
    try:
        for off in (None, "1"):
            set_option("DFMIR_NO_TINYVOL", off)
            xg, wg, bg = (t.clone().to(DEV).requires_grad_() for t in (x, w, b))
            yg = ops.conv(xg, wg, bg, None, stride, 1, 0, 1, 0.2)
            (yg * cot.to(DEV)).sum().backward()
            close(yg, yr, what="y"); close(xg.grad, xr.grad, rtol=3e-4, what="dx")
            close(wg.grad, wr.grad, rtol=1e-3, what="dw"); close(bg.grad, br.grad, rtol=1e-3, what="db")
            outs.append((yg.detach(), xg.grad))
    finally:
        set_option("DFMIR_NO_TINYVOL", None)
    close(outs[0][0], outs[1][0], rtol=2e-6, what="y tiny-volume vs MFMA kernel")
    close(outs[0][1], outs[1][1], rtol=2e-5, what="dx tiny-volume vs MFMA kernel")


def test_conv3d_chain_folds_leaky_relu_backward(ops):
    """A chain of LeakyReLU ConvBlocks whose outputs feed only the next conv (conv(sole=True)): the consumer's dgrad
    epilogue applies the activation's derivative (dfmir_conv3d_split_fwd_actgrad), so no act_bwd pass runs between
    them.  Same gradients as the unfused chain (same arithmetic: one multiply per element) and as torch."""
    from dfmir_amd import networks as N_
    torch.manual_seed(0)
    convs = [N_.Conv3d(8, 16, 3, padding=1), N_.Conv3d(16, 16, 3, padding=1), N_.Conv3d(16, 3, 3, padding=1)]
    for c in convs:
        c.to(DEV)
    x = C.randn(91, 1, 8, 6, 10, 16).to(DEV)
    cot = C.randn(92, 1, 3, 6, 10, 16).to(DEV)

    def run(sole):
        for c in convs:
            c.weight.grad = None
            c.bias.grad = None
        xg = x.clone().requires_grad_()
        h = convs[0](xg, act=1, slope=0.2, sole=sole)
        h = convs[1](h, act=1, slope=0.2, sole=sole)
        y = convs[2](h)
        (y * cot).sum().backward()
        torch.cuda.synchronize()
        return [y.detach(), xg.grad] + [c.weight.grad.clone() for c in convs] + [c.bias.grad.clone() for c in convs]

    L, cnt = ops.lib(), [0]
    o1, o2 = L.dfmir_act_bwd_amax, L.dfmir_act_bwd

    def counted(f):
        def g(*a):
            cnt[0] += 1
            return f(*a)
        return g

    L.dfmir_act_bwd_amax, L.dfmir_act_bwd = counted(o1), counted(o2)
    try:
        fused = run(True)
        n_fused, cnt[0] = cnt[0], 0
        plain = run(False)
        n_plain = cnt[0]
    finally:
        L.dfmir_act_bwd_amax, L.dfmir_act_bwd = o1, o2
    import os
    off = any(os.environ.get(k) for k in ("DFMIR_NO_ACTGRAD", "DFMIR_CONV3D_FP32", "DFMIR_CONV_FP32"))   # A/B switches
    # no standalone LeakyReLU backward pass in the fused chain (unless the fusion is switched off)
    assert (n_fused, n_plain) == ((2, 2) if off else (0, 2)), (n_fused, n_plain)
    for a, b in zip(fused, plain):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-12
    # torch reference
    xr = x.cpu().double().requires_grad_()
    ws = [c.weight.detach().cpu().double().requires_grad_() for c in convs]
    bs = [c.bias.detach().cpu().double().requires_grad_() for c in convs]
    h = F.leaky_relu(F.conv3d(xr, ws[0], bs[0], padding=1), 0.2)
    h = F.leaky_relu(F.conv3d(h, ws[1], bs[1], padding=1), 0.2)
    yr = F.conv3d(h, ws[2], bs[2], padding=1)
    (yr * cot.cpu().double()).sum().backward()
    close(fused[0], yr.detach().float(), what="y")
    close(fused[1], xr.grad.float(), what="dx")
    for i in range(3):
        close(fused[2 + i], ws[i].grad.float(), rtol=3e-4, what="dw%d" % i)
        close(fused[5 + i], bs[i].grad.float(), rtol=3e-4, what="db%d" % i)
    # the fused run really skipped the standalone passes: the tag protocol marks the gradient tensors
    xg = x.clone().requires_grad_()
    h0 = convs[0](xg, act=1, slope=0.2, sole=True)
    assert getattr(h0, "_df_act_sole", None) is not None


# --------------------------------------------------------------------------------- norm / resample
@pytest.mark.parametrize("shape,relu,res", [((2, 5, 16, 16), True, False), ((2, 7, 9, 11), False, True),
                                            ((1, 3, 64, 64), True, True), ((3, 4, 5, 5), False, False),
                                            ((2, 2, 128, 128), True, False), ((1, 2, 256, 256), False, True),
                                            ((1, 3, 256, 256), True, False)])
def test_instnorm(ops, shape, relu, res):
    x = C.randn(5, *shape) * 2 + 0.7
    r = C.randn(6, *shape) if res else None
    xr = x.clone().requires_grad_()
    rr = r.clone().requires_grad_() if res else None
    yr = F.instance_norm(xr, eps=1e-5)
    if relu:
        yr = F.relu(yr)
    if res:
        yr = rr + yr
    cot = C.randn(7, *shape)
    (yr * cot).sum().backward()
    xg = x.clone().to(DEV).requires_grad_()
    rg = r.clone().to(DEV).requires_grad_() if res else None
    yg = ops.instance_norm(xg, rg, relu, 1e-5)
    (yg * cot.to(DEV)).sum().backward()
    close(yg, yr, what="y")
    close(xg.grad, xr.grad, rtol=3e-4, what="dx")
    if res:
        close(rg.grad, rr.grad, what="dres")


@pytest.mark.parametrize("hw", [64, 128, 256])
def test_instnorm_statistics_only_pass(ops, hw):
    """dfmir_instnorm_stats (the statistics + range probe of InstanceNorm without the normalised tensor; the measurement
    behind profiles/r06_in_fusion_probes.txt): mean / rstd bit-identical to dfmir_instnorm_fwd's, the probe covers
    max relu(IN(x)), nothing else is written."""
    from dfmir_amd.ops import _p, _st, check, lib
    x = (C.randn(8, 2, 3, hw, hw) * 1.7 - 0.4).to(DEV)
    planes, S = 6, hw * hw
    assert lib().dfmir_instnorm_stats_ok(S) == 1 and lib().dfmir_instnorm_stats_ok(100) == 0
    y = torch.empty_like(x)
    m0, r0, m1, r1 = (torch.empty(planes, device=DEV) for _ in range(4))
    p0, p1 = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    check(lib().dfmir_instnorm_fwd(_p(x), None, _p(y), _p(m0), _p(r0), planes, S, 1e-5, 1, _p(p0), _st()))
    check(lib().dfmir_instnorm_stats(_p(x), _p(m1), _p(r1), planes, S, 1e-5, 1, _p(p1), _st()))
    torch.cuda.synchronize()
    assert torch.equal(m0, m1) and torch.equal(r0, r1) and torch.equal(p0, p1)
    assert float(p1.max()) == float(y.max()) > 0


@pytest.mark.parametrize("shape", [(2, 3, 16, 24), (1, 2, 64, 64), (1, 1, 8, 8), (1, 2, 4, 8), (1, 2, 6, 12)])
def test_blur_vectorised_vs_oracle(ops, shape):
    """Shapes that take the 4-outputs-per-thread resampling kernels (norm_resample.hip *_v4_k), against the
    oracle's Downsample / Upsample restatement, values and input gradients."""
    from oracle import dfmir_oracle as O
    for name, mod, fn in (("down", O.BlurDown(shape[1]), ops.blur_down), ("up", O.BlurUp(shape[1]), ops.blur_up)):
        x = C.randn(91, *shape)
        xr = x.clone().requires_grad_()
        yr = mod(xr)
        cot = C.randn(92, *yr.shape)
        (yr * cot).sum().backward()
        xg = x.clone().to(DEV).requires_grad_()
        yg = fn(xg)
        (yg * cot.to(DEV)).sum().backward()
        close(yg, yr.detach(), what=name)
        close(xg.grad, xr.grad, what="d" + name)


def test_blur_reflect_golden(ops, golden):
    g = golden("blur.npz")
    x = C.randn(31, 2, 8, 12, 12).to(DEV).requires_grad_()
    y = ops.blur_down(x)
    (y * C.randn(32, *y.shape).to(DEV)).sum().backward()
    close(y, g["down"], what="down"); close(x.grad, g["ddown"], what="ddown")
    x = C.randn(33, 2, 8, 12, 12).to(DEV).requires_grad_()
    y = ops.blur_up(x)
    (y * C.randn(34, *y.shape).to(DEV)).sum().backward()
    close(y, g["up"], what="up"); close(x.grad, g["dup"], what="dup")
    xo = C.randn(35, 1, 3, 7, 9).to(DEV).requires_grad_()
    yo = ops.blur_down(xo)
    (yo * C.randn(36, *yo.shape).to(DEV)).sum().backward()
    close(yo, g["down_odd"], what="down odd"); close(xo.grad, g["ddown_odd"], what="ddown odd")
    for p, shp in ((3, (2, 1, 9, 12)), (1, (1, 4, 5, 6)), (2, (1, 2, 3, 3)), (1, (2, 3, 8, 8)), (1, (1, 2, 10, 24)), (1, (1, 1, 64, 64))):
        x = C.randn(37, *shp)
        xr = x.clone().requires_grad_()
        yr = F.pad(xr, (p, p, p, p), mode='reflect')
        cot = C.randn(38, *yr.shape)
        (yr * cot).sum().backward()
        xg = x.clone().to(DEV).requires_grad_()
        yg = ops.reflect_pad2d(xg, p)
        (yg * cot.to(DEV)).sum().backward()
        close(yg, yr, what="pad"); close(xg.grad, xr.grad, what="dpad")
        # the fold with a second gradient summed in the same pass (ResnetBlock's skip branch)
        from dfmir_amd._lib import lib, check
        add = C.randn(39, *shp).to(DEV)
        cg, out = cot.to(DEV).contiguous(), torch.empty(*shp, device=DEV)
        check(lib().dfmir_reflect_pad2d_bwd_add(cg.data_ptr(), add.data_ptr(), out.data_ptr(), shp[0] * shp[1],
                                                shp[2], shp[3], p, torch.cuda.current_stream().cuda_stream))
        close(out, xr.grad + add.cpu(), what="dpad + skip")


@pytest.mark.parametrize("even", [False, True], ids=["odd-width", "even-width"])   # even: the 8- / 16-byte forms
@pytest.mark.parametrize("nd", [2, 3])
def test_upcat_cat_scale(ops, nd, even):
    sp = ((6, 8) if even else (5, 7)) if nd == 2 else ((3, 4, 8) if even else (3, 4, 5))
    a = C.randn(41, 2, 6, *sp)
    b = C.randn(42, 2, 3, *[2 * s for s in sp])
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    yr = torch.cat([F.interpolate(ar, scale_factor=2, mode='nearest'), br], dim=1)
    cot = C.randn(43, *yr.shape)
    (yr * cot).sum().backward()
    ag, bg = a.clone().to(DEV).requires_grad_(), b.clone().to(DEV).requires_grad_()
    yg = ops.upcat(ag, bg)
    (yg * cot.to(DEV)).sum().backward()
    close(yg, yr, what="upcat"); close(ag.grad, ar.grad, what="da"); close(bg.grad, br.grad, what="db")
    c = C.randn(44, 2, 2, *sp)
    ag, cg = a.clone().to(DEV).requires_grad_(), c.clone().to(DEV).requires_grad_()
    yg = ops.upcat_channels(ag, cg)
    close(yg, torch.cat([a, c], 1), what="cat")
    cot = C.randn(45, *yg.shape)
    (yg * cot.to(DEV)).sum().backward()
    close(ag.grad, cot[:, :6], what="dcat a"); close(cg.grad, cot[:, 6:], what="dcat c")
    yb = ops.cat_batch(ag, ag.detach() * 2)
    close(yb, torch.cat([a, 2 * a], 0), what="cat_batch")
    sg = ops.scale(ag, -0.125)
    close(sg, -0.125 * a, what="scale")


# ------------------------------------------------------------------------------------------ warps
def test_warp_golden(ops, golden):
    g = golden("warp.npz")
    for tag, shp, C_ in (("2d", (17, 23), 3), ("3d", (9, 11, 13), 2)):
        B = 2 if tag == "2d" else 1
        src = C.randn(11, B, C_, *shp).to(DEV).requires_grad_()
        flow = ((C.rand(12, B, len(shp), *shp) * 12) - 6).to(DEV).requires_grad_()
        cot = C.randn(13, B, C_, *shp).to(DEV)
        y = ops.warp(src, flow)
        (y * cot).sum().backward()
        close(y, g["out_" + tag], what="warp " + tag)
        close(src.grad, g["dsrc_" + tag], what="dsrc " + tag)
        close(flow.grad, g["dflow_" + tag], rtol=3e-4, what="dflow " + tag)
        yn = ops.warp(src.detach(), flow.detach(), "nearest")
        gn = g["nearest_" + tag]
        frac_bad = float((np.abs(yn.cpu().numpy() - gn) > 1e-5).mean())
        assert frac_bad < 0.01, "nearest warp differs on %.3f of voxels" % frac_bad  # ties may round apart


@pytest.mark.parametrize("shape", [(2, 3, 16, 24), (1, 2, 6, 10, 12), (1, 1, 40, 48, 56)])
def test_warp_vectorised_vs_oracle(ops, shape):
    """W % 4 == 0 takes the 4-voxels-per-thread kernel; compare with the oracle incl. out-of-bounds samples."""
    from oracle import dfmir_oracle as O
    nd = len(shape) - 2
    src = C.randn(55, *shape)
    flow = (C.rand(56, shape[0], nd, *shape[2:]) * 14) - 7
    ref = O.spatial_transform(src, flow)
    got = ops.warp(src.to(DEV), flow.to(DEV))
    close(got, ref, what="warp v4")


@pytest.mark.parametrize("shape", [(2, 3, 16, 24), (2, 2, 50, 44), (1, 2, 6, 10, 12), (1, 3, 13, 21, 36), (1, 1, 40, 48, 56)])
@pytest.mark.parametrize("amp", [1.5, 5.0, 30.0])
def test_warp_windowed_fwd_bwd_vs_oracle(ops, shape, amp):
    """The LDS-windowed kernels (warp_win.hip) on ragged tile edges, with displacement fields that stay
    inside the window (amp 1.5), mix window hits with the global fallback (5) and mostly leave it (30):
    output, d(src) and d(flow) against the oracle's grid_sample + autograd."""
    from oracle import dfmir_oracle as O
    nd = len(shape) - 2
    src = C.randn(61, *shape).requires_grad_()
    coarse = (C.rand(62, shape[0], nd, *[max(2, s // 6) for s in shape[2:]]) * 2 - 1) * amp
    flow = torch.nn.functional.interpolate(coarse, size=shape[2:], mode="trilinear" if nd == 3 else "bilinear",
                                           align_corners=True)
    flow = (flow + (C.rand(63, *flow.shape) - 0.5) * 0.5).requires_grad_()
    cot = C.randn(64, *shape)
    ref = O.spatial_transform(src, flow)
    (ref * cot).sum().backward()
    s_, f_ = src.detach().to(DEV).requires_grad_(), flow.detach().to(DEV).requires_grad_()
    got = ops.warp(s_, f_)
    (got * cot.to(DEV)).sum().backward()
    close(got, ref.detach(), what="warp win fwd")
    close(s_.grad, src.grad, what="warp win dsrc")
    close(f_.grad, flow.grad, rtol=3e-4, what="warp win dflow")


def test_warp_identity_and_oob(ops):
    x = C.randn(51, 2, 1, 20, 30).to(DEV)
    z = torch.zeros(2, 2, 20, 30, device=DEV)
    close(ops.warp(x, z), x, what="identity")
    far = torch.full((2, 2, 20, 30), 100.0, device=DEV)
    assert float(ops.warp(x, far).abs().max()) == 0.0
    one = torch.zeros(2, 2, 20, 30, device=DEV)
    one[:, 0] = 1.0   # flow channel 0 = +1 row: content moves up, last row -> 0
    y = ops.warp(x, one)
    close(y[:, :, :-1], x[:, :, 1:], what="shift")
    assert float(y[:, :, -1].abs().max()) == 0.0


def test_vecint_resize_golden(ops, golden):
    g = golden("vecint_resize.npz")
    for tag, shp in (("2d", (32, 32)), ("3d", (8, 10, 12))):
        v = (C.randn(21, 2 if tag == "2d" else 1, len(shp), *shp) * 2.0).to(DEV).requires_grad_()
        cot = C.randn(22, *v.shape).to(DEV)
        y = ops.scale(v, 1.0 / 128)
        for _ in range(7):
            y = ops.vecint_step(y)
        (y * cot).sum().backward()
        close(y, g["vecint_" + tag], what="vecint " + tag)
        close(v.grad, g["dvecint_" + tag], rtol=1e-3, what="dvecint " + tag)
        x = C.randn(23, 1, len(shp), *shp).to(DEV).requires_grad_()
        half = ops.resize_linear(x, [s // 2 for s in shp], 0.5)
        (half * C.randn(24, *half.shape).to(DEV)).sum().backward()
        close(half, g["half_" + tag], what="half"); close(x.grad, g["dhalf_" + tag], what="dhalf")
        x2 = C.randn(25, 1, len(shp), *shp).to(DEV).requires_grad_()
        dbl = ops.resize_linear(x2, [s * 2 for s in shp], 2.0)
        (dbl * C.randn(26, *dbl.shape).to(DEV)).sum().backward()
        close(dbl, g["double_" + tag], what="double"); close(x2.grad, g["ddouble_" + tag], what="ddouble")


@pytest.mark.parametrize("shp", [(12, 16, 24), (20, 32), (9, 10, 40)], ids=["3d", "2d", "3d-odd-depth"])
def test_resize_row_kernels_equal_the_gather_kernels(ops, shp):
    """resize_rows_fwd_k / resize_w_bwd4_k (csrc/warp.hip: one wave per output row from LDS; four inputs per thread in the
    adjoint's W pass) evaluate the expressions of the per-thread gather kernels in the same order: BIT-identical results,
    forward and adjoint, x0.5 and x2 (ResizeTransform, torchvoxelmorph/layers.py:71-97), and equal to F.interpolate."""
    from dfmir_amd import _lib
    nd = len(shp)
    mode = "trilinear" if nd == 3 else "bilinear"
    for factor in (0.5, 2.0):
        out_sp = [int(s * factor) for s in shp]
        x = C.randn(27, 2, nd, *shp).to(DEV)
        cot = C.randn(28, 2, nd, *out_sp).to(DEV)
        res = []
        for off in (None, "1"):
            _lib.set_option("DFMIR_RESIZE_NO_ROWS", off)
            try:
                xg = x.clone().requires_grad_()
                y = ops.resize_linear(xg, out_sp, factor)
                (y * cot).sum().backward()
                torch.cuda.synchronize()
                res.append((y.detach(), xg.grad))
            finally:
                _lib.set_option("DFMIR_RESIZE_NO_ROWS", None)
        assert torch.equal(res[0][0], res[1][0]), "forward differs from the gather kernel (x%g)" % factor
        assert torch.equal(res[0][1], res[1][1]), "adjoint differs from the gather kernel (x%g)" % factor
        xr = x.cpu().double().requires_grad_()
        yr = factor * F.interpolate(xr, size=out_sp, mode=mode, align_corners=True)
        (yr * cot.cpu().double()).sum().backward()
        close(res[0][0], yr.detach().float(), rtol=1e-5, what="resize x%g vs F.interpolate" % factor)
        close(res[0][1], xr.grad.float(), rtol=1e-5, what="resize x%g adjoint vs autograd" % factor)


# ---------------------------------------------------------------------------------------- PatchNCE
def test_patch_gather_l2norm_nce(ops):
    from oracle import dfmir_oracle as O
    B, Cc, H, W, P = 2, 24, 9, 11, 32
    feat = C.randn(61, B, Cc, H, W)
    ids = C.patch_ids(3, 0, H * W, P)
    fr = feat.clone().requires_grad_()
    rows_r = fr.permute(0, 2, 3, 1).flatten(1, 2)[:, ids, :].flatten(0, 1)        # [B*P, C]
    qn_r = O.l2_normalize(rows_r)
    k = O.l2_normalize(C.randn(62, B * P, Cc))
    lr_ = O.patchnce_loss(qn_r, k, B, 0.07)
    cot = C.rand(63, B * P) + 0.5
    (lr_ * cot).sum().backward()
    fg = feat.clone().to(DEV).requires_grad_()
    rows_g = ops.patch_gather(fg, ids.to(DEV))                                      # [C, B*P]
    close(rows_g.t(), rows_r, what="gather")
    qn_g = ops.l2norm_rows(rows_g)
    close(qn_g.t(), qn_r, what="l2norm")
    lg = ops.patchnce_rows(qn_g, k.t().contiguous().to(DEV), B, 0.07)
    close(lg, lr_, what="nce loss")
    (lg * cot.to(DEV)).sum().backward()
    close(fg.grad, fr.grad, rtol=1e-3, what="dfeat")
    # all negatives from the minibatch (groups = 1)
    l1r = O.patchnce_loss(qn_r.detach(), k, 1, 0.07)
    l1g = ops.patchnce_rows(qn_g.detach(), k.t().contiguous().to(DEV), 1, 0.07)
    close(l1g, l1r, what="nce loss G=1")
    m = ops.mean(lg.detach())
    close(m, lr_.mean(), what="mean")


@pytest.mark.parametrize("Cc", [256, 128, 32])
def test_patchnce_matrix_core_form(ops, Cc):
    """R = 256 rows per image and C a multiple of 32 take the MFMA forms of the PatchNCE kernels
    (patchnce.hip *_mfma_k): loss and d(loss)/dq against the oracle's bmm + cross-entropy."""
    from oracle import dfmir_oracle as O
    B, P = 3, 256
    q = O.l2_normalize(C.randn(65, B * P, Cc)).requires_grad_()
    k = O.l2_normalize(C.randn(66, B * P, Cc))
    lr_ = O.patchnce_loss(q, k, B, 0.07)
    cot = C.rand(67, B * P) + 0.5
    (lr_ * cot).sum().backward()
    qg = q.detach().t().contiguous().to(DEV).requires_grad_()                       # [C, B*P]
    lg = ops.patchnce_rows(qg, k.t().contiguous().to(DEV), B, 0.07)
    close(lg, lr_.detach(), what="nce loss")
    (lg * cot.to(DEV)).sum().backward()
    close(qg.grad.t(), q.grad, rtol=3e-4, what="dq")


# ------------------------------------------------------------------------------------------ losses
def test_losses_golden(ops, golden):
    g = golden("losses.npz")
    a, b = C.image_pair(71, 2, 20, 24)
    a, b = a.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    l = ops.masked_l1(a, b, None, -0.95)
    l.backward()
    close(l, g["l1"], what="l1"); close(a.grad, g["dl1_a"], what="dl1a"); close(b.grad, g["dl1_b"], what="dl1b")
    mask = ((b > -0.95) | (a > -0.95)).detach()
    close(ops.masked_l1(a.detach(), b.detach(), mask), g["l1"], what="l1 explicit mask")
    z = ops.masked_l1(a.detach() * 0 - 1, b.detach() * 0 - 1, None, -0.95)
    assert float(z) == 0.0                                     # sum(mask)==0 -> 0 (registration_model.py:259)
    f2 = (C.randn(72, 2, 2, 18, 22) * 1.5).to(DEV).requires_grad_()
    l = ops.flow_smoothness(f2); l.backward()
    close(l, g["smooth2d"], what="smooth2d"); close(f2.grad, g["dsmooth2d"], what="dsmooth2d")
    f3 = (C.randn(73, 1, 3, 7, 9, 11) * 1.5).to(DEV).requires_grad_()
    l = ops.flow_smoothness(f3); l.backward()
    close(l, g["grad3d"], what="grad3d"); close(f3.grad, g["dgrad3d"], what="dgrad3d")
    for tag, shp in (("2d", (2, 1, 24, 28)), ("3d", (1, 1, 12, 14, 16))):
        I = C.rand(75, *shp)
        J = (0.6 * I + 0.4 * C.rand(76, *shp)).to(DEV)
        I = I.to(DEV).requires_grad_()
        l = ops.ncc_loss(I, J, 9, 1e-5); l.backward()
        close(l, g["ncc" + tag], what="ncc" + tag); close(I.grad, g["dncc" + tag], rtol=1e-3, what="dncc" + tag)


def test_edge_branches_golden(ops, golden):
    """Fixture E1-E6 (tests/golden/make_golden_edges.py, the reference's own outputs): NCC_Loss with a mask
    (util/losses.py:257-261) incl. the empty mask, vxm NCC(win).loss = -mean(cc) and vxm Grad('l1' | 'l2', loss_mult)
    (torchvoxelmorph/losses.py:7-67,93-117), Grad_Loss 'l1' / `mask=` (util/losses.py:81-130), PatchNCELoss over all negatives
    of the minibatch (patchnce.py:32-38) and PatchSampleF without the MLP (--netF sample, networks.py:280-281) -- through the
    mirror's classes (dfmir_amd.losses / voxelmorph.losses / patchnce / networks)."""
    from dfmir_amd import losses as L
    from dfmir_amd import networks as N
    from dfmir_amd import voxelmorph as V
    from dfmir_amd.options import default_options
    from dfmir_amd.patchnce import PatchNCELoss
    g = golden("edges.npz")
    d = {k: v.to(DEV) for k, v in C.edge_inputs().items()}
    for tag, kv in (("2d", [9, 9]), ("3d", [9, 9, 9])):
        I = d["I" + tag].clone().requires_grad_()
        l = L.NCC_Loss(DEV, kernel_var=kv, kernel_type='mean')(I, d["J" + tag], mask=d["mask" + tag])
        l.backward()
        close(l, g["ncc_masked_" + tag], what="masked ncc " + tag)
        close(I.grad, g["dncc_masked_" + tag], rtol=1e-3, what="d masked ncc " + tag)
    I = d["I2d"].clone().requires_grad_()
    l = L.NCC_Loss(DEV, kernel_var=[9, 9])(I, d["J2d"], mask=torch.zeros_like(I, dtype=torch.bool))
    l.backward()
    assert float(l) == float(g["ncc_empty_mask"]) == 0.0 and float(I.grad.abs().max()) == 0.0
    assert V.losses.NCC is L.NCC and V.losses.Grad is L.Grad
    for tag, win in (("2d", [5, 5]), ("3d", None)):
        yp = d["I" + tag].clone().requires_grad_()
        l = V.losses.NCC(win).loss(d["J" + tag], yp)
        l.backward()
        close(l, g["vxm_ncc_" + tag], what="vxm ncc " + tag)
        close(yp.grad, g["dvxm_ncc_" + tag], rtol=1e-3, what="d vxm ncc " + tag)
    for tag, pen, mult in (("l1", 'l1', None), ("l2m", 'l2', 2.5)):
        f = d["field3"].clone().requires_grad_()
        l = V.losses.Grad(pen, loss_mult=mult).loss(None, f)
        l.backward()
        close(l, g["vxm_grad_" + tag], what="vxm grad " + tag); close(f.grad, g["dvxm_grad_" + tag], what="d vxm grad " + tag)
    with pytest.raises(IndexError):
        V.losses.Grad('l1').loss(None, d["field2"])           # the reference indexes five axes
    f = d["field3"].clone().requires_grad_()
    l = L.Grad_Loss(dim=3, penalty='l1')(f); l.backward()
    close(l, g["grad3d_l1"], what="grad3d l1"); close(f.grad, g["dgrad3d_l1"], what="d grad3d l1")
    for key, pen, mult in (("grad2d_l1_masked", 'l1', 0.5), ("grad2d_l2_masked", 'l2', None)):
        f = d["field2"].clone().requires_grad_()
        l = L.Grad_Loss(dim=2, penalty=pen, loss_mult=mult)(f, mask=d["fmask2"]); l.backward()
        close(l, g[key], what=key); close(f.grad, g["d" + key], what="d " + key)
    # --netF sample + all negatives of the minibatch
    feats = [f.to(DEV) for f in C.edge_sample_feats()]
    hpf = N.define_F(1, 'sample', 'instance', False, 'xavier', 0.02, False, [0], default_options(netF_nc=32))
    assert len(list(hpf.parameters())) == 0 and not hpf.use_mlp
    ids = [C.patch_ids(40, i, f.shape[2] * f.shape[3], 48).to(DEV) for i, f in enumerate(feats)]
    fq = [f.clone().requires_grad_() for f in feats]
    fk = [C.randn(165 + i, *f.shape).to(DEV) for i, f in enumerate(feats)]
    kpool, _ = hpf(fk, 48, ids)
    qpool, _ = hpf(fq, 48, ids)
    for name, allneg in (("all", True), ("own", False)):
        crit = PatchNCELoss(default_options(batch_size=2, nce_includes_all_negatives_from_minibatch=allneg))
        for f in fq:
            f.grad = None
        tot = 0
        for i, (q, k) in enumerate(zip(qpool, kpool)):
            l = crit(q, k)
            close(l, g["sample_%s_loss%d" % (name, i)], what="%s-negatives loss %d" % (name, i))
            close(q, g["sample_q%d" % i], what="sampled q%d" % i)
            tot = tot + ops.mean(l)
        tot.backward(retain_graph=True)
        for i, f in enumerate(fq):
            # (layer 0 has ONE channel: x / (|x| + 1e-7) is +-1 and its true gradient 0 -- 4e-5 of round-off on both sides)
            close(f.grad, g["sample_%s_dfeat%d" % (name, i)], rtol=1e-3, atol=1e-5 if i == 0 else 1e-6,
                  what="%s-negatives dfeat%d" % (name, i))


@pytest.mark.parametrize("shape", [(1, 3, 16, 24, 32), (2, 3, 9, 21, 64), (1, 3, 20, 40, 224), (1, 3, 33, 8, 40), (1, 2, 50, 19, 12)],
                         ids=["16x24x32", "batch2-9x21x64-ragged", "20x40x224", "33x8x40-segments", "50x19x12"])
def test_flow_smoothness_backward_marching_kernel(ops, shape):
    """Grad_Loss l2 backward (util/losses.py:81-130) on flow_smooth_bwd_march_k -- a strip of 8 rows marched along z, z
    neighbours from registers, y neighbours through LDS -- against the five-loads-per-row kernel (DFMIR_SMOOTH_NO_MARCH):
    BIT-identical (the same expression), ragged strips, several z segments, batch; and against autograd."""
    from dfmir_amd._lib import set_option
    flow = C.randn(341, *shape)
    out = []
    try:
        for off in (None, "1"):
            set_option("DFMIR_SMOOTH_NO_MARCH", off)
            fg = flow.clone().to(DEV).requires_grad_()
            (ops.flow_smoothness(fg, 'l2') * 3.0).backward()
            out.append(fg.grad.clone())
    finally:
        set_option("DFMIR_SMOOTH_NO_MARCH", None)
    assert torch.equal(out[0], out[1]), "marching kernel differs from flow_smooth_bwd_v4_k"
    fr = flow.double().requires_grad_()
    dz, dy, dx = fr[:, :, 1:] - fr[:, :, :-1], fr[:, :, :, 1:] - fr[:, :, :, :-1], fr[..., 1:] - fr[..., :-1]
    (3.0 * ((dz ** 2).mean() + (dy ** 2).mean() + (dx ** 2).mean()) / 3.0).backward()
    close(out[0], fr.grad.float(), rtol=1e-5, what="d smoothness vs autograd")


def test_ncc_fused_box_passes_match_the_separate_ones(ops):
    """NCC[9,9,9] (torchvoxelmorph/losses.py NCC: 5 box-filtered product fields forward, 3 gradient fields backward): the
    launches that fuse the W and H box passes (with the products / with the field evaluation) through one LDS tile against
    the separate passes (DFMIR_NCC_NO_WH_FUSE=1, the form the golden vectors pin) on a volume of several ragged tiles and a
    batch: loss and gradient agree to fp32 summation order."""
    from dfmir_amd._lib import set_option
    I = C.rand(171, 2, 1, 6, 70, 150)
    J = (0.6 * I + 0.4 * C.rand(172, 2, 1, 6, 70, 150)).to(DEV)
    res = []
    try:
        for off in (None, "1"):
            set_option("DFMIR_NCC_NO_WH_FUSE", off)
            Ig = I.clone().to(DEV).requires_grad_()
            l = ops.ncc_loss(Ig, J, 9, 1e-5)
            l.backward()
            res.append((l.detach().clone(), Ig.grad.clone()))
    finally:
        set_option("DFMIR_NCC_NO_WH_FUSE", None)
    close(res[0][0], res[1][0], rtol=1e-6, what="ncc fused vs separate")
    close(res[0][1], res[1][1], rtol=2e-5, what="dncc fused vs separate")
    # the backward's D pass + combination in one launch (ncc_boxd_combine_k): the same ring and order of additions as
    # box_axis_march_k, the same expression as ncc_combine_k -- BIT-identical to the two launches, depth of one segment,
    # of several segments (D = 6, 80) and with a mask
    for shape, masked in (((2, 1, 6, 70, 150), False), ((1, 1, 80, 20, 24), False), ((1, 1, 40, 12, 20), True)):
        I2 = C.rand(173, *shape)
        J2 = (0.6 * I2 + 0.4 * C.rand(174, *shape)).to(DEV)
        mask = (C.rand(175, *shape) > 0.3).float().to(DEV) if masked else None
        out = []
        try:
            for off in (None, "1"):
                set_option("DFMIR_NCC_NO_D_FUSE", off)
                Ig = I2.clone().to(DEV).requires_grad_()
                l = ops.ncc_loss(Ig, J2, 9, 1e-5, mask=mask) if masked else ops.ncc_loss(Ig, J2, 9, 1e-5)
                l.backward()
                out.append(Ig.grad.clone())
        finally:
            set_option("DFMIR_NCC_NO_D_FUSE", None)
        assert torch.equal(out[0], out[1]), "fused D pass differs from box_axis_march_k + ncc_combine_k %s" % (shape,)


def test_adam_matches_torch(ops):
    from dfmir_amd.optim import FlatAdam
    ps = [C.randn(81, 37, 5), C.randn(82, 129)]
    ref = [p.clone().requires_grad_() for p in ps]
    got = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
    o_ref = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    o_got = FlatAdam(got, lr=2e-4, betas=(0.5, 0.999))
    for it in range(5):
        o_ref.zero_grad(); o_got.zero_grad()
        for i, (r, q) in enumerate(zip(ref, got)):
            gr = C.randn(90 + 10 * it + i, *r.shape) * (0.1 + it)
            r.grad = gr.clone()
            q.grad.add_(gr.to(DEV))
        o_ref.step(); o_got.step()
    for r, q in zip(ref, got):
        close(q, r, rtol=1e-6, what="adam params")


def test_deferred_weight_grads_batch_flush(ops):
    """Weight gradients accumulated in tap-major buffers and flushed by ONE batched launch at the end of backward
    (ops.deferred_weight_grads) equal the per-layer path, accumulate into existing .grad, and leave the accumulators
    cleared for the next step."""
    from dfmir_amd import networks as N
    torch.manual_seed(5)
    convs = [N.Conv2d(6, 10, 3, padding=1).to(DEV), N.Conv2d(10, 4, 1).to(DEV), N.Conv2d(4, 7, 3, padding=1).to(DEV)]
    x = C.randn(71, 2, 6, 9, 12).to(DEV)
    cot = C.randn(72, 2, 7, 9, 12).to(DEV)

    def run(deferred, steps):
        for c in convs:
            c.weight.grad = torch.full_like(c.weight, 0.25)     # existing gradient: the flush must add to it
            c.bias.grad = torch.zeros_like(c.bias)
        for _ in range(steps):
            h = x
            for c in convs:
                h = c(h)
            if deferred:
                with ops.deferred_weight_grads():
                    (h * cot).sum().backward()
            else:
                (h * cot).sum().backward()
        return [c.weight.grad.clone() for c in convs]

    ref = run(False, 2)
    got = run(True, 2)
    for g, r in zip(got, ref):
        close(g, r, rtol=1e-5, what="deferred dW")
    for c in convs:
        assert float(c._dw_tcc.abs().max()) == 0.0


def test_shared_tap_sparse_gradient(ops):
    """A tapped feature that also feeds the next layer (ops.fork_tap): the sampled rows' gradient is scattered into
    the other consumer's gradient in place instead of being returned densely and summed by autograd; same result,
    also when the other consumer yields no gradient or the rows are unused."""
    x0 = C.randn(81, 2, 6, 8, 8).to(DEV)
    ids = torch.tensor([3, 17, 40, 63, 5], device=DEV)
    w = C.randn(82, 2, 6, 8, 8).to(DEV)
    cot = C.randn(83, 6, 2 * 5).to(DEV)

    def run(shared, use_main, use_rows):
        x = x0.clone().requires_grad_()
        t = x * 2.0                                   # the "feature": non-leaf
        main, tap = ops.fork_tap(t) if shared else (t, t)
        rows = ops.patch_gather(tap, ids)
        loss = (x * 0.0).sum()
        if use_main:
            loss = loss + (main * w).sum()
        if use_rows:
            loss = loss + (rows * cot).sum()
        loss.backward()
        return x.grad

    for use_main, use_rows in ((True, True), (False, True), (True, False)):
        close(run(True, use_main, use_rows), run(False, use_main, use_rows), rtol=1e-6,
              what="fork_tap grad main=%s rows=%s" % (use_main, use_rows))


@pytest.mark.parametrize("shape", [(2, 64, 16, 32), (1, 72, 12, 40), (1, 40, 8, 16), (1, 64, 16, 62), (2, 128, 64, 64)])
def test_reflect_conv_skip_gradient(ops, shape):
    """conv(reflection_pad(x)) whose input also leaves through the skip output (ResnetBlock): dx = dgrad + d skip.
    64 / 72 channels take the zero-padded dgrad with the residual in its epilogue + the ring kernel; 40 x (8 x 16) the
    ring with a separate add (tiles under-filled); 16 x 62: a width that is not a multiple of 4 (the shared-tile kernel's epilogue
    leaves its float4 form: element loads of the residual, element stores); 128 x 64 x 64: the step's own geometry; all against torch."""
    N, Cch, H, W = shape
    x = C.randn(91, *shape)
    w = C.randn(92, Cch, Cch, 3, 3) * 0.05
    b = C.randn(93, Cch) * 0.1
    c1, c2 = C.randn(94, *shape), C.randn(95, *shape)
    xr = x.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    yr = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode='reflect'), wr, b)
    ((yr * c1).sum() + (xr * c2).sum()).backward()
    xg = x.clone().to(DEV).requires_grad_()
    wg = w.clone().to(DEV).requires_grad_()
    yg, xs = ops.conv(xg, wg, b.to(DEV), None, 1, 1, 1, 0, 0.0, skip=True)
    ((yg * c1.to(DEV)).sum() + (xs * c2.to(DEV)).sum()).backward()
    close(yg, yr, what="y"); close(xg.grad, xr.grad, rtol=2e-4, what="dx with skip")
    close(wg.grad, wr.grad, rtol=1e-3, what="dw")


def test_instnorm_bwd_border_columns(ops):
    """dfmir_instnorm_bwd_cols leaves dx's first / last column in a compact buffer (consumed by the reflect ring)."""
    x = C.randn(101, 2, 3, 64, 64).to(DEV).requires_grad_()
    y = ops.instance_norm(x, relu=True)
    cot = C.randn(102, *y.shape).to(DEV)
    # run the backward function by hand to get at the tagged gradient tensor
    dx, = torch.autograd.grad(y, x, cot)
    from dfmir_amd.ops import InstNormFn
    xs = x.detach().clone().requires_grad_()
    ys = InstNormFn.apply(xs, None, True, 1e-5)
    g = ys.grad_fn.apply(cot)[0] if hasattr(ys.grad_fn, "apply") else None
    if g is not None and hasattr(g, "_df_cols"):
        cols = g._df_cols[0].view(2 * 3, 2, 64)
        assert torch.equal(cols[:, 0], g.reshape(6, 64, 64)[:, :, 0])
        assert torch.equal(cols[:, 1], g.reshape(6, 64, 64)[:, :, 63])
    close(g if g is not None else dx, dx, rtol=0, atol=0, what="same dx")


@pytest.mark.parametrize("cfg", [(2, 64, 20, 24, True), (1, 64, 33, 70, True), (3, 40, 16, 32, False), (1, 64, 256, 256, True)],
                         ids=["small", "ragged", "zero_pad_40", "full"])
def test_stem7_direct(ops, cfg):
    """nn.ReflectionPad2d(3) + Conv2d(1, C, 7) through the direct kernels (forward, dW, db) and the tap-stack adjoint
    (dx), against torch."""
    N, Cout, H, W, reflect = cfg
    x = C.randn(111, N, 1, H, W)
    w = C.randn(112, Cout, 1, 7, 7) / 7.0
    b = C.randn(113, Cout) * 0.1
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = torch_conv(xr, wr, br, 1, 3, reflect, 0, 2)
    cot = C.randn(114, *yr.shape)
    (yr * cot).sum().backward()
    xg, wg, bg = (t.clone().to(DEV).requires_grad_() for t in (x, w, b))
    yg = ops.conv_taps(xg, wg, bg, 3, 1 if reflect else 0)
    (yg * cot.to(DEV)).sum().backward()
    close(yg, yr, what="y")
    close(xg.grad, xr.grad, rtol=3e-4, what="dx")
    close(wg.grad, wr.grad, rtol=3e-4, atol=1e-4 * float(wr.grad.abs().max()), what="dw")
    close(bg.grad, br.grad, rtol=3e-4, atol=1e-4 * float(br.grad.abs().max()), what="db")


@pytest.mark.parametrize("case", ["3d_smooth", "3d_rough", "2d_self"])
def test_warp_backward_is_reproducible_and_matches_atomic_path(ops, case):
    """d(src) of the warp without device-scope atomics (owner-gather kernels, fixed-point LDS accumulation): two runs
    are bit-equal, and the result equals the atomic scatter path (DFMIR_WARP_ATOMIC semantics, ops._warp_bwd into a
    zeroed buffer) to float round-off -- on a smooth field (no voxel leaves its window), on a rough one (the slow-voxel
    list is exercised) and on the VecInt self-warp form (v + warp(v, v): own term and flow gradient into d(src))."""
    if case == "2d_self":
        shp, nd, C_ = (3, 64, 96), 2, 2
    else:
        shp, nd, C_ = (1, 24, 40, 64), 3, 2
    B, sp = shp[0], shp[1:]
    amp, cell = (6.0, 4) if case == "3d_rough" else (0.4, 8)      # smooth: no tap leaves its tile's window
    coarse = C.randn(131, B, nd, *[max(2, s_ // cell) for s_ in sp]) * amp
    flow = torch.nn.functional.interpolate(coarse, size=sp, mode="trilinear" if nd == 3 else "bilinear",
                                           align_corners=True).contiguous().to(DEV)
    self_warp = case == "2d_self"
    src = flow if self_warp else C.randn(132, B, C_, *sp).to(DEV)
    dout = C.randn(133, *src.shape).to(DEV)
    runs = []
    for _ in range(2):
        dflow = None if self_warp else torch.empty_like(flow)
        dsrc = ops._warp_bwd_dsrc(dout, src, flow, dflow, int(self_warp), int(self_warp))
        runs.append((dsrc.clone(), None if dflow is None else dflow.clone()))
    if case != "3d_rough":       # (listed slow voxels still go through float atomics)
        assert torch.equal(runs[0][0], runs[1][0]), "d(src) must be bit-reproducible"
    ref = torch.zeros_like(src)
    dflow_ref = None if self_warp else torch.empty_like(flow)
    ops._warp_bwd(dout, src, flow, ref, dflow_ref, int(self_warp), int(self_warp))
    # 32-bit fixed point, scale from the workgroup's SUM of |g| (<= 2048 max|g|): every contribution is rounded to
    # <= 2^-19 max|g|; the guaranteed bound for a cell fed by k <= 64 voxels is 64 * 2^-19 max|g|, the typical error
    # (k ~ 8, random rounding, scale from the mean rather than the max) is 100x below it
    gmax = float(dout.abs().max()) * (1.0 + (float(flow.abs().max()) if self_warp else 0.0))
    err = float((runs[0][0] - ref).abs().max())
    assert err <= 64 * 2.0 ** -19 * gmax, (err, gmax)
    close(runs[0][0], ref, rtol=1e-5, what="dsrc vs atomic path")
    if not self_warp:
        assert torch.equal(runs[0][1], runs[1][1])
        close(runs[0][1], dflow_ref, rtol=2e-6, what="dflow vs atomic path")


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_warp_backward_propagates_non_finite(ops, bad):
    """A NaN / inf in the incoming gradient must not be laundered into finite numbers by the fixed-point accumulation
    (grid_sample's backward and the atomic path propagate it): the owning tile's d(src) sub-box comes out non-finite,
    every other tile is untouched."""
    B, C_, sp = 1, 1, (16, 24, 64)
    coarse = C.randn(131, B, 3, 2, 3, 8) * 0.4
    flow = torch.nn.functional.interpolate(coarse, size=sp, mode="trilinear", align_corners=True).contiguous().to(DEV)
    src = C.randn(132, B, C_, *sp).to(DEV)
    dout = C.randn(133, B, C_, *sp).to(DEV)
    good = ops._warp_bwd_dsrc(dout, src, flow, torch.empty_like(flow), 0, 0)
    assert bool(torch.isfinite(good).all())
    dout[0, 0, 3, 5, 7] = bad
    got = ops._warp_bwd_dsrc(dout, src, flow, torch.empty_like(flow), 0, 0)
    assert not bool(torch.isfinite(got[0, 0, 2:5, 4:7, 6:9]).all()), "the non-finite gradient was lost"
    far = got[0, 0, :, :, 40:]                               # x-tiles 1.. are other workgroups
    assert torch.equal(far, good[0, 0, :, :, 40:])


@pytest.mark.parametrize("H", [256, 128])
def test_in_relu_blurdown_fused_matches_two_passes(ops, H):
    """InstanceNorm + ReLU + Downsample in one pass per plane == the two kernels it replaces (instance_norm(relu) then
    blur_down), forward and backward, and == torch (instance_norm / relu / reflect-pad + [1 2 1]^2/16 stride-2 conv,
    models/networks.py:37-60,984-996)."""
    x = (C.randn(141, 2, 3, H, H) * torch.tensor([0.5, 2.0, 1.0]).view(1, 3, 1, 1) + 0.3).to(DEV)
    cot = C.randn(142, 2, 3, H // 2, H // 2).to(DEV)
    xa = x.clone().requires_grad_()
    za = ops.instance_norm_relu_blur_down(xa)
    (za * cot).sum().backward()
    xb = x.clone().requires_grad_()
    zb = ops.blur_down(ops.instance_norm(xb, None, True))
    (zb * cot).sum().backward()
    close(za, zb, rtol=2e-6, what="z vs two passes")
    close(xa.grad, xb.grad, rtol=2e-5, what="dx vs two passes")
    xr = x.double().cpu().requires_grad_()
    y = torch.relu(F.instance_norm(xr, eps=1e-5))
    f = torch.tensor([1.0, 2.0, 1.0], dtype=torch.float64)
    k = (f[:, None] * f[None, :] / 16.0)[None, None].repeat(3, 1, 1, 1)
    zr = F.conv2d(F.pad(y, (1, 1, 1, 1), mode="reflect"), k, stride=2, groups=3)
    (zr * cot.double().cpu()).sum().backward()
    close(za, zr, rtol=1e-5, what="z vs torch")
    close(xa.grad, xr.grad, rtol=1e-4, what="dx vs torch")

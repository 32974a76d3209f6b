"""CPU: pin the oracle (oracle/dfmir_oracle.py) against the golden vectors captured from the
reference itself (tests/golden/make_golden.py).  Tolerance: 1e-4 relative with a 1e-6 absolute floor
 -- in practice the oracle reproduces the reference to ~1e-6."""
import numpy as np
import pytest
import torch

from oracle import dfmir_oracle as O
from tests.golden import common as C

RTOL, ATOL = 1e-4, 1e-6


def close(a, b, rtol=RTOL, atol=ATOL, what=""):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a - b).max())
    assert err <= atol + rtol * scale, "%s: max abs err %.3e (scale %.3e)" % (what, err, scale)


def test_warp(golden):
    g = golden("warp.npz")
    for tag, shp, C_ in (("2d", (17, 23), 3), ("3d", (9, 11, 13), 2)):
        B = 2 if tag == "2d" else 1
        src = C.randn(11, B, C_, *shp).requires_grad_()
        flow = ((C.rand(12, B, len(shp), *shp) * 12) - 6).requires_grad_()
        cot = C.randn(13, B, C_, *shp)
        y = O.spatial_transform(src, flow)
        (y * cot).sum().backward()
        close(y, g["out_" + tag], what="out " + tag)
        close(src.grad, g["dsrc_" + tag], what="dsrc " + tag)
        close(flow.grad, g["dflow_" + tag], what="dflow " + tag)
        close(O.spatial_transform(src.detach(), flow.detach(), 'nearest'), g["nearest_" + tag], what="nearest")


def test_vecint_resize(golden):
    g = golden("vecint_resize.npz")
    for tag, shp in (("2d", (32, 32)), ("3d", (8, 10, 12))):
        v = (C.randn(21, 2 if tag == "2d" else 1, len(shp), *shp) * 2.0).requires_grad_()
        cot = C.randn(22, *v.shape)
        y = O.vec_int(v, 7)
        (y * cot).sum().backward()
        close(y, g["vecint_" + tag], what="vecint")
        close(v.grad, g["dvecint_" + tag], rtol=1e-3, what="dvecint")
        x = C.randn(23, 1, len(shp), *shp).requires_grad_()
        half = O.resize_transform(x, 2)
        (half * C.randn(24, *half.shape)).sum().backward()
        close(half, g["half_" + tag]); close(x.grad, g["dhalf_" + tag])
        x2 = C.randn(25, 1, len(shp), *shp).requires_grad_()
        dbl = O.resize_transform(x2, 0.5)
        (dbl * C.randn(26, *dbl.shape)).sum().backward()
        close(dbl, g["double_" + tag]); close(x2.grad, g["ddouble_" + tag])


def test_blur(golden):
    g = golden("blur.npz")
    x = C.randn(31, 2, 8, 12, 12).requires_grad_()
    y = O.BlurDown(8)(x)
    (y * C.randn(32, *y.shape)).sum().backward()
    close(y, g["down"]); close(x.grad, g["ddown"])
    x = C.randn(33, 2, 8, 12, 12).requires_grad_()
    y = O.BlurUp(8)(x)
    (y * C.randn(34, *y.shape)).sum().backward()
    close(y, g["up"]); close(x.grad, g["dup"])
    xo = C.randn(35, 1, 3, 7, 9).requires_grad_()
    yo = O.BlurDown(3)(xo)
    (yo * C.randn(36, *yo.shape)).sum().backward()
    close(yo, g["down_odd"]); close(xo.grad, g["ddown_odd"])


def test_resblock_and_generator(golden):
    g = golden("resblock.npz")
    torch.manual_seed(41)
    ob = O.ResBlock(16)
    assert C.state_checksum(ob) == str(g["wsum"]), "seeded weights differ from the fixture's (torch RNG changed?)"
    x = C.randn(42, 2, 16, 10, 14).requires_grad_()
    y = ob(x)
    (y * C.randn(43, *y.shape)).sum().backward()
    close(y, g["out"]); close(x.grad, g["dx"])
    close(ob.conv_block[1].weight.grad, g["dw1"]); close(ob.conv_block[5].bias.grad, g["db5"])

    g = golden("generator.npz")
    og = make_tiny_generator()
    assert C.state_checksum(og) == str(g["wsum"])
    x = C.image_pair(52, 2, 64, 64)[0].requires_grad_()
    y, feats = og(x, [0, 4, 8, 12, 16], encode_only=False)
    fc = [C.randn(54 + i, *f.shape) for i, f in enumerate(feats)]
    ((y * C.randn(53, *y.shape)).sum() + sum((f * c).sum() for f, c in zip(feats, fc))).backward()
    close(y, g["out"], what="gen out")
    for i, f in enumerate(feats):
        close(f, g["feat%d" % i], what="feat%d" % i)
    close(x.grad, g["dx"], rtol=1e-3, what="gen dx")
    for k, p in og.named_parameters():
        ref = float(g["gnorm_" + k.replace(".", "_")])
        assert abs(float(p.grad.norm()) - ref) <= 1e-3 * max(ref, 1e-6), k
    enc = og(x.detach(), [0, 4, 8, 12, 16], encode_only=True)
    assert len(enc) == 5 and enc[4].shape == feats[4].shape


def make_tiny_generator():
    torch.manual_seed(51)
    og = O.Generator(1, 1, 8, 9)
    O.init_weights_xavier(og, 0.02)
    with torch.no_grad():
        for p in og.parameters():
            p.mul_(12.0)
    return og


def make_patch_sampler():
    torch.manual_seed(61)
    feats = [C.randn(62, 2, 1, 14, 14), C.randn(63, 2, 16, 12, 12), C.randn(64, 2, 32, 8, 8)]
    opf = O.PatchSampler(32, True)
    opf.create_mlp(feats)
    with torch.no_grad():
        for p in opf.parameters():
            p.mul_(20.0)
            if p.dim() == 1:
                p.add_(0.05)
    return opf, feats


def test_patchnce(golden):
    g = golden("patchnce.npz")
    opf, feats = make_patch_sampler()
    assert C.state_checksum(opf) == str(g["wsum"])
    ids = [C.patch_ids(0, i, f.shape[2] * f.shape[3], 48) for i, f in enumerate(feats)]
    fq = [f.clone().requires_grad_() for f in feats]
    fk = [C.randn(65 + i, *f.shape) for i, f in enumerate(feats)]
    kpool, _ = opf(fk, 48, ids)
    qpool, _ = opf(fq, 48, ids)
    tot = 0
    for i, (q, k) in enumerate(zip(qpool, kpool)):
        l = O.patchnce_loss(q, k, 2, 0.07)
        close(l, g["loss%d" % i], what="nce loss %d" % i)
        close(q, g["q%d" % i])
        tot = tot + l.mean()
    tot.backward()
    for i, f in enumerate(fq):
        close(f.grad, g["dfeat%d" % i], rtol=1e-3, what="dfeat%d" % i)
    for k_, p in opf.named_parameters():
        close(p.grad, g["dparam_" + k_.replace(".", "_")], rtol=1e-3, what=k_)


def test_losses(golden):
    g = golden("losses.npz")
    a, b = C.image_pair(71, 2, 20, 24)
    a.requires_grad_(); b.requires_grad_()
    l = O.masked_l1(a, b, (b > -0.95) + (a > -0.95))
    l.backward()
    close(l, g["l1"]); close(a.grad, g["dl1_a"]); close(b.grad, g["dl1_b"])
    f2 = (C.randn(72, 2, 2, 18, 22) * 1.5).requires_grad_()
    l = O.smoothing_loss(f2); l.backward()
    close(l, g["smooth2d"]); close(f2.grad, g["dsmooth2d"])
    f3 = (C.randn(73, 1, 3, 7, 9, 11) * 1.5).requires_grad_()
    l = O.grad_loss_l2(f3); l.backward()
    close(l, g["grad3d"]); close(f3.grad, g["dgrad3d"])
    f2b = C.randn(74, 2, 2, 18, 22).requires_grad_()
    l = O.grad_loss_l2(f2b); l.backward()
    close(l, g["grad2d"]); close(f2b.grad, g["dgrad2d"])
    for tag, shp in (("2d", (2, 1, 24, 28)), ("3d", (1, 1, 12, 14, 16))):
        I = C.rand(75, *shp).requires_grad_()
        J = 0.6 * I.detach() + 0.4 * C.rand(76, *shp)
        l = O.ncc_loss(I, J, 9); l.backward()
        close(l, g["ncc" + tag]); close(I.grad, g["dncc" + tag], rtol=1e-3)


def test_edges(golden):
    """Fixture E1-E6 (tests/golden/make_golden_edges.py): the masked NCC_Loss, vxm NCC / Grad, Grad_Loss l1 / masked,
    PatchNCELoss over all negatives of the minibatch and PatchSampleF without the MLP -- the oracle against the
    reference's own outputs."""
    g = golden("edges.npz")
    d = C.edge_inputs()
    for tag in ("2d", "3d"):
        I = d["I" + tag].clone().requires_grad_()
        l = O.ncc_loss(I, d["J" + tag], 9, mask=d["mask" + tag]); l.backward()
        close(l, g["ncc_masked_" + tag]); close(I.grad, g["dncc_masked_" + tag], rtol=1e-3)
    assert float(O.ncc_loss(d["I2d"], d["J2d"], 9, mask=torch.zeros_like(d["I2d"], dtype=torch.bool))) == float(g["ncc_empty_mask"]) == 0.0
    for tag, win in (("2d", 5), ("3d", 9)):
        yp = d["I" + tag].clone().requires_grad_()
        l = O.vxm_ncc_loss(d["J" + tag], yp, win); l.backward()
        close(l, g["vxm_ncc_" + tag]); close(yp.grad, g["dvxm_ncc_" + tag], rtol=1e-3)
    for tag, pen, mult in (("l1", 'l1', None), ("l2m", 'l2', 2.5)):
        f = d["field3"].clone().requires_grad_()
        l = O.grad_loss(f, pen, loss_mult=mult); l.backward()
        close(l, g["vxm_grad_" + tag]); close(f.grad, g["dvxm_grad_" + tag])
    f = d["field3"].clone().requires_grad_()
    l = O.grad_loss(f, 'l1'); l.backward()
    close(l, g["grad3d_l1"]); close(f.grad, g["dgrad3d_l1"])
    for key, pen, mult in (("grad2d_l1_masked", 'l1', 0.5), ("grad2d_l2_masked", 'l2', None)):
        f = d["field2"].clone().requires_grad_()
        l = O.grad_loss(f, pen, mask=d["fmask2"], loss_mult=mult); l.backward()
        close(l, g[key]); close(f.grad, g["d" + key])
    feats = C.edge_sample_feats()
    opf = O.PatchSampler(32, False)
    assert len(list(opf.parameters())) == 0
    ids = [C.patch_ids(40, i, f.shape[2] * f.shape[3], 48) for i, f in enumerate(feats)]
    fq = [f.clone().requires_grad_() for f in feats]
    fk = [C.randn(165 + i, *f.shape) for i, f in enumerate(feats)]
    kpool, _ = opf(fk, 48, ids)
    qpool, _ = opf(fq, 48, ids)
    for name, allneg in (("all", True), ("own", False)):
        for f in fq:
            f.grad = None
        tot = 0
        for i, (q, k) in enumerate(zip(qpool, kpool)):
            l = O.patchnce_loss(q, k, 2, 0.07, all_negatives=allneg)
            close(l, g["sample_%s_loss%d" % (name, i)], what="%s loss %d" % (name, i))
            close(q, g["sample_q%d" % i])
            tot = tot + l.mean()
        tot.backward(retain_graph=True)
        for i, f in enumerate(fq):
            close(f.grad, g["sample_%s_dfeat%d" % (name, i)], rtol=1e-3, what="%s dfeat%d" % (name, i))
    assert str(g["netF_sample_optimizer_error"]) == "ValueError: optimizer got an empty parameter list"


def make_vxm(tag):
    shp, feats_ = ((64, 64), O.PLUGIN_UNET_FEATURES) if tag == "2d" else ((32, 32, 32), None)
    torch.manual_seed(81)
    ov = O.VxmDense(shp, feats_, 7, True)
    with torch.no_grad():
        ov.flow.weight.mul_(1e5)
        ov.flow.bias.copy_(C.randn(82, *ov.flow.bias.shape) * 2.0)
    return ov, shp


@pytest.mark.parametrize("tag", ["2d", "3d"])
def test_vxm(golden, tag):
    g = golden("vxm.npz")
    ov, shp = make_vxm(tag)
    assert C.state_checksum(ov) == str(g["wsum_" + tag])
    B = 2 if tag == "2d" else 1
    s_ = C.rand(83, B, 1, *shp).requires_grad_()
    t_ = C.rand(84, B, 1, *shp)
    ys, yt, fl = ov(s_, t_)
    ((ys * C.randn(85, *ys.shape)).sum() + (fl * C.randn(86, *fl.shape) * 0.1).sum()).backward()
    close(ys, g["ys_" + tag]); close(yt, g["yt_" + tag]); close(fl, g["flow_" + tag])
    close(s_.grad, g["dsrc_" + tag], rtol=1e-3)
    close(ov.flow.weight.grad, g["gflow_w_" + tag], rtol=1e-3)
    close(ov.unet_model.downarm[0].main.weight.grad, g["gdown0_w_" + tag], rtol=1e-3)
    close(ov.unet_model.uparm[1].main.weight.grad, g["gup1_w_" + tag], rtol=1e-3)
    y2, _ = ov(s_.detach(), t_, registration=True)
    close(y2, g["reg_ys_" + tag])
    assert float(np.abs(g["flow_" + tag]).max()) > 0.5, "fixture flow must be non-trivial"


def make_step():
    size, B, ngf = 64, 2, 8
    torch.manual_seed(91)
    st = O.RegistrationStep(size, B, ngf=ngf)
    with torch.no_grad():
        st.netR.flow.weight.mul_(1e5)
        st.netR.flow.bias.copy_(C.randn(92, 2) * 1.0)
    st.ids_hook = lambda c, feats: [C.patch_ids(c, i, f.shape[2] * f.shape[3], 256) for i, f in enumerate(feats)]
    A0, B0 = C.image_pair(93, B, size, size)
    st.data_dependent_initialize(A0, B0)
    with torch.no_grad():  # see make_golden.py: avoids the reference's 0*inf NaN on exactly-zero pixels
        for p in st.netF.parameters():
            if p.dim() == 1:
                p.add_(0.01)
    return st, size, B


def test_whole_step(golden):
    g = golden("step.npz")
    st, size, B = make_step()
    assert C.multi_state_checksum((st.netG, st.netF, st.netR)) == str(g["wsum"])
    for it in range(3):
        A_, B_ = C.image_pair(100 + 2 * it, B, size, size)
        ls = st.step(A_, B_)
        got = np.array([ls[k] for k in ("G", "NCE", "R", "smooth", "local", "NCE_Y")])
        assert np.isfinite(g["losses_%d" % it]).all()
        np.testing.assert_allclose(got, g["losses_%d" % it], rtol=2e-4 * (it + 1), atol=1e-6)
        if it == 0:
            close(st.fake_B, g["fake_B"]); close(st.registered, g["registered"]); close(st.regA, g["regA"])
            close(st.flow, g["pos_flow"])                     # the deformation field of the reference's own netR call
            close(st.idt_B, g["idt_B"])
            for nm, net in (("G", st.netG), ("F", st.netF), ("R", st.netR)):
                n2 = sum(float((p.grad.double() ** 2).sum()) for p in net.parameters() if p.grad is not None) ** 0.5
                assert abs(n2 - float(g["gradnorm_" + nm])) <= 1e-3 * float(g["gradnorm_" + nm]), nm


def test_patch_gather_formulation_and_torch_cpu_instance_norm_bug():
    """The oracle gathers patches as f.flatten(2)[:, :, pid] (contiguous gradient).  (a) it equals the
    reference's permute/flatten/index formulation in value and, at batch 2, in gradient; (b) at batch 1
    the reference's formulation hands InstanceNorm a channels-last grad_output, for which torch 2.10's
    CPU backward returns wrong values -- the oracle's gradient instead equals the dense-
    cotangent gradient, which is formulation-independent."""
    import torch.nn.functional as F
    for B in (1, 2):
        x = C.randn(300 + B, B, 5, 6, 7).requires_grad_()
        pid = C.patch_ids(0, 0, 42, 20)
        cot = C.randn(310 + B, B * 20, 5)
        y = F.instance_norm(x)
        rows_ref = y.permute(0, 2, 3, 1).flatten(1, 2)[:, pid, :].flatten(0, 1)
        rows_orc = y.flatten(2)[:, :, pid].permute(0, 2, 1).flatten(0, 1)
        assert torch.equal(rows_ref, rows_orc)
        g_orc, = torch.autograd.grad((rows_orc * cot).sum(), x, retain_graph=True)
        dense = torch.zeros(B, 5, 42)
        dense[:, :, pid] = cot.view(B, 20, 5).permute(0, 2, 1)
        g_dense, = torch.autograd.grad((y * dense.view_as(y)).sum(), x, retain_graph=True)
        close(g_orc, g_dense, what="oracle vs dense, B=%d" % B)
        if B == 2:
            g_ref, = torch.autograd.grad((rows_ref * cot).sum(), x)
            close(g_ref, g_orc, what="reference formulation, B=2")

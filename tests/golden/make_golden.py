"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference; never on the GPU box).  The reference is
imported in place with four shims: a torchvision stub, `.cuda()` -> identity,
cwd = /root/reference for ./deform256.jpg, and the `dvf` batch fix for batch_size > 1.  Only
inputs-by-seed and expected OUTPUTS are stored -- no reference source, no reference weights (weights
come from torch.manual_seed through the oracle's constructors and are guarded by a checksum).

    python tests/golden/make_golden.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from tests.golden import common as C  # noqa: E402


# ------------------------------------------------------------------------------------------ shims
def install_shims():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class Compose(object):
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class CenterCrop(object):
        def __init__(self, size):
            self.size = size

        def __call__(self, im):
            w, h = im.size
            l, t = int(round((w - self.size) / 2.0)), int(round((h - self.size) / 2.0))
            return im.crop((l, t, l + self.size, t + self.size))

    class ToTensor(object):
        def __call__(self, im):
            a = np.asarray(im, dtype=np.float32) / 255.0
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(a).permute(2, 0, 1).contiguous()

    class Normalize(object):
        def __init__(self, mean, std):
            self.mean, self.std = mean[0], std[0]

        def __call__(self, t):
            return (t - self.mean) / self.std

    class _Named(object):
        def __init__(self, *a, **k):
            pass

    # The transforms data/base_dataset.py:82-131 composes, with torchvision's documented semantics on PIL images
    # (torchvision itself is not installed): Grayscale = convert('L'); Resize((h, w), method) = PIL resize;
    # RandomCrop draws the top row then the left column with torch.randint (none when the sizes already match);
    # RandomHorizontalFlip flips when torch.rand(1) < p.
    class Grayscale(object):
        def __init__(self, num_output_channels=1):
            assert num_output_channels == 1

        def __call__(self, im):
            return im.convert('L')

    class Resize(object):
        def __init__(self, size, interpolation=2):
            self.size, self.method = size, interpolation

        def __call__(self, im):
            h, w = self.size
            return im.resize((w, h), self.method)

    class RandomCrop(object):
        def __init__(self, size):
            self.size = (size, size) if isinstance(size, int) else tuple(size)

        def __call__(self, im):
            w, h = im.size
            th, tw = self.size
            if (w, h) == (tw, th):
                return im
            i = int(torch.randint(0, h - th + 1, size=(1,)).item())
            j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
            return im.crop((j, i, j + tw, i + th))

    class RandomHorizontalFlip(object):
        def __init__(self, p=0.5):
            self.p = p

        def __call__(self, im):
            from PIL import Image
            return im.transpose(Image.FLIP_LEFT_RIGHT) if float(torch.rand(1)) < self.p else im

    class Lambda(object):
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, im):
            return self.fn(im)

    tr.Compose, tr.CenterCrop, tr.ToTensor, tr.Normalize = Compose, CenterCrop, ToTensor, Normalize
    tr.Grayscale, tr.Resize, tr.RandomCrop, tr.RandomHorizontalFlip, tr.Lambda = \
        Grayscale, Resize, RandomCrop, RandomHorizontalFlip, Lambda
    tr.InterpolationMode = _Named
    tv.transforms = tr
    tv.models = types.ModuleType("torchvision.models")
    tv.utils = types.ModuleType("torchvision.utils")
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.models": tv.models,
                        "torchvision.utils": tv.utils})
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    os.chdir(REF)


def ref_options(size, batch, ngf):
    tmp = tempfile.mkdtemp(prefix="dfmir_golden_")
    sys.argv = ["train.py", "--dataroot", tmp, "--gpu_ids", "-1", "--checkpoints_dir", tmp, "--name", "g",
                "--batch_size", str(batch), "--crop_size", str(size), "--load_size", str(size), "--ngf", str(ngf),
                "--CUT_mode", "CUT", "--no_flip"]
    from options.train_options import TrainOptions
    return TrainOptions().parse()


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote %-22s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


# ------------------------------------------------------------------------------------------ main
def main():
    install_shims()
    from oracle import dfmir_oracle as O
    import models.networks as RN
    from models.patchnce import PatchNCELoss as RefNCE
    from models.voxelmorph.torchvoxelmorph import layers as RL
    from models.voxelmorph.torchvoxelmorph import networks as RV
    from util.losses import Grad_Loss as RefGrad
    from util.losses import NCC_Loss as RefNCC
    import models.registration_model as RM
    import argparse

    # ---- W1/W2 SpatialTransformer 2-D / 3-D: out + grads for a fixed cotangent
    out = {}
    for tag, shp, C_ in (("2d", (17, 23), 3), ("3d", (9, 11, 13), 2)):
        B = 2 if tag == "2d" else 1
        src = C.randn(11, B, C_, *shp).requires_grad_()
        flow = ((C.rand(12, B, len(shp), *shp) * 12) - 6).requires_grad_()
        cot = C.randn(13, B, C_, *shp)
        y = RL.SpatialTransformer(shp)(src, flow)
        (y * cot).sum().backward()
        out.update({"out_" + tag: npy(y), "dsrc_" + tag: npy(src.grad), "dflow_" + tag: npy(flow.grad)})
        yn = RL.SpatialTransformer(shp, mode='nearest')(src.detach(), flow.detach())
        out["nearest_" + tag] = npy(yn)
    save("warp.npz", **out)

    # ---- W3 VecInt, W4 ResizeTransform
    out = {}
    for tag, shp in (("2d", (32, 32)), ("3d", (8, 10, 12))):
        v = (C.randn(21, 2 if tag == "2d" else 1, len(shp), *shp) * 2.0).requires_grad_()
        cot = C.randn(22, *v.shape)
        y = RL.VecInt(shp, 7)(v)
        (y * cot).sum().backward()
        out.update({"vecint_" + tag: npy(y), "dvecint_" + tag: npy(v.grad)})
        x = C.randn(23, 1, len(shp), *shp).requires_grad_()
        half = RL.ResizeTransform(2, len(shp))(x)
        cot = C.randn(24, *half.shape)
        (half * cot).sum().backward()
        out.update({"half_" + tag: npy(half), "dhalf_" + tag: npy(x.grad)})
        x2 = C.randn(25, 1, len(shp), *shp).requires_grad_()
        dbl = RL.ResizeTransform(0.5, len(shp))(x2)
        cot = C.randn(26, *dbl.shape)
        (dbl * cot).sum().backward()
        out.update({"double_" + tag: npy(dbl), "ddouble_" + tag: npy(x2.grad)})
    save("vecint_resize.npz", **out)

    # ---- G1 Downsample / Upsample
    out = {}
    x = C.randn(31, 2, 8, 12, 12).requires_grad_()
    y = RN.Downsample(8)(x)
    cot = C.randn(32, *y.shape)
    (y * cot).sum().backward()
    out.update(down=npy(y), ddown=npy(x.grad))
    x = C.randn(33, 2, 8, 12, 12).requires_grad_()
    y = RN.Upsample(8)(x)
    cot = C.randn(34, *y.shape)
    (y * cot).sum().backward()
    out.update(up=npy(y), dup=npy(x.grad))
    xo = C.randn(35, 1, 3, 7, 9).requires_grad_()           # odd sizes
    yo = RN.Downsample(3)(xo)
    cot = C.randn(36, *yo.shape)
    (yo * cot).sum().backward()
    out.update(down_odd=npy(yo), ddown_odd=npy(xo.grad))
    save("blur.npz", **out)

    # ---- G2 ResnetBlock(16), G3 tiny ResnetGenerator (weights by seed through the oracle)
    norm = RN.get_norm_layer('instance')
    torch.manual_seed(41)
    ob = O.ResBlock(16)
    rb = RN.ResnetBlock(16, 'reflect', norm, False, True)
    rb.load_state_dict(ob.state_dict())
    x = C.randn(42, 2, 16, 10, 14).requires_grad_()
    y = rb(x)
    cot = C.randn(43, *y.shape)
    (y * cot).sum().backward()
    save("resblock.npz", wsum=np.array(C.state_checksum(ob)), out=npy(y), dx=npy(x.grad),
         dw1=npy(rb.conv_block[1].weight.grad), db1=npy(rb.conv_block[1].bias.grad),
         dw5=npy(rb.conv_block[5].weight.grad), db5=npy(rb.conv_block[5].bias.grad))

    torch.manual_seed(51)
    og = O.Generator(1, 1, 8, 9)
    O.init_weights_xavier(og, 0.02)
    with torch.no_grad():                     # larger weights so that activations are not ~0
        for p in og.parameters():
            p.mul_(12.0)
    rg = RN.ResnetGenerator(1, 1, 8, norm_layer=norm, use_dropout=False, n_blocks=9)
    rg.load_state_dict(og.state_dict())
    x = C.image_pair(52, 2, 64, 64)[0].requires_grad_()
    y, feats = rg(x, [0, 4, 8, 12, 16], encode_only=False)
    cot = C.randn(53, *y.shape)
    fc = [C.randn(54 + i, *f.shape) for i, f in enumerate(feats)]
    ((y * cot).sum() + sum((f * c).sum() for f, c in zip(feats, fc))).backward()
    gn = {("gnorm_" + k.replace(".", "_")): np.array(float(p.grad.norm())) for k, p in rg.named_parameters()}
    save("generator.npz", wsum=np.array(C.state_checksum(og)), out=npy(y), dx=npy(x.grad),
         **{"feat%d" % i: npy(f) for i, f in enumerate(feats)}, **gn)

    # ---- F1 PatchSampleF + PatchNCELoss
    torch.manual_seed(61)
    feats = [C.randn(62, 2, 1, 14, 14), C.randn(63, 2, 16, 12, 12), C.randn(64, 2, 32, 8, 8)]
    opf = O.PatchSampler(32, True)
    opf.create_mlp(feats)
    with torch.no_grad():
        for p in opf.parameters():
            p.mul_(20.0)
            if p.dim() == 1:
                p.add_(0.05)
    rpf = RN.PatchSampleF(use_mlp=True, init_type='xavier', init_gain=0.02, nc=32, gpu_ids=[])
    rpf.create_mlp(feats)
    rpf.load_state_dict(opf.state_dict())
    ids = [C.patch_ids(0, i, f.shape[2] * f.shape[3], 48) for i, f in enumerate(feats)]
    fq = [f.clone().requires_grad_() for f in feats]
    fk = [C.randn(65 + i, *f.shape) for i, f in enumerate(feats)]
    kpool, _ = rpf(fk, 48, ids)
    qpool, _ = rpf(fq, 48, ids)
    nopt = argparse.Namespace(nce_includes_all_negatives_from_minibatch=False, batch_size=2, nce_T=0.07)
    crit = RefNCE(nopt)
    tot = 0
    out = dict(wsum=np.array(C.state_checksum(opf)))
    for i, (q, k) in enumerate(zip(qpool, kpool)):
        l = crit(q, k)
        out["loss%d" % i] = npy(l)
        out["q%d" % i] = npy(q)
        tot = tot + l.mean()
    tot.backward()
    for i, f in enumerate(fq):
        out["dfeat%d" % i] = npy(f.grad)
    for k_, p in rpf.named_parameters():
        out["dparam_" + k_.replace(".", "_")] = npy(p.grad)
    save("patchnce.npz", **out)

    # ---- L1 losses
    out = {}
    a, b = C.image_pair(71, 2, 20, 24)
    a = a.requires_grad_()
    b = b.requires_grad_()
    mask = (b > -0.95) + (a > -0.95)
    dummy = types.SimpleNamespace()
    l = RM.REGISTRATIONModel.calculate_L1_loss(dummy, a, b, mask)
    l.backward()
    out.update(l1=npy(l), dl1_a=npy(a.grad), dl1_b=npy(b.grad))
    f2 = (C.randn(72, 2, 2, 18, 22) * 1.5).requires_grad_()
    l = RM.smooothing_loss(f2)
    l.backward()
    out.update(smooth2d=npy(l), dsmooth2d=npy(f2.grad))
    f3 = (C.randn(73, 1, 3, 7, 9, 11) * 1.5).requires_grad_()
    l = RefGrad(dim=3, penalty='l2')(f3)
    l.backward()
    out.update(grad3d=npy(l), dgrad3d=npy(f3.grad))
    f2b = (C.randn(74, 2, 2, 18, 22)).requires_grad_()
    l = RefGrad(dim=2, penalty='l2')(f2b)
    l.backward()
    out.update(grad2d=npy(l), dgrad2d=npy(f2b.grad))
    for tag, shp, kv in (("2d", (2, 1, 24, 28), [9, 9]), ("3d", (1, 1, 12, 14, 16), [9, 9, 9])):
        I = C.rand(75, *shp).requires_grad_()
        J = (0.6 * I.detach() + 0.4 * C.rand(76, *shp))
        l = RefNCC('cpu', kernel_var=kv, kernel_type='mean')(I, J)
        l.backward()
        out.update({"ncc" + tag: npy(l), "dncc" + tag: npy(I.grad)})
    save("losses.npz", **out)

    # ---- R1 VxmDense 2-D (plugin features) and 3-D (default features), flow layer rescaled
    out = {}
    for tag, shp, feats_ in (("2d", (64, 64), O.PLUGIN_UNET_FEATURES), ("3d", (32, 32, 32), None)):
        torch.manual_seed(81)
        ov = O.VxmDense(shp, feats_, 7, True)
        with torch.no_grad():
            ov.flow.weight.mul_(1e5)
            ov.flow.bias.copy_(C.randn(82, *ov.flow.bias.shape) * 2.0)
        rv = RV.VxmDense(shp, feats_, int_steps=7, bidir=True)
        rv.load_state_dict(ov.state_dict(), strict=False)
        B = 2 if tag == "2d" else 1
        s_ = C.rand(83, B, 1, *shp).requires_grad_()
        t_ = C.rand(84, B, 1, *shp)
        ys, yt, fl = rv(s_, t_)
        cot, cot2 = C.randn(85, *ys.shape), C.randn(86, *fl.shape) * 0.1
        ((ys * cot).sum() + (fl * cot2).sum()).backward()
        out.update({"wsum_" + tag: np.array(C.state_checksum(ov)), "ys_" + tag: npy(ys), "yt_" + tag: npy(yt),
                    "flow_" + tag: npy(fl), "dsrc_" + tag: npy(s_.grad),
                    "gflow_w_" + tag: npy(rv.flow.weight.grad),
                    "gdown0_w_" + tag: npy(rv.unet_model.downarm[0].main.weight.grad),
                    "gup1_w_" + tag: npy(rv.unet_model.uparm[1].main.weight.grad)})
        y2, f2_ = rv(s_.detach(), t_, registration=True)
        out["reg_ys_" + tag] = npy(y2)
    save("vxm.npz", **out)

    # ---- S1 whole train step, config 1 geometry (64x64, batch 2), ngf=8 so that weights stay small
    size, B, ngf = 64, 2, 8
    opt = ref_options(size, B, ngf)
    orig_open = RM.open_image_to_torch
    RM.open_image_to_torch = lambda path, sz: orig_open(path, sz)[:, :, :size, :size].expand(B, -1, -1, -1)
    torch.manual_seed(91)
    ostep = O.RegistrationStep(size, B, ngf=ngf)
    with torch.no_grad():
        ostep.netR.flow.weight.mul_(1e5)       # |phi| ~ px so that warps / smoothness are not vacuous
        ostep.netR.flow.bias.copy_(C.randn(92, 2) * 1.0)
    A0, B0 = C.image_pair(93, B, size, size)
    ostep.ids_hook = lambda c, feats: [C.patch_ids(c, i, f.shape[2] * f.shape[3], 256) for i, f in enumerate(feats)]
    ostep.data_dependent_initialize(A0, B0)   # creates netF from the seeded RNG stream (before the reference model draws)
    with torch.no_grad():                     # zero MLP biases + exactly-zero (zero-padded) pixels make the
        for p in ostep.netF.parameters():     # reference's Normalize backward 0*inf = NaN; keep the fixture regular
            if p.dim() == 1:
                p.add_(0.01)
    model = RM.REGISTRATIONModel(opt)
    call = [0]

    def sizes_ids(feats):
        r = [C.patch_ids(call[0], i, f.shape[2] * f.shape[3], 256) for i, f in enumerate(feats)]
        call[0] += 1
        return r

    ref_netF_forward = model.netF.forward

    def netF_forward(feats, num_patches=64, patch_ids=None):
        if patch_ids is None:
            patch_ids = sizes_ids(feats)
        return ref_netF_forward(feats, num_patches, patch_ids)

    model.netF.forward = netF_forward
    data = {"A": A0, "B": B0, "A_paths": ["a"] * B, "B_paths": ["b"] * B}
    model.netG.load_state_dict(ostep.netG.state_dict())
    model.netR.load_state_dict(ostep.netR.state_dict(), strict=False)
    model.data_dependent_initialize(data)
    model.netF.load_state_dict(ostep.netF.state_dict())
    out = dict(wsum=np.array(C.multi_state_checksum((ostep.netG, ostep.netF, ostep.netR))))
    # the deformation field: optimize_parameters keeps netR's output in a local (registration_model.py:143-147), so it
    # is taken from the module's own forward through a hook -- y_output[2] of the step's one netR call
    flows = []
    hook = model.netR.register_forward_hook(lambda mod, inp, outp: flows.append(outp[2].detach().clone()))
    for it in range(3):
        A_, B_ = C.image_pair(100 + 2 * it, B, size, size)
        model.set_input({"A": A_, "B": B_, "A_paths": ["a"] * B, "B_paths": ["b"] * B})
        model.optimize_parameters()
        ls = model.get_current_losses()
        out["losses_%d" % it] = np.array([ls[k] for k in ("G", "NCE", "R", "smooth", "local", "NCE_Y")], dtype=np.float64)
        if it == 0:
            out.update(fake_B=npy(model.fake_B), registered=npy(model.registered), regA=npy(model.regA),
                       idt_B=npy(model.idt_B), pos_flow=npy(flows[0]),
                       # row A12: the decoded test pattern the reference warped (input) and the visual it produced
                       dvf_image=npy(RM.open_image_to_torch("./deform256.jpg", 256)[:1]), dvf=npy(model.dvf))
            for nm, net in (("G", model.netG), ("F", model.netF), ("R", model.netR)):
                g2 = sum(float((p.grad.double() ** 2).sum()) for p in net.parameters() if p.grad is not None)
                out["gradnorm_" + nm] = np.array(g2 ** 0.5)
    hook.remove()
    assert len(flows) == 3, len(flows)
    save("step.npz", **out)
    RM.open_image_to_torch = orig_open

    # ---- N3: the reference's own UnalignedDataset + get_transform (data/unaligned_dataset.py:20-87,
    # data/base_dataset.py:82-131) over seeded synthetic image folders (C.write_slice_folders)
    import random
    from data.unaligned_dataset import UnalignedDataset as RefDataset
    out = {}
    for tag, flip in (("flip", False), ("noflip", True)):
        root = tempfile.mkdtemp(prefix="dfmir_data_")
        C.write_slice_folders(root)
        dopt = ref_options(24, 1, 8)
        dopt.dataroot, dopt.phase, dopt.load_size, dopt.crop_size = root, "train", 30, 24
        dopt.preprocess, dopt.no_flip, dopt.serial_batches, dopt.max_dataset_size = "resize_and_crop", flip, False, float("inf")
        ds = RefDataset(dopt)
        out["len_" + tag] = np.array(len(ds))
        for idx in range(4):
            random.seed(500 + idx)
            torch.manual_seed(700 + idx)
            item = ds[idx]
            out["A_%s_%d" % (tag, idx)] = npy(item["A"])
            out["B_%s_%d" % (tag, idx)] = npy(item["B"])
            out["names_%s_%d" % (tag, idx)] = np.array(os.path.basename(item["A_paths"]) + "|" + os.path.basename(item["B_paths"]))
    save("dataset.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()

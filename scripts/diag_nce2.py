"""Diagnostic variants: which ingredient breaks the encoder backward?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import _load
from tests.test_oracle_golden import make_tiny_generator
from dfmir_amd import networks as N, ops

def run(B, variant):
    size = 64
    og = make_tiny_generator()
    hg = N.define_G(1, 1, 8, 'resnet_9blocks', 'instance', False, 'xavier', 0.02, False, False, [0], None)
    _load(hg, og)
    x = C.image_pair(52, B, size, size)[0]
    src = C.image_pair(53, B, size, size)[1]
    layers = [0, 4, 8, 12, 16]
    xo = x.clone().requires_grad_()
    fq = og(xo, layers, encode_only=True)
    xh = x.clone().cuda().requires_grad_()
    hq = hg(xh, layers, encode_only=True)
    if variant in ("nograd_pass", "gather"):
        with torch.no_grad():
            hk = hg(src.cuda(), layers, encode_only=True)
    if variant == "gather":
        ids = [C.patch_ids(0, i, f.shape[2] * f.shape[3], 256) for i, f in enumerate(fq)]
        lo = 0; lh = 0
        for i in (3, 4):
            cot = C.randn(70 + i, fq[i].shape[1], B * ids[i].numel())
            ro = fq[i].permute(0, 2, 3, 1).flatten(1, 2)[:, ids[i], :].flatten(0, 1)     # [B*P, C]
            lo = lo + (ro.t() * cot).sum()
            rh = ops.patch_gather(hq[i], ids[i].cuda())
            lh = lh + (rh * cot.cuda()).sum()
        lo.backward(); lh.backward()
    else:
        cots = [C.randn(60 + i, *f.shape) for i, f in enumerate(fq)]
        use = (4,) if variant == "last_only" else (0, 1, 2, 3, 4)
        sum((fq[i] * cots[i]).sum() for i in use).backward()
        sum((hq[i] * cots[i].cuda()).sum() for i in use).backward()
    res = []
    for (k, po), (k2, ph) in zip(og.named_parameters(), hg.named_parameters()):
        if po.grad is not None and k.endswith("weight") and ph.grad is not None:
            a, b = po.grad, ph.grad.cpu()
            res.append((k, float((a.flatten() @ b.flatten()) / (a.norm() * b.norm() + 1e-30))))
    xc = float((xo.grad.flatten() @ xh.grad.cpu().flatten()) / (xo.grad.norm() * xh.grad.cpu().norm() + 1e-30))
    worst = min(res, key=lambda t: t[1])
    print("B=%d %-12s x.grad cos %.6f | worst weight cos %.6f (%s) | last conv cos %.6f" % (
        B, variant, xc, worst[1], worst[0], dict(res)["model.16.conv_block.5.weight"]))

for B in (1, 2):
    for v in ("dense", "last_only", "nograd_pass", "gather"):
        run(B, v)

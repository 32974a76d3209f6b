"""Deterministic inputs shared by the golden-vector generator (make_golden.py, runs only where
/root/reference exists) and the tests that replay the vectors.  Pure torch CPU RNG."""
import hashlib

import numpy as np
import torch


def gen(seed):
    g = torch.Generator()
    g.manual_seed(int(seed))
    return g


def randn(seed, *shape):
    return torch.randn(*shape, generator=gen(seed))


def rand(seed, *shape):
    return torch.rand(*shape, generator=gen(seed))


def image_pair(seed, B, H, W):
    """Synthetic [-1,1] pairs with a -1 background so the (>-0.95) masks are non-trivial."""
    a = rand(seed, B, 1, H, W) * 2 - 1
    b = rand(seed + 1, B, 1, H, W) * 2 - 1
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    r = ((yy - H / 2.0) ** 2 + (xx - W / 2.0) ** 2).sqrt()
    bg = (r > 0.42 * min(H, W))[None, None]
    a = torch.where(bg, torch.full_like(a, -1.0), a)
    b = torch.where(bg.roll(3, -1), torch.full_like(b, -1.0), b)
    return a.contiguous(), b.contiguous()


def patch_ids(call, layer, S, P):
    """Patch ids of NCE call `call` (0,1 = data-dependent init; 2,3,4 = step 0; ...) at `layer`."""
    return torch.randperm(S, generator=gen(1000003 * (call + 1) + 7919 * layer))[:min(P, S)]


def edge_inputs():
    """Seeded inputs of fixture edges.npz (make_golden_edges.py), shared with tests/test_oracle_golden.py and tests/test_gpu_ops.py."""
    d = {}
    for tag, shp in (("2d", (2, 1, 24, 28)), ("3d", (1, 1, 12, 14, 16))):
        I = rand(175, *shp)
        J = 0.6 * I + 0.4 * rand(176, *shp)
        d["I" + tag], d["J" + tag] = I, J
        d["mask" + tag] = (rand(177, *shp) > 0.35)
    d["field3"] = randn(178, 1, 3, 7, 9, 11) * 1.5
    d["field2"] = randn(179, 2, 2, 18, 22)
    d["fmask2"] = (rand(180, 2, 1, 18, 22) > 0.3).float()
    return d


def edge_sample_feats():
    return [randn(162, 2, 1, 14, 14), randn(163, 2, 16, 12, 12), randn(164, 2, 32, 8, 8)]


def checksum(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def state_checksum(module):
    sd = module.state_dict()
    return checksum([sd[k] for k in sorted(sd.keys())])


def multi_state_checksum(modules):
    return hashlib.sha256("".join(state_checksum(m) for m in modules).encode()).hexdigest()


def write_slice_folders(root):
    """Seeded synthetic image folders <root>/trainA (3 RGB PNGs, 40x36) and <root>/trainB (2 grayscale PNGs, 33x47)
    for the data-pipeline fixture (row N3); PNG is lossless, so generator and test see the same pixels."""
    import os
    from PIL import Image
    rs = np.random.RandomState(77)
    for sub, n, shape in (("trainA", 3, (36, 40, 3)), ("trainB", 2, (47, 33))):
        d = os.path.join(root, sub)
        os.makedirs(d, exist_ok=True)
        for i in range(n):
            base = rs.randint(0, 256, size=(6, 5) + shape[2:]).astype(np.uint8)
            img = Image.fromarray(base).resize((shape[1], shape[0]), Image.BILINEAR)   # smooth content
            arr = np.asarray(img).astype(np.int32) + rs.randint(-20, 21, size=shape)
            Image.fromarray(np.clip(arr, 0, 255).astype(np.uint8)).save(os.path.join(d, "%s_%02d.png" % (sub, i)))

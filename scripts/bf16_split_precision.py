"""CPU experiment (no GPU): how much accuracy would error-compensated bf16 MFMA cost the generator?
Emulates conv(x, w) = sum of products of bf16 splits of x and w (fp32 accumulate) through the full ngf-64
ResnetGenerator at 256^2 and reports the output / feature error against plain fp32.
Result (round 1): 2-way split (3 MFMA products) 3.4e-5 relative at the output, 3-way split (6 products) 1.9e-6.
"""
import sys, torch, torch.nn.functional as F
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dfmir_oracle as O
from tests.golden import common as C
torch.set_num_threads(8)
torch.manual_seed(0)
G = O.Generator(1, 1, 64, 9); O.init_weights_xavier(G)
x = C.image_pair(3, 1, 256, 256)[0]
def split(t, n):
    parts = []; r = t
    for _ in range(n):
        h = r.to(torch.bfloat16).to(torch.float32); parts.append(h); r = r - h
    return parts
MODE = [None]
orig = F.conv2d
def conv2d_emul(inp, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if MODE[0] is None or groups != 1 or w.shape[2] == 7:
        return orig(inp, w, b, stride, padding, dilation, groups)
    n = MODE[0]
    xs, ws = split(inp, n), split(w, n)
    acc = None
    for i in range(n):
        for j in range(n):
            if i + j < n:       # keep terms up to order n-1: n=2 -> hh,hl,lh (3 products); n=3 -> 6 products
                t = orig(xs[i], ws[j], None, stride, padding, dilation, groups)
                acc = t if acc is None else acc + t
    return acc + (b.view(1, -1, 1, 1) if b is not None else 0)
F.conv2d = conv2d_emul
torch.nn.functional.conv2d = conv2d_emul
import torch.nn.modules.conv as mc
mc.F.conv2d = conv2d_emul
with torch.no_grad():
    ref, feats_ref = G(x, [4, 8, 12, 16, 20], encode_only=False)
    for n in (2, 3):
        MODE[0] = n
        out, feats = G(x, [4, 8, 12, 16, 20], encode_only=False)
        MODE[0] = None
        e = float((out - ref).abs().max() / ref.abs().max())
        fe = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(feats, feats_ref)]
        print("split into %d bf16 terms (%d MFMA products): output rel err %.2e ; feature rel errs %s" % (n, n * (n + 1) // 2, e, ["%.1e" % v for v in fe]))

"""GPU: the single-launch PatchNCE head plumbing (csrc/nce_head.hip) -- device patch-id draws, the multi-source key
gather, the fused per-term loss reduction and the scalar loss algebra -- against torch / the oracle, and the model's
batched key path against the term-by-term path it replaces on identical ids."""
import numpy as np
import pytest
import torch

from tests.golden import common as C
from tests.test_gpu_ops import DEV, close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from dfmir_amd import ops as _ops
    return _ops


def test_patch_ids_draw_properties(ops):
    """What torch.randperm(S)[:P] guarantees (models/networks.py:609-610): P distinct positions in [0, S), fresh at
    every call, every position equally likely.  Plus: reproducible from (seed, counter)."""
    sizes, T, P = [68644, 65536, 16384, 4096, 4096], 3, 256
    ops.seed_patch_ids(1234, DEV)
    a = ops.draw_patch_ids(sizes, T, P, DEV)
    b = ops.draw_patch_ids(sizes, T, P, DEV)
    assert a.shape == (5, T, P) and a.dtype == torch.int64
    for ids in (a, b):
        for l, S in enumerate(sizes):
            for t in range(T):
                v = ids[l, t].cpu().numpy()
                assert v.min() >= 0 and v.max() < S
                assert len(np.unique(v)) == P, (l, t)
    assert not torch.equal(a, b)                                   # the counter advanced on the device
    assert not torch.equal(a[3, 0], a[3, 1]) and not torch.equal(a[3, 0], a[4, 0])   # sets / layers independent
    ops.seed_patch_ids(1234, DEV)
    assert torch.equal(ops.draw_patch_ids(sizes, T, P, DEV), a)   # deterministic replay
    assert torch.equal(ops.draw_patch_ids(sizes, T, P, DEV), b)
    # uniformity: 400 sets of 256 out of 1024 positions -> 100 hits per position on average (sigma ~ 8.7)
    ops.seed_patch_ids(99, DEV)
    hits = np.zeros(1024)
    for _ in range(50):
        d = ops.draw_patch_ids([1024], 8, 256, DEV).cpu().numpy().reshape(-1)
        hits += np.bincount(d, minlength=1024)
    assert abs(hits.mean() - 100.0) < 1e-9
    assert hits.min() > 100 - 6 * 8.7 and hits.max() < 100 + 6 * 8.7, (hits.min(), hits.max())
    chi2 = float(((hits - 100.0) ** 2 / 100.0).sum())             # ~ 0.75 * 1023 (sampling without replacement)
    assert 0.55 * 1023 < chi2 < 0.95 * 1023, chi2
    # the densest case the kernel takes (S = 2P), P not a power of two, P > 256 (several slots per thread)
    d = ops.draw_patch_ids([600], 2, 300, DEV).cpu().numpy()
    assert len(np.unique(d[0, 0])) == 300 and len(np.unique(d[0, 1])) == 300 and d.max() < 600
    d = ops.draw_patch_ids([5000, 1400], 1, 700, DEV).cpu().numpy()
    assert len(np.unique(d[0, 0])) == 700 and len(np.unique(d[1, 0])) == 700 and d[1, 0].max() < 1400
    # dense layers (P <= S < 2P): the sorted-permutation branch; P == S is a permutation of everything
    d = ops.draw_patch_ids([300, 256, 4096], 3, 256, DEV).cpu().numpy()
    for t in range(3):
        assert len(np.unique(d[0, t])) == 256 and d[0, t].max() < 300
        assert sorted(d[1, t]) == list(range(256))
        assert len(np.unique(d[2, t])) == 256 and d[2, t].max() < 4096
    assert not np.array_equal(d[1, 0], d[1, 1])                     # a different order per set
    hits = np.zeros(300)
    for _ in range(100):
        hits += np.bincount(ops.draw_patch_ids([300], 8, 256, DEV).cpu().numpy().reshape(-1), minlength=300)
    assert abs(hits.mean() - 800 * 256 / 300.0) < 1e-6 and hits.min() > 600 and hits.max() < 760, (hits.min(), hits.max())
    from dfmir_amd._lib import DfmirHipError
    with pytest.raises(DfmirHipError):
        ops.draw_patch_ids([200], 1, 256, DEV)                      # S < P: ragged, host path
    with pytest.raises(DfmirHipError):
        ops.draw_patch_ids([5000], 1, 3000, DEV)                    # dense and too large for the LDS sort


def test_patch_gather_multi(ops):
    Bper, Cc, H, W, P, G = 2, 5, 7, 9, 16, 3
    srcs = [C.randn(70 + g, Bper, Cc, H, W) for g in range(2)]
    srcs = [srcs[0], srcs[1], srcs[1]]                              # real_A, real_B, real_B as in the step
    ids = torch.stack([C.patch_ids(5 + g, 0, H * W, P) for g in range(G)])
    got = ops.patch_gather_multi([s.to(DEV) for s in srcs], ids.to(DEV))   # [C, G*Bper*P]
    ref = torch.cat([s.permute(0, 2, 3, 1).flatten(1, 2)[:, ids[g], :].flatten(0, 1) for g, s in enumerate(srcs)])
    assert torch.equal(got.t().cpu(), ref)


def test_nce_terms_match_sequential(ops):
    """ops.nce_terms (one PatchNCE launch per layer over the stacked terms + one reduction) == the reference's
    per-term, per-layer `loss.mean() * lambda_NCE` summed and divided by n_layers (registration_model.py:247-253)."""
    from oracle import dfmir_oracle as O
    T, B, P, L, lam = 3, 2, 256, 2, 0.25
    Cs = [64, 32]
    qs = [O.l2_normalize(C.randn(80 + l, T * B * P, Cs[l])) for l in range(L)]
    ks = [O.l2_normalize(C.randn(90 + l, T * B * P, Cs[l])) for l in range(L)]
    qr = [q.clone().requires_grad_() for q in qs]
    ref = []
    for t in range(T):
        tot = 0.0
        for l in range(L):
            sl = slice(t * B * P, (t + 1) * B * P)
            tot = tot + O.patchnce_loss(qr[l][sl], ks[l][sl], B, 0.07).mean() * lam
        ref.append(tot / L)
    w = torch.tensor([0.5, 0.5, 0.25])
    (torch.stack(ref) * w).sum().backward()
    qg = [q.t().contiguous().to(DEV).requires_grad_() for q in qs]
    got = ops.nce_terms(qg, [k.t().contiguous().to(DEV) for k in ks], T * B, 0.07, lam / L, T)
    close(got, torch.stack(ref), rtol=2e-5, what="nce terms")
    (got * w.to(DEV)).sum().backward()
    for l in range(L):
        close(qg[l].grad.t(), qr[l].grad, rtol=3e-4, what="dq layer %d" % l)


def test_scalar_combine(ops):
    M = [[0.5, 0.5, 0.0], [0.0, 0.25, 1.0], [0.5, 0.75, 1.0]]
    xs = [torch.tensor(v, device=DEV, requires_grad=True) for v in (1.5, -2.0, 0.125)]
    out = ops.scalar_combine(M, xs)
    close(out, torch.tensor([-0.25, -0.375, -0.625]), what="combine")
    (out * torch.tensor([1.0, 2.0, 3.0], device=DEV)).sum().backward()
    for x, g in zip(xs, (0.5 + 1.5, 0.5 + 0.5 + 2.25, 2.0 + 3.0)):
        assert abs(float(x.grad) - g) < 1e-6


def test_batched_key_path_matches_term_by_term():
    """The default step (device-drawn ids, one key gather / MLP / PatchNCE launch per layer for the three terms) gives
    the losses and gradients of the term-by-term key path (netF called once per term as in
    registration_model.py:237-253) when both see the same patch ids."""
    from oracle import dfmir_oracle as O      # noqa: F401  (weights come from the oracle's seeded constructors)
    from tests.test_gpu_models import _hip_model_from_oracle, _load
    from tests.test_oracle_golden import make_step
    res = []
    for batched in (True, False):
        st, size, B = make_step()
        model, opt = _hip_model_from_oracle(st, size, B, 8)
        A0, B0 = C.image_pair(93, B, size, size)
        call = [0]
        base_forward = model.netF.forward

        def pinned(feats, num_patches=64, patch_ids=None, base_forward=base_forward, call=call):
            if patch_ids is None:
                patch_ids = [C.patch_ids(call[0], i, f.shape[2] * f.shape[3], 256).to(DEV) for i, f in enumerate(feats)]
                call[0] += 1
            return base_forward(feats, num_patches, patch_ids)

        model.netF.forward = pinned
        model.data_dependent_initialize({"A": A0, "B": B0, "A_paths": [""] * B, "B_paths": [""] * B})   # calls 0, 1
        _load(model.netF, st.netF)
        model.setup(opt)
        model.parallelize()
        if batched:
            del model.netF.forward                                   # back to the class method: the default path
            model.patch_id_source = lambda sizes, n_sets, P: torch.stack(
                [torch.stack([C.patch_ids(2 + t, l, S, P) for t in range(n_sets)]) for l, S in enumerate(sizes)])
        A_, B_ = C.image_pair(100, B, size, size)
        model.set_input({"A": A_, "B": B_, "A_paths": [""] * B, "B_paths": [""] * B})
        model.optimize_parameters()
        res.append(([v for v in model.get_current_losses().values()], [o_.flat_g.clone() for o_ in model.optimizers]))
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=3e-6)
    names = ["G", "R", "F"]                                      # model.optimizers order
    arenas = dict(zip(names, zip(res[0][1], res[1][1])))
    for nm in ("G", "R"):
        a, b = arenas[nm]
        assert float((a - b).norm()) <= 2e-5 * float(b.norm()), nm
    # netF's weight gradients are sums of per-row terms orthogonal to the (L2-normalised) rows: they cancel to ~1e-5
    # of the gradient that flows through the head (|g_F| ~ 5e-3 vs |g_G| ~ 8 here), so the 1e-7 relative difference
    # between the keys of the two paths (same math, different GEMM tile walk) is visible in them.  Bound the
    # difference by fp32 round-off of the flow they are taken from (scripts/diag/diag_keypath.py prints the breakdown;
    # the batched path reproduces itself to 1e-6).
    a, b = arenas["F"]
    assert float((a - b).norm()) <= 2e-6 * float(arenas["G"][1].norm()), float((a - b).norm())


@pytest.mark.parametrize("C_in,hw,G", [(3, 24, 1), (128, 16, 3), (256, 8, 2), (37, 12, 1)])
def test_fused_head_matches_torch_and_unfused(ops, C_in, hw, G):
    """dfmir_nce_head_fwd (sample + Linear + ReLU + Linear + L2-normalise in one launch, csrc/nce_head.hip) against the
    torch restatement of PatchSampleF.forward (models/networks.py:604-619) and against the unfused launches it replaces
    -- outputs and the gradients of the features and of both Linear layers -- including a channel count that is not a
    multiple of the K batch, an odd one, and a row count that does not fill the last 32-row tile."""
    from dfmir_amd import networks as N
    torch.manual_seed(7 + C_in)
    Bper, P = 2, 50 if C_in == 37 else 64
    B = G * Bper
    S = hw * hw
    feat0 = torch.randn(B, C_in, hw, hw, device=DEV)
    ids = torch.stack([torch.randperm(S, device=DEV)[:P] for _ in range(G)])
    netF = N.PatchSampleF(use_mlp=True, init_type='normal', init_gain=0.5, gpu_ids=[0], nc=256)
    netF.create_mlp([feat0])
    mlp = netF.mlp_0
    dy = torch.randn(256, B * P, device=DEV)
    got = {}
    for fused in (True, False):
        feat = feat0.clone().requires_grad_(True)
        for p_ in mlp.parameters():
            p_.grad = None
        if fused:
            assert ops.nce_head_ok(C_in, 256, True)
            y = ops.nce_head(feat, ids, mlp[0], mlp[2])
        else:
            y = netF.project(0, ops.patch_gather(feat, ids, G))
        (y * dy).sum().backward()
        got[fused] = [y.detach().clone(), feat.grad.clone()] + [p_.grad.clone() for p_ in mlp.parameters()]
    # torch restatement in fp64
    w1, b1, w2, b2 = [p_.detach().double() for p_ in (mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)]
    w1, w2 = w1.reshape(256, C_in), w2.reshape(256, 256)
    f = feat0.double().requires_grad_(True)
    fr = f.permute(0, 2, 3, 1).flatten(1, 2)                                    # [B, S, C]
    xs = torch.cat([fr[b, ids[b // Bper]] for b in range(B)], 0)               # [B*P, C]
    h = torch.relu(xs @ w1.t() + b1)
    yp = h @ w2.t() + b2
    yr = yp / (yp.pow(2).sum(1, keepdim=True).sqrt() + 1e-7)
    (yr.t() * dy.double()).sum().backward()
    close(got[True][0], yr.t().float(), rtol=2e-5, atol=2e-6, what="fused head vs torch")
    close(got[True][1], f.grad.float(), rtol=1e-4, atol=1e-5, what="fused head d(feat) vs torch")
    for a, b, nm in zip(got[True], got[False], ("y", "dfeat", "dw1", "db1", "dw2", "db2")):
        close(a, b, rtol=1e-4, atol=1e-5 * max(1.0, float(b.abs().max())), what="fused vs unfused " + nm)
    # key side: G separate source tensors, no gradient
    srcs = [feat0[g * Bper:(g + 1) * Bper].contiguous() for g in range(G)]
    close(ops.nce_head_multi(srcs, ids, mlp[0], mlp[2]), got[True][0], rtol=0, atol=0, what="multi-source head")

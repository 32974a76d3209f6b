"""GPU box: which pieces of the step survive hipGraph capture?  One case per process (a crash must not hide the
others):  python scripts/graph_probe.py <case>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfmir_amd import ops  # noqa: E402
from dfmir_amd import networks as N  # noqa: E402

DEV = "cuda"
case = sys.argv[1]


def capture(fn, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    ops.begin_graph_capture()
    with torch.cuda.graph(g):
        out = fn()
    ops.end_graph_capture()
    g.replay()
    torch.cuda.synchronize()
    return out


torch.manual_seed(0)
if case == "scale":
    x = torch.randn(1000, device=DEV)
    print(float(capture(lambda: ops.scale(x, 2.0)).sum()))
elif case == "memset":
    x = torch.randn(1000, device=DEV)
    print(float(capture(lambda: ops.mean(x))))
elif case == "l1":
    a, b = torch.randn(2, 1, 32, 32, device=DEV), torch.randn(2, 1, 32, 32, device=DEV)
    print(float(capture(lambda: ops.masked_l1(a, b, None, -0.95))))
elif case == "conv":
    c = N.Conv2d(64, 128, 3, padding=1).to(DEV)
    x = torch.randn(2, 64, 32, 32, device=DEV)
    with torch.no_grad():
        print(float(capture(lambda: c(x)).sum()))
elif case == "convin":
    c = N.Conv2d(64, 128, 3, padding=1).to(DEV)
    inn = N.InstanceNorm2d(128)
    x = torch.randn(2, 64, 32, 32, device=DEV)
    with torch.no_grad():
        print(float(capture(lambda: c(inn(c(x).narrow(1, 0, 64).contiguous(), relu=True))).sum()))
elif case == "ids":
    ops.seed_patch_ids(3, DEV)
    print(int(capture(lambda: ops.draw_patch_ids([4096, 1024], 3, 256, DEV)).sum()))
elif case == "bwd_torch":
    w = torch.randn(100, device=DEV, requires_grad=True)
    x = torch.randn(100, device=DEV)

    def f():
        w.grad = None
        (w * x).sum().backward()
        return w.grad
    print(float(capture(f).sum()))
elif case == "bwd_scale":
    w = torch.randn(100, device=DEV, requires_grad=True)

    def f():
        w.grad = None
        ops.mean(ops.scale(w, 2.0)).backward()
        return w.grad
    print(float(capture(f).sum()))
elif case == "bwd_conv":
    c = N.Conv2d(64, 128, 3, padding=1).to(DEV)
    x = torch.randn(2, 64, 32, 32, device=DEV, requires_grad=True)

    def f():
        x.grad = None
        c.weight.grad = None
        c.bias.grad = None
        ops.mean(c(x)).backward()
        return x.grad
    print(float(capture(f).sum()))
elif case == "gen":
    g = N.define_G(1, 1, 8, 'resnet_9blocks', 'instance', False, 'xavier', 0.02, False, False, [0], None)
    x = torch.randn(2, 1, 64, 64, device=DEV)
    with torch.no_grad():
        print(float(capture(lambda: g(x)).sum()))
elif case == "gen_bwd":
    g = N.define_G(1, 1, 8, 'resnet_9blocks', 'instance', False, 'xavier', 0.02, False, False, [0], None)
    x = torch.randn(2, 1, 64, 64, device=DEV)

    def f():
        for p in g.parameters():
            p.grad = None
        y = g(x)
        with ops.deferred_weight_grads():
            ops.mean(y).backward()
        return y
    print(float(capture(f).sum()))
elif case == "warp":
    s = torch.randn(2, 1, 64, 64, device=DEV, requires_grad=True)
    fl = torch.randn(2, 2, 64, 64, device=DEV, requires_grad=True)

    def f():
        s.grad = None
        fl.grad = None
        ops.mean(ops.warp(s, fl)).backward()
        return s.grad
    print(float(capture(f).sum()))
elif case == "vxm":
    from dfmir_amd import voxelmorph as V
    hv = V.VxmDense((64, 64), [[16, 32, 32, 64, 64, 64], [64, 64, 64, 32, 32, 32, 16]], int_steps=7, bidir=True).to(DEV)
    a, b = torch.rand(2, 1, 64, 64, device=DEV), torch.rand(2, 1, 64, 64, device=DEV)

    def f():
        for p in hv.parameters():
            p.grad = None
        ys, yt, fl = hv(a, b)
        with ops.deferred_weight_grads():
            (ops.mean(ys) + ops.flow_smoothness(fl)).backward()
        return ys
    print(float(capture(f).sum()))
elif case == "nce":
    from dfmir_amd.options import default_options
    pf = N.PatchSampleF(use_mlp=True, init_type='xavier', init_gain=0.02, nc=256, gpu_ids=[0])
    feats = [torch.randn(6, 16, 32, 32, device=DEV, requires_grad=True), torch.randn(6, 32, 32, 32, device=DEV, requires_grad=True)]
    pf.create_mlp(feats)
    ops.seed_patch_ids(3, DEV)
    keys = [torch.randn(2, 16, 32, 32, device=DEV), torch.randn(2, 32, 32, 32, device=DEV)]

    def f():
        for p in pf.parameters():
            p.grad = None
        for t in feats:
            t.grad = None
        ids = ops.draw_patch_ids([1024, 1024], 3, 256, DEV)
        with torch.no_grad():
            k_cm = [pf.project(l, ops.patch_gather_multi([keys[l]] * 3, ids[l])) for l in range(2)]
        q_cm = [pf.project(l, ops.patch_gather(feats[l], ids[l], 3)) for l in range(2)]
        losses = ops.nce_terms(q_cm, k_cm, 6, 0.07, 0.25 / 2, 3)
        a, b, c = losses.unbind(0)
        out = ops.scalar_combine([[0.5, 0.5, 0.0], [0.5, 0.5, 0.25]], [a, b, c])
        with ops.deferred_weight_grads():
            out[1].backward()
        return out
    print(float(capture(f).sum()))
elif case.startswith("model"):
    from dfmir_amd.options import default_options
    from dfmir_amd.registration_model import REGISTRATIONModel
    B, S = 2, 64
    opt = default_options(batch_size=B, crop_size=S, load_size=S, ngf=8, gpu_ids=[0], checkpoints_dir="/tmp/c", name="c",
                          capture_step=(case == "model"))
    model = REGISTRATIONModel(opt)
    a, b = torch.rand(B, 1, S, S, device=DEV) * 2 - 1, torch.rand(B, 1, S, S, device=DEV) * 2 - 1
    data = {"A": a, "B": b, "A_paths": [""] * B, "B_paths": [""] * B}
    model.data_dependent_initialize(data)
    model.setup(opt)
    model.parallelize()
    if case == "model_fwd":
        model.set_input(data)
        with torch.no_grad():
            def f():
                model.forward()
                y = model.netR(model.real_A, model.real_B)
                return model.fake_B
            print(float(capture(f).sum()))
    elif case.startswith("model_stage"):
        stage = int(case[len("model_stage"):])
        from dfmir_amd.losses import smooothing_loss
        for i in range(2):
            model.set_input(data)
            model.optimize_parameters()
        model.set_input(data)

        def f():
            m = model
            m.forward()
            if stage == 1:
                return m.fake_B
            y_output = m.netR(m.real_A, m.real_B)
            m.registered = m.spatialTransformer(m.fake_B, y_output[2])
            with torch.no_grad():
                m.dvf = m.spatialTransformer(m._checkerboard(m.real_A.size(0)), y_output[2].detach())
            if stage == 2:
                return m.registered
            m.optimizer_G.zero_grad(); m.optimizer_R.zero_grad(); m.optimizer_F.zero_grad()
            if stage == 3:
                return m.registered
            terms = m.calculate_NCE_losses_stacked(((m.real_A, None), (m.real_B, None), (m.real_B, y_output[0])))
            if stage == 4:
                return terms[0]
            l1_reg = m.calculate_L1_loss(m.registered, m.real_B, mask='threshold')
            l1_idt = m.calculate_L1_loss(m.idt_B, m.registered, mask='threshold')
            smooth = smooothing_loss(y_output[2])
            out = ops.scalar_combine([[0.5, 0.5, 0.25, 1.0, 1.0, 0.20]], list(terms) + [l1_reg, l1_idt, smooth])
            if stage == 5:
                return out
            if stage == 6:      # backward of everything but the NCE terms
                o2 = ops.scalar_combine([[1.0, 1.0, 0.20]], [l1_reg, l1_idt, smooth])
                with ops.deferred_weight_grads():
                    o2[0].backward()
                return o2
            if stage == 7:      # backward of the NCE terms only
                o2 = ops.scalar_combine([[0.5, 0.5, 0.25]], list(terms))
                with ops.deferred_weight_grads():
                    o2[0].backward()
                return o2
            with ops.deferred_weight_grads():
                out[0].backward()
            return out
        print(float(capture(f, warm=1).sum()))
    else:
        opt.capture_step = True
        for i in range(5):
            model.set_input(data)
            model.optimize_parameters()
            print(i, model.get_current_losses())
elif case == "memset_order":
    # is a captured memset node ordered with the kernels around it?
    x = torch.empty(8, device=DEV)
    big = torch.empty(1 << 22, device=DEV)
    a, b = torch.randn(2, 1, 64, 64, device=DEV), torch.randn(2, 1, 64, 64, device=DEV)
    ref = float(ops.masked_l1(a, b, None, -0.95))
    outs = []

    def f():
        x.zero_()
        x.add_(1.0)
        big.zero_()
        big.add_(2.0)
        outs[:] = [ops.masked_l1(a, b, None, -0.95) for _ in range(4)]
        return x
    capture(f)
    g = torch.cuda.CUDAGraph()
    bad = 0
    with torch.cuda.graph(g):
        f()
    for it in range(300):
        g.replay()
        torch.cuda.synchronize()
        vals = [float(o) for o in outs]
        if float(x.sum()) != 8.0 or float(big[::4097].sum()) != 2.0 * len(big[::4097]) or any(abs(v - ref) > 1e-6 for v in vals):
            bad += 1
            if bad < 5:
                print("replay", it, float(x.sum()), float(big[::4097].sum()), vals, ref)
    print("bad replays:", bad, "of 300")
print("CASE", case, "OK")

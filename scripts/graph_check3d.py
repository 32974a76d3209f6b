import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from dfmir_amd.registration3d import Registration3DModel
PLUGIN = [[16, 32, 32, 64, 64, 64], [64, 64, 64, 32, 32, 32, 16]]
for shape, feats in (((64, 64, 64), PLUGIN), ((128, 128, 128), PLUGIN), ((160, 192, 224), None)):
    res = {}
    for cap in (False, True):
        torch.manual_seed(0)
        m = Registration3DModel(shape, feats, capture_step=cap)
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        A = torch.rand(1, 1, *shape, device="cuda", generator=g) * 2 - 1
        B = 0.5 * A + 0.5 * (torch.rand(1, 1, *shape, device="cuda", generator=g) * 2 - 1)
        losses = []
        for i in range(6):
            m.set_input({"A": A, "B": B}); m.optimize_parameters(); losses.append(m.get_current_losses())
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 10
        for _ in range(n):
            m.set_input({"A": A, "B": B}); m.optimize_parameters()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        res[cap] = (dt, th / n, losses, m.optimizer_R.flat_p.clone())
    e, c = res[False], res[True]
    dp = float((e[3] - c[3]).abs().max()) / float(e[3].abs().max())
    print(shape, "eager %.2f ms (host %.2f)  captured %.2f ms (host %.2f)  loss5 eager %s captured %s  max rel param diff %.2e" % (
        e[0] * 1e3, e[1] * 1e3, c[0] * 1e3, c[1] * 1e3, e[2][5], c[2][5], dp))

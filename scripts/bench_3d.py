"""GPU box: time the 3-D registration step (BASELINE configs 4 and 5 geometries, 1 volume per GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd.registration3d import Registration3DModel

PLUGIN = [[16, 32, 32, 64, 64, 64], [64, 64, 64, 32, 32, 32, 16]]
for shape, feats, name, gf in (((128, 128, 128), PLUGIN, "128^3 plugin feats", 285.0),
                               ((160, 192, 224), None, "160x192x224 default feats", 2393.0)):
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        continue
    torch.manual_seed(0)
    m = Registration3DModel(shape, feats)
    A = torch.rand(1, 1, *shape, device="cuda") * 2 - 1
    B = 0.5 * A + 0.5 * (torch.rand(1, 1, *shape, device="cuda") * 2 - 1)
    for _ in range(2):
        m.set_input({"A": A, "B": B}); m.optimize_parameters()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        m.set_input({"A": A, "B": B}); m.optimize_parameters()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-28s %.1f ms/step  %.2f pairs/s  %.1f TFLOP/s conv  losses %s  peak mem %.1f GB" % (
        name, dt * 1e3, 1 / dt, gf / dt / 1e3, m.get_current_losses(), torch.cuda.max_memory_allocated() / 2**30))

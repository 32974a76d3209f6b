"""GPU box: the 128^3 plugin-feature 3-D step alone (BASELINE configs[3]) -- for scripts/prof_3d_step.sh-style traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd.registration3d import Registration3DModel
PLUGIN = [[16, 32, 32, 64, 64, 64], [64, 64, 64, 32, 32, 32, 16]]
shape = (128, 128, 128)
torch.manual_seed(0)
m = Registration3DModel(shape, PLUGIN)
A = torch.rand(1, 1, *shape, device="cuda") * 2 - 1
B = 0.5 * A + 0.5 * (torch.rand(1, 1, *shape, device="cuda") * 2 - 1)
for _ in range(3):
    m.set_input({"A": A, "B": B}); m.optimize_parameters()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 10
for _ in range(n):
    m.set_input({"A": A, "B": B}); m.optimize_parameters()
th = time.perf_counter() - t0
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("128^3 plugin feats %.2f ms/step (host enqueue %.2f ms/step)" % (dt * 1e3, th / n * 1e3))

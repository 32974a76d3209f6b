"""Run one split conv shape back to back for 6 s while sampling rocm-smi clocks/power (profiles/r01_power_clock.md).
args: randn | relu | zero (input data)"""
import os, sys, time, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops
Cin = Cout = 256; H = 64; n = 32
mode = sys.argv[1] if len(sys.argv) > 1 else "randn"
x = torch.randn(n, Cin, 1, H, H, device="cuda")
if mode == "relu": x = torch.relu(x)
if mode == "zero": x = torch.zeros_like(x) + 1e-3
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02
wt = ops.weight_pack(w, 0)
xa = ops.absmax(x)
def run():
    ops.conv_raw(x, wt, None, Cout, (1, 3, 3), 1, (0, 1, 1), 1, 1, 0, 0.0, (1, H, H), xa)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip()[:600])
        except Exception as e:
            samples.append("ERR %r" % e)
        time.sleep(0.5)
th = threading.Thread(target=sampler); th.start()
for _ in range(10): run()
torch.cuda.synchronize()
t0 = time.time(); k = 0
while time.time() - t0 < 6.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    k += 1
    if k % 3 == 0: print("%s t=%.1fs  %.4f ms/launch (%.0f TF)" % (mode, time.time() - t0, e0.elapsed_time(e1) / 200, 154.6 / (e0.elapsed_time(e1) / 200)))
stop = True; th.join()
for s in samples[2:8]: print(s)

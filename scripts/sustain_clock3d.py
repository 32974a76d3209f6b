"""Run one 3-D split kernel (fwd | wgrad, layer Cin-Cout at 160x192x224) back to back for 5 s while sampling
rocm-smi clocks / power.  args: fwd|wgrad Cin Cout"""
import os, sys, time, subprocess, threading, json, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
Cin, Cout = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (34, 32)
sp = (160, 192, 224)
x = torch.randn(1, Cin, *sp, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") / (Cin * 27) ** 0.5
dy = torch.randn(1, Cout, *sp, device="cuda")
wt = ops.weight_pack(w, 0)
xa, da = ops.absmax(x), ops.absmax(dy)
fl = 2.0 * Cout * sp[0] * sp[1] * sp[2] * Cin * 27
def run():
    with torch.no_grad():
        if what == "fwd":
            ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 1, 0.2, sp, xa)
        else:
            ops.conv_wgrad_raw(x, dy, (3, 3, 3), 1, (1, 1, 1), 0, x_amax=xa, dy_amax=da)
samples = []; stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out)
        except Exception as e:
            samples.append("ERR %r" % e)
        time.sleep(0.4)
th = threading.Thread(target=sampler); th.start()
for _ in range(5): run()
torch.cuda.synchronize()
t0 = time.time(); k = 0
while time.time() - t0 < 5.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): run()
    e1.record(); torch.cuda.synchronize(); k += 1
    if k % 3 == 0:
        ms = e0.elapsed_time(e1) / 100
        print("%s %d->%d t=%.1fs %.4f ms/launch (%.0f TF algorithmic)" % (what, Cin, Cout, time.time() - t0, ms, fl / ms / 1e9))
stop = True; th.join()
for s in samples[3:8]:
    m = re.findall(r'"(sclk clock speed:|Current Socket Graphics Package Power \(W\)|Average Graphics Package Power \(W\))": "([^"]+)"', s)
    print(m if m else s[:300])

"""GPU box: Grad_Loss l2 (util/losses.py:81-130) forward / backward on a [1,3,160,192,224] flow, HIP-event timed with a
rotating set of buffers (cold)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops
dev = "cuda"
sp = (160, 192, 224)
flows = [torch.randn(1, 3, *sp, device=dev).requires_grad_() for _ in range(4)]
def timeit(fn, reps=20):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
with torch.no_grad():
    mf = timeit(lambda i: ops.flow_smoothness(flows[i % 4], 'l2'))
ls = [ops.flow_smoothness(f, 'l2') for f in flows]
mb = timeit(lambda i: torch.autograd.grad(ls[i % 4], flows[i % 4], retain_graph=True))
n = flows[0].numel() * 4
print("FWD_CAP %s BWD_CAP %s: forward %.3f ms (%.2f TB/s), backward %.3f ms (%.2f TB/s)" % (
    os.environ.get("DFMIR_FS_FWD_CAP"), os.environ.get("DFMIR_FS_BWD_CAP"), mf, n / mf / 1e9, mb, 2 * n / mb / 1e9))

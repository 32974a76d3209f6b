"""GPU box: how long does the HOST sit in hipGraphLaunch for (a) ONE graph with a forked second branch and (b) the same
work as separate single-stream graphs launched on their own streams with event edges between the launches?
(round 6: opt.overlap_registration blocks the host for ~60 % of a step in form (a); registration_model's staged step is
form (b).)   python scripts/graph_split_probe.py [n_main] [n_side] [mbytes]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfmir_amd import ops  # noqa: E402

n_main = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 150
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda")
big = torch.empty(mb << 18, device=dev)          # n floats: one fill ~ mb MB of writes
small = torch.empty(4 << 18, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def main_work(n=n_main):
    for _ in range(n):
        ops.zero_(big)


def side_work(n=n_side):
    for _ in range(n):
        ops.zero_(small)


def timed(label, launch, reps=5):
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        launch()
        host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        total.append(time.perf_counter() - t0)
    print("%-46s host in launch %8.3f ms   step %8.3f ms" % (label, 1e3 * sorted(host)[len(host) // 2], 1e3 * sorted(total)[len(total) // 2]))


main_work(2); side_work(2)
torch.cuda.synchronize()

# (a) one graph, fork / join inside
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s1):
    s2.wait_stream(s1)
    with torch.cuda.stream(s2):
        side_work()
    main_work()
    s1.wait_stream(s2)
timed("(a) one graph with a forked branch", g.replay)

# (a') one chain
gc_ = torch.cuda.CUDAGraph()
with torch.cuda.graph(gc_, stream=s1):
    side_work()
    main_work()
timed("(a') one graph, one chain", gc_.replay)

# (b) two single-stream graphs, event edges between the launches
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(ga, stream=s1):
    main_work()
with torch.cuda.graph(gb, stream=s2):
    side_work()


def launch_b():
    s2.wait_stream(s1)
    with torch.cuda.stream(s2):
        gb.replay()
    with torch.cuda.stream(s1):
        ga.replay()
    s1.wait_stream(s2)


timed("(b) two graphs on two streams + events", launch_b)

# (c) eight single-stream graphs alternating streams (the staged step's shape)
parts = []
for i in range(8):
    gi = torch.cuda.CUDAGraph()
    s = s1 if i % 2 == 0 else s2
    with torch.cuda.graph(gi, stream=s):
        (main_work if i % 2 == 0 else side_work)((n_main if i % 2 == 0 else n_side) // 4)
    parts.append((s, gi))


def launch_c():
    for i, (s, gi) in enumerate(parts):
        other = s2 if s is s1 else s1
        if i in (2, 3, 5, 6):
            s.wait_stream(other)
        with torch.cuda.stream(s):
            gi.replay()
    s1.wait_stream(s2)


timed("(c) eight graphs alternating streams + events", launch_c)

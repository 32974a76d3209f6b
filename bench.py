#!/usr/bin/env python
"""bench.py -- throughput of the DFMIR `--model registration` train step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" = REGISTRATIONModel.set_input + optimize_parameters (reference train.py:46-47): forward,
backward and Adam of G/F/R on one batch of synthetic pairs already resident in HBM.  Workload at
every N: BASELINE.json configs[1] geometry -- 2-D 256x256, batch 16 PER GPU, ngf 64, fp32 (weak
scaling; configs[2] is the same thing at N = 8).  One JSON line on rank 0 with
  `roofline`      dominant kernel = conv3x3_split_cs_k (3x3 forward/dgrad convs, fp32 operands split into scaled
                  fp16 pairs, 3 products per MAC on v_mfma_f32_32x32x16_f16), timed live with HIP events on its
                  launch stream; achieved = ISSUED 16-bit matrix FLOP/s (3 x algorithmic), peak = 2.5 PFLOP/s dense
                  fp16, frac <= 1; the algorithmic fp32 rate is reported beside it
  `roofline_hbm`  the trilinear warp (grid_sample) forward / backward at 160x192x224, algorithmic bytes / HIP-event time
                  against 8 TB/s, measured in the same process
  `also_3d`       the 3-D step of configs[4] geometry on one GPU with its own roofline (conv3d_split_k forward / dgrad and
                  conv3d_wgrad_tr_k, both scaled fp16x2 on the 16-bit matrix pipe, issued FLOP/s against 2.5 PFLOP/s)
  `cpu_baseline`  (N = 1) the CPU oracle = a port of the reference's PyTorch-CPU path, timed on the host cores on a
                  bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz
FP16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (v_mfma_f32_32x32x16_f16), 2.4 GHz
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E
CS_TRAFFIC_BYTES = 480.5e6      # FETCH_SIZE 346.3 MB + WRITE_SIZE 134.2 MB (profiles/r01_conv3x3s_pmc.md)


def synth_pairs(B, H, W, device, seed):
    """Uniform [-1,1] slices with a -1 background disc complement (non-trivial >-0.95 masks)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    a = torch.rand(B, 1, H, W, generator=g) * 2 - 1
    b = torch.rand(B, 1, H, W, generator=g) * 2 - 1
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    bg = (((yy - H / 2.0) ** 2 + (xx - W / 2.0) ** 2).sqrt() > 0.45 * min(H, W))[None, None]
    a = torch.where(bg, torch.full_like(a, -1.0), a)
    b = torch.where(bg, torch.full_like(b, -1.0), b)
    return a.to(device), b.to(device)


class KernelTimer(object):
    """Brackets selected kernel launches with HIP events on the launch (= torch current) stream."""

    def __init__(self, kinds, every=1):
        self.kinds = set(kinds)
        self.records = {}
        self.enabled = False
        self.every = every          # bracket every n-th launch of a kind (keeps the events out of the step's way)
        self.seen = {}

    def __call__(self, kind, flops, launch):
        if not self.enabled or kind not in self.kinds:
            launch()
            return
        n = self.seen.get(kind, 0)
        self.seen[kind] = n + 1
        if n % self.every:
            launch()
            return
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        launch()
        e.record()
        self.records.setdefault(kind, []).append((s, e, flops))

    def summary(self):
        out = {}
        for kind, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            fl = sum(f for _, _, f in recs)
            out[kind] = dict(launches=len(recs), ms=ms, flops=fl)
        return out


def pick_cpu_threads():
    """Threads for the CPU baseline: the host cores this process may actually use (affinity and
    cgroup quota), refined by a 1-second calibration of the path's dominant conv shape -- a box that
    advertises 256 CPUs but schedules far fewer is 100x slower when oversubscribed."""
    import torch.nn.functional as F
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail = max(1, min(avail, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    x = torch.randn(1, 256, 64, 64)
    w = torch.randn(256, 256, 3, 3)
    best, best_t = 1, float("inf")
    for n in sorted(set(min(c, avail) for c in (avail, 128, 64, 32, 16, 8))):
        torch.set_num_threads(n)
        F.conv2d(x, w, padding=1)
        t0 = time.time()
        for _ in range(3):
            F.conv2d(x, w, padding=1)
        dt = time.time() - t0
        if dt < best_t * 0.95:
            best, best_t = n, dt
    return best


def cpu_baseline(size, max_steps=3):
    """The oracle (CPU port of the reference path) at the same 256x256 geometry, batch 1."""
    from oracle import dfmir_oracle as O
    cores = pick_cpu_threads()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    st = O.RegistrationStep(size, 1, ngf=64)
    a, b = synth_pairs(1, size, size, "cpu", 1)
    st.data_dependent_initialize(a, b)      # warm-up (also creates netF), like train.py:43
    t0 = time.time()
    n = 0
    while n < max_steps:
        a, b = synth_pairs(1, size, size, "cpu", 2 + n)
        st.step(a, b)
        n += 1
        if time.time() - t0 > 30.0:
            break
    dt = time.time() - t0
    return dict(value=n / dt, unit="image-pairs/s", cores=cores, kind="port",
                sample="%d train steps of the CPU oracle (PyTorch fp32, %d threads) at 2-D %dx%d batch 1, ngf 64, "
                       "after 1 warm-up fwd+bwd" % (n, cores, size, size))


def bench_3d(dev, steps=5, warmup=2):
    """Auxiliary line: the 3-D step of BASELINE configs[4] geometry on ONE GPU -- 160x192x224, batch 1,
    VxmDense(default features, int_steps 7, bidir) + NCC[9,9,9] + Grad l2, fwd+bwd+Adam.
    Its roofline: conv3d_mfma16_k (every stride-1 3x3x3 forward / dgrad conv), fp32 MFMA, timed with HIP events."""
    from dfmir_amd import ops
    from dfmir_amd.registration3d import Registration3DModel
    shape = (160, 192, 224)
    torch.manual_seed(0)
    m = Registration3DModel(shape, None, device=dev)
    A = torch.rand(1, 1, *shape, device=dev) * 2 - 1
    B = 0.5 * A + 0.5 * (torch.rand(1, 1, *shape, device=dev) * 2 - 1)
    sizes = ("small", "S", "M", "L")
    sp_kinds = ["conv3ds_" + z for z in sizes]          # split fp16x2 kernel (csrc/conv3ds.hip)
    fw_kinds = ["conv3d_" + z for z in sizes]           # fp32-MFMA kernel (csrc/conv3d.hip): Cout < 8 layers, or A/B env
    wg_kinds = ["wgrad3d_" + z for z in ("S", "M", "L")]      # fp32-MFMA wgrad (Cout < 8 / Cin > 48 layers)
    wgs_kinds = ["wgrad3ds_" + z for z in ("S", "M", "L")]    # split fp16x2 wgrad
    timer = KernelTimer(sp_kinds + fw_kinds + wg_kinds + wgs_kinds)
    ops.set_conv_profiler(timer)
    for _ in range(warmup):
        m.set_input({"A": A, "B": B})
        m.optimize_parameters()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(steps):
        m.set_input({"A": A, "B": B})
        m.optimize_parameters()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    timer.enabled = False
    ops.set_conv_profiler(None)
    losses = m.get_current_losses()
    assert all(v == v and abs(v) < 1e6 for v in losses.values()), losses   # finite
    ks = timer.summary()

    def agg(kinds):
        ms = sum(ks[k]["ms"] for k in kinds if k in ks)
        fl = sum(ks[k]["flops"] for k in kinds if k in ks)
        n = sum(ks[k]["launches"] for k in kinds if k in ks)
        return (fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0), n, ms
    sp_tf, sp_n, sp_ms = agg(sp_kinds)
    fw_tf, fw_n, fw_ms = agg(fw_kinds)
    wg_tf, wg_n, wg_ms = agg(wg_kinds)
    wgs_tf, wgs_n, wgs_ms = agg(wgs_kinds)
    if sp_n:
        roof = {"bound": "mfma", "achieved": 3.0 * sp_tf, "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": 3.0 * sp_tf / FP16_MFMA_PEAK_TFLOPS, "traffic": None,
                "achieved_note": "issued 16-bit matrix FLOP/s = 3 x the algorithmic fp32 FLOP/s (scaled fp16x2 split, 3 "
                                 "products per MAC); algorithmic FLOPs = 2*N*Cout*D*H*W*Cin*27 of the timed launches",
                "algorithmic_tflops": sp_tf, "products_per_mac": 3.0,
                "kernel": "conv3d_split_k (v_mfma_f32_32x32x16_f16, 4x8x16-voxel x 32-cout tiles, LDS halo patch split into "
                          "fp16 pairs per 8-channel chunk): forward + dgrad of every stride-1 3x3x3 conv",
                "launches_timed": sp_n, "avg_launch_ms": sp_ms / max(sp_n, 1)}
    else:
        roof = {"bound": "mfma", "achieved": fw_tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fw_tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                "kernel": "conv3d_mfma16_k (v_mfma_f32_16x16x4_f32): forward + dgrad of every stride-1 3x3x3 conv",
                "launches_timed": fw_n, "avg_launch_ms": fw_ms / max(fw_n, 1)}
    if wgs_n:
        roof.update({"wgrad_kernel": "conv3d_wgrad_tr_k (voxels as the matrix K; operands read with ds_read_b64_tr_b16 from "
                                     "channel-major fp16-pair LDS images of the patch and of dY; plane-pair columns for "
                                     "<= 16 output channels)",
                     "wgrad_achieved": 3.0 * wgs_tf, "wgrad_frac": 3.0 * wgs_tf / FP16_MFMA_PEAK_TFLOPS,
                     "wgrad_algorithmic_tflops": wgs_tf, "wgrad_launches_timed": wgs_n})
    else:
        roof.update({"wgrad_kernel": "conv3d_wgrad16_k (v_mfma_f32_16x16x4_f32)", "wgrad_kernel_tflops": wg_tf,
                     "wgrad_frac_of_fp32_peak": wg_tf / FP32_MFMA_PEAK_TFLOPS, "wgrad_launches_timed": wg_n})
    return {"workload": "3-D 160x192x224 volume pair, batch 1, VxmDense default features + NCC[9,9,9] + Grad-l2, "
                        "fwd+bwd+Adam (BASELINE configs[4] geometry, one GPU)",
            "value": 1.0 / dt, "unit": "image-pairs/s", "ms_per_step": 1e3 * dt, "steps": steps, "warmup": warmup,
            "conv_tflops": 2393.0 / dt / 1e3, "dtype": "f32",
            "losses": {k: round(v, 6) for k, v in losses.items()}, "roofline": roof}


def bench_warp_hbm(dev, reps=20):
    """roofline_hbm: the trilinear displacement-field warp (SpatialTransformer = grid_sample, reference
    models/voxelmorph/torchvoxelmorph/layers.py:30-48) at 160x192x224, C = 1, on a registration-like smooth field
    (control points every 32 voxels, ~1 voxel rms).  Algorithmic bytes: fwd 4*(C+nd+C)*N_vox,
    bwd 4*(C [dOut] + C [src] + nd [flow] + C [dSrc] + nd [dFlow])*N_vox; time = HIP events on the launch stream."""
    from dfmir_amd import ops
    sp, C, nd = (160, 192, 224), 1, 3
    g = torch.Generator(device="cpu")
    g.manual_seed(5)
    src = torch.randn(1, C, *sp, generator=g).to(dev)
    coarse = torch.randn(1, nd, *[s // 32 for s in sp], generator=g).to(dev)
    flow = torch.nn.functional.interpolate(coarse, size=sp, mode='trilinear', align_corners=True).contiguous()
    dout = torch.randn(1, C, *sp, generator=g).to(dev)
    dsrc, dflow = torch.zeros_like(src), torch.empty_like(flow)
    nv = src.numel() // C

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps

    def bwd():
        ops._warp_bwd_dsrc(dout, src, flow, dflow, 0, 0)

    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
           "workload": "160x192x224, C=1, smooth field", "kernel": "warp_win_fwd_k<3>"}
    b_f = 4 * (C + nd + C) * nv
    ms = timeit(lambda: ops._warp_fwd(src, flow, 0, 0))
    out.update(achieved=b_f / ms / 1e6, frac=b_f / ms / 1e6 / HBM_PEAK_GBS, bytes=b_f, avg_launch_ms=ms)
    b_b = 4 * (C + 2 * C + 2 * nd) * nv
    ms = timeit(bwd)
    out["bwd"] = {"kernel": "warp_win_bwd_own_k<3> + warp_win_gather_k<3> (+ the empty slow-voxel pass): d(src) + d(flow) "
                            "without device-scope atomics, bit-reproducible", "achieved": b_b / ms / 1e6,
                  "frac": b_b / ms / 1e6 / HBM_PEAK_GBS, "bytes": b_b, "avg_launch_ms": ms}
    b_bf = 4 * (C + C + 2 * nd) * nv
    ms = timeit(lambda: ops._warp_bwd(dout, src, flow, None, dflow, 0, 0))
    out["bwd_dflow_only"] = {"achieved": b_bf / ms / 1e6, "frac": b_bf / ms / 1e6 / HBM_PEAK_GBS, "bytes": b_bf,
                             "avg_launch_ms": ms}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--ngf", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-3d", action="store_true", help="skip the auxiliary 3-D (config 5 geometry) measurement")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every step eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--roofline-steps", type=int, default=3, help="eager steps after the timed region that time the dominant kernels")
    ap.add_argument("--host-input-steps", type=int, default=5,
                    help="extra steps fed from pinned HOST memory (PCIe-inclusive rate, reported beside the headline)")
    args = ap.parse_args()

    from dfmir_amd import distributed as dfdist
    rank, world, dev_index = dfdist.init_from_env("nccl")     # RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env
    dev = torch.device("cuda", dev_index)

    from dfmir_amd import ops
    from dfmir_amd.options import default_options
    from dfmir_amd.registration_model import REGISTRATIONModel

    B, S = args.batch, args.size
    opt = default_options(batch_size=B, crop_size=S, load_size=S, ngf=args.ngf, gpu_ids=[dev.index],
                          checkpoints_dir="/tmp/dfmir_bench", name="bench", capture_step=not args.no_graph)
    torch.manual_seed(0)                      # same weights on every rank (also broadcast in parallelize())
    model = REGISTRATIONModel(opt)
    timer = KernelTimer(["conv3x3_L", "wgrad3x3_L"])
    ops.set_conv_profiler(timer)

    batches = [synth_pairs(B, S, S, dev, 1000 * rank + i) for i in range(4)]   # resident in HBM
    paths = [""] * B

    def feed(i):
        a, b = batches[i % len(batches)]
        return {"A": a, "B": b, "A_paths": paths, "B_paths": paths}

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # keep stdout = the one JSON line
        model.data_dependent_initialize(feed(0))
        model.setup(opt)
        model.parallelize()
    # the step's hipGraph is captured on the third optimize_parameters call: with --warmup < 3 the missing steps are
    # run here as well (untimed), so that the timed region holds K steady-state steps and nothing else
    n_warm = max(args.warmup, 3) if opt.capture_step else args.warmup
    for i in range(n_warm):
        model.set_input(feed(i))
        model.optimize_parameters()

    def fence():
        torch.cuda.synchronize()
        dfdist.barrier()
        torch.cuda.synchronize()

    graphed = bool(opt.capture_step)
    if graphed:
        assert model._graph['graph'] is not None
    fence()
    timer.enabled = not graphed                   # HIP events cannot be recorded inside a graph replay
    t0 = time.perf_counter()
    for i in range(args.steps):
        model.set_input(feed(i))
        model.optimize_parameters()
    host_dt = time.perf_counter() - t0            # host-side enqueue time of the K steps (no device wait inside a step)
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
    losses = model.get_current_losses()
    assert all(v == v and abs(v) < 1e6 for v in losses.values()), losses   # finite
    if graphed:
        # the dominant kernels, timed live with HIP events on their launch stream over eager steps of the same
        # workload right after the timed region (same process, same weights stream, same kernels and shapes)
        model._graph['force_eager'] = True
        timer.enabled = True
        for i in range(args.roofline_steps):
            model.set_input(feed(args.steps + i))
            model.optimize_parameters()
        torch.cuda.synchronize()
        timer.enabled = False
        model._graph['force_eager'] = False

    dt = dfdist.allreduce_max(dt, dev)            # the slowest rank's clock

    # PCIe-inclusive rate: the same step when set_input() is handed pinned host tensors (what a DataLoader with
    # pin_memory delivers, dfmir_amd/data.py): never the headline value, reported as `value_host_inputs`
    host_rate = None
    if args.host_input_steps > 0:
        hb = [tuple(t.cpu().pin_memory() for t in pair) for pair in batches]
        torch.cuda.synchronize()
        th = time.perf_counter()
        for i in range(args.host_input_steps):
            a, b = hb[i % len(hb)]
            model.set_input({"A": a, "B": b, "A_paths": paths, "B_paths": paths})
            model.optimize_parameters()
        torch.cuda.synchronize()
        host_rate = B * world * args.host_input_steps / (time.perf_counter() - th)

    ks = timer.summary()
    result = None
    if rank == 0:
        pairs = B * world * args.steps
        dom = ks.get("conv3x3_L", dict(launches=0, ms=0.0, flops=0.0))
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        wg = ks.get("wgrad3x3_L", dict(launches=0, ms=0.0, flops=0.0))
        wg_tf = wg["flops"] / (wg["ms"] * 1e-3) / 1e12 if wg["ms"] > 0 else 0.0
        # conv FLOPs per pair: 1581 G in the reference's step; 3 of its 6 NCE encoder
        # passes recompute activations the forward pass already holds and are not re-executed here (349 G)
        step_tflop = (1581e9 - 349e9) * B / 1e12
        split = os.environ.get("DFMIR_CONV_FP32") is None
        nprod = 6.0 if os.environ.get("DFMIR_CONV_SPLIT", "f").startswith("b") else 3.0   # bf16x3 / fp16x2 (default)
        form = "bf16x3 (6 products)" if nprod == 6.0 else "scaled fp16x2 (3 products)"
        # The split kernels issue `nprod` 16-bit MFMA products per algorithmic fp32 MAC, with no padded k-step
        # (conv3x3_split_cs_k: 9 taps x 16 channels = 9 k-steps of K = 16 per chunk): issued = nprod x algorithmic.
        peak = FP16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        issued = ach * nprod if split else ach
        wg_issued = wg_tf * nprod if split else wg_tf
        result = {
            "metric": "train-step image-pairs/sec (fwd+bwd): 2D 256x256 bs=16 and 3D 160^3 bs=1",
            "value": pairs / dt, "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "host_enqueue_ms_per_step": 1e3 * host_dt / args.steps,
            "step_submission": "hipGraph replay + eager all-reduce/Adam" if graphed else "eager",
            "value_host_inputs": host_rate, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "2-D %dx%d T1<->T2-shaped synthetic slice pairs, batch %d per GPU, ngf %d: "
                                   "REGISTRATIONModel.set_input+optimize_parameters (ResnetGenerator-9 + PatchNCE + 2-D "
                                   "VoxelMorph + bilinear warps, fwd+bwd+Adam), BASELINE configs[1]" % (S, S, B, args.ngf),
                       "global_batch": B * world, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "achieved": issued, "peak": peak, "unit": "TFLOP/s",
                         "frac": issued / peak,
                         # HBM-side bytes of ONE launch of the dominant shape (256->256 3x3 @64^2, n = 32: 154.6 GFLOP,
                         # 272 MB algorithmic), rocprofv3 PMC passes of scripts/prof_conv.sh (profiles/*_conv3x3s_pmc.md):
                         # FETCH_SIZE + WRITE_SIZE
                         "traffic": CS_TRAFFIC_BYTES if split and nprod == 3.0 else None,
                         "traffic_note": "bytes per launch of the 154.6-GFLOP shape, PMC (FETCH_SIZE + WRITE_SIZE); "
                                         "algorithmic 272 MB",
                         "achieved_note": ("achieved = issued 16-bit matrix FLOP/s = %d x the algorithmic fp32 FLOP/s of the "
                                           "timed launches (%s); peak = dense fp16 MFMA" % (int(nprod), form)) if split else
                                          "achieved = algorithmic fp32 FLOP/s; peak = dense fp32 MFMA",
                         "algorithmic_tflops": ach, "products_per_mac": nprod if split else 1.0,
                         "kernel": ("conv3x3_split_cs_k<true,64,8> (fp32 operands split into 16-bit terms, %s on "
                                    "v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate; 128 couts x 8x32-pixel tile shared by "
                                    "two ping-pong wave groups; reflect dgrads = zero-padded form + ring kernel)" % form if split else
                                    "conv3x3_mfma_k<2,2,2,2,400> (v_mfma_f32_32x32x2_f32; 128 couts x 128 pixels)") +
                                   ": forward + dgrad of every 3x3 conv with Cout > 64",
                         "clock_note": "these kernels hold the package at its 1400 W cap: sclk ~1.75 GHz sustained "
                                       "(profiles/r01_power_clock.md), i.e. a 1.84 PFLOP/s 16-bit roof at that clock",
                         "launches_timed": dom["launches"], "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
                         "timed_over": ("%d eager steps right after the timed region (which replays one hipGraph per step; "
                                        "events cannot be recorded inside a replay)" % args.roofline_steps) if graphed else
                                       "the timed region",
                         "wgrad_kernel": ("conv3x3_wgrad_split2_k (same split; 64 ci x 128 co x 9 taps per workgroup, runs of "
                                          "2 rows x 16 px, double-buffered LDS, staggered wave groups)" if split else "conv3x3_wgrad_k<1,4> (v_mfma_f32_32x32x2_f32)"),
                         "wgrad_achieved": wg_issued, "wgrad_frac": wg_issued / peak,
                         "wgrad_algorithmic_tflops": wg_tf, "wgrad_launches_timed": wg["launches"],
                         "whole_step_tflops": step_tflop * args.steps / dt if S == 256 and args.ngf == 64 else None},
            "losses": {k: round(v, 6) for k, v in losses.items()},
        }
    dfdist.barrier()
    if rank == 0:
        ops.set_conv_profiler(None)
        if world == 1 and not args.no_3d:
            model = None
            batches.clear()
            torch.cuda.empty_cache()
            result["roofline_hbm"] = bench_warp_hbm(dev)
            torch.cuda.empty_cache()
            result["also_3d"] = bench_3d(dev)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(S)
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

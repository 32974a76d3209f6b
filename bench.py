#!/usr/bin/env python
"""bench.py -- throughput of the DFMIR `--model registration` train step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" = REGISTRATIONModel.set_input + optimize_parameters (reference train.py:46-47): forward,
backward and Adam of G/F/R on one batch of synthetic pairs already resident in HBM.  Workload at
every N: BASELINE.json configs[1] geometry -- 2-D 256x256, batch 16 PER GPU, ngf 64, fp32 (weak
scaling; configs[2] is the same thing at N = 8).  One JSON line on rank 0 with
  `roofline`      dominant kernel = conv3x3_split_cs_k (3x3 forward/dgrad convs, fp32 operands split into scaled
                  fp16 pairs, 3 products per MAC on v_mfma_f32_32x32x16_f16), timed live with HIP events on its
                  launch stream; achieved = ALGORITHMIC fp32 FLOP/s, peak = 2.5 PFLOP/s dense fp16;
                  `issued_frac` (= 3 x frac, what the matrix pipe executes) beside it; `traffic` = HBM-side bytes per
                  launch from the committed PMC passes (profiles/rNN_pmc.json)
  `roofline_hbm`  the trilinear warp (grid_sample) forward / backward at 160x192x224, algorithmic bytes / HIP-event time
                  against 8 TB/s, measured in the same process, on a smooth and on a rough field -- COLD (four rotating
                  buffer sets: > 256 MiB between two uses of a buffer) with the cache-warm figure beside it as `warm`
  `also_3d`       the 3-D step of configs[4] geometry on one GPU with its own roofline (the tiled conv3d_split_k /
                  conv3d_split_m16_k and the z-marching conv3d_march_k forward / dgrad kernels, conv3d_wgrad_tr_k, all scaled
                  fp16x2 on the 16-bit matrix pipe)
  `value_host_inputs`, `value_pil_loader`  the same step fed from pinned host tensors / from the plugin's own PNG -> PIL ->
                  transforms DataLoader (worker processes): PCIe- and pipeline-inclusive rates, never the headline
  `also_3d_128`   configs[3]: the 3-D step at 128^3 with the plugin's 6-level U-Net features
  `cpu_baseline`  (N = 1) the CPU oracle = a port of the reference's PyTorch-CPU path, timed on the host cores on a
                  bounded sample: the 2-D step at 256^2 batch 1, (`also_3d_128`) the 3-D step at 128^3 and (`also_3d_big`) one
                  3-D step at 160x192x224.
  `step_ms`       median / min / max of the K timed steps (HIP events between steps).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz
FP16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (v_mfma_f32_32x32x16_f16), 2.4 GHz
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E
G_FULL_GF, G_ENC_GF = 126.61, 68.06   # ResnetGenerator conv GFLOP per image: full pass / encoder-only pass
STEP_GF_REFERENCE = 1581.0            # per pair, the reference's step
PLUGIN_FEATS = [[16, 32, 32, 64, 64, 64], [64, 64, 64, 32, 32, 32, 16]]   # registration_model.py:93-96


def load_pmc():
    """HBM-side bytes per launch of the priced kernels: the newest profiles/rNN_pmc.json (written by
    scripts/pmc_json.py from the rocprofv3 --pmc passes of scripts/collect_profiles.sh; FETCH_SIZE x2 gfx950
    correction, KiB -> bytes).  Counters cannot be read from inside the process, so `traffic` is the committed
    measurement of the same kernels on the same shapes; {} when no file is there (traffic: null)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_pmc.json")))
    if not files:
        return {}, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], REPO)
    except Exception:
        return {}, None


def pmc_traffic(pmc, key):
    e = pmc.get(key) or {}
    return e.get("traffic_bytes")


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.lower().startswith("model name"):
                return l.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


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
    accepts_issued = True       # ops passes the products a split kernel really issues (tiling padding included)

    def __init__(self, kinds, every=1):
        self.kinds = set(kinds)
        self.records = {}
        self.enabled = False
        self.every = every          # bracket every n-th launch of a kind (keeps the events out of the step's way)
        self.seen = {}

    def __call__(self, kind, flops, launch, issued=None):
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
        self.records.setdefault(kind, []).append((s, e, flops, issued if issued is not None else 0.0))

    def summary(self):
        out = {}
        for kind, recs in self.records.items():
            ms = sum(r[0].elapsed_time(r[1]) for r in recs)
            fl = sum(r[2] for r in recs)
            out[kind] = dict(launches=len(recs), ms=ms, flops=fl, issued=sum(r[3] for r in recs))
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


def cpu_baseline(size, max_steps=3, cores=None):
    """The oracle (CPU port of the reference path) at the same 256x256 geometry, batch 1."""
    from oracle import dfmir_oracle as O
    cores = cores or pick_cpu_threads()
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
    return dict(value=n / dt, unit="image-pairs/s", cores=cores, kind="port", cpu_model=cpu_model(),
                sample="%d train steps of the CPU oracle (PyTorch fp32, %d threads) at 2-D %dx%d batch 1, ngf 64, "
                       "after 1 warm-up fwd+bwd" % (n, cores, size, size))


def cpu_baseline_3d(cores, shape=(128, 128, 128), budget_s=45.0):
    """The 3-D half of the metric on the host cores: the oracle's Registration3DStep (VxmDense with the plugin's
    6-level features + NCC_Loss[9,9,9] + Grad_Loss l2 + Adam; SURVEY section 8 A13 / BASELINE.md section 4) at
    BASELINE configs[3] geometry, 128^3, batch 1.  Bounded: one warm-up step, then steps until `budget_s`."""
    from oracle import dfmir_oracle as O
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    st = O.Registration3DStep(shape, PLUGIN_FEATS)
    g = torch.Generator(device="cpu")
    g.manual_seed(1)
    A = torch.rand(1, 1, *shape, generator=g) * 2 - 1
    B = 0.5 * A + 0.5 * (torch.rand(1, 1, *shape, generator=g) * 2 - 1)
    tw = time.time()
    st.step(A, B)                            # warm-up
    tw = time.time() - tw
    t0 = time.time()
    n = 0
    while n < 3:
        st.step(A, B)
        n += 1
        if time.time() - t0 + tw > budget_s:
            break
    dt = time.time() - t0
    return dict(value=n / dt, unit="image-pairs/s", cores=cores, kind="port", cpu_model=cpu_model(),
                sample="%d train steps of the CPU oracle's Registration3DStep (PyTorch fp32, %d threads) at 3-D %dx%dx%d "
                       "batch 1, plugin 6-level features, NCC[9,9,9] + Grad-l2 + Adam, after 1 warm-up step"
                       % ((n, cores) + tuple(shape)))


def cpu_baseline_3d_big(cores, shape=(160, 192, 224)):
    """BASELINE.md section 4's headline 3-D shape on the host cores: ONE bounded train step of the oracle's Registration3DStep
    at 160x192x224 with the default U-Net features (the 6-level list does not divide this shape), no warm-up
    step (a step is 15-40 s of CPU work; the first step also pays the allocator's first touches)."""
    from oracle import dfmir_oracle as O
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    st = O.Registration3DStep(shape, None)
    g = torch.Generator(device="cpu")
    g.manual_seed(1)
    A = torch.rand(1, 1, *shape, generator=g) * 2 - 1
    B = 0.5 * A + 0.5 * (torch.rand(1, 1, *shape, generator=g) * 2 - 1)
    t0 = time.time()
    st.step(A, B)
    dt = time.time() - t0
    return dict(value=1.0 / dt, unit="image-pairs/s", cores=cores, kind="port", cpu_model=cpu_model(),
                sample="1 train step (no warm-up) of the CPU oracle's Registration3DStep (PyTorch fp32, %d threads) at 3-D "
                       "%dx%dx%d batch 1, default U-Net features, NCC[9,9,9] + Grad-l2 + Adam" % ((cores,) + tuple(shape)))


def bench_3d(dev, pmc, shape=(160, 192, 224), feats=None, gflop_step=2393.0, label="BASELINE configs[4] geometry, one GPU",
             pmc_key="conv3d_split_k_34_32", steps=5, warmup=3, capture=False, roofline_steps=2, rough_steps=0):
    """Auxiliary line: the 3-D step on ONE GPU -- batch 1, VxmDense(int_steps 7, bidir) + NCC[9,9,9] + Grad l2,
    fwd+bwd+Adam: 160x192x224 with the default U-Net features (configs[4]'s per-GPU shard) or
    128^3 with the plugin's 6-level features (configs[3]).  Its roofline: conv3d_split_k (every stride-1 3x3x3 forward /
    dgrad conv) and conv3d_wgrad_tr_k, timed with HIP events on the launch stream -- inside the timed region when the
    step is enqueued eagerly, over `roofline_steps` eager steps right after it when it is one hipGraph replay."""
    from dfmir_amd import ops
    from dfmir_amd.registration3d import Registration3DModel
    torch.manual_seed(0)
    m = Registration3DModel(shape, feats, device=dev, capture_step=capture)
    A = torch.rand(1, 1, *shape, device=dev) * 2 - 1
    B = 0.5 * A + 0.5 * (torch.rand(1, 1, *shape, device=dev) * 2 - 1)
    sizes = ("small", "S", "M", "L")
    sp_kinds = ["conv3ds_" + z for z in sizes]          # split fp16x2 kernel (csrc/conv3ds.hip)
    fw_kinds = ["conv3d_" + z for z in sizes]           # fp32-MFMA kernel (csrc/conv3d.hip): A/B env only
    wg_kinds = ["wgrad3d_" + z for z in ("S", "M", "L")]      # fp32-MFMA wgrad (Cin > 128 layers)
    wgs_kinds = ["wgrad3ds_" + z for z in ("S", "M", "L")]    # split fp16x2 wgrad
    wgu_kinds = ["wgrad3dup_" + z for z in ("S", "M", "L")]   # ... of cat(up2(a), b) in parity classes (csrc/conv3duw.hip)
    up_kinds = ["conv3dup_" + z for z in sizes]         # parity-class kernels of the nearest_up2 + cat layers
    timer = KernelTimer(sp_kinds + fw_kinds + wg_kinds + wgs_kinds + wgu_kinds + up_kinds)
    ops.set_conv_profiler(timer)
    for _ in range(max(warmup, 3) if capture else warmup):
        m.set_input({"A": A, "B": B})
        m.optimize_parameters()
    torch.cuda.synchronize()
    graphed = bool(capture and m._graph['graph'] is not None)
    timer.enabled = not graphed
    t0 = time.perf_counter()
    for _ in range(steps):
        m.set_input({"A": A, "B": B})
        m.optimize_parameters()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if dt * steps < 0.25 and graphed:
        # a 3 ms step timed over 20 steps is 60 ms of wall clock: one host hiccup is 10 % (3.28 ms on the line of a run whose
        # A/B processes measured 2.81 three times).  Short timed regions are repeated over five times the steps.
        steps *= 5
        t0 = time.perf_counter()
        for _ in range(steps):
            m.set_input({"A": A, "B": B})
            m.optimize_parameters()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    timer.enabled = False
    if graphed:
        m._graph['force_eager'] = True
        timer.enabled = True
        for _ in range(roofline_steps):
            m.set_input({"A": A, "B": B})
            m.optimize_parameters()
        torch.cuda.synchronize()
        timer.enabled = False
        m._graph['force_eager'] = False
    ops.set_conv_profiler(None)
    losses = m.get_current_losses()
    assert all(v == v and abs(v) < 1e6 for v in losses.values()), losses   # finite
    rough = None
    if rough_steps > 0:
        # The headline step runs on freshly initialised weights, whose flow head N(0, 1e-5) makes the
        # field ~0: the windowed warp kernels' best case.  Second figure: the same step with the flow head rescaled as in the
        # parity runs (weight x 1e5, bias N(0, 1)): |phi| of a few voxels through the 7 integration steps and both warps.
        # Calibrated, not guessed: weight x 1e5 alone gave |phi| ~ 47 voxels (mean) through the integration; the head is
        # rescaled twice by 2 / mean|phi| of a probe step, so the timed field is ~2 voxels mean (reported below).
        with torch.no_grad():
            m.netR.flow.weight.mul_(1e5)
            m.netR.flow.bias.copy_(0.25 * torch.randn(m.netR.flow.bias.shape, generator=torch.Generator().manual_seed(8)).to(dev))
        for _ in range(3):
            ops.bump_weights_epoch()
            with torch.no_grad():
                cur = float(m.netR(A, B)[2].abs().mean())
            with torch.no_grad():
                f = 2.0 / max(cur, 1e-6)
                m.netR.flow.weight.mul_(f)
                m.netR.flow.bias.mul_(f)
        ops.bump_weights_epoch()
        m._graph['force_eager'] = True            # (other weights than the captured step's history; timed eagerly)
        for _ in range(2):
            m.set_input({"A": A, "B": B})
            m.optimize_parameters()
        torch.cuda.synchronize()
        tr = time.perf_counter()
        for _ in range(rough_steps):
            m.set_input({"A": A, "B": B})
            m.optimize_parameters()
        torch.cuda.synchronize()
        dtr = (time.perf_counter() - tr) / rough_steps
        with torch.no_grad():
            fl = m.flow.detach() if hasattr(m, "flow") and torch.is_tensor(getattr(m, "flow", None)) else None
        rl = m.get_current_losses()
        rough = {"ms_per_step": 1e3 * dtr, "value": 1.0 / dtr, "unit": "image-pairs/s", "steps": rough_steps,
                 "flow_abs_mean_voxels": float(fl.abs().mean()) if fl is not None else None,
                 "flow_abs_max_voxels": float(fl.abs().max()) if fl is not None else None,
                 "losses": {k: round(v, 6) for k, v in rl.items()},
                 "note": "same step, flow head rescaled so that mean |phi| of the integrated field is ~2 voxels when the timing "
                         "starts (calibrated on probe forwards; Adam moves it during the timed steps); eager submission"}
        m._graph['force_eager'] = False
    ks = timer.summary()

    def agg(kinds):
        ms = sum(ks[k]["ms"] for k in kinds if k in ks)
        fl = sum(ks[k]["flops"] for k in kinds if k in ks)
        n = sum(ks[k]["launches"] for k in kinds if k in ks)
        return (fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0), n, ms
    sp_tf, sp_n, sp_ms = agg(sp_kinds)
    sp_issued = sum(ks[k]["issued"] for k in sp_kinds if k in ks) / (sp_ms * 1e-3) / 1e12 if sp_ms > 0 else 0.0
    fw_tf, fw_n, fw_ms = agg(fw_kinds)
    wg_tf, wg_n, wg_ms = agg(wg_kinds)
    wgs_tf, wgs_n, wgs_ms = agg(wgs_kinds + wgu_kinds)
    wgs_issued = (sum(ks[k]["issued"] for k in wgs_kinds + wgu_kinds if k in ks) / (wgs_ms * 1e-3) / 1e12) if wgs_ms > 0 else 0.0
    wgu_tf, wgu_n, wgu_ms = agg(wgu_kinds)
    up_tf, up_n, up_ms = agg(up_kinds)
    timed_over = ("%d eager steps right after the timed region (one hipGraph replay per step)" % roofline_steps) if graphed \
        else "the timed region"
    if sp_n:
        roof = {"bound": "mfma", "achieved": sp_tf, "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": sp_tf / FP16_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(pmc, pmc_key),
                "traffic_note": "HBM-side bytes of one launch of %s (PMC FETCH_SIZE x2 + WRITE_SIZE)" % pmc_key
                                if pmc_traffic(pmc, pmc_key) else None,
                "achieved_note": "achieved = ALGORITHMIC fp32 FLOP/s (2*N*Cout*D*H*W*Cin*27 of the timed launches / their HIP-event "
                                 "time) over the dense fp16 MFMA peak; the kernel issues 3 fp16 products per fp32 MAC (scaled "
                                 "fp16x2 split) and pads taps 27 -> 28 (plane-pair form for <= 16 output channels: 36 taps', "
                                 "32 rows for 2 x cout), input channels to chunks of 8 and output channels to 32 rows: "
                                 "issued_* count every product the matrix pipe executes",
                "issued_tflops": sp_issued, "issued_frac": sp_issued / FP16_MFMA_PEAK_TFLOPS, "products_per_mac": 3.0,
                "frac_of_fp32_mfma_peak": sp_tf / FP32_MFMA_PEAK_TFLOPS,
                "kernel": "conv3d_split_k / conv3d_split_m16_k (4x8x16-voxel tiles, LDS halo patch split into fp16 pairs per "
                          "8-channel chunk) and, for the full-resolution layers with Cin x Cout <= 512, conv3d_march_k (16x32 "
                          "columns marched along z, weights resident in LDS): forward + dgrad of every stride-1 3x3x3 conv",
                "march_traffic": {k_: pmc_traffic(pmc, k_) for k_ in ("conv3d_march_k_32_16", "conv3d_march_k_16_16", "conv3d_march_k_16_32")},
                "launches_timed": sp_n, "avg_launch_ms": sp_ms / max(sp_n, 1), "timed_over": timed_over}
    else:
        roof = {"bound": "mfma", "achieved": fw_tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fw_tf / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                "kernel": "conv3d_mfma16_k (v_mfma_f32_16x16x4_f32): forward + dgrad of every stride-1 3x3x3 conv",
                "launches_timed": fw_n, "avg_launch_ms": fw_ms / max(fw_n, 1), "timed_over": timed_over}
    if wgs_n:
        roof.update({"wgrad_kernel": "conv3d_wgrad_tr_k (voxels as the matrix K; operands read with ds_read_b64_tr_b16 from "
                                     "channel-major fp16-pair LDS images of the patch and of dY; plane-pair columns for "
                                     "<= 16 output channels)",
                     "wgrad_achieved": wgs_tf, "wgrad_frac": wgs_tf / FP16_MFMA_PEAK_TFLOPS,
                     "wgrad_issued_frac": wgs_issued / FP16_MFMA_PEAK_TFLOPS, "wgrad_launches_timed": wgs_n,
                     "wgrad_note": "achieved = REFERENCE-EQUIVALENT FLOP/s; issued counts the products executed: 3 per MAC x tile "
                                   "padding, and for the top-level layer over cat(up2(a), b) 8/27 of the up-sampled share "
                                   "(conv3d_upwgrad4_k: parity classes, skip channels and bias gradient fused)",
                     "wgrad_upcat": {"kernel": "conv3d_upwgrad4_k", "launches_timed": wgu_n,
                                     "avg_launch_ms": wgu_ms / max(wgu_n, 1), "reference_equivalent_tflops": wgu_tf}})
    else:
        roof.update({"wgrad_kernel": "conv3d_wgrad16_k (v_mfma_f32_16x16x4_f32)", "wgrad_achieved": wg_tf,
                     "wgrad_frac_of_fp32_peak": wg_tf / FP32_MFMA_PEAK_TFLOPS, "wgrad_launches_timed": wg_n})
    if up_n:
        roof["up_phase"] = {
            "kernel": "conv3d_up_phase_k / conv3d_up_dgrad_k: forward and d(a) of the ConvBlocks that consume "
                      "cat(nearest_up2(a), b) in parity classes (the 27 taps over the up-sampled tensor fall on 2x2x2 "
                      "low-resolution voxels: 8/27 of the reference's products, no up-sampled / concatenated tensor)",
            "reference_equivalent_tflops": up_tf, "executed_fraction_of_reference_flops": 8.0 / 27.0,
            "executed_tflops": up_tf * 8.0 / 27.0, "issued_frac": 3.0 * up_tf * 8.0 / 27.0 / FP16_MFMA_PEAK_TFLOPS,
            "launches_timed": up_n, "ms": up_ms}
    return {"workload": "3-D %dx%dx%d volume pair, batch 1, VxmDense %s features + NCC[9,9,9] + Grad-l2, fwd+bwd+Adam (%s)"
                        % (tuple(shape) + ("default" if feats is None else "plugin 6-level", label)),
            "value": 1.0 / dt, "unit": "image-pairs/s", "ms_per_step": 1e3 * dt, "steps": steps, "warmup": warmup,
            "step_submission": "hipGraph replay + eager Adam" if graphed else "eager",
            "conv_tflops": gflop_step / dt / 1e3, "conv_gflop_per_step": gflop_step,
            "dtype": "f32 (fp32 MFMA)" if (os.environ.get("DFMIR_CONV3D_FP32") or os.environ.get("DFMIR_CONV_FP32")) else "f32 (fp16x2-split MFMA, fp32 accumulate)",
            "losses": {k: round(v, 6) for k, v in losses.items()}, "roofline": roof,
            "field_note": "timed on freshly initialised weights: flow head N(0, 1e-5), i.e. a near-identity field (SURVEY section 8 "
                          "D2 keeps the headline on init weights); `rough_field` is the same step on a deformation of a few voxels",
            "rough_field": rough}


def bench_warp_hbm(dev, pmc, reps=20, nset=4):
    """roofline_hbm: the trilinear displacement-field warp (SpatialTransformer = grid_sample, reference
    models/voxelmorph/torchvoxelmorph/layers.py:30-48) at 160x192x224, C = 1, on a registration-like smooth field
    (control points every 32 voxels, ~1 voxel rms).  Algorithmic bytes: fwd 4*(C+nd+C)*N_vox,
    bwd 4*(C [dOut] + C [src] + nd [flow] + C [dSrc] + nd [dFlow])*N_vox; time = HIP events on the launch stream.

    COLD figures (`achieved`, `frac`): the launches cycle through `nset` = 4 independent buffer sets (inputs AND outputs), so
    that 413 MB (forward) / 744 MB (backward) of other data pass through the chip between two uses of the same buffer -- more
    than the 256 MiB Infinity Cache, whose hits the fabric counters do not separate from HBM.  `warm` = the round-4 way (the
    same 138 / 248 MB working set every launch, largely cache-served) is kept beside it for comparison."""
    from dfmir_amd import ops
    sp, C, nd = (160, 192, 224), 1, 3
    g = torch.Generator(device="cpu")
    g.manual_seed(5)

    def field(cell, amp):
        coarse = torch.randn(1, nd, *[s // cell for s in sp], generator=g).to(dev) * amp
        return torch.nn.functional.interpolate(coarse, size=sp, mode='trilinear', align_corners=True).contiguous()

    sets = []
    for i in range(nset):
        src = torch.randn(1, C, *sp, generator=g).to(dev)
        sets.append({"src": src, "flow": field(32, 1.0), "rough": field(16, 3.0), "dout": torch.randn(1, C, *sp, generator=g).to(dev),
                     "dflow": torch.empty(1, nd, *sp, device=dev)})
    nv = sets[0]["src"].numel() // C
    keep = []                                                  # results stay alive: the allocator cannot hand the same block back

    def timeit(fn, cold):
        n_ = nset if cold else 1
        for i in range(3):
            keep.append(fn(sets[i % n_]))
            del keep[:-nset]
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(reps):
            keep.append(fn(sets[i % n_]))
            del keep[:-(nset if cold else 1)]
        e.record()
        torch.cuda.synchronize()
        del keep[:]
        return s.elapsed_time(e) / reps

    def fwd(fl):
        return lambda S: ops._warp_fwd(S["src"], S[fl], 0, 0)

    def bwd(fl):
        return lambda S: ops._warp_bwd_dsrc(S["dout"], S["src"], S[fl], S["dflow"], 0, 0)

    def rate(nbytes, ms):
        return {"achieved": nbytes / ms / 1e6, "frac": nbytes / ms / 1e6 / HBM_PEAK_GBS, "avg_launch_ms": ms}

    b_f, b_b, b_bf = 4 * (C + nd + C) * nv, 4 * (C + 2 * C + 2 * nd) * nv, 4 * (C + C + 2 * nd) * nv
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": pmc_traffic(pmc, "warp_win_fwd_k"),
           "workload": "160x192x224, C=1, smooth field; %d rotating buffer sets (cold: > 256 MiB between reuses)" % nset,
           "kernel": "warp_win_fwd_k<3>", "bytes": b_f,
           "cache_note": "achieved / frac are COLD (every launch's inputs and outputs were evicted from the 256 MiB Infinity "
                         "Cache by the launches in between); `warm` re-uses one 138 MB working set (round 4's figure)"}
    out.update(rate(b_f, timeit(fwd("flow"), True)))
    out["warm"] = rate(b_f, timeit(fwd("flow"), False))
    out["bwd"] = dict(rate(b_b, timeit(bwd("flow"), True)), bytes=b_b, traffic=pmc_traffic(pmc, "warp_bwd_dsrc_dflow"),
                      kernel="warp_win_bwd_own_k<3> + warp_win_gather_k<3> (+ the empty slow-voxel pass): d(src) + d(flow) "
                             "without device-scope atomics, bit-reproducible",
                      warm=rate(b_b, timeit(bwd("flow"), False)))
    out["bwd_dflow_only"] = dict(rate(b_bf, timeit(lambda S: ops._warp_bwd(S["dout"], S["src"], S["flow"], None, S["dflow"], 0, 0), True)),
                                 bytes=b_bf)
    # the same launches on a ROUGH field (control points every 16 voxels, 3 voxels rms: many taps leave their tile's
    # window) -- the worst case of the windowed kernels; a field under a smoothness loss is of the first kind
    out["rough_field"] = {"workload": "160x192x224, C=1, control points every 16 voxels, 3 voxels rms (cold)",
                          "fwd": rate(b_f, timeit(fwd("rough"), True)), "bwd": rate(b_b, timeit(bwd("rough"), True))}
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
    ap.add_argument("--pil-workers", type=int, default=4, help="DataLoader worker processes of the value_pil_loader measurement (0 = skip)")
    ap.add_argument("--bucket-allreduce", action="store_true",
                    help="N > 1: G's late layers (45 %% of its arena) start their all-reduce inside backward (opt.bucket_allreduce; "
                         "inside a captured step only with DFMIR_BUCKET_IN_GRAPH=1)")
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
                          checkpoints_dir="/tmp/dfmir_bench", name="bench", capture_step=not args.no_graph,
                          bucket_allreduce=args.bucket_allreduce)
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

    graphed = bool(opt.capture_step) and model._graph['graph'] is not None
    if opt.capture_step and not graphed:
        print("bench: the step was not captured (%s); timing the eager step" % model._graph.get('capture_error', 'n/a'),
              file=sys.stderr)
    distributed = dfdist.is_distributed()         # world > 1 (or a forced single-rank group: tests)
    if distributed:
        model._collective_timing = []             # event pairs around every wait for a gradient exchange
    fence()
    timer.enabled = not graphed                   # HIP events cannot be recorded inside a graph replay
    # one HIP event between consecutive steps (recorded on the step's stream, no host wait): per-step device times for
    # the median / spread; `value` stays the wall clock of the whole region
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        model.set_input(feed(i))
        model.optimize_parameters()
        marks[i + 1].record()
    host_dt = time.perf_counter() - t0            # host-side enqueue time of the K steps (no device wait inside a step)
    fence()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    timer.enabled = False
    coll_pairs = getattr(model, '_collective_timing', None)
    model._collective_timing = None
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
        # ... and once more with netR on the SAME stream (opt.overlap_registration off): the launch durations without
        # the share of the CUs the second stream takes -> roofline.single_stream
        ks_overlapped = timer.summary()
        timer.records = {}
        prev_ov = getattr(opt, 'overlap_registration', True)
        opt.overlap_registration = False
        timer.enabled = True
        for i in range(args.roofline_steps):
            model.set_input(feed(args.steps + args.roofline_steps + i))
            model.optimize_parameters()
        torch.cuda.synchronize()
        timer.enabled = False
        ks_single = timer.summary()
        timer.records = {}
        opt.overlap_registration = prev_ov
        model._graph['force_eager'] = False

    dt = dfdist.allreduce_max(dt, dev)            # the slowest rank's clock
    collective = None
    if distributed:
        exposed = sum(s_.elapsed_time(e_) for s_, e_ in (coll_pairs or [])) / max(args.steps, 1)
        ones = torch.ones(1, device=dev)
        if dfdist._staged(ones):                  # gloo over one device (tests): staged through the host
            h_ = ones.cpu()
            torch.distributed.all_reduce(h_)
            ones = h_
        else:
            torch.distributed.all_reduce(ones)    # every rank contributes 1 through the data-path backend
        collective = {"backend": torch.distributed.get_backend(), "rccl_ranks_seen": int(round(float(ones))),
                      "patch_ids": "one draw shared by all ranks (same generator seed on every rank): the reference draws "
                                   "one id set per forward and applies it to its whole gathered DataParallel batch "
                                   "(models/networks.py:577,609-611)",
                      "payload_bytes": 4 * sum(o.flat_g.numel() for o in model.optimizers),
                      "arenas_bytes": [4 * o.flat_g.numel() for o in model.optimizers],
                      "exposed_ms_per_step": dfdist.allreduce_max(exposed, dev),
                      # per-rank median step time (HIP events on each rank's stream), rank order: the spread says whether one
                      # GPU of the node holds the others back (value is computed from the slowest rank's wall clock)
                      "step_ms_by_rank": dfdist.allgather_float(
                          step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2]), dev),
                      "early_bucket": (None if getattr(model, '_bucket', None) is None else
                                       {"arena": "G", "bytes": 4 * (model.optimizer_G.flat_g.numel() - model._bucket['off']),
                                        "fired_steps": model._bucket['fired'],
                                        "what": "modules %d.. of G all-reduced from inside backward" % model._bucket['idx']}),
                      "note": "one all-reduce per network arena (G, R, F) issued back to back after the graph replay; "
                              "exposed = the compute stream's wait for each arena (HIP event pairs), max over ranks; R's and "
                              "F's exchange ride under G's Adam launch"}

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

    # ... and the same step fed by the plugin's own data pipeline (row N3): PNG slices on disk -> PIL decode -> grayscale ->
    # bicubic resize 286 -> random crop 256 -> flip -> [-1, 1] -> pinned batch, in DataLoader worker processes beside the
    # training process (dfmir_amd/data.py = data/unaligned_dataset.py + base_dataset.py:82-145) -> `value_pil_loader`
    pil_rate = None
    if args.host_input_steps > 0 and args.pil_workers > 0:
        import copy
        import tempfile
        from PIL import Image
        from dfmir_amd.data import create_dataset
        root = tempfile.mkdtemp(prefix="dfmir_bench_data_")
        rs = np.random.RandomState(11 + rank)
        n_img = B * (args.host_input_steps + 3)
        for sub in ("trainA", "trainB"):
            os.makedirs(os.path.join(root, sub))
            for i in range(n_img):
                base = rs.randint(0, 256, size=(9, 9)).astype(np.uint8)
                Image.fromarray(base).resize((300, 300), Image.BICUBIC).save(os.path.join(root, sub, "%04d.png" % i))
        lo = copy.copy(opt)
        lo.dataroot, lo.phase, lo.num_threads, lo.serial_batches = root, "train", args.pil_workers, False
        lo.load_size, lo.crop_size, lo.preprocess, lo.no_flip, lo.max_dataset_size = 286, args.size, "resize_and_crop", False, float("inf")
        it = iter(create_dataset(lo))
        first = next(it)                                        # workers started, first batches in flight
        model.set_input(first); model.optimize_parameters()
        torch.cuda.synchronize()
        th = time.perf_counter()
        n_done = 0
        for data in it:
            model.set_input(data)
            model.optimize_parameters()
            n_done += 1
            if n_done >= args.host_input_steps:
                break
        torch.cuda.synchronize()
        pil_rate = B * world * n_done / (time.perf_counter() - th) if n_done else None
        del it
    pil_by_rank = None
    if distributed and world > 1:
        # every rank's own loader-fed rate (x world, i.e. comparable with `value`): 8 ranks x (1 training process + pil_workers
        # loader processes) share the host's cores
        pil_by_rank = dfdist.allgather_float(float(pil_rate or 0.0), dev)

    ks = ks_overlapped if graphed else timer.summary()
    ks1 = ks_single if graphed else {}
    result = None
    if rank == 0:
        pairs = B * world * args.steps
        dom = ks.get("conv3x3_L", dict(launches=0, ms=0.0, flops=0.0))
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        wg = ks.get("wgrad3x3_L", dict(launches=0, ms=0.0, flops=0.0))
        wg_tf = wg["flops"] / (wg["ms"] * 1e-3) / 1e12 if wg["ms"] > 0 else 0.0
        # conv FLOPs per pair: 1581 G in the reference's step; its 3 key-side NCE encoder passes
        # (forward only, 3 x 68.06 G: SURVEY 3.3 / section 8 A4) recompute activations forward() already holds and are
        # not re-executed here
        step_gf = STEP_GF_REFERENCE - 3.0 * G_ENC_GF
        step_tflop = step_gf * B / 1e3
        split = os.environ.get("DFMIR_CONV_FP32") is None
        nprod = 6.0 if os.environ.get("DFMIR_CONV_SPLIT", "f").startswith("b") else 3.0   # bf16x3 / fp16x2 (default)
        form = "bf16x3 (6 products)" if nprod == 6.0 else "scaled fp16x2 (3 products)"
        # The split kernels issue `nprod` 16-bit MFMA products per algorithmic fp32 MAC, with no padded k-step
        # (conv3x3_split_cs_k: 9 taps x 16 channels = 9 k-steps of K = 16 per chunk): issued = nprod x algorithmic.
        # achieved = ALGORITHMIC FLOP/s, peak = that of the pipe the kernel issues on.
        peak = FP16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
        npm = nprod if split else 1.0
        pmc, pmc_file = load_pmc()
        traffic = pmc_traffic(pmc, "conv3x3_split_cs_k") if split and nprod == 3.0 else None
        result = {
            "metric": "train-step image-pairs/sec (fwd+bwd): 2D 256x256 bs=16 and 3D 160^3 bs=1",
            "value": pairs / dt, "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "step_ms": {"median": step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2]),
                        "min": step_ms[0], "max": step_ms[-1],
                        "note": "rank 0's device time between HIP events recorded after every step of the timed region (no host "
                                "wait between steps); ms_per_step / value are the barrier-to-barrier wall clock, max over ranks"},
            "host_enqueue_ms_per_step": 1e3 * host_dt / args.steps,
            "step_submission": (("hipGraph replay (%d single-stream graphs) + eager all-reduce/Adam" % len(model._graph['graph'])
                                 if isinstance(model._graph['graph'], list) else "hipGraph replay + eager all-reduce/Adam")
                                if graphed else "eager"),
            "host_enqueue_note": ("time the host spends inside the submission calls of a step.  The two-stream step is submitted as "
                                  "eight single-stream hipGraphs with event edges between the launches (opt.staged_step, "
                                  "registration_model._step_pieces): every launch returns at once.  DFMIR_NO_STAGED=1 = ONE graph "
                                  "with parallel branches, whose hipGraphLaunch (ROCm 7.2) returns only when the second branch has "
                                  "been handed to the GPU: ~39 ms of a 74 ms step (profiles/r06_ab_staged.txt)"),
            "value_host_inputs": host_rate,
            "value_pil_loader": pil_rate,       # fed by dfmir_amd.data's DataLoader (PNG decode + transforms in worker processes)
            "value_pil_loader_by_rank": pil_by_rank,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (fp16x2-split MFMA, fp32 accumulate)" if (split and nprod == 3.0) else
                      ("f32 (bf16x3-split MFMA, fp32 accumulate)" if split else "f32 (fp32 MFMA)")),
            "data": "synthetic",
            "config": {"workload": "2-D %dx%d T1<->T2-shaped synthetic slice pairs, batch %d per GPU, ngf %d: "
                                   "REGISTRATIONModel.set_input+optimize_parameters (ResnetGenerator-9 + PatchNCE + 2-D "
                                   "VoxelMorph + bilinear warps, fwd+bwd+Adam), BASELINE configs[1]" % (S, S, B, args.ngf),
                       "global_batch": B * world, "parallelism": "dp%d" % world},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach / peak,
                         "issued_tflops": ach * npm, "issued_frac": ach * npm / peak, "products_per_mac": npm,
                         "frac_of_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                         # HBM-side bytes of ONE launch of the dominant shape (256->256 3x3 @64^2, n = 32: 154.6 GFLOP,
                         # 272 MB algorithmic), rocprofv3 PMC passes of scripts/prof_conv.sh -> profiles/rNN_pmc.json
                         "traffic": traffic,
                         "traffic_note": ("bytes per launch of the 154.6-GFLOP shape, PMC FETCH_SIZE x2 (gfx950 correction) + "
                                          "WRITE_SIZE, from %s; algorithmic 272 MB" % pmc_file) if traffic else None,
                         "achieved_note": ("achieved = ALGORITHMIC fp32 FLOP/s of the timed launches (2*N*Cout*H*W*Cin*9 / "
                                           "HIP-event time); peak = dense fp16 MFMA.  The kernel computes in fp32-grade "
                                           "precision by issuing %d 16-bit products per MAC (%s): the matrix pipe runs at "
                                           "issued_frac = %d x frac" % (int(nprod), form, int(nprod))) if split else
                                          "achieved = algorithmic fp32 FLOP/s; peak = dense fp32 MFMA",
                         "kernel": ("conv3x3_split_cs_k<true,64,8> (fp32 operands split into 16-bit terms, %s on "
                                    "v_mfma_f32_32x32x16_{f16,bf16}, fp32 accumulate; 128 couts x 8x32-pixel tile shared by "
                                    "two ping-pong wave groups; reflect dgrads = zero-padded form + ring kernel)" % form if split else
                                    "conv3x3_mfma_k<2,2,2,2,400> (v_mfma_f32_32x32x2_f32; 128 couts x 128 pixels)") +
                                   ": forward + dgrad of every 3x3 conv with Cout > 64",
                         "clock_note": "these kernels hold the package at its 1400 W cap: sclk ~1.75 GHz sustained "
                                       "(profiles/r01_power_clock.md), i.e. a 1.84 PFLOP/s 16-bit roof at that clock",
                         "overlap_note": ("since round 3 netR's kernels run on a second stream beside these launches "
                                          "(opt.overlap_registration): a launch's duration includes the share of the CUs they take -- "
                                          "with DFMIR_NO_OVERLAP_R=1 the same kernel measures ~371 TF / issued 0.445 "
                                          "(profiles/README.md) and the step is 1.3 ms longer"),
                         "single_stream": ({k2: (ks1[k1]["flops"] / (ks1[k1]["ms"] * 1e-3) / 1e12 * npm / peak) if k1 in ks1 and ks1[k1]["ms"] > 0 else None
                                            for k1, k2 in (("conv3x3_L", "issued_frac"), ("wgrad3x3_L", "wgrad_issued_frac"))}
                                           if ks1 else None),
                         "single_stream_note": "the same launches over as many eager steps with opt.overlap_registration off (netR on "
                                               "the generator's stream): the kernels' own rate, without the CUs the second stream takes",
                         "launches_timed": dom["launches"], "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
                         "timed_over": ("%d eager steps right after the timed region (which replays one hipGraph per step; "
                                        "events cannot be recorded inside a replay); the rocprofv3 kernel trace of the "
                                        "replays themselves is profiles/*_bench_b16_kernel_stats.csv and agrees to ~3 %%"
                                        % args.roofline_steps) if graphed else "the timed region",
                         "wgrad_kernel": ("conv3x3_wgrad_split2_k (same split; 64 ci x 128 co x 9 taps per workgroup, runs of "
                                          "2 rows x 16 px, double-buffered LDS, staggered wave groups)" if split else "conv3x3_wgrad_k<1,4> (v_mfma_f32_32x32x2_f32)"),
                         "wgrad_achieved": wg_tf, "wgrad_frac": wg_tf / peak, "wgrad_issued_frac": wg_tf * npm / peak,
                         "wgrad_traffic": pmc_traffic(pmc, "conv3x3_wgrad_split2_k") if split and nprod == 3.0 else None,
                         "wgrad_launches_timed": wg["launches"],
                         "whole_step_tflops": step_tflop * args.steps / dt if S == 256 and args.ngf == 64 else None,
                         "whole_step_note": "(%.0f - 3 x %.2f) = %.1f conv GFLOP per pair executed (key-side encoder passes "
                                            "reused from forward()), x batch / step time" % (STEP_GF_REFERENCE, G_ENC_GF, step_gf)},
            "losses": {k: round(v, 6) for k, v in losses.items()},
        }
        if collective is not None:
            result["collective"] = collective
    dfdist.barrier()
    if rank == 0:
        ops.set_conv_profiler(None)
        if world == 1 and not args.no_3d:
            model = None
            batches.clear()
            torch.cuda.empty_cache()
            result["roofline_hbm"] = bench_warp_hbm(dev, pmc)
            torch.cuda.empty_cache()
            result["also_3d"] = bench_3d(dev, pmc, rough_steps=5)
            torch.cuda.empty_cache()
            # BASELINE configs[3]: 128^3, the plugin's 6-level features (SURVEY section 8 D2 "state which"), one hipGraph
            # replay per step (eager, this size is host-bound: 3.7 ms of enqueue per 5.3 ms step)
            result["also_3d_128"] = bench_3d(dev, pmc, shape=(128, 128, 128), feats=PLUGIN_FEATS, gflop_step=285.0,
                                             label="BASELINE configs[3]", pmc_key="conv3d_split_k_34_16_128", steps=20,
                                             warmup=3, capture=True)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(S)
            result["cpu_baseline"]["also_3d_128"] = cpu_baseline_3d(result["cpu_baseline"]["cores"])
            result["cpu_baseline"]["also_3d_big"] = cpu_baseline_3d_big(result["cpu_baseline"]["cores"])
        print(json.dumps(result))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

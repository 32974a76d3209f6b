"""torch.autograd.Function wrappers over the C ABI of libdfmir_hip.so.

Every op here launches hand-written gfx950 kernels on torch's *current* HIP stream with raw device
pointers; torch is used only for memory, streams and the autograd tape.  There is no CPU or
eager-PyTorch fallback: a non-CUDA tensor raises.
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ._lib import DfConvGeom, DfmirHipError, check, lib

_VP = ctypes.c_void_p


def _p(t):
    return None if t is None else _VP(t.data_ptr())


def _st():
    return _VP(torch.cuda.current_stream().cuda_stream)


def _need(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise DfmirHipError("dfmir_amd ops run only on the HIP device (got a %s tensor); "
                                "there is no CPU fallback" % t.device)
        if t.dtype not in (torch.float32, torch.int64, torch.bool, torch.uint8):
            raise DfmirHipError("unsupported dtype %s" % t.dtype)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def zero_(t):
    """t.zero_() as a kernel launch.  torch zeroes contiguous tensors with hipMemsetAsync, which a captured step turns
    into hipGraph memset nodes; those are not reliably ordered with neighbouring kernel nodes on ROCm 7.2
    (scripts/graph_probe.py memset_order), so nothing on the step path uses torch.zeros / zero_ / zeros_like."""
    if t.numel():
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise DfmirHipError("ops.zero_: contiguous fp32 device tensors only")
        check(lib().dfmir_fill_zero(_p(t), t.numel(), _st()))
    return t


def zeros(shape, device):
    return zero_(torch.empty(shape, device=device, dtype=torch.float32))


def zeros_like(t):
    return zero_(torch.empty_like(t, memory_format=torch.contiguous_format))


def _env_on(name):
    """A/B switch from the environment, read at import: on unless unset, "" or "0" (the semantics of the library's own
    DfOptFlag, include/dfmir_hip.h "Options").  These Python-side switches select host-side call paths and are
    ENVIRONMENT-ONLY: dfmir_set_option() after import changes the C side alone."""
    v = os.environ.get(name)
    return v is not None and v not in ("", "0")


# ------------------------------------------------------------------------------------------------
# raw launches
# ------------------------------------------------------------------------------------------------
# Optional launch profiler (bench.py): called as prof(kind, flops, launch) where launch() issues
# the kernel; lets the bench bracket the dominant kernel with HIP events on the launch stream.
_CONV_PROFILER = [None]


def set_conv_profiler(fn):
    _CONV_PROFILER[0] = fn


# Range probes (max |t|) of the fp16x2 split kernels (csrc/conv3x3s.hip).  Slots come zero-initialised from
# a pool (one fill per 4096 probes).  A producer that computes the maximum as a by-product (InstanceNorm
# forward / backward) tags its output tensor with it; `amax_of` reuses the tag while the tensor is unmodified.
PROBE_SLOTS = 64          # DFMIR_PROBE_SLOTS: floats of an InstanceNorm by-product probe
_AMAX_POOL = {"buf": None, "next": 0, "gen": 0}


def amax_slot(device, n=1):
    """n floats of zero-initialised device memory (a probe = n partial maxima)."""
    pool = _AMAX_POOL
    if pool["buf"] is None or pool["next"] + n > pool["buf"].numel() or pool["buf"].device != device:
        if pool.get("capturing") and pool["buf"] is not None:
            raise DfmirHipError("probe pool exhausted inside a graph capture (a replay would not re-zero the new pool)")
        pool["buf"] = zeros(max(1 << 18, n), device)
        pool["next"] = 0
    i = pool["next"]
    pool["next"] = i + n
    return pool["buf"][i:i + n]


def prepare_step(device):
    """What a step triggers lazily on first use and every stream of the step then relies on -- the batched re-pack of all
    weights after an optimizer step, a probe pool with room for a whole step -- done NOW, on the current stream: a step
    that forks work onto a second stream (registration_model._forward_backward) calls this before the fork."""
    ep = weights_epoch()
    if _PACKS["epoch"] != ep and _PACKS["entries"]:
        _repack_all(ep)
    pool = _AMAX_POOL
    if pool["buf"] is None or pool["buf"].device != device:
        amax_slot(device, 0)
    elif pool["next"] + (1 << 16) > pool["buf"].numel() and not pool.get("capturing"):
        pool["buf"] = zeros(1 << 18, device)
        pool["next"] = 0


def begin_graph_capture():
    """Called right before a hipGraph capture of the step: probe slots handed out during the capture must be zeroed by
    the graph itself at every replay, so the pool is re-created (allocation + zero fill) inside the capture."""
    _AMAX_POOL["buf"] = None
    _AMAX_POOL["capturing"] = True


def end_graph_capture():
    _AMAX_POOL["capturing"] = False


def invalidate_after_failed_capture():
    """A capture that raised has RECORDED launches without running them, while the host-side caches were updated as if
    they had run: the batched weight re-pack (and the 3-D split workspaces derived from it) is marked current for this
    weights epoch, and the probe pool handed out during the capture was allocated and zero-filled only by the dead graph.
    Forget both: the next eager step re-packs every weight, re-splits every 3-D workspace and starts a fresh, really
    zeroed probe pool."""
    _AMAX_POOL["buf"] = None
    _AMAX_POOL["capturing"] = False
    _AMAX_POOL["gen"] += 1                                         # tags that point into the dead capture's pool expire
    _PACKS["epoch"] = None
    _PACKS["key"] = _WS3D["key"] = None                           # job tables uploaded inside the capture never arrived
    for j_ in _JOBS_BY_GROUP.values():
        j_["key"] = None
    for e in _PACKS["entries"].values():
        e["epoch"] = None
    for e in _WS3D["entries"].values():
        w = e["ref"]()
        cache = getattr(w, "_df_ws3d", None) if w is not None else None
        if cache is not None and e["key"] in cache:
            cache[e["key"]] = (None, cache[e["key"]][1])          # generation None never matches: re-split in place


def absmax(t):
    out = amax_slot(t.device)
    check(lib().dfmir_absmax(_p(t), t.numel(), _p(out), _st()))
    return out


def tag_amax(t, slot):
    t._df_amax = (slot, t._version, t.data_ptr(), _AMAX_POOL["gen"])
    return t


def _amax_ok(t, tag):
    """The tag still describes t (same version, same storage) AND its slot comes from a probe pool that really ran: a
    capture that failed hands out slots of a graph-private pool that no launch ever filled (invalidate_after_failed_capture
    bumps the pool generation)."""
    return tag[1] == t._version and tag[2] == t.data_ptr() and tag[3] == _AMAX_POOL["gen"]


def amax_of(t):
    tag = getattr(t, "_df_amax", None)
    if tag is not None and _amax_ok(t, tag):
        return tag[0]
    return absmax(t)


_NO_SPLIT3D = _env_on("DFMIR_CONV3D_FP32") or _env_on("DFMIR_CONV_FP32")

# Probe audit (DFMIR_PROBE_AUDIT=1 or set_probe_audit(True); debugging / test mode, one device sync per conv launch).
# The fp16x2 split scales a tensor by its range PROBE, and probes are inherited (blur outputs, nearest_up2 + cat,
# sampled-feature scatters, dgrad epilogues); a probe below the true maximum would overflow fp16 silently.  In audit
# mode every split launch first measures the true max |t| of each operand with dfmir_absmax and raises if the probe
# it was handed is smaller (per-plane dY maxima of the weight gradient included).
_PROBE_AUDIT = {"on": _env_on("DFMIR_PROBE_AUDIT"), "log": []}


def set_probe_audit(on):
    _PROBE_AUDIT["on"] = bool(on)
    _PROBE_AUDIT["log"] = []
    return _PROBE_AUDIT["log"]


def _audit_probe(t, probe, what, plane_max=None):
    true = torch.zeros(1, device=t.device, dtype=torch.float32)
    tc = _c(t)
    check(lib().dfmir_absmax(_p(tc), tc.numel(), _p(true), _st()))
    pv, tv = float(probe.max()), float(true)
    _PROBE_AUDIT["log"].append((what, pv, tv))
    if tv == tv and not pv >= tv:
        raise DfmirHipError("probe audit: %s has max|t| = %.6g but its range probe says %.6g" % (what, tv, pv))
    if plane_max is not None:                 # per-(n, c) maxima of dY (split weight gradient)
        pm = tc.reshape(plane_max.numel(), -1).abs().amax(dim=1)
        bad = (plane_max.reshape(-1) < pm)
        if bool(bad.any()):
            i = int(bad.nonzero()[0])
            raise DfmirHipError("probe audit: %s plane %d has max %.6g but its per-plane probe says %.6g"
                                % (what, i, float(pm[i]), float(plane_max.reshape(-1)[i])))


def _wants_amax(K, stride, dil, Di, Cin, Cout):
    """Shapes the split kernels take (the C side decides; this only avoids useless probes): 2-D 3x3 with more than
    32 output channels (csrc/conv3x3s.hip), 3-D 3x3x3 with at least 8 channels on both sides (csrc/conv3ds.hip)."""
    if tuple(K) == (3, 3, 3):   # Cout < 8 (the flow conv): only its weight gradient is split (swapped operand roles)
        return stride == 1 and dil == 1 and Di > 1 and Cin >= 8 and Cout >= 1 and not _NO_SPLIT3D
    return tuple(K) == (1, 3, 3) and stride == 1 and dil == 1 and Di == 1 and Cout > 32 and Cin >= 16


def conv_raw(x5, w_tcc, bias, Cout, K, stride, pad, dil, pad_mode, act, slope, out_sp, x_amax=None, res=None, ring=None,
             cout_used=None, act_src=None, act_slope=0.0):
    """res (optional, shape of the output): y = act(conv + bias) + res, inside the kernel's epilogue where the
    library has one (dfmir_conv3x3_res_ok), else by a separate add.  ring = (buffer, row length) of
    dfmir_conv3x3_reflect_ring: folded in by the same epilogue (the caller checked dfmir_conv3x3_res_ok)."""
    N, Cin, Di, Hi, Wi = x5.shape
    y = torch.empty((N, Cout) + tuple(out_sp), device=x5.device, dtype=torch.float32)
    g = DfConvGeom(N, Cin, Cout, Di, Hi, Wi, out_sp[0], out_sp[1], out_sp[2], K[0], K[1], K[2],
                   stride, dil, pad[0], pad[1], pad[2], pad_mode, act, float(slope))
    fuse_res = (res is not None and not _NO_RES and x_amax is not None and res.is_contiguous() and tuple(res.shape) == tuple(y.shape)
                and lib().dfmir_conv3x3_res_ok(ctypes.byref(g)))
    # the flow head / its data gradient (16 -> 3, 3 -> 16): plain fp32 FMAs (csrc/conv3dt.hip)
    tiny3d = (tuple(K) == (3, 3, 3) and res is None and ring is None and not _NO_TINY3D
              and (cout_used is None or cout_used == Cout) and bool(lib().dfmir_conv3d_tiny_ok(ctypes.byref(g))))
    # ... the head itself on the z-marching kernel's FLOW form when the volume is large enough for it (csrc/conv3dm.hip)
    flow_march = (tiny3d and Cin == 16 and Cout == 3 and act == 0 and act_src is None and x_amax is not None
                  and not _NO_FLOW_MARCH and Wi % 4 == 0 and bool(lib().dfmir_conv3d_march_ok(ctypes.byref(g))))
    # the first encoder level (2 -> 16, stride 2) from an LDS-staged patch (csrc/conv3dt.hip)
    s2c2 = (not tiny3d and tuple(K) == (3, 3, 3) and stride == 2 and Cin == 2 and res is None and ring is None and not _NO_TINY3D
            and cout_used is None and bool(lib().dfmir_conv3d_s2c2_ok(ctypes.byref(g))))
    # the deeper stride-2 encoder levels and their data gradients (fp32 MFMA from LDS-staged patches: csrc/conv3ds2.hip)
    plain = res is None and ring is None and cout_used is None and act_src is None and tuple(K) == (3, 3, 3) and not _NO_S2
    s2m = plain and not s2c2 and stride == 2 and dil == 1 and bool(lib().dfmir_conv3d_s2_ok(ctypes.byref(g)))
    s2d = (plain and stride == 1 and dil == 2 and bias is None and act == 0
           and bool(lib().dfmir_conv3d_s2_dgrad_ok(ctypes.byref(g))))
    split3d = (not tiny3d and not s2c2 and x_amax is not None and tuple(K) == (3, 3, 3) and res is None and ring is None
               and bool(lib().dfmir_conv3d_split_ok(ctypes.byref(g))))
    # <= 512 output voxels per image (the deepest levels of the 6-level U-Net: 8^3, 4^3, 2^3): a split launch costs its
    # 45-80 us of prologue whatever the volume; conv_tinyvol_k (behind dfmir_conv_fwd_scaled) takes 20-30
    if (split3d and cout_used is None and not _NO_TINYVOL and Cout <= 64
            and out_sp[0] * out_sp[1] * out_sp[2] <= 512 and N * out_sp[0] * out_sp[1] * out_sp[2] <= 8192):
        split3d = False
    if cout_used is not None and not split3d:
        cout_used = None                                     # only the split 3-D kernel computes a channel subset
    # the full-resolution layers with Cin x Cout <= 512: the z-marching kernel (csrc/conv3dm.hip)
    march3d = (split3d and (cout_used is None or cout_used == Cout) and Wi % 4 == 0
               and bool(lib().dfmir_conv3d_march_ok(ctypes.byref(g))))
    if act_src is not None and not ((split3d or tiny3d) and act == 0 and tuple(act_src.shape) == tuple(y.shape)
                                    and act_src.is_contiguous()):
        act_src = None                                       # ... and only these epilogues apply an activation derivative
    _LAST_ACTGRAD[0] = act_src is not None
    if _PROBE_AUDIT["on"] and x_amax is not None:
        _audit_probe(x5, x_amax, "conv input %s -> %d ch, k=%s" % (tuple(x5.shape), Cout, tuple(K)))

    def launch():
        if flow_march:
            slot = amax_slot(x5.device, PROBE_SLOTS)
            check(lib().dfmir_conv3d_march_fwd(ctypes.byref(g), _p(x5), _p(x_amax), x_amax.numel(), _p(w_tcc), _p(bias),
                                               _p(y), _p(slot), None, 0.0, _st()))
            tag_amax(y, slot)
            _LAST_CONV_AMAX[0] = slot
        elif tiny3d:
            slot = amax_slot(x5.device, PROBE_SLOTS)
            check(lib().dfmir_conv3d_tiny_fwd(ctypes.byref(g), _p(x5), _p(w_tcc), _p(bias), _p(y), _p(slot), _p(act_src),
                                              float(act_slope), _st()))
            tag_amax(y, slot)
            _LAST_CONV_AMAX[0] = slot
        elif s2c2:
            slot = amax_slot(x5.device, PROBE_SLOTS)
            check(lib().dfmir_conv3d_s2c2_fwd(ctypes.byref(g), _p(x5), _p(w_tcc), _p(bias), _p(y), _p(slot), _st()))
            tag_amax(y, slot)
            _LAST_CONV_AMAX[0] = slot
        elif s2m:
            slot = amax_slot(x5.device, PROBE_SLOTS)
            check(lib().dfmir_conv3d_s2_fwd(ctypes.byref(g), _p(x5), _p(w_tcc), _p(bias), _p(y), _p(slot), _st()))
            tag_amax(y, slot)
            _LAST_CONV_AMAX[0] = slot
        elif s2d:
            check(lib().dfmir_conv3d_s2_dgrad(ctypes.byref(g), _p(x5), _p(w_tcc), _p(y), _st()))
        elif march3d:
            slot = amax_slot(x5.device, PROBE_SLOTS)
            check(lib().dfmir_conv3d_march_fwd(ctypes.byref(g), _p(x5), _p(x_amax), x_amax.numel(), _p(w_tcc), _p(bias),
                                               _p(y), _p(slot), _p(act_src), float(act_slope), _st()))
            tag_amax(y, slot)
            _LAST_CONV_AMAX[0] = slot
        elif split3d:
            # fp16x2 split form on the 16-bit matrix pipe; the kernel leaves the range probe of y for the next layer.
            # The split weights are kept per packed-weight buffer and re-made only when that buffer was re-packed.
            cu = Cout if cout_used is None else cout_used
            # (the generation and the cache live ON the persistent packed buffer's tensor object: a temporary packing
            # has neither, and an address recycled by the allocator cannot alias a stale entry)
            pair_ = lib().dfmir_conv3d_split_is_pair(cu)
            ws, need_ = _ws_cached(w_tcc, (Cin, Cout, cu), lib().dfmir_conv3d_split_ws_floats(Cin, Cout), x5.device,
                                   job=(0, Cin, cu if pair_ else Cout, pair_, Cin, 0, 0))
            w_arg = w_tcc if need_ else None
            slot = amax_slot(x5.device, PROBE_SLOTS)
            if act_src is not None:
                # dgrad into the output of a LeakyReLU: the epilogue applies the activation's derivative (csrc/conv3ds.hip)
                check(lib().dfmir_conv3d_split_fwd_actgrad(ctypes.byref(g), _p(x5), _p(x_amax), x_amax.numel(), _p(w_arg),
                                                           _p(ws), _p(bias), _p(y), _p(slot), cu, _p(act_src),
                                                           float(act_slope), _st()))
            else:
                check(lib().dfmir_conv3d_split_fwd_sub(ctypes.byref(g), _p(x5), _p(x_amax), x_amax.numel(), _p(w_arg), _p(ws),
                                                       _p(bias), _p(y), _p(slot), cu, _st()))
            tag_amax(y, slot)              # survives as is when y is a backward result (dgrad) ...
            _LAST_CONV_AMAX[0] = slot      # ... and is re-attached by conv() to the tensor Function.apply returns
        elif fuse_res or ring is not None:
            check(lib().dfmir_conv3x3_fwd_scaled_res(ctypes.byref(g), _p(x5), _p(x_amax), x_amax.numel(), _p(w_tcc),
                                                     _p(bias), _p(res) if fuse_res else None,
                                                     _p(ring[0]) if ring is not None else None,
                                                     ring[1] if ring is not None else 0, _p(y), _st()))
            if res is not None and not fuse_res:
                y.add_(res.reshape(y.shape))
        else:
            check(lib().dfmir_conv_fwd_scaled(ctypes.byref(g), _p(x5), _p(x_amax),
                                              0 if x_amax is None else x_amax.numel(), _p(w_tcc), _p(bias), _p(y), _st()))
            if res is not None:
                y.add_(res.reshape(y.shape))

    prof = _CONV_PROFILER[0]
    if prof is None:
        launch()
    else:
        flops = 2.0 * N * Cout * out_sp[0] * out_sp[1] * out_sp[2] * Cin * K[0] * K[1] * K[2] / (dil ** 3 if K[0] > 1 else dil ** 2)
        size = "L" if Cout > 64 else ("M" if Cout > 32 else ("S" if Cout > 4 else "small"))
        is3x3 = (tuple(K) == (1, 3, 3) and stride == 1 and dil == 1 and Di == 1 and pad[1] == pad[2]
                 and pad[1] in (1, 2) and Cout > 4)
        is3d = (tuple(K) == (3, 3, 3) and stride == 1 and dil == 1 and tuple(pad) == (1, 1, 1) and pad_mode == 0
                and not (Cout <= 4 and Cin < 8))          # what csrc/conv3d.hip::df_conv3d_fwd_try takes
        kind = ("conv3x3_" if is3x3 else (("conv3ds_" if split3d else "conv3d_") if is3d else "conv_mfma_")) + size
        if tiny3d or s2c2 or s2m or s2d:
            kind = "conv3dt_" + size
        if split3d:
            # 16-bit products the kernel ISSUES per algorithmic MAC: 3 (a0b0 + a0b1 + a1b0) x the padding of its tiling --
            # taps 27 -> 28 (plane-pair rows: 27 -> 36 taps'), input channels to whole chunks of 8, output channels to 32 rows
            cu = Cout if cout_used is None else cout_used
            form_ = lib().dfmir_conv3d_split_is_pair(cu)      # 0: 32 rows, 1: plane pairs (36 taps'), 2: 16 rows
            pad_k = (36.0 if form_ == 1 else 28.0) / 27.0 * (8.0 * ((Cin + 7) // 8)) / Cin
            pad_m = (16.0 / cu) if form_ else (32.0 * ((cu + 31) // 32)) / cu
            if march3d:                                       # no padded rows; 16 -> 16 walks 10 tap slots per 9 taps
                pad_k, pad_m = ((10.0 / 9.0) if (Cin == 16 and Cout == 16) else 1.0), 1.0
            if getattr(prof, "accepts_issued", False):
                prof(kind, flops * cu / Cout, launch, 3.0 * pad_k * pad_m * flops * cu / Cout)
            else:
                prof(kind, flops * cu / Cout, launch)
        else:
            prof(kind, flops, launch)
    return y


_LAST_ACTGRAD = [False]     # did the last conv_raw() apply an activation derivative in its epilogue?
_NO_TINY3D = _env_on("DFMIR_CONV3D_NO_TINY")  # A/B switch: the flow head on the split kernels
_NO_TINYVOL = _env_on("DFMIR_NO_TINYVOL")     # A/B switch (also read by the library): no conv_tinyvol_k
_NO_S2 = _env_on("DFMIR_CONV3D_NO_S2")        # A/B switch: the stride-2 encoder levels on the generic gather kernels
_NO_FLOW_MARCH = _env_on("DFMIR_CONV3D_NO_FLOW_MARCH")  # A/B switch: the flow head's forward on the fp32-FMA kernel
_NO_ACTGRAD = _env_on("DFMIR_NO_ACTGRAD")     # A/B switch: LeakyReLU backward always as its own pass


def _tag_ok(t, tag):
    return tag is not None and tag[-2] == t._version and tag[-1] == t.data_ptr()


# dfmir_conv3d_upwgrad below this many low-resolution voxels per image: the direct kernel over both parts.  With MORE than two
# skip channels their share runs on the direct kernel anyway and the split only pays at the largest volumes
# (profiles/r05_bench_upwgrad.txt: 48 -> 32 at 80x96x112 0.32 vs 0.29 ms); two skip channels ride in the same launch.
_UPWGRAD_MIN_VOX = int(os.environ.get("DFMIR_UPWGRAD_MIN_VOX", "400000"))
_UPWGRAD_MIN_VOX_FUSED = 50000
_UPWGRAD_WS = {}


def _upwgrad_ws(device):
    """64 accumulator tiles of dfmir_conv3d_upwgrad: one buffer per (device, stream) -- the call zeroes it, fills it and
    folds it on the caller's stream, so launches of one stream may share it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _UPWGRAD_WS.get(key)
    if ws is None:
        ws = _UPWGRAD_WS[key] = torch.zeros(int(lib().dfmir_conv3d_upwgrad_ws_floats()), device=device)
    return ws


# Deterministic weight gradients (opt.deterministic_wgrad -> set_deterministic_wgrad; build-defined).  The weight- and
# bias-gradient kernels split their reduction over workgroups and meet in fp32 atomics: every run rounds in a different
# order (2 of 30 runs of one test exceeded a 4e-5 bound on step 4's gradients because of it).  In this mode each
# gradient launch accumulates 64-bit fixed-point integers into a per-call scratch (include/dfmir_hip.h "Deterministic
# weight gradients") -- integer addition is associative -- and a finaliser adds the converted sums to the real target: two
# runs of a step from the same state give bit-identical gradient arenas.  Cost: one zero-fill, one scale launch and one
# finaliser per gradient launch, and the bias gradients as their own pass over dY.
_DET = {"on": _env_on("DFMIR_DETERMINISTIC_WGRAD")}


def set_deterministic_wgrad(on):
    _DET["on"] = bool(on)


class _DetAcc(object):
    def __init__(self, n_dw, n_db, count, x, x_amax, dy, dy_amax):
        head = int(lib().dfmir_det_head_floats())
        self.n_dw, self.n_db = int(n_dw), int(n_db)
        self.scr = torch.empty(head + 2 * (self.n_dw + self.n_db), device=dy.device, dtype=torch.float32)
        check(lib().dfmir_det_begin(_p(self.scr), self.n_dw + self.n_db, _p(x), 0 if x is None else x.numel(), _p(x_amax),
                                    0 if x_amax is None else x_amax.numel(), _p(dy), dy.numel(), _p(dy_amax),
                                    0 if dy_amax is None else dy_amax.numel(), float(count), _st()))
        self.dw = self.scr[head:head + 2 * self.n_dw]              # the entry points see 8-byte slots behind this pointer
        self.db = self.scr[head + 2 * self.n_dw:] if self.n_db else None

    def end(self, dw_out, db_out):
        check(lib().dfmir_det_end(_p(self.scr), _p(dw_out), self.n_dw, _p(db_out), self.n_db if db_out is not None else 0, _st()))

    def abort(self):
        check(lib().dfmir_det_end(_p(self.scr), None, 0, None, 0, _st()))


def bias_grad(dy5, db):
    """db[c] += sum over (n, voxels) of dy5[n, c] (accumulates); deterministic mode: through the fixed-point scratch."""
    N, C = dy5.shape[0], dy5.shape[1]
    S = dy5.numel() // (N * C)
    if not _DET["on"]:
        check(lib().dfmir_bias_grad(_p(dy5), _p(db), N, C, S, _st()))
        return
    acc = _DetAcc(0, C, N * S, dy5, None, dy5, None)         # (x plays no part: the bound is count * max|dy|)
    try:
        check(lib().dfmir_bias_grad(_p(dy5), _p(acc.db), N, C, S, _st()))
    except BaseException:
        acc.abort()
        raise
    acc.end(None, db)


def conv_wgrad_raw(x5, dy5, K, stride, pad, pad_mode, out=None, x_amax=None, dy_amax=None, db=None, dy_pmax=None,
                   parts=None):
    """dW in the tap-major packing; `out` (same packing) is accumulated into when given.  parts = (a, b): the operand is
    cat(nearest_up2(a), b), never built (x5 is None; the caller checked upcat_wgrad_ok)."""
    if parts is not None:
        a_, b_ = parts
        N, Cin = b_.shape[0], a_.shape[1] + b_.shape[1]
        Di, Hi, Wi = b_.shape[2:]
        x5 = b_
    else:
        N, Cin, Di, Hi, Wi = x5.shape
    _, Cout, Do, Ho, Wo = dy5.shape
    T = K[0] * K[1] * K[2]
    dw = zeros((T, Cin, Cout), x5.device) if out is None else out
    g = DfConvGeom(N, Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, K[0], K[1], K[2], stride, 1, pad[0], pad[1],
                   pad[2], pad_mode, 0, 0.0)
    split3d = (x_amax is not None and dy_amax is not None and tuple(K) == (3, 3, 3)
               and bool(lib().dfmir_conv3d_split_wgrad_ok(ctypes.byref(g))))
    if parts is not None and not split3d:
        raise DfmirHipError("conv_wgrad_raw(parts=...) needs the split 3-D weight-gradient kernel")
    if _PROBE_AUDIT["on"]:
        if x_amax is not None:
            for t_ in (parts if parts is not None else (x5,)):
                _audit_probe(t_, x_amax, "wgrad x %s" % (tuple(t_.shape),))
        if dy_amax is not None:
            pm_ = dy_pmax if (dy_pmax is not None and dy_pmax.numel() == N * Cout and not split3d) else None
            _audit_probe(dy5, dy_amax, "wgrad dY %s" % (tuple(dy5.shape),), plane_max=pm_)

    # the up-sampled channels in parity classes (csrc/conv3duw.hip) where the volume fills the chip
    upw = (parts is not None
           and parts[0][0, 0].numel() >= (min(_UPWGRAD_MIN_VOX, _UPWGRAD_MIN_VOX_FUSED) if parts[1].shape[1] == 2 else _UPWGRAD_MIN_VOX)
           and bool(lib().dfmir_conv3d_upwgrad_ok(ctypes.byref(g), parts[0].shape[1])))
    s2c2 = (parts is None and tuple(K) == (3, 3, 3) and stride == 2 and Cin == 2 and not _NO_TINY3D
            and bool(lib().dfmir_conv3d_s2c2_ok(ctypes.byref(g))))
    s2m = (parts is None and not s2c2 and tuple(K) == (3, 3, 3) and stride == 2 and not _NO_S2
           and bool(lib().dfmir_conv3d_s2_ok(ctypes.byref(g))))
    # deterministic mode: the kernels add 64-bit fixed-point sums into a scratch, never a bias gradient (its own pass below)
    det = _DetAcc(dw.numel(), Cout if db is not None else 0, N * Do * Ho * Wo, x5 if parts is None else None, x_amax,
                  dy5, dy_amax) if _DET["on"] else None
    dw_real, db_real = dw, db
    if det is not None:
        dw, db = det.dw, None

    def launch():
        if s2c2:
            if db is not None:
                check(lib().dfmir_bias_grad(_p(dy5), _p(db), dy5.shape[0], Cout, Do * Ho * Wo, _st()))   # accumulates
            check(lib().dfmir_conv3d_s2c2_wgrad(ctypes.byref(g), _p(x5), _p(dy5), _p(dw), _st()))
            return
        if s2m:
            check(lib().dfmir_conv3d_s2_wgrad(ctypes.byref(g), _p(x5), _p(dy5), _p(dw), _p(db), _st()))   # db fused
            return
        if parts is not None:
            if upw:
                check(lib().dfmir_conv3d_upwgrad(ctypes.byref(g), _p(parts[0]), _p(parts[1]), parts[0].shape[1],
                                                 _p(x_amax), x_amax.numel(), _p(dy5), _p(dy_amax), dy_amax.numel(),
                                                 _p(dw), _p(db), _p(_upwgrad_ws(dy5.device)), _st()))
                return
            check(lib().dfmir_conv3d_split_wgrad_upcat(ctypes.byref(g), _p(parts[0]), _p(parts[1]), parts[0].shape[1],
                                                       _p(x_amax), x_amax.numel(), _p(dy5), _p(dy_amax), dy_amax.numel(),
                                                       _p(dw), _p(db), _st()))
            return
        if split3d:
            check(lib().dfmir_conv3d_split_wgrad_db(ctypes.byref(g), _p(x5), _p(x_amax), x_amax.numel(), _p(dy5),
                                                    _p(dy_amax), dy_amax.numel(), _p(dw), _p(db), _st()))   # db fused
            return
        pm = dy_pmax if (dy_pmax is not None and dy_pmax.numel() == N * Cout and dy_amax is not None) else None
        check(lib().dfmir_conv_wgrad_scaled_ch(ctypes.byref(g), _p(x5), _p(x_amax),
                                               0 if x_amax is None else x_amax.numel(), _p(dy5), _p(dy_amax),
                                               0 if dy_amax is None else dy_amax.numel(), _p(pm), _p(dw), _p(db), _st()))

    if det is not None:
        try:
            launch()
            if db_real is not None:
                check(lib().dfmir_bias_grad(_p(dy5), _p(det.db), dy5.shape[0], Cout, Do * Ho * Wo, _st()))
        except BaseException:
            det.abort()
            raise
        det.end(dw_real, db_real)
        return dw_real
    prof = _CONV_PROFILER[0]
    if prof is None:
        launch()
    else:
        is3x3 = (tuple(K) == (1, 3, 3) and stride == 1 and Di == 1 and pad[1] == 1 and pad[2] == 1
                 and Cout >= 64 and Cin >= 32 and (Hi * Wi) % 32 == 0 and Wi % 32 == 0)
        size = "L" if Cout > 64 else ("M" if Cout > 32 else "S")
        is3d = (tuple(K) == (3, 3, 3) and stride == 1 and tuple(pad) == (1, 1, 1) and pad_mode == 0 and Cout <= 32
                and Wi % 4 == 0)                          # csrc/conv3d.hip::df_conv3d_wgrad_try
        flops = 2.0 * N * Cout * Do * Ho * Wo * Cin * T
        kind = ("wgrad3dt_S" if (s2c2 or s2m) else
                ("wgrad3x3_" if is3x3 else (("wgrad3ds_" if split3d else "wgrad3d_") if is3d else "conv_wgrad_")) + size)
        if split3d and is3d and not s2c2 and getattr(prof, "accepts_issued", False):
            # 16-bit products the kernel ISSUES per algorithmic MAC (cf. conv_raw): 3 x the padding of its tiling
            if upw:     # parity classes: 8 / 27 of the up-sampled share, 32 columns; fused skip pair: 54 -> 64 rows
                Ca_, Cb_ = parts[0].shape[1], parts[1].shape[1]
                skip_ = (64.0 / 54.0) if Cb_ == 2 else (28.0 / 27.0) * (8.0 * ((Cb_ + 7) // 8)) / Cb_
                pad_ = ((8.0 / 27.0) * Ca_ + skip_ * Cb_) / Cin * (32.0 / Cout)
                kind = "wgrad3dup_" + size
            elif Cout < 8:          # the flow head: roles swapped, rows = (tap, co of a chunk of 8), columns = ci of 32
                pad_ = (28.0 / 27.0) * (8.0 / Cout) * (32.0 / Cin)
            elif parts is None and lib().dfmir_conv3d_wgrad_is_march_at(ctypes.byref(g), _p(x5), _p(dy5)):   # 14 tiles of two taps for 27
                pad_ = 28.0 / 27.0
            elif Cout <= 16:        # plane-pair columns: 36 taps' of 27, 16 columns per plane
                pad_ = (36.0 / 27.0) * (8.0 * ((Cin + 7) // 8)) / Cin * (16.0 / Cout)
            else:
                pad_ = (28.0 / 27.0) * (8.0 * ((Cin + 7) // 8)) / Cin * (32.0 / Cout)
            prof(kind, flops, launch, 3.0 * pad_ * flops)
        else:
            prof(kind, flops, launch)
    return dw


def weight_pack(w, mode):
    Cout, Cin = w.shape[0], w.shape[1]
    T = w.numel() // (Cout * Cin)
    out = torch.empty(lib().dfmir_weight_pack_floats(Cout, Cin, T), device=w.device, dtype=torch.float32)
    check(lib().dfmir_weight_pack(_p(_c(w)), _p(out), Cout, Cin, T, mode, _st()))
    return out


# Packed weights are kept per (module, mode) in persistent buffers.  All weights change together (the optimizer step
# bumps the weights epoch), so the first request after a bump re-packs EVERY registered buffer whose parameter is still
# in place with one batched call (2 launches instead of ~170 per train step).
_PACKS = {"epoch": None, "entries": {}, "key": None, "descs": None, "dev": None}
def _bump_gen(buf):
    """Generation of a persistent packed-weight buffer (bumped whenever it is re-packed): keys what is derived from it."""
    buf._df_gen = getattr(buf, "_df_gen", 0) + 1
_KEEP_TABLES = []     # device job tables are never freed (a few KB each; see _repack_all / _flush_deferred)


def packed_weight(owner, w3, mode):
    """Packed form of owner's weight (w3 = the parameter viewed [Cout, Cin, T...]) for mode 0 (forward) / 1 (dgrad).
    The buffer of an (owner, mode) pair is allocated once and refreshed IN PLACE for as long as the parameter keeps its
    address and shape -- a captured hipGraph holds the buffer's address, and an in-place change of the weights through
    torch (load_state_dict, a test restoring a snapshot) must not move it."""
    import weakref
    ents = _PACKS["entries"]
    key = (id(owner), mode)
    ep = weights_epoch()
    place = (w3.data_ptr(), tuple(w3.shape), w3.device)
    ent = ents.get(key)
    if ent is not None and (ent["ref"]() is not owner or ent["place"] != place):
        ent = None
    if ent is None:
        buf = weight_pack(w3, mode)
        _bump_gen(buf)
        ents[key] = {"ref": weakref.ref(owner), "place": place, "version": w3._version, "epoch": ep, "buf": buf,
                     "w": w3.detach(), "mode": mode}
        _PACKS["key"] = None
        return buf
    if ent["version"] != w3._version:            # modified in place through torch since the last pack
        ent["version"] = w3._version
        ent["epoch"] = None
    if ent["epoch"] != ep and _PACKS["epoch"] != ep:
        _repack_all(ep)
    if ent["epoch"] == ep:
        return ent["buf"]
    # the batched pass did not cover it (it ran before this entry existed / before the in-place change): refresh
    Cout, Cin = w3.shape[0], w3.shape[1]
    check(lib().dfmir_weight_pack(_p(_c(w3)), _p(ent["buf"]), Cout, Cin, w3.numel() // (Cout * Cin), mode, _st()))
    _bump_gen(ent["buf"])
    ent["epoch"] = ep
    return ent["buf"]


def _repack_all(ep):
    import numpy as np
    ents = _PACKS["entries"]
    for k in [k for k, e in ents.items() if e["ref"]() is None]:
        del ents[k]
        _PACKS["key"] = None
    live = [e for e in ents.values() if e["place"][0] == e["w"].data_ptr() and e["w"].is_contiguous()]
    _PACKS["epoch"] = ep
    if not live:
        return
    dev = live[0]["w"].device
    live = [e for e in live if e["w"].device == dev]
    key = tuple((e["w"].data_ptr(), e["buf"].data_ptr(), e["mode"]) for e in live)
    upload = 0
    if _PACKS["key"] != key:
        tab = np.zeros((len(live), 4), dtype=np.int64)             # struct DfPackJob = 2 pointers + 4 ints
        for i, e in enumerate(live):
            Cout, Cin = int(e["w"].shape[0]), int(e["w"].shape[1])
            T = e["w"].numel() // (Cout * Cin)
            tab[i, 0], tab[i, 1] = e["w"].data_ptr(), e["buf"].data_ptr()
            tab[i, 2] = Cout | (Cin << 32)
            tab[i, 3] = T | (e["mode"] << 32)
        _PACKS["descs"] = tab
        _PACKS["dev"] = torch.empty(len(live) * 8, device=dev, dtype=torch.int64)   # 64 B per job
        _KEEP_TABLES.append(_PACKS["dev"])       # a captured hipGraph may hold the address of an earlier table
        _PACKS["key"] = key
        upload = 1
    check(lib().dfmir_weight_pack_batch(_PACKS["descs"].ctypes.data, len(live), _p(_PACKS["dev"]), upload, _st()))
    for e in live:
        e["epoch"] = ep
        e["version"] = e["w"]._version
        _bump_gen(e["buf"])
    _resplit_all_3d(dev)


def weight_unpack(g_tcc, shape):
    Cout, Cin = shape[0], shape[1]
    T = g_tcc.numel() // (Cout * Cin)
    out = torch.empty(shape, device=g_tcc.device, dtype=torch.float32)
    check(lib().dfmir_weight_unpack(_p(g_tcc), _p(out), Cout, Cin, T, _st()))
    return out


# A global "weights epoch": the fused Adam kernel updates parameters through raw pointers, which
# torch's version counters cannot see; packed-weight caches are keyed on (version, epoch).
_EPOCH = [0]


def bump_weights_epoch():
    _EPOCH[0] += 1


# Deferred weight gradients.  A train step back-propagates through the same conv modules several times
# (G once, its encoder three more times for the NCE terms).  Inside `with deferred_weight_grads():` the
# wgrad kernels of all those passes accumulate into ONE persistent tap-major buffer per module and the
# bias gradients straight into `bias.grad`; the buffers are unpacked into `weight.grad` once, on exit.
# Without it every pass pays a zero-fill, an unpack and an autograd `add` per parameter (~900 tiny launches).
_NO_RES = _env_on("DFMIR_NO_RES")       # A/B switch: residual added by a separate kernel
_NO_DEAD_TAIL = _env_on("DFMIR_NO_DEAD_TAIL")   # A/B switch: dgrad also for skip channels that need none
_NO_CH_SCALE = _env_on("DFMIR_NO_CH_SCALE")   # A/B switch: one dY scale per tensor in the split wgrad
_NO_RING = _env_on("DFMIR_NO_RING")     # A/B switch: reflect dgrad as padded-frame conv + fold
_DEFER = {"on": False, "pending": {}}


class deferred_weight_grads:
    def __enter__(self):
        _DEFER["on"] = not _env_on("DFMIR_NO_DEFER")   # A/B switch
        return self

    def __exit__(self, *exc):
        _DEFER["on"] = False
        pending, _DEFER["pending"] = _DEFER["pending"], {}
        if exc[0] is None and pending:
            _flush_deferred(list(pending.values()))
        elif pending:
            # backward raised: the accumulators hold partial sums that only the flush kernel would clear; a caller
            # that catches the error and goes on (skip-batch / OOM-retry loops) must not inherit them
            for _, buf, _ in pending.values():
                zero_(buf)
        return False


def begin_deferred():
    """deferred_weight_grads as two calls, for a backward that runs in several pieces (registration_model's staged step:
    the pieces are separate autograd calls on two streams, captured into separate hipGraphs): begin_deferred() before the
    first piece, end_deferred() -- the one flush -- after the last, on a stream that has waited for all of them."""
    _DEFER["on"] = not _env_on("DFMIR_NO_DEFER")


def end_deferred(failed=False):
    _DEFER["on"] = False
    pending, _DEFER["pending"] = _DEFER["pending"], {}
    if not pending:
        return
    if failed:
        for _, buf, _ in pending.values():
            zero_(buf)
    else:
        _flush_deferred(list(pending.values()))


_JOBS = {"key": None, "dev": None, "max": 0}
_JOBS_BY_GROUP = {"all": _JOBS}


def flush_deferred_subset(owner_ids, group):
    """Flush NOW the deferred accumulators of the modules in `owner_ids` (ids) whose weight gradients are final -- a
    gradient bucket that leaves for its all-reduce before backward has finished (registration_model._bucket_ready).
    `group` names the job table (one device table per group: a captured step holds it by address)."""
    pend = _DEFER["pending"]
    items = [pend.pop(i) for i in list(pend.keys()) if i in owner_ids]
    if items:
        _flush_deferred(items, group)
    return len(items)


def _flush_deferred(items, group="all"):
    """grad += unpack(accumulator); accumulator = 0 for every deferred conv, one launch (the job table is uploaded
    once and reused while the same buffers come back, i.e. every step after the first)."""
    import numpy as np
    _JOBS = _JOBS_BY_GROUP.setdefault(group, {"key": None, "dev": None, "max": 0})
    key = tuple((buf.data_ptr(), owner.weight.grad.data_ptr(), tuple(shape)) for owner, buf, shape in items)
    if _JOBS["key"] != key:
        tab = np.zeros((len(items), 4), dtype=np.int64)            # struct DfUnpackJob = 2 pointers + 4 ints
        mx = 0
        for i, (owner, buf, shape) in enumerate(items):
            Cout, Cin = int(shape[0]), int(shape[1])
            T = buf.numel() // (Cout * Cin)
            tab[i, 0], tab[i, 1] = buf.data_ptr(), owner.weight.grad.data_ptr()
            tab[i, 2] = Cout | (Cin << 32)
            tab[i, 3] = T
            mx = max(mx, Cout * Cin * T)
        _JOBS["dev"] = torch.from_numpy(tab).to(items[0][1].device)
        _KEEP_TABLES.append(_JOBS["dev"])
        _JOBS["key"], _JOBS["max"] = key, mx
    check(lib().dfmir_weight_unpack_add_batch(_p(_JOBS["dev"]), len(items), _JOBS["max"], _st()))


def _deferred_buffer(owner, T, Cin, Cout, shape, device):
    buf = getattr(owner, "_dw_tcc", None)
    if buf is None or buf.shape != (T, Cin, Cout) or buf.device != device:
        buf = zeros((T, Cin, Cout), device)
        owner._dw_tcc = buf
    _DEFER["pending"][id(owner)] = (owner, buf, shape)
    return buf


def weights_epoch():
    return _EPOCH[0]


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
class ConvFn(Function):
    """y = act(conv(x, w) + b).  `owner` supplies cached packed weights (owner.packed(mode))."""

    @staticmethod
    def forward(ctx, x, weight, bias, owner, stride, pad, pad_mode, act, slope, skip=False):
        # skip=True (reflect-padded stride-1 convs): also return x itself as a second output for a residual
        # branch, so that backward receives that branch's gradient and sums it inside the halo-fold kernel
        # (ResnetBlock: out = x + conv_block(x), models/networks.py:1219-1221) instead of autograd's add kernel.
        _need(x, weight, bias)
        nd = x.dim() - 2
        x5 = _c(x) if nd == 3 else _c(x).unsqueeze(2)
        K = tuple(weight.shape[2:]) if nd == 3 else (1,) + tuple(weight.shape[2:])
        p3 = (pad,) * 3 if nd == 3 else (0, pad, pad)
        sp = x5.shape[2:]
        out_sp = tuple((sp[i] + 2 * p3[i] - K[i]) // stride + 1 for i in range(3))
        w_tcc = owner.packed(0) if owner is not None else weight_pack(weight, 0)
        x_amax = None
        if _wants_amax(K, stride, 1, x5.shape[2], x5.shape[1], weight.shape[0]):
            x_amax = amax_of(x) if x.is_contiguous() else absmax(x5)
        y5 = conv_raw(x5, w_tcc, bias, weight.shape[0], K, stride, p3, 1, pad_mode, act, slope, out_sp, x_amax)
        ctx.x_amax = x_amax
        # trailing input channels that need no gradient (cat([up2(a), b]) with b = the network's input images)
        ctx.dead_tail = int(getattr(x, "_df_nograd_tail", 0))
        # x = the output of a LeakyReLU ConvBlock that feeds nothing but this conv (declared by the caller, conv(sole=True)):
        # this conv's dgrad can then return the gradient w.r.t. that block's PRE-activation (see backward)
        ta = getattr(x, "_df_act_sole", None)
        ctx.in_act = (ta[0], ta[1]) if (_tag_ok(x, ta) and x.is_contiguous() and not _NO_ACTGRAD) else None
        ctx.cfg = (nd, K, stride, p3, pad_mode, act, slope, owner)
        ctx.save_for_backward(x5, weight, y5 if act else None)
        ctx.has_bias = bias is not None
        ctx.skip = bool(skip)
        y = y5 if nd == 3 else y5.squeeze(2)
        if skip:
            if not (stride == 1 and pad_mode == 1 and x.is_contiguous()):
                raise DfmirHipError("conv(skip=True) is the reflect-padded stride-1 case on a contiguous input")
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dskip=None):
        if dy is None:              # only the skip output was used downstream
            return (dskip,) + (None,) * 9
        x5, weight, y5 = ctx.saved_tensors
        dx, dw, db = _conv_backward_impl(ctx, dy, dskip, x5, weight, y5)
        return dx, dw, db, None, None, None, None, None, None, None


def _conv_backward_impl(ctx, dy, dskip, x5, weight, y5):
    """Backward of y = act(conv(x5, weight) + bias) for a ctx-like object carrying cfg, x_amax, dead_tail, in_act,
    has_bias, needs_input_grad (x, weight, bias); returns (dx, dw, db).  Shared by ConvFn and UpCatConv3dFn."""
    if True:
        nd, K, stride, p3, pad_mode, act, slope, owner = ctx.cfg
        dy5 = _c(dy) if nd == 3 else _c(dy).unsqueeze(2)
        Cout, Cin = weight.shape[0], weight.shape[1]
        want_probe = _wants_amax(K, stride, 1, dy5.shape[2], Cout, Cin) or ctx.x_amax is not None
        dy_amax = None
        if act and _tag_ok(dy, getattr(dy, "_df_premasked", None)) and dy.is_contiguous():
            pass      # the producing dgrad already multiplied by this activation's derivative (and left the range probe)
        elif act:
            dpre = torch.empty_like(dy5)
            if want_probe and nd == 3 and not ((dy5.data_ptr() | y5.data_ptr() | dpre.data_ptr()) & 15):
                # the activation's backward leaves the range probe of what it writes (3-D tensors are 0.1-1 GB: a
                # separate absmax pass would cost 10 % of the dgrad it serves)
                dy_amax = amax_slot(dy5.device, PROBE_SLOTS)
                check(lib().dfmir_act_bwd_amax(_p(dy5), _p(y5), _p(dpre), dy5.numel(), act, float(slope), _p(dy_amax), _st()))
            else:
                check(lib().dfmir_act_bwd(_p(dy5), _p(y5), _p(dpre), dy5.numel(), act, float(slope), _st()))
            dy5 = dpre
        dx = dw = db = None
        # dY feeds the dgrad conv (as its input) and the wgrad: one range probe for both
        if want_probe and dy_amax is None:
            dy_amax = amax_of(dy) if (dy.is_contiguous() and (not act or dy5.data_ptr() == dy.data_ptr())) else absmax(dy5)
        if ctx.needs_input_grad[0]:
            wd = owner.packed(1) if owner is not None else weight_pack(weight, 1)
            in_sp = tuple(x5.shape[2:])
            gf = None
            if stride == 1 and pad_mode == 1 and dy_amax is not None and not _NO_RING:
                gf = DfConvGeom(x5.shape[0], Cin, Cout, 1, in_sp[1], in_sp[2], 1, in_sp[1], in_sp[2], K[0], K[1], K[2],
                                1, 1, p3[0], p3[1], p3[2], 1, 0, 0.0)
                gd = DfConvGeom(x5.shape[0], Cout, Cin, 1, in_sp[1], in_sp[2], 1, in_sp[1], in_sp[2], K[0], K[1], K[2],
                                1, 1, p3[0], p3[1], p3[2], 0, 0, 0.0)      # the zero-padded dgrad as a conv
                if not (lib().dfmir_conv3x3_reflect_ring_ok(ctypes.byref(gf))
                        and lib().dfmir_conv3x3_res_ok(ctypes.byref(gd))):
                    gf = None
            if gf is not None:
                # the one-pixel ring of the padded frame first (four 1-D convolutions of dY's border lines, into a
                # compact buffer), then the frame's interior = the zero-padded "same" dgrad on full tiles, whose
                # epilogue folds the ring (and the skip branch's gradient) in   (csrc/conv3x3s.hip)
                rl = lib().dfmir_conv3x3_reflect_ring_len(ctypes.byref(gf))
                ringbuf = torch.empty(x5.shape[0] * 4 * Cin * rl, device=dy5.device, dtype=torch.float32)
                ctag = getattr(dy, "_df_cols", None)
                cols = ctag[0] if (ctag is not None and ctag[1] == dy._version and ctag[2] == dy.data_ptr()
                                   and dy.is_contiguous() and not act) else None
                check(lib().dfmir_conv3x3_reflect_ring(ctypes.byref(gf), _p(dy5), _p(cols), _p(dy_amax), dy_amax.numel(),
                                                       _p(wd), _p(ringbuf), _st()))
                res5 = None
                if dskip is not None:
                    res5 = _c(dskip) if nd == 3 else _c(dskip).unsqueeze(2)
                    dskip = None
                dx5 = conv_raw(dy5, wd, None, Cin, K, 1, p3, 1, 0, 0, 0.0, in_sp, dy_amax, res=res5, ring=(ringbuf, rl))
            elif stride == 1 and pad_mode == 1:
                # full correlation onto the reflect-padded frame, then fold the halo back
                padp = tuple(K[i] - 1 for i in range(3))
                out_sp = tuple(in_sp[i] + 2 * p3[i] for i in range(3))
                dxp = conv_raw(dy5, wd, None, Cin, K, 1, padp, 1, 0, 0, 0.0, out_sp, dy_amax)
                if p3[0] != 0 or p3[1] != p3[2]:
                    raise DfmirHipError("reflect padding is 2-D, symmetric only")
                dx5 = torch.empty_like(x5)
                if dskip is not None:
                    check(lib().dfmir_reflect_pad2d_bwd_add(_p(dxp), _p(_c(dskip)), _p(dx5), x5.shape[0] * Cin,
                                                            in_sp[1], in_sp[2], p3[1], _st()))
                    dskip = None
                else:
                    check(lib().dfmir_reflect_pad2d_bwd(_p(dxp), _p(dx5), x5.shape[0] * Cin, in_sp[1], in_sp[2],
                                                        p3[1], _st()))
            else:
                padp = tuple(K[i] - 1 - p3[i] for i in range(3))
                used = Cin - ctx.dead_tail if (ctx.dead_tail and stride == 1 and not _NO_DEAD_TAIL) else None
                ia = ctx.in_act if (stride == 1 and nd == 3) else None
                dx5 = conv_raw(dy5, wd, None, Cin, K, 1, padp, stride, 0, 0, 0.0, in_sp, dy_amax, cout_used=used,
                               act_src=x5 if ia else None, act_slope=ia[1] if ia else 0.0)
                if ia and _LAST_ACTGRAD[0]:
                    dx5._df_premasked = (dx5._version, dx5.data_ptr())
            dx = dx5 if nd == 3 else dx5.squeeze(2)
        if dskip is not None:       # not folded in above (x needs no conv-path gradient, or non-reflect dgrad)
            dx = dskip if dx is None else dx + dskip
        defer = (_DEFER["on"] and owner is not None and getattr(owner, "weight", None) is not None
                 and owner.weight.grad is not None and owner.weight.grad.is_contiguous())
        # bias gradient target: straight into bias.grad when deferring, else a fresh buffer (returned to autograd);
        # it rides along with the wgrad call (which reads dY anyway) when there is one
        db_buf = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            bg = getattr(owner, "bias", None).grad if defer and getattr(owner, "bias", None) is not None else None
            if bg is not None and bg.is_contiguous():
                db_buf = bg
            else:
                db = db_buf = zeros(Cout, dy5.device)
        if ctx.needs_input_grad[1]:
            ptag = getattr(dy, "_df_pmax", None) if not act else None     # per-plane maxima of dY (InstanceNorm backward)
            dy_pmax = ptag[0] if (ptag is not None and ptag[1] == dy._version and ptag[2] == dy.data_ptr()
                                  and dy.is_contiguous() and not _NO_CH_SCALE) else None
            parts = getattr(ctx, "x_parts", None)
            if defer:
                T = K[0] * K[1] * K[2]
                conv_wgrad_raw(x5, dy5, K, stride, p3, pad_mode,
                               out=_deferred_buffer(owner, T, Cin, Cout, tuple(weight.shape), dy5.device),
                               x_amax=ctx.x_amax, dy_amax=dy_amax, db=db_buf, dy_pmax=dy_pmax, parts=parts)
            else:
                dwt = conv_wgrad_raw(x5, dy5, K, stride, p3, pad_mode, x_amax=ctx.x_amax, dy_amax=dy_amax, db=db_buf,
                                     dy_pmax=dy_pmax, parts=parts)
                dw = weight_unpack(dwt, tuple(weight.shape))
        elif db_buf is not None:
            bias_grad(dy5, db_buf)                                                           # accumulates
        return dx, dw, db


_LAST_CONV_AMAX = [None]


def conv(x, weight, bias=None, owner=None, stride=1, pad=0, pad_mode=0, act=0, slope=0.0, skip=False, sole=False):
    """sole=True: the caller guarantees that the returned tensor feeds exactly ONE consumer (the next conv of a chain);
    with a LeakyReLU epilogue that consumer's dgrad may then fold this activation's backward into its own epilogue."""
    _LAST_CONV_AMAX[0] = None
    out = ConvFn.apply(x, weight, bias, owner, stride, pad, pad_mode, act, slope, skip)
    if _LAST_CONV_AMAX[0] is not None and not skip:
        tag_amax(out, _LAST_CONV_AMAX[0])
        _LAST_CONV_AMAX[0] = None
    if sole and act == 1 and not skip and x.dim() == 5:
        out._df_act_sole = (act, float(slope), out._version, out.data_ptr())
    return out


# ------------------------------------------------------------------------------------------------
# 7x7 convolutions as tap-stack / tap-sum + 1x1 GEMM
# ------------------------------------------------------------------------------------------------
class TapStackFn(Function):
    """x [N,C,H,W] -> S [N, C*K*K, H, W],  S[c*T+tap][p] = x[c][map(p + tap - pad)]."""

    @staticmethod
    def forward(ctx, x, K, pad, pad_mode):
        _need(x)
        x = _c(x)
        N, C, H, W = x.shape
        s = torch.empty((N, C * K * K, H, W), device=x.device, dtype=torch.float32)
        check(lib().dfmir_tapstack_fwd(_p(x), _p(s), N, C, H, W, K, pad, pad_mode, _st()))
        ctx.meta = (N, C, H, W, K, pad, pad_mode)
        return s

    @staticmethod
    @once_differentiable
    def backward(ctx, ds):
        N, C, H, W, K, pad, pad_mode = ctx.meta
        ds = _c(ds)
        dx = torch.empty((N, C, H, W), device=ds.device, dtype=torch.float32)
        check(lib().dfmir_tapstack_bwd(_p(ds), _p(dx), N, C, H, W, K, pad, pad_mode, _st()))
        return dx, None, None, None


class TapSumFn(Function):
    """Z [N, C*K*K, H, W] -> y [N,C,H,W] = act(bias + sum_tap Z[c*T+tap][map(p + tap - pad)])."""

    @staticmethod
    def forward(ctx, z, bias, C, K, pad, pad_mode, act, slope):
        _need(z, bias)
        z = _c(z)
        N, CT, H, W = z.shape
        y = torch.empty((N, C, H, W), device=z.device, dtype=torch.float32)
        check(lib().dfmir_tapsum_fwd(_p(z), _p(bias), _p(y), N, C, H, W, H, W, K, pad, pad_mode, act, float(slope), _st()))
        ctx.meta = (N, C, H, W, K, pad, pad_mode, act, float(slope))
        ctx.save_for_backward(y if act else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        N, C, H, W, K, pad, pad_mode, act, slope = ctx.meta
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        if act:
            dpre = torch.empty_like(dy)
            check(lib().dfmir_act_bwd(_p(dy), _p(y), _p(dpre), dy.numel(), act, slope, _st()))
            dy = dpre
        dz = db = None
        if ctx.needs_input_grad[0]:
            dz = torch.empty((N, C * K * K, H, W), device=dy.device, dtype=torch.float32)
            check(lib().dfmir_tapsum_bwd(_p(dy), _p(dz), N, C, H, W, H, W, K, pad, pad_mode, _st()))
        if ctx.has_bias and ctx.needs_input_grad[1]:
            db = zeros(C, dy.device)
            bias_grad(dy.view(N, C, H * W), db)
        return dz, db, None, None, None, None, None, None


class Stem7Fn(Function):
    """7x7 conv of a single-channel image (Cout <= 64, pad 3) straight from the image (csrc/taps.hip), not through
    the 49-plane tap stack.  Backward: dW, db by the direct kernel; dx (only when the image needs a gradient) through
    the tap-stack adjoint."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad_mode):
        _need(x, weight, bias)
        x = _c(x)
        N, _, H, W = x.shape
        Cout = weight.shape[0]
        y = torch.empty((N, Cout, H, W), device=x.device, dtype=torch.float32)
        check(lib().dfmir_conv7x7_c1_fwd(_p(x), _p(_c(weight)), _p(bias), _p(y), N, H, W, Cout, pad_mode, _st()))
        ctx.save_for_backward(x, weight)
        ctx.pad_mode = pad_mode
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _c(dy)
        N, _, H, W = x.shape
        Cout = weight.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dwb = zeros(Cout * 49 + Cout, dy.device)
            dbp = dwb[Cout * 49:] if ctx.has_bias else None
            if _DET["on"]:
                det = _DetAcc(Cout * 49, Cout if ctx.has_bias else 0, N * H * W, x, None, dy, None)
                try:
                    check(lib().dfmir_conv7x7_c1_wgrad(_p(x), _p(dy), _p(det.dw), None, N, H, W, Cout, ctx.pad_mode, _st()))
                    if ctx.has_bias:
                        check(lib().dfmir_bias_grad(_p(dy), _p(det.db), N, Cout, H * W, _st()))
                except BaseException:
                    det.abort()
                    raise
                det.end(dwb, dbp)
            else:
                check(lib().dfmir_conv7x7_c1_wgrad(_p(x), _p(dy), _p(dwb), _p(dbp), N, H, W, Cout, ctx.pad_mode, _st()))
            dw = dwb[:Cout * 49].view(Cout, 1, 7, 7)
            db = dbp if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if ctx.needs_input_grad[0]:
            wd = weight_pack(weight.detach().view(Cout, 49, 1), 1)
            ds = conv_raw(dy.unsqueeze(2), wd, None, 49, (1, 1, 1), 1, (0, 0, 0), 1, 0, 0, 0.0, (1, H, W))
            dx = torch.empty_like(x)
            check(lib().dfmir_tapstack_bwd(_p(ds), _p(dx), N, 1, H, W, 7, 3, ctx.pad_mode, _st()))
        return dx, dw, db, None


_NO_STEM7 = _env_on("DFMIR_NO_STEM7")    # A/B switch: the stem through tap stack + 1x1 GEMM


def conv_taps(x, weight, bias, pad, pad_mode, act=0, slope=0.0):
    """KxK conv with Cin == 1 or Cout <= 4 as a 1x1 GEMM over K*K tap planes (see csrc/taps.hip)."""
    Cout, Cin, K = weight.shape[0], weight.shape[1], weight.shape[2]
    T = K * K
    if Cin == 1 and K == 7 and pad == 3 and Cout <= 64 and act == 0 and x.dim() == 4 and not _NO_STEM7:
        return Stem7Fn.apply(x, weight, bias, pad_mode)
    if Cin == 1:
        s = TapStackFn.apply(x, K, pad, pad_mode)
        return conv(s, weight.view(Cout, T, 1, 1), bias, None, 1, 0, 0, act, slope)
    wz = weight.view(Cout, Cin, T).permute(0, 2, 1).reshape(Cout * T, Cin, 1, 1)   # [c*T+tap][ci]
    z = conv(x, wz, None, None, 1, 0, 0, 0, 0.0)
    return TapSumFn.apply(z, bias, Cout, K, pad, pad_mode, act, slope)


# ------------------------------------------------------------------------------------------------
# InstanceNorm (+ReLU, +residual)
# ------------------------------------------------------------------------------------------------
class InstNormFn(Function):
    @staticmethod
    def forward(ctx, x, res, relu, eps):
        _need(x, res)
        x = _c(x)
        planes = x.shape[0] * x.shape[1]
        S = x.numel() // planes
        y = torch.empty_like(x)
        mean = torch.empty(planes, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        r = _c(res) if res is not None else None
        slot = amax_slot(x.device, PROBE_SLOTS)
        check(lib().dfmir_instnorm_fwd(_p(x), _p(r), _p(y), _p(mean), _p(rstd), planes, S, float(eps),
                                       int(relu), _p(slot), _st()))
        _LAST_AMAX[0] = slot
        ctx.save_for_backward(x, mean, rstd)
        ctx.relu = int(relu)
        ctx.has_res = res is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        planes = x.shape[0] * x.shape[1]
        S = x.numel() // planes
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            slot = amax_slot(x.device, PROBE_SLOTS)
            W = x.shape[-1]
            want_cols = x.dim() == 4 and W <= 94 and S // W <= 94 and lib().dfmir_instnorm_bwd_cols_ok(S, W)   # ring-kernel sizes
            # first / last column of dx on the side: the dgrad of a reflect-padded conv reads them (ring kernel)
            cols = torch.empty(planes * 2 * (S // W), device=x.device, dtype=torch.float32) if want_cols else None
            if x.dim() == 4 and lib().dfmir_instnorm_bwd_pmax_ok(S):
                # + the maximum of every (n, c) plane: per-output-channel dY scales for the split weight gradient
                pmax = torch.empty(planes, device=x.device, dtype=torch.float32)
                check(lib().dfmir_instnorm_bwd_pmax(_p(dy), _p(x), _p(mean), _p(rstd), _p(dx), planes, S, ctx.relu,
                                                    _p(slot), _p(cols), W if want_cols else 0, _p(pmax), _st()))
                dx._df_pmax = (pmax, dx._version, dx.data_ptr())
            elif want_cols:
                check(lib().dfmir_instnorm_bwd_cols(_p(dy), _p(x), _p(mean), _p(rstd), _p(dx), planes, S, ctx.relu,
                                                    _p(slot), _p(cols), W, _st()))
            else:
                check(lib().dfmir_instnorm_bwd(_p(dy), _p(x), _p(mean), _p(rstd), _p(dx), planes, S, ctx.relu,
                                               _p(slot), _st()))
            if want_cols:
                dx._df_cols = (cols, dx._version, dx.data_ptr())
            tag_amax(dx, slot)
        dres = dy if (ctx.has_res and ctx.needs_input_grad[1]) else None
        return dx, dres, None, None


_LAST_AMAX = [None]


def instance_norm(x, res=None, relu=False, eps=1e-5):
    y = InstNormFn.apply(x, res, relu, eps)
    return tag_amax(y, _LAST_AMAX[0])


_NO_IN_BLUR = _env_on("DFMIR_NO_IN_BLUR")     # A/B switch: InstanceNorm+ReLU and Downsample as two passes


class InstNormReluBlurDownFn(Function):
    """Downsample(ReLU(InstanceNorm2d(x))) in one pass per plane (csrc/norm_resample.hip in_relu_blurdown_*): the
    full-resolution normalised tensor is never written -- it feeds only the blur and no backward needs it."""

    @staticmethod
    def forward(ctx, x, eps):
        _need(x)
        x = _c(x)
        N, C, H, W = x.shape
        planes = N * C
        z = torch.empty((N, C, H // 2, W // 2), device=x.device, dtype=torch.float32)
        mean = torch.empty(planes, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        slot = amax_slot(x.device, PROBE_SLOTS)
        check(lib().dfmir_in_relu_blurdown_fwd(_p(x), _p(z), _p(mean), _p(rstd), planes, H, W, float(eps), _p(slot), _st()))
        _LAST_AMAX[0] = slot
        ctx.save_for_backward(x, mean, rstd)
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        x, mean, rstd = ctx.saved_tensors
        dz = _c(dz)
        N, C, H, W = x.shape
        dx = torch.empty_like(x)
        slot = amax_slot(x.device, PROBE_SLOTS)
        pmax = torch.empty(N * C, device=x.device, dtype=torch.float32)
        check(lib().dfmir_in_relu_blurdown_bwd(_p(dz), _p(x), _p(mean), _p(rstd), _p(dx), N * C, H, W, _p(slot), _p(pmax),
                                               _st()))
        dx._df_pmax = (pmax, dx._version, dx.data_ptr())
        tag_amax(dx, slot)
        return dx, None


def in_relu_blurdown_ok(x):
    return (not _NO_IN_BLUR and x.dim() == 4 and x.is_cuda and x.dtype == torch.float32
            and bool(lib().dfmir_in_relu_blurdown_ok(int(x.shape[2]), int(x.shape[3]))))


def instance_norm_relu_blur_down(x, eps=1e-5):
    z = InstNormReluBlurDownFn.apply(x, eps)
    return tag_amax(z, _LAST_AMAX[0])


# ------------------------------------------------------------------------------------------------
# plane-wise resampling
# ------------------------------------------------------------------------------------------------
def _plane_op(fwd_name, bwd_name, out_hw):
    class _Fn(Function):
        @staticmethod
        def forward(ctx, x, *extra):
            _need(x)
            x = _c(x)
            N, C, H, W = x.shape
            Ho, Wo = out_hw(H, W, *extra)
            y = torch.empty((N, C, Ho, Wo), device=x.device, dtype=torch.float32)
            check(getattr(lib(), fwd_name)(_p(x), _p(y), N * C, H, W, *extra, _st()))
            ctx.shape = (N, C, H, W)
            ctx.extra = extra
            return y

        @staticmethod
        @once_differentiable
        def backward(ctx, dy):
            N, C, H, W = ctx.shape
            dy = _c(dy)
            dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32)
            check(getattr(lib(), bwd_name)(_p(dy), _p(dx), N * C, H, W, *ctx.extra, _st()))
            return (dx,) + (None,) * len(ctx.extra)

    _Fn.__name__ = fwd_name
    return _Fn


BlurDownFn = _plane_op("dfmir_blur_down_fwd", "dfmir_blur_down_bwd",
                       lambda H, W: ((H - 1) // 2 + 1, (W - 1) // 2 + 1))
BlurUpFn = _plane_op("dfmir_blur_up_fwd", "dfmir_blur_up_bwd", lambda H, W: (2 * H, 2 * W))
ReflectPadFn = _plane_op("dfmir_reflect_pad2d_fwd", "dfmir_reflect_pad2d_bwd",
                         lambda H, W, p: (H + 2 * p, W + 2 * p))


def _bound_from(y, x):
    """y is a convex combination of x's values (the [1,2,1] blurs: non-negative taps summing to 1 per output), so
    max|x| bounds max|y|: x's range probe serves y's split conversion (no standalone absmax launch)."""
    tag = getattr(x, "_df_amax", None)
    if tag is not None and _amax_ok(x, tag):
        tag_amax(y, tag[0])
    return y


def blur_down(x):
    return _bound_from(BlurDownFn.apply(x), x)


def blur_up(x):
    return _bound_from(BlurUpFn.apply(x), x)


def reflect_pad2d(x, p):
    return ReflectPadFn.apply(x, int(p))


class UpCatFn(Function):
    """cat([nearest_up2(a), b], dim=1)  (torchvoxelmorph/networks.py:97-100)."""

    @staticmethod
    def forward(ctx, a, b):
        _need(a, b)
        a, b = _c(a), _c(b)
        nd = a.dim() - 2
        N, Ca = a.shape[0], a.shape[1]
        Cb = b.shape[1]
        Da, Ha, Wa = (a.shape[2:] if nd == 3 else (1,) + tuple(a.shape[2:]))
        sd = 2 if nd == 3 else 1
        exp = (Da * sd, Ha * 2, Wa * 2) if nd == 3 else (Ha * 2, Wa * 2)
        if tuple(b.shape[2:]) != tuple(exp):
            raise DfmirHipError("upcat: skip tensor %s does not match upsampled %s" % (tuple(b.shape), exp))
        y = torch.empty((N, Ca + Cb) + tuple(b.shape[2:]), device=a.device, dtype=torch.float32)
        check(lib().dfmir_upcat_fwd(_p(a), _p(b), _p(y), N, Ca, Cb, Da, Ha, Wa, sd, _st()))
        ctx.meta = (a.shape, b.shape, N, Ca, Cb, Da, Ha, Wa, sd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        ash, bsh, N, Ca, Cb, Da, Ha, Wa, sd = ctx.meta
        dy = _c(dy)
        da = torch.empty(ash, device=dy.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        db = torch.empty(bsh, device=dy.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:   # (the skip part of dy may be unwritten when it needs no gradient)
            check(lib().dfmir_upcat_bwd(_p(dy), _p(da), _p(db), N, Ca, Cb, Da, Ha, Wa, sd, _st()))
        return da, db


def _valid_amax(t):
    tag = getattr(t, "_df_amax", None)
    if tag is not None and _amax_ok(t, tag) and tag[0].numel() == PROBE_SLOTS:
        return tag[0]
    return None


def _probe64(t):
    """A PROBE_SLOTS-wide range probe of t by a standalone pass (slot 0 = max |t|, the rest stay 0); tags t."""
    slot = amax_slot(t.device, PROBE_SLOTS)
    tc = _c(t)
    check(lib().dfmir_absmax(_p(tc), tc.numel(), _p(slot), _st()))
    if tc.data_ptr() == t.data_ptr():
        tag_amax(t, slot)
    return slot


def upcat(a, b):
    y = UpCatFn.apply(a, b)
    if torch.is_grad_enabled() and a.requires_grad and not b.requires_grad:
        y._df_nograd_tail = int(b.shape[1])    # the consumer conv's dgrad skips these channels (ConvFn)
    pa, pb = _valid_amax(a), _valid_amax(b)
    if a.dim() == 5 and not _NO_SPLIT3D:
        # 3-D: the consumer is a split conv that needs the range of y.  A part without a probe (skip tensors made by the
        # stride-2 convs, the 2-channel input) is measured on its own: 1/17 .. 1/3 of a pass over the concatenation
        pa = _probe64(a) if pa is None else pa
        pb = _probe64(b) if pb is None else pb
    if pa is not None and pb is not None:      # nearest up-sampling + concatenation create no new values
        slot = amax_slot(y.device, PROBE_SLOTS)
        check(lib().dfmir_probe_merge(_p(pa), _p(pb), _p(slot), _st()))
        tag_amax(y, slot)
    return y


_NO_UPWGRAD = _env_on("DFMIR_CONV3D_NO_UPWGRAD")     # A/B switch: weight gradient from the materialised cat
_NO_UPDGRAD = _env_on("DFMIR_CONV3D_NO_UPDGRAD")     # A/B switch: d(a) through the full-resolution dgrad + pool
_NO_UPSKIP2 = _env_on("DFMIR_CONV3D_NO_UPSKIP2")     # A/B switch: <= 2 skip channels as a second launch
_NO_UPPHASE = _env_on("DFMIR_CONV3D_NO_UPPHASE")     # A/B switch: materialise nearest_up2 + cat, one conv


# Registry of the split-weight workspaces of persistent packed buffers: after the batched re-pack of an optimizer step
# (_repack_all) every registered split is re-made by ONE launch (dfmir_conv3d_wsplit_batch) instead of one launch per
# layer and mode at its first use.
_WS3D = {"entries": {}, "key": None, "dev": None}
_NO_WS_BATCH = _env_on("DFMIR_NO_WSPLIT_BATCH")


def _ws3d_register(w_tcc, key, ws, job):
    """job = (kind, K, M, pair, Ktot, koff, Cb) as dfmir_conv3d_wsplit_batch reads them."""
    import weakref
    _WS3D["entries"][(id(w_tcc), key)] = {"ref": weakref.ref(w_tcc), "key": key, "ws": ws, "job": job}
    _WS3D["key"] = None


def _resplit_all_3d(dev):
    import numpy as np
    ents = _WS3D["entries"]
    for k_ in [k_ for k_, e in ents.items() if e["ref"]() is None]:
        del ents[k_]
        _WS3D["key"] = None
    live = [e for e in ents.values() if e["ws"].device == dev and getattr(e["ref"](), "_df_gen", None) is not None]
    if not live or _NO_WS_BATCH:
        return
    key = tuple((e["ref"]().data_ptr(), e["ws"].data_ptr()) + tuple(e["job"]) for e in live)
    if _WS3D["key"] != key:
        tab = np.zeros((len(live), 7), dtype=np.int64)
        for i, e in enumerate(live):
            kind, K, M, pair, Ktot, koff, Cb = e["job"]
            tab[i, 0] = e["ref"]().data_ptr()
            tab[i, 1] = e["ws"].data_ptr()
            tab[i, 2] = e["ws"].data_ptr() + 4 * (e["ws"].numel() - 4)
            tab[i, 3] = kind | (K << 32)
            tab[i, 4] = M | (pair << 32)
            tab[i, 5] = Ktot | (koff << 32)
            tab[i, 6] = Cb
        _WS3D["dev"] = torch.from_numpy(tab).to(dev)
        _KEEP_TABLES.append(_WS3D["dev"])
        _WS3D["key"] = key
    check(lib().dfmir_conv3d_wsplit_batch(_p(_WS3D["dev"]), len(live), _st()))
    for e in live:
        w_tcc = e["ref"]()
        w_tcc._df_ws3d[e["key"]] = (w_tcc._df_gen, e["ws"])


def _ws_cached(w_tcc, key, floats, device, job=None):
    """(ws, needs_split): a per-(packed-weight buffer, key) workspace for split weights, re-made only when the packed
    buffer was re-packed (its generation changed).  A temporary packing (no generation) gets a fresh workspace."""
    gen = getattr(w_tcc, "_df_gen", None)
    cache = getattr(w_tcc, "_df_ws3d", None) if gen is not None else None
    ent = cache.get(key) if cache is not None else None
    if ent is not None and ent[0] == gen:
        return ent[1], False
    ws = ent[1] if ent is not None else torch.empty(floats, device=device, dtype=torch.float32)   # in place: graphs hold it
    if gen is not None:
        if cache is None:
            cache = w_tcc._df_ws3d = {}
        cache[key] = (gen, ws)
        if ent is None and job is not None:
            _ws3d_register(w_tcc, key, ws, job)
    return ws, True


class _Ctx(object):
    pass


class UpCatConv3dFn(Function):
    """y = act(conv3x3x3(cat([nearest_up2(a), b], 1), weight) + bias)  (torchvoxelmorph/networks.py:64,97-100 + the next
    ConvBlock, :1506-1521) without the up-sampled / concatenated tensor in the forward pass: the up-sampled channels
    through the parity-class form (8 of the 27 products, csrc/conv3ds.hip conv3d_up_phase_k), the skip channels
    through the split kernel whose epilogue adds both, the bias and the activation.  Backward: the concatenation is
    materialised there (for the weight gradient's operand) and the existing dgrad / wgrad / pooling kernels run."""

    @staticmethod
    def forward(ctx, a, b, weight, bias, owner, act, slope):
        _need(a, b, weight, bias)
        a, b = _c(a), _c(b)
        N, Ca, D, H, W = a.shape
        Cb = b.shape[1]
        Cout = weight.shape[0]
        if tuple(b.shape[2:]) != (2 * D, 2 * H, 2 * W) or weight.shape[1] != Ca + Cb:
            raise DfmirHipError("upcat_conv3d: a %s, b %s, weight %s do not fit" % (tuple(a.shape), tuple(b.shape), tuple(weight.shape)))
        w_tcc = owner.packed(0) if owner is not None else weight_pack(weight, 0)
        pa = _valid_amax(a)
        pa = _probe64(a) if pa is None else pa
        pb = _valid_amax(b)
        pb = _probe64(b) if pb is None else pb
        y = torch.empty((N, Cout, 2 * D, 2 * H, 2 * W), device=a.device, dtype=torch.float32)
        g = DfConvGeom(N, Cb, Cout, 2 * D, 2 * H, 2 * W, 2 * D, 2 * H, 2 * W, 3, 3, 3, 1, 1, 1, 1, 1, 0, act, float(slope))
        fused_ = Cb <= 2 and act in (0, 1) and not _NO_UPSKIP2
        ws_up, split_up = _ws_cached(w_tcc, ("up", Ca, Cout, Cb if fused_ else 0), lib().dfmir_conv3d_up_ws_floats(Ca, Cout),
                                     a.device, job=(1, Ca, Cout, 0, Ca + Cb, 0, Cb if fused_ else 0))
        ws_sk, split_sk = (None, False) if (Cb <= 2 and act in (0, 1) and not _NO_UPSKIP2) else \
            _ws_cached(w_tcc, ("skip", Ca, Cb, Cout), lib().dfmir_conv3d_split_ws_floats(Cb, Cout), a.device,
                       job=(0, Cb, Cout, lib().dfmir_conv3d_split_is_pair(Cout), Ca + Cb, Ca, 0))
        slot = amax_slot(a.device, PROBE_SLOTS)
        if _PROBE_AUDIT["on"]:
            _audit_probe(a, pa, "upcat conv a %s" % (tuple(a.shape),))
            _audit_probe(b, pb, "upcat conv b %s" % (tuple(b.shape),))

        fused = Cb <= 2 and act in (0, 1) and not _NO_UPSKIP2

        def launch_fused():
            check(lib().dfmir_conv3d_up_skip2_fwd(_p(a), _p(pa), pa.numel(), _p(b), _p(pb), pb.numel(), Cb,
                                                  _p(w_tcc) if split_up else None, _p(ws_up), _p(bias), _p(y), _p(slot),
                                                  N, Ca, Cout, D, H, W, act, float(slope), _st()))

        def launch_up():
            check(lib().dfmir_conv3d_up_fwd(_p(a), _p(pa), pa.numel(), _p(w_tcc) if split_up else None, Ca + Cb, _p(ws_up),
                                            _p(y), N, Ca, Cout, D, H, W, _st()))

        def launch_skip():
            check(lib().dfmir_conv3d_split_fwd_add(ctypes.byref(g), _p(b), _p(pb), pb.numel(), _p(w_tcc) if split_sk else None,
                                                   Ca + Cb, Ca, _p(ws_sk), _p(bias), _p(y), _p(slot), _st()))

        prof = _CONV_PROFILER[0]
        vox = 8.0 * N * D * H * W
        size = "L" if Cout > 64 else ("M" if Cout > 32 else ("S" if Cout > 4 else "small"))
        if fused:
            if prof is None:
                launch_fused()
            else:
                prof("conv3dup_" + size, 2.0 * vox * Cout * (Ca + Cb) * 27, launch_fused)   # reference-equivalent FLOPs
        elif prof is None:
            launch_up()
            launch_skip()
        else:
            prof("conv3dup_" + size, 2.0 * vox * Cout * Ca * 27, launch_up)        # reference-equivalent FLOPs (27 taps)
            prof("conv3ds_" + size, 2.0 * vox * Cout * Cb * 27, launch_skip)
        _LAST_CONV_AMAX[0] = slot
        ctx.save_for_backward(a, b, weight, y if act else None)
        ctx.probes = (pa, pb)
        ctx.cfg = (3, (3, 3, 3), 1, (1, 1, 1), 0, act, slope, owner)
        ctx.has_bias = bias is not None
        ta = getattr(a, "_df_act_sole", None)          # a = the output of a LeakyReLU ConvBlock feeding only this layer
        ctx.a_act = (ta[0], ta[1]) if (_tag_ok(a, ta) and not _NO_ACTGRAD) else None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        a, b, weight, y = ctx.saved_tensors
        N, Ca, D, H, W = a.shape
        Cb = b.shape[1]
        Cout = weight.shape[0]
        nd, K, stride, p3, pad_mode, act, slope, owner = ctx.cfg
        dy = _c(dy)
        # gradient w.r.t. the conv's result (LeakyReLU backward, unless the consumer's dgrad epilogue already applied it)
        if act and not _tag_ok(dy, getattr(dy, "_df_premasked", None)):
            dpre = torch.empty_like(dy)
            if not ((dy.data_ptr() | y.data_ptr() | dpre.data_ptr()) & 15):
                slot = amax_slot(dy.device, PROBE_SLOTS)
                check(lib().dfmir_act_bwd_amax(_p(dy), _p(y), _p(dpre), dy.numel(), act, float(slope), _p(slot), _st()))
                tag_amax(dpre, slot)
            else:
                check(lib().dfmir_act_bwd(_p(dy), _p(y), _p(dpre), dy.numel(), act, float(slope), _st()))
            dy = dpre
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        up_dgrad = need_a and not need_b and Cout % 8 == 0 and not _NO_UPDGRAD
        xp = amax_slot(a.device, PROBE_SLOTS)
        check(lib().dfmir_probe_merge(_p(ctx.probes[0]), _p(ctx.probes[1]), _p(xp), _st()))
        c = _Ctx()
        gfull = DfConvGeom(N, Ca + Cb, Cout, 2 * D, 2 * H, 2 * W, 2 * D, 2 * H, 2 * W, 3, 3, 3, 1, 1, 1, 1, 1, 0, 0, 0.0)
        gather = (up_dgrad and not _NO_UPWGRAD and W % 4 == 0 and ctx.needs_input_grad[2]
                  and bool(lib().dfmir_conv3d_split_wgrad_ok(ctypes.byref(gfull))) and Cout >= 8)
        if gather:
            # the weight gradient stages its operand patch from a (at half the coordinates) and b: no concatenation
            x = None
            c.x_parts = (a, b)
        else:
            # the concatenation, for the weight gradient's operand (and the dgrad's reference shape)
            x = torch.empty((N, Ca + Cb) + tuple(b.shape[2:]), device=a.device, dtype=torch.float32)
            check(lib().dfmir_upcat_fwd(_p(a), _p(b), _p(x), N, Ca, Cb, D, H, W, 2, _st()))
        c.cfg, c.x_amax, c.has_bias = (nd, K, stride, p3, pad_mode, 0, 0.0, owner), xp, ctx.has_bias
        c.dead_tail = 0 if need_b else Cb
        c.in_act = None
        c.needs_input_grad = ((need_a or need_b) and not up_dgrad, ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        dx, dw, dbias = _conv_backward_impl(c, dy, None, x, weight, None)
        da = db = None
        if up_dgrad:
            # d(a) directly at low resolution: the sum pool of nearest_up2's adjoint composed with the dgrad is a 4x4x4
            # stride-2 conv of dy, run in parity classes (csrc/conv3ds.hip conv3d_up_dgrad_k)
            w_tcc = owner.packed(0) if owner is not None else weight_pack(weight, 0)
            ws, split = _ws_cached(w_tcc, ("updgrad", Ca, Cout), lib().dfmir_conv3d_up_dgrad_ws_floats(Ca, Cout), a.device,
                                   job=(2, Ca, Cout, 0, Ca + Cb, 0, 0))
            dya = amax_of(dy)
            if _PROBE_AUDIT["on"]:
                _audit_probe(dy, dya, "upcat conv dY %s" % (tuple(dy.shape),))
            da = torch.empty_like(a)
            slot = amax_slot(a.device, PROBE_SLOTS)
            ia = ctx.a_act

            def launch():
                check(lib().dfmir_conv3d_up_dgrad(_p(dy), _p(dya), dya.numel(), _p(w_tcc) if split else None, Ca + Cb, _p(ws),
                                                  _p(da), _p(slot), _p(a) if ia else None, float(ia[1]) if ia else 0.0,
                                                  N, Ca, Cout, D, H, W, _st()))
            prof = _CONV_PROFILER[0]
            if prof is None:
                launch()
            else:
                size = "L" if Ca > 64 else ("M" if Ca > 32 else ("S" if Ca > 4 else "small"))
                prof("conv3dup_" + size, 2.0 * 8.0 * N * D * H * W * Cout * Ca * 27, launch)     # reference-equivalent FLOPs
            tag_amax(da, slot)
            if ia:
                da._df_premasked = (da._version, da.data_ptr())
        elif dx is not None:
            dx = _c(dx)
            da = torch.empty_like(a) if need_a else None
            db = torch.empty_like(b) if need_b else None
            if da is not None or db is not None:
                check(lib().dfmir_upcat_bwd(_p(dx), _p(da), _p(db), N, Ca, Cb, D, H, W, 2, _st()))
        return da, db, dw, dbias, None, None, None


def upcat_conv3d_ok(a, b, weight):
    if _NO_UPPHASE or _NO_SPLIT3D or a.dim() != 5 or not (a.is_cuda and b.is_cuda):
        return False
    N, Ca, D, H, W = a.shape
    return (tuple(weight.shape[2:]) == (3, 3, 3) and tuple(b.shape[2:]) == (2 * D, 2 * H, 2 * W)
            and bool(lib().dfmir_conv3d_up_ok(N, Ca, int(weight.shape[0]), D, H, W)))


def upcat_conv3d(a, b, weight, bias, owner, act=0, slope=0.0, sole=False):
    """The next ConvBlock applied to cat([nearest_up2(a), b], 1); see UpCatConv3dFn."""
    _LAST_CONV_AMAX[0] = None
    out = UpCatConv3dFn.apply(a, b, weight, bias, owner, act, slope)
    if _LAST_CONV_AMAX[0] is not None:
        tag_amax(out, _LAST_CONV_AMAX[0])
        _LAST_CONV_AMAX[0] = None
    if sole and act == 1:
        out._df_act_sole = (act, float(slope), out._version, out.data_ptr())
    return out


class CatChannelsFn(Function):
    """torch.cat([a, b], dim=1) for same-shape-but-channels tensors."""

    @staticmethod
    def forward(ctx, a, b):
        _need(a, b)
        a, b = _c(a), _c(b)
        if a.shape[0] != b.shape[0] or a.shape[2:] != b.shape[2:]:
            raise DfmirHipError("cat: shapes %s / %s differ outside dim 1" % (tuple(a.shape), tuple(b.shape)))
        N = a.shape[0]
        SA, SB = a.numel() // N, b.numel() // N
        y = torch.empty((N, a.shape[1] + b.shape[1]) + tuple(a.shape[2:]), device=a.device, dtype=torch.float32)
        check(lib().dfmir_cat_channels_fwd(_p(a), _p(b), _p(y), N, SA, SB, _st()))
        ctx.meta = (a.shape, b.shape, N, SA, SB)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        ash, bsh, N, SA, SB = ctx.meta
        dy = _c(dy)
        da = torch.empty(ash, device=dy.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        db = torch.empty(bsh, device=dy.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            check(lib().dfmir_cat_channels_bwd(_p(dy), _p(da), _p(db), N, SA, SB, _st()))
        return da, db


def upcat_channels(a, b):
    return CatChannelsFn.apply(a, b)


class CatBatchFn(Function):
    """torch.cat([a, b], dim=0) (registration_model.py:187) as one copy launch."""

    @staticmethod
    def forward(ctx, a, b):
        _need(a, b)
        a, b = _c(a), _c(b)
        if a.shape[1:] != b.shape[1:]:
            raise DfmirHipError("cat_batch: shapes %s / %s differ outside dim 0" % (tuple(a.shape), tuple(b.shape)))
        y = torch.empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), device=a.device, dtype=torch.float32)
        check(lib().dfmir_cat_channels_fwd(_p(a), _p(b), _p(y), 1, a.numel(), b.numel(), _st()))
        ctx.meta = (a.shape, b.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        ash, bsh = ctx.meta
        dy = _c(dy)
        na = 1
        for s in ash:
            na *= s
        da = dy[:ash[0]] if ctx.needs_input_grad[0] else None
        db = dy[ash[0]:] if ctx.needs_input_grad[1] else None
        return da, db


def cat_batch(a, b):
    return CatBatchFn.apply(a, b)


class ScaleFn(Function):
    @staticmethod
    def forward(ctx, x, mult):
        _need(x)
        x = _c(x)
        y = torch.empty_like(x)
        check(lib().dfmir_scale(_p(x), _p(y), x.numel(), float(mult), _st()))
        ctx.mult = float(mult)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib().dfmir_scale(_p(dy), _p(dx), dy.numel(), ctx.mult, _st()))
        return dx, None


def scale(x, mult):
    return ScaleFn.apply(x, mult)


# ------------------------------------------------------------------------------------------------
# warps
# ------------------------------------------------------------------------------------------------
def _warp_fwd(src, flow, mode, add_identity):
    nd = src.dim() - 2
    out = torch.empty_like(src)
    if nd == 2:
        B, C, H, W = src.shape
        check(lib().dfmir_warp2d_fwd(_p(src), _p(flow), _p(out), B, C, H, W, mode, add_identity, _st()))
    else:
        B, C, D, H, W = src.shape
        check(lib().dfmir_warp3d_fwd(_p(src), _p(flow), _p(out), B, C, D, H, W, mode, add_identity, _st()))
    return out


def _warp_bwd(dout, src, flow, dsrc, dflow, add_identity, into_src):
    nd = src.dim() - 2
    if nd == 2:
        B, C, H, W = src.shape
        check(lib().dfmir_warp2d_bwd(_p(dout), _p(src), _p(flow), _p(dsrc), _p(dflow), B, C, H, W,
                                     add_identity, into_src, _st()))
    else:
        B, C, D, H, W = src.shape
        check(lib().dfmir_warp3d_bwd(_p(dout), _p(src), _p(flow), _p(dsrc), _p(dflow), B, C, D, H, W,
                                     add_identity, into_src, _st()))


_WARP_ATOMIC = _env_on("DFMIR_WARP_ATOMIC")     # A/B switch: d(src) through global atomics


def _warp_bwd_dsrc(dout, src, flow, dflow, add_identity, into_src):
    """Backward of a warp that needs d(src): the atomic-free, bit-reproducible owner-gather kernels where the shape is
    eligible (W % 4 == 0), else the scatter with global atomics into a zeroed buffer.  Returns d(src)."""
    nd = src.dim() - 2
    B, C = src.shape[0], src.shape[1]
    D, H, W = (src.shape[2:] if nd == 3 else (1,) + tuple(src.shape[2:]))
    n = 0 if _WARP_ATOMIC else lib().dfmir_warp_bwd_own_ws_floats(nd, B, C, D, H, W)
    if n > 0:
        dsrc = torch.empty_like(src)
        ws = torch.empty(n, device=src.device, dtype=torch.float32)
        check(lib().dfmir_warp_bwd_own(nd, _p(dout), _p(src), _p(flow), _p(dsrc), _p(dflow), B, C, D, H, W, add_identity,
                                       into_src, _p(ws), _st()))
        return dsrc
    dsrc = zeros_like(src)
    _warp_bwd(dout, src, flow, dsrc, dflow, add_identity, into_src)
    return dsrc


class WarpFn(Function):
    @staticmethod
    def forward(ctx, src, flow, mode):
        _need(src, flow)
        src, flow = _c(src), _c(flow)
        nd = src.dim() - 2
        if flow.shape[1] != nd or flow.shape[0] != src.shape[0] or flow.shape[2:] != src.shape[2:]:
            raise DfmirHipError("warp: src %s / flow %s mismatch (expected grid and input to have same "
                                "batch size and spatial shape)" % (tuple(src.shape), tuple(flow.shape)))
        ctx.save_for_backward(src, flow)
        ctx.mode = mode
        return _warp_fwd(src, flow, mode, 0)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        src, flow = ctx.saved_tensors
        dout = _c(dout)
        dflow = None
        if ctx.needs_input_grad[1]:
            dflow = zeros_like(flow) if ctx.mode == 1 else torch.empty_like(flow)
        if ctx.mode == 1:
            if ctx.needs_input_grad[0]:
                raise DfmirHipError("nearest-mode warp backward is not implemented (inference only)")
            return None, dflow, None
        if ctx.needs_input_grad[0]:
            return _warp_bwd_dsrc(dout, src, flow, dflow, 0, 0), dflow, None
        _warp_bwd(dout, src, flow, None, dflow, 0, 0)
        return None, dflow, None


def warp(src, flow, mode="bilinear"):
    return WarpFn.apply(src, flow, 1 if mode == "nearest" else 0)


class VecIntStepFn(Function):
    """v -> v + warp(v, v)   (one scaling-and-squaring step, layers.py:66-67)."""

    @staticmethod
    def forward(ctx, v):
        _need(v)
        v = _c(v)
        ctx.save_for_backward(v)
        return _warp_fwd(v, v, 0, 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (v,) = ctx.saved_tensors
        dout = _c(dout)
        return _warp_bwd_dsrc(dout, v, v, None, 1, 1)


def vecint_step(v):
    return VecIntStepFn.apply(v)


_RESIZE_ONEPASS = _env_on("DFMIR_RESIZE_ONEPASS")     # A/B switch: the one-pass gather adjoint


class ResizeFn(Function):
    @staticmethod
    def forward(ctx, x, out_sp, mult):
        _need(x)
        x = _c(x)
        nd = x.dim() - 2
        isp = tuple(x.shape[2:]) if nd == 3 else (1,) + tuple(x.shape[2:])
        osp = tuple(out_sp) if nd == 3 else (1,) + tuple(out_sp)
        planes = x.shape[0] * x.shape[1]
        y = torch.empty(tuple(x.shape[:2]) + tuple(out_sp), device=x.device, dtype=torch.float32)
        check(lib().dfmir_resize_fwd(_p(x), _p(y), planes, isp[0], isp[1], isp[2], osp[0], osp[1], osp[2],
                                     float(mult), _st()))
        ctx.meta = (x.shape, planes, isp, osp, float(mult))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xs, planes, isp, osp, mult = ctx.meta
        dy = _c(dy)
        dx = torch.empty(xs, device=dy.device, dtype=torch.float32)
        n = lib().dfmir_resize_bwd_ws_floats(planes, isp[0], isp[1], isp[2], osp[0], osp[1], osp[2])
        if n > 0 and not _RESIZE_ONEPASS:
            ws = torch.empty(n + 4, device=dy.device, dtype=torch.float32)
            check(lib().dfmir_resize_bwd_sep(_p(dy), _p(dx), planes, isp[0], isp[1], isp[2], osp[0], osp[1], osp[2],
                                             mult, _p(ws), _st()))
        else:
            check(lib().dfmir_resize_bwd(_p(dy), _p(dx), planes, isp[0], isp[1], isp[2], osp[0], osp[1], osp[2],
                                         mult, _st()))
        return dx, None, None


def resize_linear(x, out_sp, mult=1.0):
    return ResizeFn.apply(x, tuple(int(s) for s in out_sp), mult)


# ------------------------------------------------------------------------------------------------
# PatchNCE
# ------------------------------------------------------------------------------------------------
class TapForkFn(Function):
    """feat -> (main, tap): two aliases of a feature map that is tapped for PatchNCE sampling AND feeds the next
    layer.  The sampled rows' gradient is a scatter of a few hundred positions per plane; returned densely
    (zeros + scatter) autograd would then sum two full-size tensors.  `patch_gather` on `tap` instead parks
    (d rows, ids) in `stash` and returns no gradient; this node's backward scatters them into the next layer's
    gradient in place (or into zeros if `main` received none)."""

    @staticmethod
    def forward(ctx, feat, stash):
        ctx.stash = stash
        ctx.shape = tuple(feat.shape)
        ctx.set_materialize_grads(False)
        return feat.view_as(feat), feat.view_as(feat)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_main, g_tap):
        stash = ctx.stash
        g = g_main
        if stash:
            if g is None:
                g = zeros(ctx.shape, stash[0][0].device)
            elif not g.is_contiguous():
                g = g.contiguous()
            atag = getattr(g, "_df_amax", None)
            if atag is not None and not _amax_ok(g, atag):
                atag = None
            ptag = getattr(g, "_df_pmax", None)
            if ptag is not None and not (ptag[1] == g._version and ptag[2] == g.data_ptr()):
                ptag = None
            for dout, ids, (shape, B, C, S, Pn, G, distinct) in stash:
                if atag is not None and atag[0].numel() != PROBE_SLOTS:
                    atag = None
                if ptag is not None and (atag is None or ptag[0].numel() != B * C):
                    ptag = None
                # with a probe: keep g's range probes (from the InstanceNorm backward that produced it) valid
                if not distinct:          # caller-supplied ids with repeats: accumulate with atomics
                    check(lib().dfmir_patch_gather_bwd_any(_p(dout), _p(ids), _p(g), B, C, S, Pn, G,
                                                           _p(atag[0]) if atag is not None else None,
                                                           _p(ptag[0]) if ptag is not None else None, _st()))
                elif ptag is not None:
                    check(lib().dfmir_patch_gather_bwd_gp(_p(dout), _p(ids), _p(g), B, C, S, Pn, G, _p(atag[0]),
                                                          _p(ptag[0]), _st()))
                else:
                    check(lib().dfmir_patch_gather_bwd_g(_p(dout), _p(ids), _p(g), B, C, S, Pn, G,
                                                         _p(atag[0]) if atag is not None else None, _st()))
            del stash[:]
            # modified through the raw pointer: tags that were not maintained would be stale
            stale = ["_df_cols"]
            if atag is None:
                stale.append("_df_amax")
            if ptag is None:
                stale.append("_df_pmax")
            for tag in stale:
                if hasattr(g, tag):
                    delattr(g, tag)
        if g_tap is not None:                    # a dense gradient on the tap (some other use of it)
            g = g_tap if g is None else g + g_tap
        return g, None


def fork_tap(feat):
    """(main, tap) for a tapped feature that also feeds the next layer; see TapForkFn.  Without grad: (feat, feat)."""
    if not (torch.is_grad_enabled() and feat.requires_grad and feat.is_contiguous()):
        return feat, feat
    stash = []
    main, tap = TapForkFn.apply(feat, stash)
    tap._df_tap_stash = stash
    tag = getattr(feat, "_df_amax", None)
    if tag is not None and _amax_ok(feat, tag):
        tag_amax(main, tag[0])
        tag_amax(tap, tag[0])
    return main, tap


class PatchGatherFn(Function):
    """feat [B,C,*sp], ids int64 [P] (or [G,P]: image b uses row b // (B/G)) -> channel-major rows [C, B*P]."""

    @staticmethod
    def forward(ctx, feat, ids, groups=1, distinct=True):
        _need(feat, ids)
        ctx.stash = getattr(feat, "_df_tap_stash", None) if feat.is_contiguous() else None
        feat = _c(feat)
        ids = _c(ids.to(torch.int64))
        B, C = feat.shape[0], feat.shape[1]
        S = feat.numel() // (B * C)
        G = int(groups)
        if ids.numel() % G or B % G:
            raise DfmirHipError("patch_gather: ids [G,P] with G dividing the batch")
        Pn = ids.numel() // G
        out = torch.empty((C, B * Pn), device=feat.device, dtype=torch.float32)
        check(lib().dfmir_patch_gather_fwd_g(_p(feat), _p(ids), _p(out), B, C, S, Pn, G, _st()))
        ctx.save_for_backward(ids)
        ctx.meta = (feat.shape, B, C, S, Pn, G, bool(distinct))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        shape, B, C, S, Pn, G, distinct = ctx.meta
        dout = _c(dout)
        if ctx.stash is not None:                # collected by TapForkFn.backward, which runs after this node
            ctx.stash.append((dout, ids, ctx.meta))
            return None, None, None, None
        dfeat = zeros(shape, dout.device)
        if distinct:
            check(lib().dfmir_patch_gather_bwd_g(_p(dout), _p(ids), _p(dfeat), B, C, S, Pn, G, None, _st()))
        else:
            check(lib().dfmir_patch_gather_bwd_any(_p(dout), _p(ids), _p(dfeat), B, C, S, Pn, G, None, None, _st()))
        return dfeat, None, None, None


def mark_distinct(ids):
    """Tag an id tensor as a P-subset per group (what torch.randperm / dfmir_patch_ids_draw yield)."""
    ids._df_distinct = (ids._version, ids.data_ptr())
    return ids


def ids_distinct(ids, groups=1):
    """Are the ids of every group pairwise distinct?  Generated ids carry the tag of mark_distinct; a caller-supplied
    tensor (PatchSampleF.forward(patch_ids=...), model.patch_id_source) is checked on the host ONCE per tensor state
    (one device sync).  The non-atomic scatter of patch_gather's backward is a data race on repeated ids, so those go
    through the accumulating kernel instead."""
    tag = getattr(ids, "_df_distinct", None)
    if tag is not None and tag == (ids._version, ids.data_ptr()):
        return True
    # the verdict lives ON the tensor object (an address + version key does not identify content: a fresh tensor per
    # step usually gets the allocator's previous address with version 0)
    key = (ids._version, ids.data_ptr(), int(groups))
    seen = getattr(ids, "_df_distinct_checked", None)
    if seen is not None and seen[0] == key:
        return seen[1]
    if torch.cuda.is_current_stream_capturing():
        return False                          # no host round trip inside a capture: take the safe (atomic) form
    srt = torch.sort(ids.reshape(int(groups), -1), dim=1).values
    hit = bool((srt[:, 1:] != srt[:, :-1]).all()) if srt.shape[1] > 1 else True
    ids._df_distinct_checked = (key, hit)
    return hit


def patch_gather(feat, ids, groups=1):
    distinct = ids_distinct(ids, groups) if (torch.is_grad_enabled() and feat.requires_grad) else True
    return PatchGatherFn.apply(feat, ids, groups, distinct)


class L2NormFn(Function):
    """x [C, rows] -> x / (||x||_2 over C + eps)   (networks.py:499-502)."""

    @staticmethod
    def forward(ctx, x, eps):
        _need(x)
        x = _c(x)
        C, rows = x.shape
        y = torch.empty_like(x)
        nrm = torch.empty(rows, device=x.device, dtype=torch.float32)
        check(lib().dfmir_l2norm_fwd(_p(x), _p(y), _p(nrm), C, rows, float(eps), _st()))
        ctx.save_for_backward(x, nrm)
        ctx.eps = float(eps)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, nrm = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        check(lib().dfmir_l2norm_bwd(_p(dy), _p(x), _p(nrm), _p(dx), x.shape[0], x.shape[1], ctx.eps, _st()))
        return dx, None


def l2norm_rows(x, eps=1e-7):
    return L2NormFn.apply(x, eps)


class PatchNCEFn(Function):
    """q, k [C, rows] -> per-row InfoNCE loss [rows]; gradient flows to q only (k is detached,
    patchnce.py:17)."""

    @staticmethod
    def forward(ctx, q, k, groups, T):
        _need(q, k)
        q, k = _c(q), _c(k)
        C, rows = q.shape
        if rows % groups != 0:
            raise DfmirHipError("patchnce: rows %d not divisible by groups %d" % (rows, groups))
        R = rows // groups
        loss = torch.empty(rows, device=q.device, dtype=torch.float32)
        probs = torch.empty((rows, R + 1), device=q.device, dtype=torch.float32)
        check(lib().dfmir_patchnce_fwd(_p(q), _p(k), _p(loss), _p(probs), rows, C, groups, float(T), _st()))
        ctx.save_for_backward(probs, k)
        ctx.meta = (rows, C, groups, float(T))
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        probs, k = ctx.saved_tensors
        rows, C, groups, T = ctx.meta
        dloss = _c(dloss)
        dq = torch.empty_like(k)
        check(lib().dfmir_patchnce_bwd(_p(dloss), _p(probs), _p(k), _p(dq), rows, C, groups, T, _st()))
        return dq, None, None, None


def patchnce_rows(q, k, groups, T):
    return PatchNCEFn.apply(q, k, groups, T)


class NCETermsFn(Function):
    """Per-row NCE losses of L layers and T stacked terms -> the T per-term losses
    scale * sum_l mean(rows_l[t])  (registration_model.py:247-253 for every term at once).  q[l], k[l]: channel-major
    [C, T*seg]; one PatchNCE launch per layer writes into one [L, T*seg] buffer, one launch reduces it."""

    @staticmethod
    def forward(ctx, groups, temp, scale, n_terms, *qk):
        L = len(qk) // 2
        qs = [_c(t) for t in qk[:L]]
        ks = [_c(t) for t in qk[L:]]
        _need(*qs)
        _need(*ks)
        rows = qs[0].shape[1]
        if rows % n_terms or rows % groups:
            raise DfmirHipError("nce_terms: %d rows, %d terms, %d groups" % (rows, n_terms, groups))
        seg = rows // n_terms
        R = rows // groups
        dev = qs[0].device
        buf = torch.empty((L, rows), device=dev, dtype=torch.float32)
        probs = []
        for l in range(L):
            C = qs[l].shape[0]
            pr = torch.empty((rows, R + 1), device=dev, dtype=torch.float32)
            check(lib().dfmir_patchnce_fwd(_p(qs[l]), _p(ks[l]), _VP(buf.data_ptr() + 4 * l * rows), _p(pr), rows, C,
                                           groups, float(temp), _st()))
            probs.append(pr)
        out = torch.empty(n_terms, device=dev, dtype=torch.float32)
        check(lib().dfmir_segment_means_fwd(_p(buf), _p(out), L, n_terms, seg, float(scale), _st()))
        ctx.save_for_backward(*(probs + ks))
        ctx.meta = (L, rows, seg, groups, float(temp), float(scale), n_terms, [q.shape[0] for q in qs])
        ctx.rows_buf = buf            # kept for inspection (tests read the per-row losses)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        L, rows, seg, groups, temp, scale, n_terms, Cs = ctx.meta
        saved = ctx.saved_tensors
        probs, ks = saved[:L], saved[L:]
        g = _c(g)
        drows = torch.empty((L, rows), device=g.device, dtype=torch.float32)
        check(lib().dfmir_segment_means_bwd(_p(g), _p(drows), L, n_terms, seg, scale, _st()))
        dqs = []
        for l in range(L):
            dq = torch.empty_like(ks[l])
            check(lib().dfmir_patchnce_bwd(_VP(drows.data_ptr() + 4 * l * rows), _p(probs[l]), _p(ks[l]), _p(dq), rows,
                                           Cs[l], groups, temp, _st()))
            dqs.append(dq)
        return (None, None, None, None) + tuple(dqs) + (None,) * L


def nce_terms(qs, ks, groups, temp, scale, n_terms):
    """qs, ks: lists (per layer) of channel-major [C, T*seg] rows -> tensor [n_terms] of per-term losses."""
    return NCETermsFn.apply(int(groups), temp, scale, int(n_terms), *(list(qs) + [k.detach() for k in ks]))


class ScalarCombineFn(Function):
    """out[j] = sum_i M[j][i] * in_i for 0-dim device scalars in_i (one launch; gradients are M^T g)."""

    @staticmethod
    def forward(ctx, M, *ins):
        import numpy as np
        _need(*ins)
        n_in, n_out = len(ins), len(M)
        Mh = np.ascontiguousarray(np.asarray(M, dtype=np.float32).reshape(n_out, n_in))
        ptrs = (ctypes.c_void_p * n_in)(*[t.data_ptr() for t in ins])
        out = torch.empty(n_out, device=ins[0].device, dtype=torch.float32)
        check(lib().dfmir_scalar_combine_fwd(ptrs, n_in, Mh.ctypes.data, n_out, _p(out), _st()))
        ctx.M = Mh
        ctx.shapes = [tuple(t.shape) for t in ins]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _c(g)
        n_out, n_in = ctx.M.shape
        din = torch.empty(n_in, device=g.device, dtype=torch.float32)
        check(lib().dfmir_scalar_combine_bwd(_p(g), n_in, ctx.M.ctypes.data, n_out, _p(din), _st()))
        return (None,) + tuple(din[i].view(ctx.shapes[i]) for i in range(n_in))


def scalar_combine(M, ins):
    """M: n_out rows of n_in coefficients; ins: device scalars (0-dim or 1-element tensors) -> tensor [n_out]."""
    return ScalarCombineFn.apply(M, *ins)


# device-side patch-id generator state: {seed, draw counter, finished-workgroup counter}
_IDS_STATE = {}


def _cuda_device(device):
    device = torch.device("cuda" if device is None else device)
    if device.type != "cuda":
        raise DfmirHipError("patch ids are drawn on the HIP device, not on %s" % device)
    return torch.device("cuda", torch.cuda.current_device() if device.index is None else device.index)


def seed_patch_ids(seed, device=None):
    """(Re)seed the device patch-id generator (default seed: torch's initial_seed at first use).  The state tensor is
    updated in place once it exists: a captured hipGraph holds its address."""
    device = _cuda_device(device)
    new = torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64)
    st = _IDS_STATE.get(device)
    if st is None:
        st = _IDS_STATE[device] = new.to(device)
    else:
        st.copy_(new)
    return st


def draw_patch_ids(sizes, n_sets, P, device):
    """[n_layers, n_sets, P] int64: for each layer l and set t a uniformly random P-subset of [0, sizes[l])
    (PatchSampleF's torch.randperm(S)[:P], models/networks.py:609-610), all in ONE launch, no host round trip."""
    device = _cuda_device(device)
    st = _IDS_STATE.get(device)
    if st is None:
        st = seed_patch_ids(torch.initial_seed(), device)
    L = len(sizes)
    out = torch.empty((L, n_sets, P), device=device, dtype=torch.int64)
    sz = (ctypes.c_longlong * L)(*[int(s) for s in sizes])
    check(lib().dfmir_patch_ids_draw(_p(st), sz, L, int(n_sets), int(P), _p(out), _st()))
    return mark_distinct(out)


def patch_gather_multi(srcs, ids):
    """srcs: G tensors [Bper, C, *sp] (no gradient: the detached key side); ids [G, P] -> [C, G*Bper*P]."""
    _need(*srcs)
    _need(ids)
    G = len(srcs)
    srcs = [_c(s) for s in srcs]
    Bper, C = srcs[0].shape[0], srcs[0].shape[1]
    S = srcs[0].numel() // (Bper * C)
    for s_ in srcs:
        if tuple(s_.shape) != tuple(srcs[0].shape):
            raise DfmirHipError("patch_gather_multi: sources differ in shape")
    ids = _c(ids.to(torch.int64))
    if ids.dim() != 2 or ids.shape[0] != G:
        raise DfmirHipError("patch_gather_multi: ids must be [G, P]")
    Pn = ids.shape[1]
    out = torch.empty((C, G * Bper * Pn), device=srcs[0].device, dtype=torch.float32)
    ptrs = (ctypes.c_void_p * G)(*[s_.data_ptr() for s_ in srcs])
    check(lib().dfmir_patch_gather_fwd_multi(ptrs, G, _p(ids), _p(out), Bper, C, S, Pn, _st()))
    return out


_NO_NCE_FUSED = _env_on("DFMIR_NO_NCE_FUSED")     # A/B switch: gather, two 1x1 convs and l2norm as 4 launches


def nce_head_ok(C, nc, use_mlp):
    return use_mlp and not _NO_NCE_FUSED and nc == 256 and 1 <= C <= 256


def _nce_head_launch(ptrs, G, ids, mlp0, mlp2, Bper, C, S, Pn, eps, device, save):
    rows = G * Bper * Pn
    out = torch.empty((256, rows), device=device, dtype=torch.float32)
    nrm = xs = hs = ypre = None
    if save:
        nrm = torch.empty(rows, device=device, dtype=torch.float32)
        xs = torch.empty((C, rows), device=device, dtype=torch.float32)
        hs = torch.empty((256, rows), device=device, dtype=torch.float32)
        ypre = torch.empty((256, rows), device=device, dtype=torch.float32)
    arr = (ctypes.c_void_p * G)(*ptrs)
    check(lib().dfmir_nce_head_fwd(arr, G, _p(ids), _p(mlp0.packed(0)), _p(mlp0.bias), _p(mlp2.packed(0)), _p(mlp2.bias),
                                   _p(out), _p(nrm), _p(xs), _p(hs), _p(ypre), Bper, C, S, Pn, float(eps), _st()))
    return out, nrm, xs, hs, ypre


def nce_head_multi(srcs, ids, mlp0, mlp2, eps=1e-7):
    """Key side (no gradient): G source tensors [Bper, C, *sp], ids [G, P] -> L2-normalised projections [256, G*Bper*P]."""
    _need(*srcs)
    srcs = [_c(s_) for s_ in srcs]
    Bper, C = srcs[0].shape[0], srcs[0].shape[1]
    S = srcs[0].numel() // (Bper * C)
    ids = _c(ids.to(torch.int64))
    out, _, _, _, _ = _nce_head_launch([s_.data_ptr() for s_ in srcs], len(srcs), ids, mlp0, mlp2, Bper, C, S, ids.shape[1], eps,
                                       srcs[0].device, False)
    return out


class NceHeadFn(Function):
    """Query side: feat [B, C, *sp] (the G terms' images stacked along the batch), ids [G, P] -> [256, B*P]; the backward
    runs the chain the unfused modules would run (l2norm, the two Linear layers' dgrad / wgrad / bias gradients, the
    sampled-feature scatter) on the intermediates the fused forward saved."""

    @staticmethod
    def forward(ctx, feat, ids, w1, b1, w2, b2, mlp0, mlp2, eps, distinct):
        _need(feat, ids)
        ctx.stash = getattr(feat, "_df_tap_stash", None) if feat.is_contiguous() else None
        feat = _c(feat)
        ids = _c(ids.to(torch.int64))
        G, Pn = ids.shape
        B, C = feat.shape[0], feat.shape[1]
        if B % G:
            raise DfmirHipError("nce_head: ids [G,P] with G dividing the batch")
        Bper = B // G
        S = feat.numel() // (B * C)
        base = feat.data_ptr()
        ptrs = [base + 4 * g * Bper * C * S for g in range(G)]
        out, nrm, xs, hs, ypre = _nce_head_launch(ptrs, G, ids, mlp0, mlp2, Bper, C, S, Pn, eps, feat.device, True)
        ctx.save_for_backward(ids, nrm, xs, hs, ypre, w1, w2)
        ctx.meta = (tuple(feat.shape), B, C, S, Pn, G, bool(distinct), float(eps))
        ctx.mods = (mlp0, mlp2)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        ids, nrm, xs, hs, ypre, w1, w2 = ctx.saved_tensors
        shape, B, C, S, Pn, G, distinct, eps = ctx.meta
        mlp0, mlp2 = ctx.mods
        rows = B * Pn
        dout = _c(dout)
        dy = torch.empty_like(ypre)
        check(lib().dfmir_l2norm_bwd(_p(dout), _p(ypre), _p(nrm), _p(dy), 256, rows, eps, _st()))
        # Linear(256, 256): y = W2 h + b2
        c2 = _Ctx()
        c2.cfg, c2.x_amax, c2.dead_tail, c2.in_act, c2.has_bias = (2, (1, 1, 1), 1, (0, 0, 0), 0, 0, 0.0, mlp2), None, 0, None, True
        c2.needs_input_grad = (True, ctx.needs_input_grad[4], ctx.needs_input_grad[5])
        dh, dw2, db2 = _conv_backward_impl(c2, dy.view(1, 256, 1, rows), None, hs.view(1, 256, 1, 1, rows),
                                           w2.view(256, 256, 1, 1), None)
        # Linear(C, 256) + ReLU: h = relu(W1 x + b1)
        c1 = _Ctx()
        c1.cfg, c1.x_amax, c1.dead_tail, c1.in_act, c1.has_bias = (2, (1, 1, 1), 1, (0, 0, 0), 0, 1, 0.0, mlp0), None, 0, None, True
        c1.needs_input_grad = (ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        dx, dw1, db1 = _conv_backward_impl(c1, dh, None, xs.view(1, C, 1, 1, rows), w1.view(256, C, 1, 1),
                                           hs.view(1, 256, 1, 1, rows))
        dfeat = None
        if dx is not None:
            dxr = _c(dx).view(C, rows)
            meta = (shape, B, C, S, Pn, G, distinct)
            if ctx.stash is not None:
                ctx.stash.append((dxr, ids, meta))
            else:
                dfeat = zeros(shape, dxr.device)
                if distinct:
                    check(lib().dfmir_patch_gather_bwd_g(_p(dxr), _p(ids), _p(dfeat), B, C, S, Pn, G, None, _st()))
                else:
                    check(lib().dfmir_patch_gather_bwd_any(_p(dxr), _p(ids), _p(dfeat), B, C, S, Pn, G, None, None, _st()))
        dw1 = dw1.view(256, C) if dw1 is not None else None          # the Linear parameters are [out, in]
        dw2 = dw2.view(256, 256) if dw2 is not None else None
        return dfeat, None, dw1, db1, dw2, db2, None, None, None, None


def nce_head(feat, ids, mlp0, mlp2, eps=1e-7):
    """Query side with gradient; see NceHeadFn."""
    groups = ids.shape[0] if ids.dim() == 2 else 1
    ids2 = ids if ids.dim() == 2 else ids.view(1, -1)
    distinct = ids_distinct(ids, groups) if (torch.is_grad_enabled() and feat.requires_grad) else True
    return NceHeadFn.apply(feat, ids2, mlp0.weight, mlp0.bias, mlp2.weight, mlp2.bias, mlp0, mlp2, eps, distinct)


# ------------------------------------------------------------------------------------------------
# scalar losses
# ------------------------------------------------------------------------------------------------
_LAST_L1_WS = [None]


class MaskedL1Fn(Function):
    @staticmethod
    def forward(ctx, a, b, mask, thr):
        _need(a, b, mask)
        a, b = _c(a), _c(b)
        m = None
        if mask is not None:
            m = _c(mask.to(torch.bool))
            if m.shape != a.shape:
                m = _c(m.expand_as(a))
        ws = torch.empty(8, device=a.device, dtype=torch.float32)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        check(lib().dfmir_masked_l1_fwd(_p(a), _p(b), _p(m), float(thr), _p(ws), _p(out), a.numel(), _st()))
        _LAST_L1_WS[0] = ws
        ctx.save_for_backward(a, b, m, ws)
        ctx.thr = float(thr)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b, m, ws = ctx.saved_tensors
        g = _c(g)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            check(lib().dfmir_masked_l1_bwd(_p(a), _p(b), _p(m), ctx.thr, _p(ws), _p(g), _p(da), _p(db),
                                            a.numel(), _st()))
        return da, db, None, None


def masked_l1(a, b, mask=None, thr=-0.95):
    """sum(|a-b|*m)/sum(m); m = mask if given else (a>thr)|(b>thr).  The result carries `_df_mask_sum` = sum(m) (a
    0-dim view of the kernel's workspace) for the global-batch normalisation of the data-parallel form."""
    out = MaskedL1Fn.apply(a, b, mask, thr)
    out._df_mask_sum = _LAST_L1_WS[0][1]
    return out


class FlowSmoothFn(Function):
    @staticmethod
    def forward(ctx, flow, penalty=2):
        _need(flow)
        flow = _c(flow)
        nd = flow.dim() - 2
        B, C = flow.shape[0], flow.shape[1]
        D, H, W = flow.shape[2:] if nd == 3 else (1,) + tuple(flow.shape[2:])
        ws = torch.empty(int(lib().dfmir_flow_smooth_ws_floats()), device=flow.device, dtype=torch.float32)
        out = torch.empty((), device=flow.device, dtype=torch.float32)
        if penalty == 2:
            check(lib().dfmir_flow_smooth_fwd(_p(flow), _p(ws), _p(out), B, C, D, H, W, _st()))
        else:
            check(lib().dfmir_flow_smooth_fwd_p(_p(flow), _p(ws), _p(out), B, C, D, H, W, int(penalty), _st()))
        ctx.save_for_backward(flow)
        ctx.meta = (B, C, D, H, W, int(penalty))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (flow,) = ctx.saved_tensors
        B, C, D, H, W, penalty = ctx.meta
        g = _c(g)
        df = torch.empty_like(flow)
        if penalty == 2:
            check(lib().dfmir_flow_smooth_bwd(_p(flow), _p(g), _p(df), B, C, D, H, W, _st()))
        else:
            check(lib().dfmir_flow_smooth_bwd_p(_p(flow), _p(g), _p(df), B, C, D, H, W, penalty, _st()))
        return df, None


def flow_smoothness(flow, penalty='l2'):
    """mean over axes of mean(|forward difference|^p): smooothing_loss (registration_model.py:25-32), Grad_Loss
    (util/losses.py:81-130) and vxm Grad (torchvoxelmorph/losses.py:93-117); penalty 'l2' (p = 2) or 'l1' (p = 1)."""
    if penalty not in ('l1', 'l2'):
        raise DfmirHipError("gradient penalty must be 'l1' or 'l2', got %r" % (penalty,))
    return FlowSmoothFn.apply(flow, 2 if penalty == 'l2' else 1)


class MulFn(Function):
    """a * b element-wise, same shapes (`prediction * mask`, util/losses.py:120-121)."""

    @staticmethod
    def forward(ctx, a, b):
        _need(a, b)
        a, b = _c(a), _c(b)
        if a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32:
            raise DfmirHipError("ops.mul: fp32 tensors of one shape (got %s, %s)" % (tuple(a.shape), tuple(b.shape)))
        out = torch.empty_like(a)
        check(lib().dfmir_mul(_p(a), _p(b), _p(out), a.numel(), _st()))
        ctx.save_for_backward(a if ctx.needs_input_grad[1] else None, b if ctx.needs_input_grad[0] else None)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _c(g)
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(g)
            check(lib().dfmir_mul(_p(g), _p(b), _p(da), g.numel(), _st()))
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(g)
            check(lib().dfmir_mul(_p(g), _p(a), _p(db), g.numel(), _st()))
        return da, db


def mul(a, b):
    return MulFn.apply(a, b)


class NCCFn(Function):
    """Windowed NCC with a win^nd mean window; gradient w.r.t. the prediction I only (J is the fixed target).
    mode 0: -sqrt(mean(cc)) (NCC_Loss, util/losses.py:248-256), with a mask -sqrt(sum(cc * mask) / sum(mask)) and 0 for
    an empty mask (:257-261); mode 1: -mean(cc) (vxm NCC.loss, torchvoxelmorph/losses.py:67)."""

    @staticmethod
    def forward(ctx, I, J, win, eps, mask=None, mode=0):
        _need(I, J, mask)
        I, J = _c(I), _c(J)
        if I.shape[1] != 1:
            raise DfmirHipError("NCC expects single-channel volumes")
        if mask is not None and (mask.dtype != torch.float32 or mask.shape != I.shape or not mask.is_contiguous()):
            raise DfmirHipError("NCC mask: a contiguous fp32 tensor of the volumes' shape")
        nd = I.dim() - 2
        B = I.shape[0]
        D, H, W = I.shape[2:] if nd == 3 else (1,) + tuple(I.shape[2:])
        N = I.numel()
        sums = torch.empty(5 * N, device=I.device, dtype=torch.float32)
        tmp2 = torch.empty(5 * N, device=I.device, dtype=torch.float32)
        ws = torch.empty(8, device=I.device, dtype=torch.float32)
        out = torch.empty((), device=I.device, dtype=torch.float32)
        if mask is None and mode == 0:
            check(lib().dfmir_ncc_fwd(_p(I), _p(J), _p(sums), _p(tmp2), _p(ws), _p(out), B, D, H, W, int(win),
                                      float(eps), _st()))
        else:
            check(lib().dfmir_ncc_fwd_m(_p(I), _p(J), _p(mask), int(mode), _p(sums), _p(tmp2), _p(ws), _p(out), B, D, H, W,
                                        int(win), float(eps), _st()))
        ctx.save_for_backward(I, J, sums, ws, mask)
        ctx.meta = (B, D, H, W, int(win), float(eps), int(mode))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        I, J, sums, ws, mask = ctx.saved_tensors
        B, D, H, W, win, eps, mode = ctx.meta
        g = _c(g)
        N = I.numel()
        t1 = torch.empty(3 * N, device=I.device, dtype=torch.float32)
        t2 = torch.empty(3 * N, device=I.device, dtype=torch.float32)
        dI = torch.empty_like(I)
        if mask is None and mode == 0:
            check(lib().dfmir_ncc_bwd(_p(I), _p(J), _p(sums), _p(t1), _p(t2), _p(ws), _p(g), _p(dI), B, D, H, W,
                                      win, eps, _st()))
        else:
            check(lib().dfmir_ncc_bwd_m(_p(I), _p(J), _p(mask), mode, _p(sums), _p(t1), _p(t2), _p(ws), _p(g), _p(dI),
                                        B, D, H, W, win, eps, _st()))
        return dI, None, None, None, None, None


def ncc_loss(I, J, win=9, eps=1e-5, mask=None, reduction='neg_sqrt_mean'):
    """reduction 'neg_sqrt_mean' (NCC_Loss) or 'neg_mean' (vxm NCC); mask: any tensor that broadcasts to I's shape
    (bool / byte / float: the reference multiplies cc by it, util/losses.py:261)."""
    if mask is not None:
        mask = mask.to(device=I.device, dtype=torch.float32).expand_as(I).contiguous()
    return NCCFn.apply(I, J, win, eps, mask, {'neg_sqrt_mean': 0, 'neg_mean': 1}[reduction])


class MeanFn(Function):
    @staticmethod
    def forward(ctx, x):
        _need(x)
        x = _c(x)
        out = torch.empty((), device=x.device, dtype=torch.float32)
        check(lib().dfmir_sum_scaled(_p(x), _p(out), x.numel(), 1.0 / x.numel(), _st()))
        ctx.shape = x.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _c(g)
        n = 1
        for s in ctx.shape:
            n *= s
        dx = torch.empty(ctx.shape, device=g.device, dtype=torch.float32)
        check(lib().dfmir_fill_from_scalar(_p(g), _p(dx), n, 1.0 / n, _st()))
        return dx


def mean(x):
    return MeanFn.apply(x)


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    _need(p, g, m, v)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    check(lib().dfmir_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2),
                                float(eps), float(bc1), float(bc2), float(grad_scale), _st()))
    bump_weights_epoch()

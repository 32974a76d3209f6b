/*
 * dfmir_hip.h -- C ABI of libdfmir_hip.so: the MI355X (gfx950) compute library behind the
 * DFMIR `--model registration` training step.
 *
 * The reference (heyblackC/DFMIR) has no native boundary: its hot path calls torch.nn /
 * torch.nn.functional ops from Python.  Each entry point below replaces ONE such torch op call
 * site on the path REGISTRATIONModel.set_input -> optimize_parameters
 * (reference models/registration_model.py:138-183); the reference line each one stands in for is
 * cited next to it.  The host side (the dfmir_amd Python package) binds these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to caller-allocated, contiguous fp32 memory laid out
 *     NCHW / NCDHW exactly like the reference's tensors (2-D tensors are D == 1);
 *   - `stream` is a hipStream_t passed as void*; nothing allocates, nothing synchronises; the only
 *     state the library keeps is the (thread-local) last-error string and the PROCESS-GLOBAL option
 *     table below ("Options") -> re-entrant across streams / processes, but two models in one
 *     process share one set of options;
 *   - return value 0 = launched; >0 = hipError_t; <0 = bad argument.  dfmir_last_error() gives text;
 *   - "accumulates" means the kernel atomically adds into a buffer the caller has zeroed/initialised.
 */
#ifndef DFMIR_HIP_H
#define DFMIR_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DFMIR_ABI_VERSION 14

int dfmir_abi_version(void);
const char* dfmir_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Options -- process-global A/B switches of the kernel selection (lab knobs; the defaults are the
 * product).  dfmir_set_option(name, value) sets `name` (must start with "DFMIR_") to `value`;
 * value == NULL marks it explicitly UNSET (also hiding an environment variable of that name).
 * A name never set through this call falls back to the environment variable of the same name.
 * The change applies to launches issued after the call returns, on every thread (no per-stream or
 * per-model scope).  Options that change a DATA LAYOUT the caller holds (DFMIR_CONV_SPLIT,
 * DFMIR_CONV_FP32, DFMIR_CONV3D_FP32: the split sections of packed weights) must be set before the
 * first dfmir_weight_pack* call, or every packed buffer re-packed after the change.
 * dfmir_get_option copies the current value into buf (NUL-terminated, truncated to buf_len) and
 * returns its length, or -1 when the option is unset.  Flag options (everything below without "=n") are ON when set to
 * anything but "" or "0" -- set_option(name, "0") switches a flag off like set_option(name, NULL) does.  Reads are
 * thread-safe: values are copied out under the table's lock, the per-site caches are single atomic words.
 *   kernel selection: DFMIR_CONV_FP32, DFMIR_CONV_SPLIT=bf16x3, DFMIR_CONV3D_FP32, DFMIR_CONV_GENERIC=1,
 *     DFMIR_CONV_NO_CS, DFMIR_CONV_CS_PLAIN, DFMIR_CS_XCD_PAIR=n, DFMIR_WGRAD_V1, DFMIR_WGRAD_NO_SWAP,
 *     DFMIR_NO_SMALL_WGRAD, DFMIR_NO_DIL2, DFMIR_NO_SMALL_TILES, DFMIR_GEMM_BIG_MIN=n,
 *     DFMIR_CONV3D_NO_PAIR, DFMIR_CONV3D_NO_M16, DFMIR_CONV3D_NO_TINY, DFMIR_CONV3D_NO_VEC,
 *     DFMIR_CONV3D_NO_MULTI, DFMIR_CONV3D_WGS=n, DFMIR_WSPLIT_WGS=n, DFMIR_CONV3D_NO_UPPHASE,
 *     DFMIR_CONV3D_WGRAD_COPIES, DFMIR_CONV3D_WGRAD_NO_PAIR, DFMIR_IN_BLUR_BANDED, DFMIR_CONV3D_NO_MARCH,
 *     DFMIR_MARCH_NSEG=n, DFMIR_CS_DEPHASE=n, DFMIR_CONV_W1 (lab builds with -DDFMIR_BUILD_W1 only),
 *     DFMIR_UPWGRAD_DIRECT, DFMIR_UPWGRAD_NO_FUSEB, DFMIR_UPWGRAD_8WAVE, DFMIR_UPWGRAD_NSEG=n (dfmir_conv3d_upwgrad),
 *     DFMIR_NCC_NO_WH_FUSE, DFMIR_CONV3D_NO_WGRAD_MARCH, DFMIR_WGRAD_MARCH_NSEG=n, DFMIR_NO_TINYVOL,
 *     DFMIR_NO_1X1_WGRAD, DFMIR_CONV3D_NO_FLOW_WGRAD, DFMIR_CONV3D_NO_S2, DFMIR_RESIZE_NO_ROWS, DFMIR_NCC_NO_D_FUSE, DFMIR_SMOOTH_NO_MARCH.
 * ---------------------------------------------------------------------------------------- */
int dfmir_set_option(const char* name, const char* value);
int dfmir_get_option(const char* name, char* buf, int buf_len);

/* ------------------------------------------------------------------------------------------
 * Convolutions (implicit GEMM on the matrix cores: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, and -- for the
 * 2-D 3x3 stride-1 layers with more than 32 output channels -- fp32 emulated on the 16-bit matrix pipe by
 * operand splitting, csrc/conv3x3s.hip: scaled fp16x2 (default) or bf16x3 (DFMIR_CONV_SPLIT=bf16x3);
 * DFMIR_CONV_FP32=1 keeps the fp32 instruction there too).
 * Replaces nn.Conv2d / nn.Conv3d (+ the ReflectionPad2d in front of it, + the LeakyReLU/Tanh
 * behind it) at: models/networks.py:982-983, 995, 1016-1023 (ResnetGenerator),
 * models/networks.py:1201,1214 (ResnetBlock), models/networks.py:587-595 (PatchSampleF MLP,
 * as 1x1 convs), models/voxelmorph/torchvoxelmorph/networks.py:1506-1521 (ConvBlock),
 * :1077-1081 (flow conv).
 *
 * Weights are consumed in "tap-major" packing  w_tcc[tap][Cin][Cout]  (tap = (kd*KH+kh)*KW+kw),
 * produced from the reference's [Cout][Cin][KD][KH][KW] by dfmir_weight_pack.
 * Input index along an axis:  c = o*stride - pad + t ;  dil>1 treats the input as zero-dilated
 * (transposed convolution = dgrad of a strided conv).  pad_mode: 0 zeros, 1 reflect.
 * act: 0 none, 1 leaky-relu(slope) (slope 0 = ReLU), 2 tanh.
 * ---------------------------------------------------------------------------------------- */
typedef struct DfConvGeom {
  int N, Cin, Cout;
  int Di, Hi, Wi;
  int Do, Ho, Wo;
  int KD, KH, KW;
  int stride, dil;
  int pd, ph, pw;
  int pad_mode;
  int act;
  float slope;
} DfConvGeom;

/* y[N,Cout,Do,Ho,Wo] = act(conv(x[N,Cin,Di,Hi,Wi], w) + bias).  bias may be NULL. */
int dfmir_conv_fwd(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias,
                   float* y, void* stream);
/* The same with a range probe of x supplied: x_amax[0..x_amax_n) are partial maxima whose maximum is max|x|
 * (one value from dfmir_absmax, or the per-plane values dfmir_instnorm_fwd/bwd leave behind; written on the same
 * stream).  The probe lets the 2-D 3x3 stride-1 kernels take the scaled fp16x2 split form (csrc/conv3x3s.hip:
 * 3 matrix products per fp32 product instead of bf16x3's 6); without it -- dfmir_conv_fwd -- those layers run
 * bf16x3 (DFMIR_CONV_SPLIT=bf16x3) or fp32 MFMA.  x_amax may be NULL. */
int dfmir_conv_fwd_scaled(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                          const float* w_tcc, const float* bias, float* y, void* stream);
/* out[0] = max(out[0], max_i |x[i]|) (NaN counts as +inf; the caller zero-initialises out).  Replaces nothing in
 * the reference: it is the range probe of the fp16x2 split.  dfmir_instnorm_fwd/bwd produce a probe of their output
 * as a by-product: y_amax / dx_amax = DFMIR_PROBE_SLOTS floats (caller zero-initialises; NULL to skip) whose maximum
 * is max|output|.  x_amax_n of the conv entry points is the number of floats to reduce (1 after dfmir_absmax,
 * DFMIR_PROBE_SLOTS after the InstanceNorm entry points). */
#define DFMIR_PROBE_SLOTS 64
int dfmir_absmax(const float* x, long long n, float* out, void* stream);
/* 3-D 3x3x3 stride-1 "same" convolution (Cin, Cout >= 8) on the 16-bit matrix pipe in the scaled fp16x2 split form
 * (csrc/conv3ds.hip) -- the VoxelMorph U-Net's stride-1 ConvBlocks and their input gradients
 * (models/voxelmorph/torchvoxelmorph/networks.py:73-86,1506-1521).  x_amax as for dfmir_conv_fwd_scaled (required);
 * ws: dfmir_conv3d_split_ws_floats(Cin, Cout) floats of 16-byte aligned scratch (the call splits the weights into it);
 * y_amax (may be NULL): DFMIR_PROBE_SLOTS zero-initialised floats that receive the range probe of y.
 * dfmir_conv3d_split_ok: 1 when the geometry is taken (0 under DFMIR_CONV3D_FP32=1 / DFMIR_CONV_FP32=1). */
int dfmir_conv3d_split_ok(const DfConvGeom* g);
long long dfmir_conv3d_split_ws_floats(int Cin, int Cout);
int dfmir_conv3d_split_fwd(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n, const float* w_tcc,
                           float* ws, const float* bias, float* y, float* y_amax, void* stream);
/* The same, computing output channels [0, cout_used) only (y keeps Cout planes; the others are left untouched): the
 * input gradient of a layer fed by cat([up2(a), b]) whose skip part b needs no gradient (the two image channels at the
 * top of the U-Net, networks.py:97-100). */
/* (w_tcc may be NULL in the _sub / _actgrad forms: `ws` then already holds the split of these weights for this
 * cout_used, left by an earlier call -- weights only change at the optimizer step.) */
int dfmir_conv3d_split_fwd_sub(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n, const float* w_tcc,
                               float* ws, const float* bias, float* y, float* y_amax, int cout_used, void* stream);
/* The weight gradient of the same layers in the same split form (8 <= Cin <= 48, 8 <= Cout <= 32, or Cout < 8 with Cin <= 32; W % 4 == 0), voxels as the
 * matrix K: dw_tcc[tap][Cin][Cout] += ...   (accumulates, like dfmir_conv_wgrad). */
/* As dfmir_conv3d_split_fwd_sub for a dgrad (g->act == 0) whose result is the gradient w.r.t. the OUTPUT of a
 * LeakyReLU: act_src = that output (shape of y); the epilogue multiplies by the activation's derivative
 * (act_src > 0 ? 1 : act_slope), so y receives the gradient w.r.t. the pre-activation and y_amax its range probe.
 * Replaces the `F.leaky_relu` backward pass between two ConvBlocks
 * (models/voxelmorph/torchvoxelmorph/networks.py:1506-1521, autograd of nn.LeakyReLU(0.2)). */
int dfmir_conv3d_split_fwd_actgrad(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                   const float* w_tcc, float* ws, const float* bias, float* y, float* y_amax,
                                   int cout_used, const float* act_src, float act_slope, void* stream);
/* conv3x3x3(cat(nearest_up2(a), b)) WITHOUT the up-sampled / concatenated tensor -- nn.Upsample(scale_factor=2,
 * mode='nearest') + torch.cat feeding the next ConvBlock (models/voxelmorph/torchvoxelmorph/networks.py:64,97-100).
 * Per output parity class the 27 taps over nearest_up2(a) fall on 2x2x2 voxels of `a`: dfmir_conv3d_up_fwd computes the
 * up-sampled channels' share as eight 8-tap convolutions of `a` with summed weights (8/27 of the products) and leaves
 * the PARTIAL sum in y [N, Cout, 2D, 2H, 2W]; dfmir_conv3d_split_fwd_add then runs the skip channels b (g: the conv
 * over b alone, Cin = Cb; its weights are rows koff .. koff + Cb - 1 of the layer's [27][Ktot][Cout] forward packing)
 * and its epilogue adds the partial sum, the bias, the activation of g and writes y and its range probe.
 * a [N, Ca, D, H, W] with Ca % 8 == 0, W % 4 == 0 (dfmir_conv3d_up_ok); ws: dfmir_conv3d_up_ws_floats /
 * dfmir_conv3d_split_ws_floats(Cb, Cout) floats; w_tcc NULL = ws already holds the split of the current weights. */
int dfmir_conv3d_up_ok(int N, int Ca, int Cout, int D, int H, int W);
/* Every weight split of a network's 3-D convs in ONE launch (all weights change together at the optimizer step;
 * torch.optim.Adam.step, registration_model.py:168-171).  jobs_dev: device array of njobs records of 7 x int64 --
 * {w_tcc, ws, trailer (= ws + its ws_floats - 4), kind | K << 32, M | pair << 32, Ktot | koff << 32, Cb}; kind 0 = the
 * split dfmir_conv3d_split_fwd* makes (K = Cin, M = Cout or cout_used in the plane-pair form), 1 = dfmir_conv3d_up_fwd /
 * _up_skip2_fwd's (K = Ca, Cb = skip channels or 0), 2 = dfmir_conv3d_up_dgrad's.  The entry points then take w_tcc = NULL. */
int dfmir_conv3d_wsplit_batch(const void* jobs_dev, int njobs, void* stream);
int dfmir_conv3d_split_is_pair(int cout_used);
/* The full-resolution 3x3x3 stride-1 layers with Cin x Cout <= 512 -- 32 -> 16, 16 -> 16 and the input gradients 16 <- 16,
 * 32 <- 16 of the `extras` chain (models/voxelmorph/torchvoxelmorph/networks.py:73-86,1506-1521) -- as a z-MARCHING kernel
 * (csrc/conv3dm.hip): a workgroup owns a 16 x 32 column of the volume and walks a segment of planes; every input plane is
 * staged once (in-plane halo only), the accumulators of the three open output planes roll through the registers, the
 * layer's weights are split by the workgroup itself and stay in LDS (no workspace, no weight-split launch).  Same
 * arithmetic as dfmir_conv3d_split_fwd (scaled fp16x2 products, fp32 accumulation); x_amax as there; y_amax
 * (DFMIR_PROBE_SLOTS floats, zeroed by the caller, or NULL) receives the range probe of y.  act_src != NULL (g->act == 0):
 * the result is multiplied by the LeakyReLU derivative act_src > 0 ? 1 : act_slope, as dfmir_conv3d_split_fwd_actgrad.
 * Cin = 16, Cout = 3, g->act == 0, act_src == NULL: the flow head (torchvoxelmorph/networks.py:1076-1080) in the FLOW form --
 * MFMA columns = (open output plane, channel), one accumulator set for the three planes of the march.
 * Needs W % 4 == 0 and 16-byte aligned x, y, act_src, w_tcc; dfmir_conv3d_march_ok says whether the geometry is taken
 * (0 under DFMIR_CONV3D_NO_MARCH / DFMIR_CONV3D_FP32 / DFMIR_CONV_FP32). */
int dfmir_conv3d_march_ok(const DfConvGeom* g);
int dfmir_conv3d_march_fwd(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n, const float* w_tcc,
                           const float* bias, float* y, float* y_amax, const float* act_src, float act_slope,
                           void* stream);
/* The flow head (models/voxelmorph/torchvoxelmorph/networks.py:1076-1080: Conv3d(16, 3, 3, padding=1)) and its data
 * gradient (3 -> 16) as plain fp32 FMAs -- neither side fills a matrix-core tile (3 of 16 rows / 3 of 8 K channels).
 * w_tcc: the fp32 tap-major packing [27][Cin][Cout] of dfmir_weight_pack (mode 0 forward, mode 1 data gradient);
 * y = act(conv + bias), times the derivative of the LeakyReLU whose OUTPUT is act_src when act_src != NULL (g->act == 0);
 * y_amax: NULL or the 64 accumulating range-probe slots of y. */
int dfmir_conv3d_tiny_ok(const DfConvGeom* g);
int dfmir_conv3d_tiny_fwd(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias, float* y,
                          float* y_amax, const float* act_src, float act_slope, void* stream);
/* The first encoder level of the 3-D U-Net (torchvoxelmorph/networks.py:66-86: Conv3d(2, 16, 3, stride=2, padding=1) +
 * LeakyReLU over cat(source, target)): forward as fp32 FMAs from an LDS-staged patch, weight gradient on fp32 MFMA with the
 * operand gathered from the same patch.  w_tcc / dw_tcc: [27][2][16] tap-major (dfmir_weight_pack mode 0); dw accumulates. */
int dfmir_conv3d_s2c2_ok(const DfConvGeom* g);
int dfmir_conv3d_s2c2_fwd(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias, float* y, float* y_amax,
                          void* stream);
int dfmir_conv3d_s2c2_wgrad(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc, void* stream);
/* The deeper stride-2 encoder levels (torchvoxelmorph/networks.py:66-71,1506-1521: ConvBlock(ndims, prev_nf, nf, stride=2),
 * Cin a multiple of 4, Cout a multiple of 16 up to 64) on fp32 MFMA from LDS-staged patches (csrc/conv3ds2.hip).
 * _fwd: y = act(conv(x) + bias), y_amax NULL or the 64 accumulating range-probe slots of y; _wgrad: dw_tcc [27][Cin][Cout]
 * and db [Cout] (may be NULL) accumulate (honours dfmir_det_begin); _dgrad: g = the geometry of the data-gradient call as a convolution of dy (Cin =
 * channels of dy, Cout = channels of dx, stride 1, dil 2, pad 1), w_tcc = the dgrad packing (dfmir_weight_pack mode 1),
 * evaluated in the 8 parity classes of the dx voxels (nothing multiplies an inserted zero). */
int dfmir_conv3d_s2_ok(const DfConvGeom* g);
int dfmir_conv3d_s2_fwd(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias, float* y, float* y_amax,
                        void* stream);
int dfmir_conv3d_s2_wgrad(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc, float* db, void* stream);
int dfmir_conv3d_s2_dgrad_ok(const DfConvGeom* g);
int dfmir_conv3d_s2_dgrad(const DfConvGeom* g, const float* dy, const float* w_tcc, float* dx, void* stream);
long long dfmir_conv3d_up_ws_floats(int Ca, int Cout);
int dfmir_conv3d_up_fwd(const float* a, const float* a_amax, int a_amax_n, const float* w_tcc, int Ktot, float* ws,
                        float* y, int N, int Ca, int Cout, int D, int H, int W, void* stream);
/* Cb <= 2 (the two input images at the top of the U-Net, networks.py:1105 `x = torch.cat([source, target], dim=1)`): the
 * whole layer in ONE launch -- after the up-sampled channels the same workgroup runs the skip channels from its own
 * full-resolution patch (one MFMA k-step = 2 tap rows x 4 x-taps x 2 channels), then bias, LeakyReLU (act 1) and the
 * range probe.  Ktot of w_tcc == Ca + Cb; ws as dfmir_conv3d_up_ws_floats. */
int dfmir_conv3d_up_skip2_fwd(const float* a, const float* a_amax, int a_amax_n, const float* b, const float* b_amax,
                              int b_amax_n, int Cb, const float* w_tcc, float* ws, const float* bias, float* y,
                              float* y_amax, int N, int Ca, int Cout, int D, int H, int W, int act, float slope,
                              void* stream);
/* d(a) of that layer, directly at low resolution (autograd of nn.Upsample(nearest) + Conv3d: a 2x2x2 sum pool of the
 * conv's input gradient = a 4x4x4 stride-2 convolution of dy, run in parity classes: 64 instead of 216 taps per
 * low-resolution voxel, no up-sampled gradient tensor, no pooling pass).  dy [N, Cout, 2D, 2H, 2W] = gradient w.r.t. the
 * conv's result (after the activation's backward), Cout % 8 == 0; w_tcc = the FORWARD packing [27][Ktot][Cout] or NULL (ws
 * current); act_src (optional) = a when a is the output of a LeakyReLU(act_slope) feeding only this layer: da then is the
 * gradient w.r.t. that activation's input; da_amax (optional) = range probe of da. */
long long dfmir_conv3d_up_dgrad_ws_floats(int Ca, int Cout);
int dfmir_conv3d_up_dgrad(const float* dy, const float* dy_amax, int dy_amax_n, const float* w_tcc, int Ktot, float* ws,
                          float* da, float* da_amax, const float* act_src, float act_slope, int N, int Ca, int Cout, int D,
                          int H, int W, void* stream);
int dfmir_conv3d_split_fwd_add(const DfConvGeom* g, const float* b, const float* b_amax, int b_amax_n, const float* w_tcc,
                               int Ktot, int koff, float* ws, const float* bias, float* y, float* y_amax, void* stream);
int dfmir_conv3d_split_wgrad_ok(const DfConvGeom* g);
int dfmir_conv3d_split_wgrad(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n, const float* dy,
                             const float* dy_amax, int dy_amax_n, float* dw_tcc, void* stream);
/* The weight (and bias) gradient of conv3x3x3 over X = cat(nearest_up2(a), b) without building X (autograd of
 * nn.Upsample + torch.cat + Conv3d, networks.py:64,97-100): a [N, Ca, D/2, H/2, W/2] is read at (z>>1, y>>1, x>>1) while
 * the operand patch is staged.  g = the whole layer (Cin = Ca + Cb); Ca % 8 == 0; D, H even, W % 8 == 0; x_amax = a range
 * probe valid for both parts; db may be NULL. */
int dfmir_conv3d_split_wgrad_upcat(const DfConvGeom* g, const float* a, const float* b, int Ca, const float* x_amax,
                                   int x_amax_n, const float* dy, const float* dy_amax, int dy_amax_n, float* dw_tcc,
                                   float* db, void* stream);
/* The same gradient, the up-sampled channels in PARITY CLASSES (csrc/conv3duw.hip): a 3x3x3 tap of output voxel 2V + p over
 * nearest_up2(a) reads a[V + floor((p + d - 1) / 2)], so G[p][i] = sum_V a[V + p - 1 + i] (x) dy[2V + p] (64 matrices per
 * layer, K = the LOW-resolution voxels) holds the whole gradient: 8 / 27 of the direct form's products and one pass over dy
 * with no halo.  The skip channels b run on the direct kernel (which also sums db).  Ca == 32, Cout % 8 == 0, Cout <= 32,
 * D, H even, W % 8 == 0; ws = dfmir_conv3d_upwgrad_ws_floats() floats, owned by the call until the stream has passed it.
 * Two skip channels (the network's input images at the top level) and db are FUSED into the same launch (rows = (tap,
 * channel) from x- and y-paired images of b; db as the row of a constant operand); wider skips and db go through the direct
 * kernel on rows Ca.. of dw_tcc.  A/B switches: DFMIR_UPWGRAD_DIRECT=1 (dfmir_conv3d_upwgrad_ok then answers 0: the direct
 * kernel over both parts), DFMIR_UPWGRAD_NO_FUSEB=1 (skip share always direct), DFMIR_UPWGRAD_8WAVE=1 (two waves per SIMD, not
 * fused), DFMIR_UPWGRAD_NSEG=n forces the z split. */
int dfmir_conv3d_upwgrad_ok(const DfConvGeom* g, int Ca);
long long dfmir_conv3d_upwgrad_ws_floats(void);
int dfmir_conv3d_upwgrad(const DfConvGeom* g, const float* a, const float* b, int Ca, const float* x_amax, int x_amax_n,
                         const float* dy, const float* dy_amax, int dy_amax_n, float* dw_tcc, float* db, float* ws,
                         void* stream);
/* The same, also accumulating the bias gradient db[Cout] += sum dy from the units it stages anyway (db may be NULL).
 * Layers with fewer than 8 output channels (the 16 -> 3 flow conv, networks.py:1077) are taken with the operand roles
 * swapped: rows = (tap, co) from shifted dy, columns = ci. */
int dfmir_conv3d_split_wgrad_db(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n, const float* dy,
                                const float* dy_amax, int dy_amax_n, float* dw_tcc, float* db, void* stream);
/* 1 when dfmir_conv3d_split_wgrad / _db run this layer on conv3d_wgrad_march_k (csrc/conv3dwm.hip): Cin 32, Cout 16 (the
 * full-resolution ConvBlock of `extras`, networks.py:73-86) -- all 27 tap matrices resident in one wave's accumulators, the
 * workgroup marching along z, both operands read once.  A/B: DFMIR_CONV3D_NO_WGRAD_MARCH=1; DFMIR_WGRAD_MARCH_NSEG=n. */
int dfmir_conv3d_wgrad_is_march(const DfConvGeom* g);
/* The launcher's own predicate: the geometry above AND x, dy 16-byte aligned (they are read as quads; unaligned operands
 * run on the tiled kernel).  _is_march(g) answers for aligned operands. */
int dfmir_conv3d_wgrad_is_march_at(const DfConvGeom* g, const float* x, const float* dy);
/* out[0..DFMIR_PROBE_SLOTS) = max(a, b): the probe of cat([nearest_up2(a), b]) from its inputs' probes. */
int dfmir_probe_merge(const float* a, const float* b, float* out, void* stream);
/* dw_tcc[tap][Cin][Cout] += sum_{n,o} x(gathered) * dy      (accumulates; same packing as w_tcc). */
int dfmir_conv_wgrad(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc,
                     void* stream);
/* db (may be NULL): the bias gradient db[Cout] += sum_{n,o} dy of the same layer, accumulated by this call -- inside
 * the wgrad kernel where it reads dY anyway (split 3x3 kernels), otherwise by a dfmir_bias_grad pass. */
int dfmir_conv_wgrad_scaled(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                            const float* dy, const float* dy_amax, int dy_amax_n, float* dw_tcc, float* db,
                            void* stream);
/* The same with the maxima of |dy| PER PLANE, dy_pmax[N][Cout] (as dfmir_instnorm_bwd_pmax leaves them; NULL = as
 * above): the fp16x2 weight-gradient kernel (Cout > 64) then scales dY per output channel -- the scale is uniform along
 * its reduction axis -- so a channel far below the tensor maximum keeps its 22 bits. */
int dfmir_conv_wgrad_scaled_ch(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n, const float* dy,
                               const float* dy_amax, int dy_amax_n, const float* dy_pmax, float* dw_tcc, float* db,
                               void* stream);
/* db[C] += sum_{n,s} dy[n,C,s]   (accumulates). */
int dfmir_bias_grad(const float* dy, float* db, int N, int C, long long S, void* stream);
/* mode 0: w_tcc[t][ci][co] = w[co][ci][t]           (forward packing)
 * mode 1: w_tcc[t][co][ci] = w[co][ci][T-1-t]       (dgrad packing: roles of Cin/Cout swapped, taps flipped)
 * The packed buffer holds dfmir_weight_pack_floats(Cout, Cin, T) floats: the tap-major fp32 weights, then
 * (T == 9 only) the same weights pre-split into bf16 triples for the 3x3 kernels.  Callers treat it as
 * opaque and hand it to dfmir_conv_fwd unchanged. */
long long dfmir_weight_pack_floats(int Cout, int Cin, int T);
int dfmir_weight_pack(const float* w, float* w_tcc, int Cout, int Cin, int T, int mode, void* stream);
/* g[co][ci][t] = g_tcc[t][ci][co]  (gradient back to the reference's parameter layout). */
int dfmir_weight_unpack(const float* g_tcc, float* g, int Cout, int Cin, int T, void* stream);
/* The deferred form of the same step for every conv of a backward pass at once (one launch per train step instead
 * of unpack + add + clear per layer): job j adds its tap-major accumulator src[T][Cin][Cout] into the gradient in the
 * reference layout, dst[Cout][Cin][T] += src (torch accumulates p.grad the same way, autograd/functions/
 * accumulate_grad.h), and zeroes src.  `jobs` is DEVICE memory; max_total = max_j Cout*Cin*T. */
/* dfmir_conv_fwd_scaled with an epilogue y = act(conv(x) + bias) + res + fold(ring), for the geometries
 * dfmir_conv3x3_res_ok(g) accepts (the shared-tile split 3x3 kernel).  res[N][Cout][Ho][Wo] (optional): a second
 * gradient of the same tensor -- the skip branch of a ResnetBlock, dx = dgrad(dy) + d out (models/networks.py:1219-1221).
 * ring (optional): see dfmir_conv3x3_reflect_ring. */
int dfmir_conv3x3_res_ok(const DfConvGeom* g);
int dfmir_conv3x3_fwd_scaled_res(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                 const float* w_tcc, const float* bias, const float* res, const float* ring,
                                 int ring_len, float* y, void* stream);

/* Input gradient of conv3x3(ReflectionPad2d(1)(x)) (ResnetBlock, models/networks.py:1190-1214) in two parts.  The
 * gradient is the fold of the full correlation of dy on the (H+2) x (W+2) padded frame; the frame's interior is the
 * zero-padded "same" dgrad (dfmir_conv3x3_fwd_scaled_res with the dgrad packing, pad 1), and its one-pixel ring is
 * four 3-tap 1-D convolutions of the border lines of dy, computed here into ring[N][4][Cin][ring_len] (strips: top,
 * bottom, left, right; index = frame coordinate along the strip), which the interior call folds in through its
 * `ring` argument.  g = the FORWARD geometry; wd_packed = dfmir_weight_pack(mode 1); dy_amax as for
 * dfmir_conv_fwd_scaled.  dfmir_conv3x3_reflect_ring_ok(g) != 0 iff this library build takes the geometry;
 * dfmir_conv3x3_reflect_ring_len(g) = ring_len. */
int dfmir_conv3x3_reflect_ring_ok(const DfConvGeom* g);
int dfmir_conv3x3_reflect_ring_len(const DfConvGeom* g);
int dfmir_conv3x3_reflect_ring(const DfConvGeom* g, const float* dy, const float* dy_cols, const float* dy_amax,
                               int dy_amax_n, const float* wd_packed, float* ring, void* stream);
/* dy_cols (optional): [N*Cout][2][H], the first and last column of dy as left by dfmir_instnorm_bwd_cols. */

/* Every packing of a train step at once (the weights of all layers change together, at the optimizer step): the same
 * result as njobs dfmir_weight_pack calls, in two launches.  `jobs` is HOST memory; `table_dev` is njobs * 64 bytes of
 * device scratch owned by the caller, (re)written when upload != 0 -- pass 0 while the same jobs come back. */
typedef struct DfPackJob {
  const float* w;      /* [Cout][Cin][T] */
  float* packed;       /* dfmir_weight_pack_floats(Cout, Cin, T) floats */
  int Cout, Cin, T, mode;
} DfPackJob;
int dfmir_weight_pack_batch(const DfPackJob* jobs, int njobs, void* table_dev, int upload, void* stream);

typedef struct DfUnpackJob {
  float* src;
  float* dst;
  int Cout, Cin, T, reserved;
} DfUnpackJob;
int dfmir_weight_unpack_add_batch(const DfUnpackJob* jobs, int njobs, long long max_total, void* stream);

/* The Cin == 1 stem directly: nn.ReflectionPad2d(3) + nn.Conv2d(1, ngf, 7) (models/networks.py:982-983).
 * x[N][1][H][W], w[Cout][1][7][7] (reference layout), Cout <= 64; pad 3, pad_mode 0 zero / 1 reflect.
 * fwd: y = conv + bias.  wgrad: dw[Cout][49] += sum dy (x) x_padded-shifted, db[Cout] += sum dy (db optional); both
 * ACCUMULATE (zero the buffers first). */
int dfmir_conv7x7_c1_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout,
                         int pad_mode, void* stream);
int dfmir_conv7x7_c1_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cout,
                           int pad_mode, void* stream);

/* 7x7 convs of the generator as 1x1 GEMMs (models/networks.py:982-983 Conv2d(1,64,7) and :1022-1024
 * Conv2d(64,1,7)+Tanh, both behind ReflectionPad2d(3)):
 *   tapstack: S[n][c*K*K+tap][p] = x[n][c][map(p + tap - pad)]            (then a 1x1 conv K*K -> Cout)
 *   tapsum  : y[n][c][p] = act(bias[c] + sum_tap Z[n][c*K*K+tap][map(p + tap - opad)])   (after a 1x1 conv Cin -> K*K)
 * map = reflect (pad_mode 1) or zero (pad_mode 0); *_bwd are the exact adjoints (gather form). */
int dfmir_tapstack_fwd(const float* x, float* s, int N, int C, int H, int W, int K, int pad,
                       int pad_mode, void* stream);
int dfmir_tapstack_bwd(const float* ds, float* dx, int N, int C, int H, int W, int K, int pad,
                       int pad_mode, void* stream);
int dfmir_tapsum_fwd(const float* z, const float* bias, float* y, int N, int C, int Hz, int Wz, int Ho,
                     int Wo, int K, int opad, int pad_mode, int act, float slope, void* stream);
int dfmir_tapsum_bwd(const float* dy, float* dz, int N, int C, int Hz, int Wz, int Ho, int Wo, int K,
                     int opad, int pad_mode, void* stream);

/* ------------------------------------------------------------------------------------------
 * InstanceNorm2d(affine=False, eps) [+ ReLU] [+ residual add]  -- models/networks.py:113-131,
 * 984-985, 1190-1221 (x + conv_block(x)).  One (n,c) plane of S elements per workgroup.
 * y = res + relu?((x-mean)*rstd) ; mean/rstd [planes] are saved for backward.
 * ---------------------------------------------------------------------------------------- */
/* Statistics of InstanceNorm only (round 6): mean / rstd per plane and the range probe of relu?(IN(x)); y is NOT written --
 * a consumer that normalises while it stages its operand follows.  S in {4096, 16384, 65536} (dfmir_instnorm_stats_ok). */
int dfmir_instnorm_stats_ok(long long S);
int dfmir_instnorm_stats(const float* x, float* mean, float* rstd, int planes, long long S, float eps, int relu,
                         float* y_amax, void* stream);
int dfmir_instnorm_fwd(const float* x, const float* res, float* y, float* mean, float* rstd,
                       int planes, long long S, float eps, int relu, float* y_amax, void* stream);
int dfmir_instnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                       float* dx, int planes, long long S, int relu, float* dx_amax, void* stream);
/* The same, also writing dx's first and last column compactly, dx_cols[planes][2][H] (S = H*W; supported iff
 * dfmir_instnorm_bwd_cols_ok(S, W)): the reflect ring of the next conv's dgrad reads them from there. */
int dfmir_instnorm_bwd_cols_ok(long long S, int W);
int dfmir_instnorm_bwd_cols(const float* dy, const float* x, const float* mean, const float* rstd, float* dx, int planes,
                            long long S, int relu, float* dx_amax, float* dx_cols, int W, void* stream);
/* Superset of the two (S in {4096, 16384, 65536}; dx_cols may be NULL): also dx_pmax[planes] <- max |dx| of each plane. */
int dfmir_instnorm_bwd_pmax_ok(long long S);
int dfmir_instnorm_bwd_pmax(const float* dy, const float* x, const float* mean, const float* rstd, float* dx, int planes,
                            long long S, int relu, float* dx_amax, float* dx_cols, int W, float* dx_pmax, void* stream);

/* InstanceNorm2d + ReLU + Downsample (blur-pool) as one pass per plane: models/networks.py:984-996, modules
 * (5,6,7) and (9,10,11) of the generator -- the full-resolution normalised tensor feeds only the blur and no backward
 * needs it.  H x W = 256 x 256 or 128 x 128 (dfmir_in_relu_blurdown_ok).  z [planes, H/2, W/2]; mean / rstd [planes]
 * saved for backward; z_amax (may be NULL): DFMIR_PROBE_SLOTS zero-initialised floats, a bound of max|z|.
 * bwd: dx from dz, x, mean, rstd; dx_amax / dx_pmax as dfmir_instnorm_bwd_pmax (may be NULL). */
int dfmir_in_relu_blurdown_ok(int H, int W);
int dfmir_in_relu_blurdown_fwd(const float* x, float* z, float* mean, float* rstd, int planes, int H, int W, float eps,
                               float* z_amax, void* stream);
int dfmir_in_relu_blurdown_bwd(const float* dz, const float* x, const float* mean, const float* rstd, float* dx,
                               int planes, int H, int W, float* dx_amax, float* dx_pmax, void* stream);

/* elementwise activation backward from the saved OUTPUT y: act 1 leaky(slope), 2 tanh. */
int dfmir_act_bwd(const float* dy, const float* y, float* dx, long long n, int act, float slope,
                  void* stream);
/* The same, also leaving the range probe of dx in dx_amax[DFMIR_PROBE_SLOTS] (zero-initialised by the caller; pointers
 * 16-byte aligned): dx feeds the split dgrad / wgrad of the layer in front of the activation. */
int dfmir_act_bwd_amax(const float* dy, const float* y, float* dx, long long n, int act, float slope,
                       float* dx_amax, void* stream);

/* ------------------------------------------------------------------------------------------
 * Anti-aliased resampling of the generator -- models/networks.py:37-60 (Downsample: reflect pad 1,
 * depthwise [1 2 1]x[1 2 1]/16, stride 2) and :73-93 (Upsample: replicate pad 1, depthwise
 * conv_transpose 4x4 [1 3 3 1]^2/16 stride 2, cropped to 2H x 2W).
 * ---------------------------------------------------------------------------------------- */
int dfmir_blur_down_fwd(const float* x, float* y, int planes, int H, int W, void* stream);
int dfmir_blur_down_bwd(const float* dy, float* dx, int planes, int H, int W, void* stream);
int dfmir_blur_up_fwd(const float* x, float* y, int planes, int H, int W, void* stream);
int dfmir_blur_up_bwd(const float* dy, float* dx, int planes, int H, int W, void* stream);
/* nn.ReflectionPad2d(p) -- models/networks.py:982,1022 (materialised only for NCE layer 0). */
int dfmir_reflect_pad2d_fwd(const float* x, float* y, int planes, int H, int W, int p, void* stream);
int dfmir_reflect_pad2d_bwd(const float* dy, float* dx, int planes, int H, int W, int p, void* stream);
/* dx = fold(dy) + add: the adjoint of ReflectionPad2d summed with a second gradient of the same tensor in one pass --
 * ResnetBlock's `out = x + self.conv_block(x)` (models/networks.py:1219-1221): x feeds the padded conv AND the skip,
 * so its gradient is fold(d conv path) + d out.  add[planes][H][W]. */
int dfmir_reflect_pad2d_bwd_add(const float* dy, const float* add, float* dx, int planes, int H, int W, int p,
                                void* stream);

/* nn.Upsample(scale 2, nearest) + torch.cat([up(a), b], 1) -- torchvoxelmorph/networks.py:64,97-100.
 * a[N,Ca,Da,Ha,Wa] -> factor sd (1 for 2-D, 2 for 3-D) in D and 2 in H,W; b[N,Cb,Da*sd,2Ha,2Wa]. */
int dfmir_upcat_fwd(const float* a, const float* b, float* y, int N, int Ca, int Cb, int Da, int Ha,
                    int Wa, int sd, void* stream);
int dfmir_upcat_bwd(const float* dy, float* da, float* db, int N, int Ca, int Cb, int Da, int Ha,
                    int Wa, int sd, void* stream);

/* torch.cat([a, b], dim=1) -- torchvoxelmorph/networks.py:1110 (source|target into the U-Net).
 * SA = Ca*S, SB = Cb*S elements per sample.  bwd: da/db may be NULL. */
int dfmir_cat_channels_fwd(const float* a, const float* b, float* y, long long N, long long SA,
                           long long SB, void* stream);
int dfmir_cat_channels_bwd(const float* dy, float* da, float* db, long long N, long long SA,
                           long long SB, void* stream);
/* y = mult * x -- `vec * self.scale` (layers.py:65), `-pos_flow` (networks.py:1125). */
int dfmir_scale(const float* x, float* y, long long n, float mult, void* stream);

/* ------------------------------------------------------------------------------------------
 * SpatialTransformer.forward  (torchvoxelmorph/layers.py:30-48): out = grid_sample(src, grid+flow)
 * with align_corners=True, padding_mode='zeros'; flow is a displacement in voxels, channel d =
 * axis d.  mode 0 bilinear/trilinear, 1 nearest.  add_identity: out += src (VecInt step
 * v + warp(v,v), layers.py:64-68; needs C == ndims).
 * Backward: dsrc accumulates (atomic scatter); dflow is written, or (flow_into_src) accumulated
 * into dsrc for the VecInt self-warp where src == flow.
 * ---------------------------------------------------------------------------------------- */
int dfmir_warp2d_fwd(const float* src, const float* flow, float* out, int B, int C, int H, int W,
                     int mode, int add_identity, void* stream);
int dfmir_warp2d_bwd(const float* dout, const float* src, const float* flow, float* dsrc,
                     float* dflow, int B, int C, int H, int W, int add_identity, int flow_into_src,
                     void* stream);
int dfmir_warp3d_fwd(const float* src, const float* flow, float* out, int B, int C, int D, int H,
                     int W, int mode, int add_identity, void* stream);
int dfmir_warp3d_bwd(const float* dout, const float* src, const float* flow, float* dsrc,
                     float* dflow, int B, int C, int D, int H, int W, int add_identity,
                     int flow_into_src, void* stream);
/* The same adjoint WITHOUT device-scope atomics on d(src), bit-reproducible (csrc/warp_win.hip, "owner gathers"): every
 * workgroup accumulates its tile's scatter in a fixed-point LDS window, writes it densely to a scratch slot, and a
 * second pass lets each d(src) cell sum the neighbouring tiles' windows that cover it, in fixed order; voxels displaced
 * beyond the 3x3(x3)-tile neighbourhood go through a list and the scalar routine (atomics).  nd = 2 (D ignored) or 3;
 * W % 4 == 0.  dsrc is written entirely (no zero-fill needed); ws = dfmir_warp_bwd_own_ws_floats(...) floats of
 * 16-byte aligned scratch (0: shape not eligible -> use dfmir_warp{2,3}d_bwd). */
long long dfmir_warp_bwd_own_ws_floats(int nd, int B, int C, int D, int H, int W);
int dfmir_warp_bwd_own(int nd, const float* dout, const float* src, const float* flow, float* dsrc, float* dflow, int B,
                       int C, int D, int H, int W, int add_identity, int flow_into_src, float* ws, void* stream);

/* ResizeTransform (layers.py:71-97): F.interpolate(align_corners=True, bi/tri-linear) fused with the
 * scalar rescale `mult`.  D == 1 for 2-D.  bwd is the adjoint in gather form (writes dx; no atomics). */
int dfmir_resize_fwd(const float* x, float* y, int planes, int Di, int Hi, int Wi, int Do, int Ho,
                     int Wo, float mult, void* stream);
int dfmir_resize_bwd(const float* dy, float* dx, int planes, int Di, int Hi, int Wi, int Do, int Ho,
                     int Wo, float mult, void* stream);
/* The same adjoint as three 1-D passes (W, H, D; the trilinear weights are separable): ws = dfmir_resize_bwd_ws_floats
 * floats (16-B aligned) for the two intermediates.  5x faster than the one-pass gather on the x2 flow up-sampling. */
long long dfmir_resize_bwd_ws_floats(int planes, int Di, int Hi, int Wi, int Do, int Ho, int Wo);
int dfmir_resize_bwd_sep(const float* dy, float* dx, int planes, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                         float mult, float* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * PatchNCE path -- models/networks.py:602-619 (PatchSampleF.forward), :493-502 (Normalize),
 * models/patchnce.py:14-55 (PatchNCELoss.forward).
 * Sampled features are kept channel-major [C][rows] (rows = B*P, row r = b*P + p) so the 2-layer
 * MLP runs as 1x1 convs over `rows` pixels; the reference's [B*P, C] tensors are the transposed
 * VIEW of this buffer.
 * ---------------------------------------------------------------------------------------- */
/* out[c][b*P+p] = feat[b][c][ids[p]]   (feat [B,C,S], ids int64 [P] shared by the batch). */
int dfmir_patch_gather_fwd(const float* feat, const long long* ids, float* out, int B, int C,
                           long long S, int P, void* stream);
int dfmir_patch_gather_bwd(const float* dout, const long long* ids, float* dfeat, int B, int C,
                           long long S, int P, void* stream);
/* The same scatter into a gradient that already holds another consumer's contribution (ids must be DISTINCT -- a
 * P-subset, as torch.randperm yields them, networks.py:609-610: plain load + store instead of atomics; this also
 * holds for the grouped forms below), keeping its per-plane range
 * probe valid: dfeat_amax[DFMIR_PROBE_SLOTS] (the dx_amax of dfmir_instnorm_bwd) is raised to |new value| where needed. */
int dfmir_patch_gather_bwd_amax(const float* dout, const long long* ids, float* dfeat, int B, int C, long long S,
                                int P, float* dfeat_amax, void* stream);
/* Grouped forms: ids[G][P]; image b uses the ids of group b / (B / G) -- the query images of G NCE terms stacked along
 * the batch, each sampled at its own term's positions (registration_model.py:237-253 called once per term in the
 * reference).  bwd_g scatters (dfeat must hold zeros or another gradient); dfeat_amax optional as above. */
int dfmir_patch_gather_fwd_g(const float* feat, const long long* ids, float* out, int B, int C, long long S, int P,
                             int G, void* stream);
int dfmir_patch_gather_bwd_g(const float* dout, const long long* ids, float* dfeat, int B, int C, long long S, int P,
                             int G, float* dfeat_amax, void* stream);
/* bwd_g that also keeps the per-plane maxima dfeat_pmax[B*C] (see dfmir_instnorm_bwd_pmax) valid.  accumulates */
int dfmir_patch_gather_bwd_gp(const float* dout, const long long* ids, float* dfeat, int B, int C, long long S, int P,
                              int G, float* dfeat_amax, float* dfeat_pmax, void* stream); /* accumulates */
/* The grouped scatter for ARBITRARY ids (repeats allowed within a group: a caller-supplied `patch_ids` list,
 * models/networks.py:602-608): accumulates with atomics, as torch's index backward does.  dfeat_amax / dfeat_pmax
 * optional (NULL), kept valid as above. */
int dfmir_patch_gather_bwd_any(const float* dout, const long long* ids, float* dfeat, int B, int C, long long S, int P,
                               int G, float* dfeat_amax, float* dfeat_pmax, void* stream); /* accumulates */
/* Key side of several NCE terms in one launch: group g gathers its Bper images from srcs[g] ([Bper,C,S]; srcs is a HOST
 * array of G <= 8 device pointers, copied by value into the launch) at ids[g][0..P) -> out[c][(g*Bper+b)*P+p].
 * Replaces `feat_k_pool, sample_ids = self.netF(feat_k, num_patches, None)` called once per term
 * (models/registration_model.py:244-245; gather = models/networks.py:604-611). */
int dfmir_patch_gather_fwd_multi(const float* const* srcs, int G, const long long* ids, float* out, int Bper, int C,
                                 long long S, int P, void* stream);
/* The PatchNCE head of one layer in one launch (PatchSampleF.forward, models/networks.py:602-619): sample ids[g][0..P) from
 * group g's images srcs[g] [Bper, C, S], Linear(C, 256) + ReLU, Linear(256, 256), x / (||x||_2 + eps) -> out [256][G*Bper*P]
 * (channel-major rows, row = (g*Bper + b)*P + p).  w1 [C][256], w2 [256][256]: forward packings of dfmir_weight_pack
 * (T = 1); C <= 256.  Optional outputs for the backward: nrm [rows], xs [C][rows] (sampled features), hs [256][rows]
 * (after the ReLU), ypre [256][rows] (before the normalisation).  srcs: HOST array of G <= 8 device pointers. */
int dfmir_nce_head_fwd(const float* const* srcs, int G, const long long* ids, const float* w1, const float* b1,
                       const float* w2, const float* b2, float* out, float* nrm, float* xs, float* hs, float* ypre,
                       int Bper, int C, long long S, int P, float eps, void* stream);
/* Patch positions drawn on the device: out[layer][set][0..P) = a uniformly random P-subset of [0, sizes[layer]) --
 * what `torch.randperm(H*W)[:num_patches]` (models/networks.py:609-610) yields per layer and per netF call, for
 * n_sets calls at once (the set, not its order, is what PatchNCELoss sees).  sizes: HOST array of n_layers <= 8
 * counts.  state: 3 device uint64 {seed, draw counter, 0}; the kernel advances the counter itself, so a captured
 * hipGraph draws fresh ids at every replay.  Deterministic for a given (seed, counter).  P <= 1024; sizes[l] >= 2P
 * (rejection sampling) or P <= sizes[l] <= 4096 (a sorted permutation). */
int dfmir_patch_ids_draw(unsigned long long* state, const long long* sizes, int n_layers, int n_sets, int P,
                         long long* out, void* stream);
/* per row: y = x / (sqrt(sum_c x^2) + eps); norm[rows] saved. */
int dfmir_l2norm_fwd(const float* x, float* y, float* norm, int C, long long rows, float eps, void* stream);
int dfmir_l2norm_bwd(const float* dy, const float* x, const float* norm, float* dx, int C,
                     long long rows, float eps, void* stream);
/* q,k: [C][rows]; G consecutive groups of R = rows/G rows share negatives (R % 16 == 0).
 * loss[rows]; probs[rows][R+1] = softmax of [pos | neg]/T, saved for backward. */
int dfmir_patchnce_fwd(const float* q, const float* k, float* loss, float* probs, long long rows,
                       int C, int G, float T, void* stream);
int dfmir_patchnce_bwd(const float* dloss, const float* probs, const float* k, float* dq,
                       long long rows, int C, int G, float T, void* stream);

/* ------------------------------------------------------------------------------------------
 * Scalar losses.  `out` is a 1-float device scalar; `ws` is a small zero-able workspace (8 floats).
 * ---------------------------------------------------------------------------------------- */
/* calculate_L1_loss (registration_model.py:255-263) with mask = (a>thr)|(b>thr)
 * (registration_model.py:160-161) when mask == NULL, else the given byte mask.
 * out = sum(|a-b|*m)/sum(m)  (0 when sum(m)==0). */
int dfmir_masked_l1_fwd(const float* a, const float* b, const unsigned char* mask, float thr,
                        float* ws, float* out, long long n, void* stream);
int dfmir_masked_l1_bwd(const float* a, const float* b, const unsigned char* mask, float thr,
                        const float* ws, const float* gout, float* da, float* db, long long n,
                        void* stream);
/* smooothing_loss (registration_model.py:25-32) / Grad_Loss l2 (util/losses.py:81-130):
 * mean over axes of mean(squared forward difference) (D==1: 2 axes). flow [B,C,D,H,W].
 * ws: dfmir_flow_smooth_ws_floats() floats of scratch (need not be zeroed): every workgroup leaves its three partial sums in
 * its own slots and the finaliser adds them in index order -- the loss is bit-reproducible. */
long long dfmir_flow_smooth_ws_floats(void);
int dfmir_flow_smooth_fwd(const float* flow, float* ws, float* out, int B, int C, int D, int H,
                          int W, void* stream);
int dfmir_flow_smooth_bwd(const float* flow, const float* gout, float* dflow, int B, int C, int D,
                          int H, int W, void* stream);
/* The same with the penalty chosen: 1 = 'l1' (mean |forward difference|), 2 = 'l2' -- Grad_Loss(penalty=...)
 * (util/losses.py:81-130) and vxm Grad(penalty).loss (models/voxelmorph/torchvoxelmorph/losses.py:93-117). */
int dfmir_flow_smooth_fwd_p(const float* flow, float* ws, float* out, int B, int C, int D, int H,
                            int W, int penalty, void* stream);
int dfmir_flow_smooth_bwd_p(const float* flow, const float* gout, float* dflow, int B, int C, int D,
                            int H, int W, int penalty, void* stream);
/* out = a * b element-wise: `prediction * mask` of Grad_Loss.forward (util/losses.py:120-121). */
int dfmir_mul(const float* a, const float* b, float* out, long long n, void* stream);
/* NCC_Loss (util/losses.py:183-261), mean kernel of `win` per axis (odd), zero padding:
 * out = -sqrt(mean(cross^2/(Ivar*Jvar+eps))).  tmp: 5*numel floats of scratch (box sums, kept for
 * backward). I = prediction, J = target, [B,1,D,H,W].  3-D, win 9: the W and H box passes run in one launch each way
 * (products / gradient fields formed on a haloed LDS tile); DFMIR_NCC_NO_WH_FUSE=1 = one launch per axis. */
int dfmir_ncc_fwd(const float* I, const float* J, float* tmp, float* tmp2, float* ws, float* out,
                  int B, int D, int H, int W, int win, float eps, void* stream);
int dfmir_ncc_bwd(const float* I, const float* J, const float* sums, float* tmp, float* tmp2,
                  const float* ws, const float* gout, float* dI, int B, int D, int H, int W,
                  int win, float eps, void* stream);
/* The same with a weight per voxel and the reduction chosen.  mask (numel floats, may be NULL): the masked branch of
 * NCC_Loss.forward (util/losses.py:257-261): out = -sqrt(sum(cc * mask) / sum(mask)), 0 when sum(mask) == 0.
 * mode 0 = that form; mode 1 = -sum(cc [* mask]) / n, n = numel (or sum(mask)): vxm NCC(win).loss = -mean(cc)
 * (models/voxelmorph/torchvoxelmorph/losses.py:15-67).  ws[0] = sum(cc * mask), ws[1] = sum(mask) (kept for backward). */
int dfmir_ncc_fwd_m(const float* I, const float* J, const float* mask, int mode, float* tmp, float* tmp2,
                    float* ws, float* out, int B, int D, int H, int W, int win, float eps, void* stream);
int dfmir_ncc_bwd_m(const float* I, const float* J, const float* mask, int mode, const float* sums,
                    float* tmp, float* tmp2, const float* ws, const float* gout, float* dI, int B, int D,
                    int H, int W, int win, float eps, void* stream);
/* out[t] = scale * sum_l mean(rows[l][t*seg..(t+1)*seg)), rows [L][T*seg]: the per-term
 * `total_nce_loss += loss.mean() * lambda_NCE` ... `/ n_layers` of calculate_NCE_loss (registration_model.py:247-253)
 * for T terms and L layers at once (scale = lambda_NCE / n_layers); bwd fills drows from gout[T]. */
int dfmir_segment_means_fwd(const float* rows, float* out, int L, int T, long long seg, float scale, void* stream);
int dfmir_segment_means_bwd(const float* gout, float* drows, int L, int T, long long seg, float scale, void* stream);
/* out[j] = sum_i M[j*n_in+i] * *in[i]: the scalar algebra of the step's loss terms (loss_G, loss_R, loss_local,
 * loss_smooth and their sum; registration_model.py:163-166,230-234) as one launch.  in: HOST array of n_in <= 8 device
 * scalar pointers, M: HOST row-major [n_out][n_in], both copied by value.  bwd: din[i] = sum_j M[j*n_in+i] * gout[j]. */
int dfmir_scalar_combine_fwd(const float* const* in, int n_in, const float* M, int n_out, float* out, void* stream);
int dfmir_scalar_combine_bwd(const float* gout, int n_in, const float* M, int n_out, float* din, void* stream);
/* p[0..n) = 0 (p 16-byte aligned).  Replaces torch.zeros / Tensor.zero_ on the step (gradient arenas, scatter targets):
 * those issue hipMemsetAsync, which a captured step turns into hipGraph memset nodes -- not reliably ordered with the
 * kernel nodes around them on ROCm 7.2. */
int dfmir_fill_zero(float* p, long long n, void* stream);
/* out = scale * sum(x) (out zeroed by this call). */
int dfmir_sum_scaled(const float* x, float* out, long long n, float scale, void* stream);
/* dx[i] = gout[0]*scale */
int dfmir_fill_from_scalar(const float* gout, float* dx, long long n, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Deterministic weight gradients (build-defined, opt.deterministic_wgrad; the reference's cuDNN backward is not
 * deterministic either, models/base_model.py:37-38 sets cudnn.benchmark).  The weight- and bias-gradient kernels split
 * their reduction over workgroups and add the partial sums with fp32 atomics: every run rounds in a different order.
 * Between dfmir_det_begin and dfmir_det_end EVERY gradient entry point called by this host thread (dfmir_conv_wgrad*,
 * dfmir_conv3d_split_wgrad*, dfmir_conv3d_upwgrad, dfmir_conv3d_s2c2_wgrad, dfmir_conv7x7_c1_wgrad, dfmir_bias_grad) adds
 * 64-bit FIXED-POINT integers instead -- associative, hence bit-reproducible -- into `scratch`:
 *   scratch: dfmir_det_head_floats() + 2 * slots floats, 16-byte aligned; the call zeroes it.  The caller passes
 *     scratch + dfmir_det_head_floats() as the entry point's dw_tcc (8 bytes per element: element i of the gradient is
 *     slot i) and NULL as its db; a bias gradient is taken with dfmir_bias_grad into slots n_dw .. n_dw + n_db - 1
 *     (pointer scratch + head + 2 * n_dw).
 *   scale: a power of two chosen on the device such that count * max(max|x|, 1) * max|dy| -- a bound of every sum formed --
 *     maps below 2^61.  x_amax / dy_amax: range probes as for the split kernels, or NULL with x / dy given (measured here).
 *   dfmir_det_end: dw_out[i] += slot[i] / scale (i < n_dw), db_out[j] += slot[n_dw + j] / scale, and ends the mode.
 * ---------------------------------------------------------------------------------------- */
int dfmir_det_head_floats(void);
int dfmir_det_begin(float* scratch, long long slots, const float* x, long long nx, const float* x_amax, int x_amax_n,
                    const float* dy, long long ndy, const float* dy_amax, int dy_amax_n, double count, void* stream);
int dfmir_det_end(const float* scratch, float* dw_out, long long n_dw, float* db_out, long long n_db, void* stream);

/* ------------------------------------------------------------------------------------------
 * torch.optim.Adam(lr, betas) step over one flat parameter arena
 * (registration_model.py:114-117,135,168-171).  grad is pre-scaled by grad_scale (1/world under DDP).
 * ---------------------------------------------------------------------------------------- */
int dfmir_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                    float beta2, float eps, float bc1, float bc2, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DFMIR_HIP_H */

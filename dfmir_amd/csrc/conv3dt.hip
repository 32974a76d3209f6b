// 3x3x3 stride-1 zero-pad convolutions with a TINY channel count on one side: the VoxelMorph flow head
// (torchvoxelmorph/networks.py:1076-1080: Conv3d(16, 3, kernel_size=3, padding=1)) and its data gradient (3 -> 16).
//
// Neither side of these layers fills a matrix-core tile: 3 of the 16 rows of v_mfma_f32_16x16x32_f16 carry output
// channels in the forward conv (19 % of the issued products useful, and three fp16 products per fp32 MAC on top), 3 of the
// 8 channels of a K chunk carry input in the data gradient -- 34-37 TFLOP/s on the split kernels, 0.5 ms per launch at
// 160 x 192 x 224 for 17.8 GFLOP and 0.52 GB of traffic.  Plain fp32 FMAs need no padding and no operand split: 1 296 FMAs per
// voxel against 76 bytes, i.e. the vector ALUs (157 TFLOP/s) and HBM are balanced within a factor of two.
//
// Workgroup = 4 x 8 RY x 32 output voxels, 256 threads, thread = 4 consecutive x of RY consecutive rows, ALL output channels
// (3 x 8 or 16 x 4 accumulators; two rows per thread in the 3-channel form: 4 patch rows and 27 x 3 weights per 216 FMAs --
// with one row the loop ran at half the vector rate, 0.47 ms against 0.25 ms for the same FMAs in the 16-channel form).
// Input channels arrive in chunks of CC: the (6 x (8 RY + 2) x 34) halo patch of each channel goes global ->
// registers (16-byte rows + two halo columns, one chunk ahead of the FMAs) -> LDS [c][z][y][40] (interior at column 4, so
// the 4 centre values of a thread are one aligned ds_read_b128, the two neighbours a ds_read_b32 each).  Weights are
// read through wave-uniform addresses from the fp32 tap-major packing [27][Cin][Cout] (scalar loads: an SGPR operand per FMA).
// Epilogue as the split kernels': bias, activation, optionally the derivative of the LeakyReLU whose output is `act_src`
// (the data gradient lands in front of that activation), the range probe of the result for the next split conv.
#include "common.h"

namespace {
constexpr int T_TZ = 4, T_TX = 32, T_PZ = T_TZ + 2, T_PITCH = 40;
struct C3tP {
  int N, D, H, W, nz, ny, nx, act;
  float slope;
  const float* act_src;
  float act_slope;
  long long ntile;
};

template <int CIN, int COUT, int CC, int RY>
__global__ __launch_bounds__(256) void conv3d_tiny_k(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     float* __restrict__ y_amax, C3tP k) {
  static_assert(CIN % CC == 0, "whole chunks");
  constexpr int T_TY = 8 * RY, T_PY = T_TY + 2;          // a thread owns RY consecutive rows x 4 consecutive x
  constexpr int ROWS = CC * T_PZ * T_PY;
  constexpr int NV = (ROWS * 8 + 255) / 256, NH = (ROWS * 2 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Xs[ROWS * T_PITCH];
  __shared__ unsigned smax;
  const int tid = threadIdx.x;
  if (tid == 0) smax = 0u;
  // workgroup ids go round-robin over the 8 XCDs: XCD e = id & 7 walks a contiguous eighth of the tiles (shared halos stay
  // in one L2)
  const long long per = (k.ntile + 7) / 8;
  long long bt = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const bool live = (long long)(blockIdx.x >> 3) < per && bt < k.ntile;
  if (!live) bt = 0;
  const int tx = (int)(bt % k.nx); bt /= k.nx;
  const int ty = (int)(bt % k.ny); bt /= k.ny;
  const int tz = (int)(bt % k.nz);
  const int n = (int)(bt / k.nz);
  const int z0 = tz * T_TZ, y0 = ty * T_TY, x0 = tx * T_TX;
  const long long DHW = (long long)k.D * k.H * k.W;
  const float* xn = x + (long long)n * CIN * DHW;

  // staging plan of one chunk (the same for every chunk: only the channel base moves)
  int offv[NV], offh[NH];                                 // element offset inside the chunk's first channel, -1 = zero
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i = tid + 256 * j, row = i >> 3, q = i & 7;
    const int c = row / (T_PZ * T_PY), r = row - c * (T_PZ * T_PY), pz = r / T_PY, py = r - pz * T_PY;
    const int z = z0 - 1 + pz, yy = y0 - 1 + py, xx = x0 + 4 * q;
    const bool ok = live && row < ROWS && z >= 0 && z < k.D && yy >= 0 && yy < k.H && xx < k.W;
    offv[j] = ok ? (int)(c * DHW + ((long long)z * k.H + yy) * k.W + xx) : -1;
  }
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int i = tid + 256 * j, row = i >> 1, side = i & 1;
    const int c = row / (T_PZ * T_PY), r = row - c * (T_PZ * T_PY), pz = r / T_PY, py = r - pz * T_PY;
    const int z = z0 - 1 + pz, yy = y0 - 1 + py, xx = side ? x0 + T_TX : x0 - 1;
    const bool ok = live && row < ROWS && z >= 0 && z < k.D && yy >= 0 && yy < k.H && xx >= 0 && xx < k.W;
    offh[j] = ok ? (int)(c * DHW + ((long long)z * k.H + yy) * k.W + xx) : -1;
  }
  float4 rv[NV];
  float rh[NH];
#define T_GLOAD(c0_)                                                                               \
  {                                                                                                \
    const float* xc = xn + (long long)(c0_) * DHW;                                                 \
    _Pragma("unroll") for (int j = 0; j < NV; ++j)                                                 \
      rv[j] = offv[j] >= 0 ? *reinterpret_cast<const float4*>(xc + offv[j]) : make_float4(0.f, 0.f, 0.f, 0.f); \
    _Pragma("unroll") for (int j = 0; j < NH; ++j) rh[j] = offh[j] >= 0 ? xc[offh[j]] : 0.f;      \
  }
#define T_LSTORE()                                                                                 \
  {                                                                                                \
    _Pragma("unroll") for (int j = 0; j < NV; ++j) {                                               \
      const int i = tid + 256 * j;                                                                 \
      if (ROWS * 8 % 256 == 0 || i < ROWS * 8)                                                     \
        *reinterpret_cast<float4*>(&Xs[(i >> 3) * T_PITCH + 4 + 4 * (i & 7)]) = rv[j];             \
    }                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NH; ++j) {                                               \
      const int i = tid + 256 * j;                                                                 \
      if (ROWS * 2 % 256 == 0 || i < ROWS * 2) Xs[(i >> 1) * T_PITCH + ((i & 1) ? 4 + T_TX : 3)] = rh[j]; \
    }                                                                                              \
  }

  const int lx = tid & 7, ly = (tid >> 3) & 7, lz = tid >> 6;
  float acc[COUT][4 * RY];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int p = 0; p < 4 * RY; ++p) acc[co][p] = 0.f;

  T_GLOAD(0)
  for (int c0 = 0; c0 < CIN; c0 += CC) {
    __syncthreads();                                       // the previous chunk's readers are done
    T_LSTORE()
    __syncthreads();
    if (c0 + CC < CIN) T_GLOAD(c0 + CC)
    // the weights of one loop body must fit the scalar registers: with these loops unrolled the compiler hoisted all
    // 27 CC COUT scalar loads and spilled thousands of SGPRs
#pragma unroll 1
    for (int c = 0; c < CC; ++c) {
#pragma unroll 1
      for (int dz = 0; dz < 3; ++dz) {
        const float* plane = &Xs[((c * T_PZ + lz + dz) * T_PY + RY * ly) * T_PITCH + 4 * lx + 3];
        if constexpr (COUT <= 4) {
          // few output channels: RY + 2 patch rows serve the 3 dy taps of RY output rows; 27 COUT weights per (c, dz)
          float v[RY + 2][6];
#pragma unroll
          for (int r = 0; r < RY + 2; ++r) {
            const float* row = plane + r * T_PITCH;
            v[r][0] = row[0];
            const float4 m = *reinterpret_cast<const float4*>(row + 1);
            v[r][1] = m.x; v[r][2] = m.y; v[r][3] = m.z; v[r][4] = m.w;
            v[r][5] = row[5];
          }
          const float* wt = w + (dz * 9 * CIN + (c0 + c)) * COUT;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int co = 0; co < COUT; ++co) {
                const float wv = wt[(dy * 3 + dx) * CIN * COUT + co];
#pragma unroll
                for (int r = 0; r < RY; ++r)
#pragma unroll
                  for (int p = 0; p < 4; ++p) acc[co][r * 4 + p] = fmaf(wv, v[r + dy][p + dx], acc[co][r * 4 + p]);
              }
        } else {
          static_assert(COUT <= 4 || RY == 1, "many output channels: one row per thread");
#pragma unroll 1
          for (int dy = 0; dy < 3; ++dy) {
            const float* row = plane + dy * T_PITCH;
            float v[6];
            v[0] = row[0];
            const float4 m = *reinterpret_cast<const float4*>(row + 1);
            v[1] = m.x; v[2] = m.y; v[3] = m.z; v[4] = m.w;
            v[5] = row[5];
            const float* wt = w + ((dz * 3 + dy) * 3 * CIN + (c0 + c)) * COUT;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int co = 0; co < COUT; ++co) {
                const float wv = wt[dx * CIN * COUT + co];
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[co][p] = fmaf(wv, v[p + dx], acc[co][p]);
              }
          }
        }
      }
    }
  }
#undef T_GLOAD
#undef T_LSTORE

  const int z = z0 + lz, xx = x0 + 4 * lx;
  float pm = 0.f;
#pragma unroll
  for (int rr = 0; rr < RY; ++rr) {
    const int yy = y0 + RY * ly + rr;
    if (!(live && z < k.D && yy < k.H && xx < k.W)) continue;
    const long long sp = ((long long)z * k.H + yy) * k.W + xx;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const long long o = ((long long)n * COUT + co) * DHW + sp;
      const float b = bias ? bias[co] : 0.f;
      float r[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float t = acc[co][rr * 4 + p] + b;
        if (k.act == 1) t = t > 0.f ? t : t * k.slope;
        else if (k.act == 2) t = tanhf(t);
        r[p] = t;
      }
      if (k.act_src) {
        const float4 a = *reinterpret_cast<const float4*>(k.act_src + o);
        r[0] = a.x > 0.f ? r[0] : r[0] * k.act_slope; r[1] = a.y > 0.f ? r[1] : r[1] * k.act_slope;
        r[2] = a.z > 0.f ? r[2] : r[2] * k.act_slope; r[3] = a.w > 0.f ? r[3] : r[3] * k.act_slope;
      }
      *reinterpret_cast<float4*>(y + o) = make_float4(r[0], r[1], r[2], r[3]);
      pm = fmaxf(fmaxf(pm, fmaxf(fabsf(r[0]), fabsf(r[1]))), fmaxf(fabsf(r[2]), fabsf(r[3])));
    }
  }
  if (y_amax) {
    __syncthreads();
    publish_block_absmax_acc(pm, &smax, y_amax);
  }
}
}  // namespace

static bool tiny_off() {
  static DfOptFlag o{"DFMIR_CONV3D_NO_TINY"};
  return o.get();
}
extern "C" int dfmir_conv3d_tiny_ok(const DfConvGeom* g) {
  if (!g || tiny_off()) return 0;
  if (!(g->KD == 3 && g->KH == 3 && g->KW == 3 && g->stride == 1 && g->dil == 1 && g->pd == 1 && g->ph == 1 && g->pw == 1 &&
        g->pad_mode == 0 && g->Do == g->Di && g->Ho == g->Hi && g->Wo == g->Wi && g->Di > 1 && (g->Wi & 3) == 0))
    return 0;
  if (!((g->Cin == 16 && g->Cout == 3) || (g->Cin == 3 && g->Cout == 16))) return 0;
  const long long dhw = (long long)g->Di * g->Hi * g->Wi;
  return (long long)16 * dhw < 0x7FFFFFFFLL ? 1 : 0;       // int offsets inside a chunk of <= 4 channels, with room
}
// y = act(conv3x3x3(x, w) + bias) [* LeakyReLU'(act_src)], w = the fp32 tap-major packing [27][Cin][Cout] of
// dfmir_weight_pack (mode 0: forward, mode 1: the data gradient as a forward conv); y_amax (may be NULL): DF_PROBE_SLOTS
// accumulating range-probe slots of y.
extern "C" int dfmir_conv3d_tiny_fwd(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias, float* y,
                                     float* y_amax, const float* act_src, float act_slope, void* stream) {
  DF_ARG_CHECK(g && x && w_tcc && y && dfmir_conv3d_tiny_ok(g));
  DF_ARG_CHECK((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(act_src) & 15) == 0);
  DF_ARG_CHECK(!act_src || g->act == 0);
  const int t_ty = g->Cin == 16 ? 16 : 8;                  // the forward form: two rows per thread
  C3tP k{g->N, g->Di, g->Hi, g->Wi, (g->Di + T_TZ - 1) / T_TZ, (g->Hi + t_ty - 1) / t_ty, (g->Wi + T_TX - 1) / T_TX,
         g->act, g->slope, act_src, act_slope, 0};
  k.ntile = (long long)g->N * k.nz * k.ny * k.nx;
  DF_ARG_CHECK(k.ntile < (1LL << 30));
  const unsigned grid = (unsigned)(8 * ((k.ntile + 7) / 8));
  hipStream_t st = (hipStream_t)stream;
  if (g->Cin == 16) conv3d_tiny_k<16, 3, 2, 2><<<grid, 256, 0, st>>>(x, w_tcc, bias, y, y_amax, k);
  else conv3d_tiny_k<3, 16, 3, 1><<<grid, 256, 0, st>>>(x, w_tcc, bias, y, y_amax, k);
  DF_LAUNCH_CHECK();
  return 0;
}

// ================================================================================================
// The first encoder level of the 3-D U-Net: Conv3d(2, 16, kernel_size=3, stride=2, padding=1) + LeakyReLU over
// cat(source, target) at full resolution (torchvoxelmorph/networks.py:66-86,1105).  The
// layer is pure memory traffic (55 MB in, 55 MB out at 160 x 192 x 224) but ran on the generic gather kernels (per-element
// index arithmetic, 4-byte loads): 0.18 ms forward at 8 TFLOP/s, 0.25 ms weight gradient at 6 TFLOP/s.
//
// Forward (conv3d_s2c2_fwd_k): workgroup = 2 x 8 x 32 output voxels, 128 threads, thread = 4 consecutive output x, all 16
// output channels; the (5 x 17 x 65) input patch of both channels is staged once (16-byte rows + the left halo column,
// interior at LDS column 4: a thread's 9 input columns are one ds_read_b32 + two ds_read_b128); weights by wave-uniform
// scalar loads ([27][2][16] tap-major packing); epilogue bias + activation + range probe.
//
// Weight gradient (conv3d_s2c2_wgrad_k): dW[j][co] += sum_p Xg[j][p] dY[co][p], j = tap * 2 + ci (54 rows), on
// v_mfma_f32_16x16x4_f32 (exact fp32 products): wave w of a workgroup owns rows 16 w .. 16 w + 15, all 16 output channels;
// K = output voxels, 4 per MFMA.  Per tile the input patch and the 16 x 512 dY tile are staged in LDS; the A operand is
// gathered from the patch with a per-lane (tap, channel) base + per-k voxel offset, B is dY[co][voxel].  Persistent
// workgroups (one per CU), one atomicAdd per result and workgroup at the end.
// ================================================================================================
namespace {
constexpr int S2_TZ = 2, S2_TY = 8, S2_TX = 32, S2_PZ = 2 * S2_TZ + 1, S2_PY = 2 * S2_TY + 1, S2_PITCH = 68;
struct C3s2P {
  int N, D, H, W, Do, Ho, Wo, nz, ny, nx, act;
  float slope;
  long long ntile;
};
// stage the (2 ch x S2_PZ x S2_PY) rows of tile (n, z0, y0, x0) [output coordinates]: LDS row = [c][pz][py], column 3 =
// input x = 2 x0 - 1, columns 4 .. 67 = input x = 2 x0 .. 2 x0 + 63
template <int NT>
__device__ __forceinline__ void s2_stage_patch(const float* __restrict__ xn, float* __restrict__ Xs, const C3s2P& k, int z0,
                                               int y0, int x0, long long DHW) {
  constexpr int ROWS = 2 * S2_PZ * S2_PY;
#pragma unroll 4
  for (int i = threadIdx.x; i < ROWS * 16; i += NT) {
    const int row = i >> 4, q = i & 15;
    const int c = row / (S2_PZ * S2_PY), r = row - c * (S2_PZ * S2_PY), pz = r / S2_PY, py = r - pz * S2_PY;
    const int z = 2 * z0 - 1 + pz, yy = 2 * y0 - 1 + py, xx = 2 * x0 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (z >= 0 && z < k.D && yy >= 0 && yy < k.H && xx < k.W)
      v = *reinterpret_cast<const float4*>(xn + c * DHW + ((long long)z * k.H + yy) * k.W + xx);
    *reinterpret_cast<float4*>(&Xs[row * S2_PITCH + 4 + 4 * q]) = v;
  }
  for (int row = threadIdx.x; row < ROWS; row += NT) {
    const int c = row / (S2_PZ * S2_PY), r = row - c * (S2_PZ * S2_PY), pz = r / S2_PY, py = r - pz * S2_PY;
    const int z = 2 * z0 - 1 + pz, yy = 2 * y0 - 1 + py, xx = 2 * x0 - 1;
    float v = 0.f;
    if (z >= 0 && z < k.D && yy >= 0 && yy < k.H && xx >= 0) v = xn[c * DHW + ((long long)z * k.H + yy) * k.W + xx];
    Xs[row * S2_PITCH + 3] = v;
  }
}

__global__ __launch_bounds__(128) void conv3d_s2c2_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         float* __restrict__ y_amax, C3s2P k) {
  constexpr int COUT = 16;
  __shared__ __attribute__((aligned(16))) float Xs[2 * S2_PZ * S2_PY * S2_PITCH];
  __shared__ unsigned smax;
  const int tid = threadIdx.x;
  if (tid == 0) smax = 0u;
  const long long per = (k.ntile + 7) / 8;
  long long bt = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const bool live = (long long)(blockIdx.x >> 3) < per && bt < k.ntile;
  if (!live) bt = 0;
  const int tx = (int)(bt % k.nx); bt /= k.nx;
  const int ty = (int)(bt % k.ny); bt /= k.ny;
  const int tz = (int)(bt % k.nz);
  const int n = (int)(bt / k.nz);
  const int z0 = tz * S2_TZ, y0 = ty * S2_TY, x0 = tx * S2_TX;
  const long long DHW = (long long)k.D * k.H * k.W, DHWo = (long long)k.Do * k.Ho * k.Wo;
  s2_stage_patch<128>(x + (long long)n * 2 * DHW, Xs, k, z0, y0, x0, DHW);
  __syncthreads();
  const int lx = tid & 7, ly = (tid >> 3) & 7, lz = tid >> 6;
  float acc[COUT][4];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[co][p] = 0.f;
#pragma unroll 1
  for (int c = 0; c < 2; ++c) {
#pragma unroll 1
    for (int dz = 0; dz < 3; ++dz) {
#pragma unroll 1
      for (int dy = 0; dy < 3; ++dy) {
        const float* row = &Xs[((c * S2_PZ + 2 * lz + dz) * S2_PY + 2 * ly + dy) * S2_PITCH + 8 * lx + 3];
        float v[9];
        v[0] = row[0];
        const float4 m0 = *reinterpret_cast<const float4*>(row + 1), m1 = *reinterpret_cast<const float4*>(row + 5);
        v[1] = m0.x; v[2] = m0.y; v[3] = m0.z; v[4] = m0.w; v[5] = m1.x; v[6] = m1.y; v[7] = m1.z; v[8] = m1.w;
        const float* wt = w + ((dz * 3 + dy) * 3 * 2 + c) * COUT;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int co = 0; co < COUT; ++co) {
            const float wv = wt[dx * 2 * COUT + co];
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[co][p] = fmaf(wv, v[2 * p + dx], acc[co][p]);
          }
      }
    }
  }
  const int z = z0 + lz, yy = y0 + ly, xx = x0 + 4 * lx;
  float pm = 0.f;
  if (live && z < k.Do && yy < k.Ho && xx < k.Wo) {
    const long long sp = ((long long)z * k.Ho + yy) * k.Wo + xx;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float b = bias ? bias[co] : 0.f;
      float r[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float t = acc[co][p] + b;
        if (k.act == 1) t = t > 0.f ? t : t * k.slope;
        else if (k.act == 2) t = tanhf(t);
        r[p] = t;
      }
      *reinterpret_cast<float4*>(y + ((long long)n * COUT + co) * DHWo + sp) = make_float4(r[0], r[1], r[2], r[3]);
      pm = fmaxf(fmaxf(pm, fmaxf(fabsf(r[0]), fabsf(r[1]))), fmaxf(fabsf(r[2]), fabsf(r[3])));
    }
  }
  if (y_amax) {
    __syncthreads();
    publish_block_absmax_acc(pm, &smax, y_amax);
  }
}

typedef float s2_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void conv3d_s2c2_wgrad_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dwt, C3s2P k, const float* __restrict__ fx) {
  constexpr int COUT = 16, NV = S2_TZ * S2_TY * S2_TX;     // 512 output voxels per tile
  constexpr int DP = NV + 4;                               // dY row stride: 16 channels x 4 voxels of a B read on 64 banks
  __shared__ __attribute__((aligned(16))) float Xs[2 * S2_PZ * S2_PY * S2_PITCH];
  __shared__ __attribute__((aligned(16))) float Ds[COUT * DP];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const long long DHW = (long long)k.D * k.H * k.W, DHWo = (long long)k.Do * k.Ho * k.Wo;
  // A operand: lane (m = lane & 15, kk = lane >> 4): row j = 16 wid + m = tap * 2 + ci -> patch offset of (ci, dz, dy, dx);
  // voxel 4 s + kk of the tile (x fastest): (vz, vy, vx) -> patch offset 2 vz * PY * PITCH + 2 vy * PITCH + 2 vx
  const int j = 16 * wid + (lane & 15), kk = lane >> 4;
  const bool jok = j < 54;
  const int tap = jok ? j >> 1 : 0, ci = j & 1;
  const int dz = tap / 9, dy_ = (tap / 3) % 3, dx = tap % 3;
  const int abase = jok ? ((ci * S2_PZ + dz) * S2_PY + dy_) * S2_PITCH + dx + 3 : 0;
  s2_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long long t = blockIdx.x; t < k.ntile; t += gridDim.x) {
    long long bt = t;
    const int tx = (int)(bt % k.nx); bt /= k.nx;
    const int ty = (int)(bt % k.ny); bt /= k.ny;
    const int tz = (int)(bt % k.nz);
    const int n = (int)(bt / k.nz);
    const int z0 = tz * S2_TZ, y0 = ty * S2_TY, x0 = tx * S2_TX;
    __syncthreads();                                       // the previous tile's readers are done
    s2_stage_patch<256>(x + (long long)n * 2 * DHW, Xs, k, z0, y0, x0, DHW);
    // dY tile [co][vz][vy][vx]: 16 x 2 x 8 rows of 32 floats (8 float4)
    for (int i = tid; i < COUT * S2_TZ * S2_TY * 8; i += 256) {
      const int q = i & 7, r = i >> 3, vy = r % S2_TY, r2 = r / S2_TY, vz = r2 % S2_TZ, co = r2 / S2_TZ;
      const int z = z0 + vz, yy = y0 + vy, xx = x0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (z < k.Do && yy < k.Ho && xx < k.Wo)
        v = *reinterpret_cast<const float4*>(dy + ((long long)n * COUT + co) * DHWo + ((long long)z * k.Ho + yy) * k.Wo + xx);
      *reinterpret_cast<float4*>(&Ds[co * DP + (vz * S2_TY + vy) * S2_TX + 4 * q]) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < NV / 4; ++s) {
      const int v = 4 * s + kk, vx = v & 31, vy = (v >> 5) & 7, vz = v >> 8;
      const float a = jok ? Xs[abase + (2 * vz * S2_PY + 2 * vy) * S2_PITCH + 2 * vx] : 0.f;
      const float b = Ds[(lane & 15) * DP + v];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
  }
  // D: register r of lane l = row 4 (l >> 4) + r, column l & 15
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 16 * wid + 4 * (lane >> 4) + r;
    if (row < 54) df_acc(dwt, row * COUT + (lane & 15), acc[r], fx);
  }
}

// ------------------------------------------------------------------------------------------------
// conv3d_flow_wgrad_k (round 6): weight + bias gradient of the flow head Conv3d(16, Cout <= 4, 3, padding 1)
// (torchvoxelmorph/networks.py:1076-1080) as a z-marching streaming kernel.
//   dW[dz][dy][dx][ci][co] = sum_u x[ci][u] * dY[co][u - (dz - 1, dy - 1, dx - 1)]
// With the (co, dx) pairs as the MFMA COLUMNS (3 Cout <= 12 of 16) and the x-row of 32 voxels as K, one
// v_mfma_f32_16x16x32_f16 per (dz, dy) and split product turns an x-row of all 16 input channels into its share of nine
// [16 ci x (co, dx)] tiles: 27 MFMAs per 32 voxels (the swapped-role tiled kernel this replaces walked 28 tap rows x 8
// padded channels against 32 columns: 0.33 ms for 0.52 GB).  A operand = 8 consecutive voxels of one channel: one 16-byte
// read of a channel-major fp16 image, no transposing read.  The dx shift of the B operand would be a 2-byte misalignment:
// the three shifted copies of every dY row are made when the row is staged (3 channels: cheap).  A workgroup owns an
// 8 x 32 column and marches along z: every x plane and every dY plane (ring of three: the taps dz reach z - 1 .. z + 1) is
// staged once as scaled fp16 pairs, in-plane halo of the 3-channel operand only.  The four waves (two rows each) meet in
// LDS in wave order, then one set of df_acc adds per workgroup; the bias gradient is summed from the staged dY.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_pair_t(float x0, float x1, float s, unsigned& h, unsigned& r) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(x1), "v"(s), "v"(h));
}
__device__ __forceinline__ void split8_t(const float* v, float s, u32x4_t& h, u32x4_t& r) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned hh, rr;
    split_pair_t(v[2 * q], v[2 * q + 1], s, hh, rr);
    h[q] = hh; r[q] = rr;
  }
}
__device__ __forceinline__ f32x4_t mma16t(u32x4_t a, u32x4_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ int scale_exp_t(float amax) {
  const int be = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  int e = (amax > 0.f) ? 14 - be : 0;
  return e < -100 ? -100 : (e > 100 ? 100 : e);
}
struct FwP {
  int N, D, H, W, Cout;
  int x_n, dy_n;                 // floats of the two range probes
  int ncy, ncx, nseg, zlen;      // 8 x 32 columns per plane, z segments of zlen planes
  float* db;                     // optional bias gradient
  const float* fx;               // deterministic mode (common.h df_acc)
};
constexpr int FW_XCS = 33;                       // 16-byte units per channel of an x plane slot: 8 rows x 4 + 1 (bank spread)
constexpr int FW_XSLOT = 2 * 16 * FW_XCS;        // [split][ci]
constexpr int FW_DCS = 41;                       // units per (co, dx) column of a dY plane slot: 10 rows x 4 + 1
// CO = the most output channels the instance holds: 3 (the flow head; 69 KB of LDS: two workgroups per CU) or 4
template <int CO>
__global__ __launch_bounds__(256, CO <= 3 ? 3 : 2) void conv3d_flow_wgrad_k(const float* __restrict__ x, const float* __restrict__ x_amax,
                                                              const float* __restrict__ dy, const float* __restrict__ dy_amax,
                                                              float* __restrict__ dwt, FwP k) {
  constexpr int FW_NCOL = 3 * CO;                  // (co, dx) columns held
  constexpr int FW_DSLOT = 2 * FW_NCOL * FW_DCS;   // [split][col]
  // ONE x slot: plane z + 1 is written after the barrier that ends the reads of plane z (16.9 KB + 35.4 KB: three workgroups per CU)
  __shared__ __attribute__((aligned(16))) u32x4_t Xa[FW_XSLOT];
  __shared__ __attribute__((aligned(16))) u32x4_t Dsh[3 * FW_DSLOT];     // 35.4 KB for CO = 3 (also the epilogue's reduction buffer)
  __shared__ float red[17];
  __shared__ float bsum[160];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, kg = lane >> 4;
  int it = blockIdx.x;
  const int cx = it % k.ncx; it /= k.ncx;
  const int cy = it % k.ncy; it /= k.ncy;
  const int seg = it % k.nseg;
  const int n = it / k.nseg;
  const int y0 = cy * 8, x0 = cx * 32, zs = seg * k.zlen;
  int ze = zs + k.zlen;
  if (ze > k.D) ze = k.D;
  const long long HW = (long long)k.H * k.W, S = HW * k.D;
  const float* xn = x + (long long)n * 16 * S;
  const float* dn = dy + (long long)n * k.Cout * S;
  const int ex = scale_exp_t(reduce_absmax(x_amax, k.x_n, red));
  __syncthreads();
  const int ed = scale_exp_t(reduce_absmax(dy_amax, k.dy_n, red));
  const float xscale = __uint_as_float((unsigned)(ex + 127) << 23), dscale = __uint_as_float((unsigned)(ed + 127) << 23);
  const float osc = __uint_as_float((unsigned)(-ex + 127) << 23) * __uint_as_float((unsigned)(-ed + 127) << 23);
  const int ncol = 3 * k.Cout;

  // x staging: two items per thread, item = (ci, row, 8 voxels)
  float4 rx[2][2];
  // dY staging: threads 0 .. Cout * 40 - 1, item = (co, patch row 0 .. 9 <-> y0 - 1 .. y0 + 8, 8-voxel group): voxels
  // x0 + 8 g - 1 .. x0 + 8 g + 8
  const int d_co = tid / 40, d_row = (tid % 40) >> 2, d_g = tid & 3;
  const bool d_act = tid < k.Cout * 40;
  float4 rd[2];
  float rdl = 0.f, rdr = 0.f;
  float bacc = 0.f;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define FW_GLOAD_X(z_)                                                                                     \
  {                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                        \
      const int i_ = tid + 256 * j, ci_ = i_ >> 5, row_ = (i_ >> 2) & 7, g_ = i_ & 3;                      \
      const int yy_ = y0 + row_, xx_ = x0 + 8 * g_;                                                        \
      const bool ok_ = (z_) < k.D && yy_ < k.H;                                                            \
      const float* p_ = xn + (long long)ci_ * S + (long long)(z_) * HW + (long long)yy_ * k.W + xx_;       \
      rx[j][0] = (ok_ && xx_ < k.W) ? *reinterpret_cast<const float4*>(p_) : z4;                           \
      rx[j][1] = (ok_ && xx_ + 4 < k.W) ? *reinterpret_cast<const float4*>(p_ + 4) : z4;                   \
    }                                                                                                      \
  }
#define FW_LSTORE_X(buf_)                                                                                  \
  {                                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                        \
      const int i_ = tid + 256 * j, ci_ = i_ >> 5, row_ = (i_ >> 2) & 7, g_ = i_ & 3;                      \
      const float v_[8] = {rx[j][0].x, rx[j][0].y, rx[j][0].z, rx[j][0].w, rx[j][1].x, rx[j][1].y, rx[j][1].z, rx[j][1].w}; \
      u32x4_t h_, r_;                                                                                      \
      split8_t(v_, xscale, h_, r_);                                                                        \
      Xa[ci_ * FW_XCS + row_ * 4 + g_] = h_;                                                               \
      Xa[(16 + ci_) * FW_XCS + row_ * 4 + g_] = r_;                                                        \
    }                                                                                                      \
  }
#define FW_GLOAD_D(z_)                                                                                     \
  {                                                                                                        \
    rd[0] = z4; rd[1] = z4; rdl = 0.f; rdr = 0.f;                                                          \
    const int yy_ = y0 - 1 + d_row, xx_ = x0 + 8 * d_g;                                                    \
    if (d_act && (z_) >= 0 && (z_) < k.D && yy_ >= 0 && yy_ < k.H) {                                       \
      const float* p_ = dn + (long long)d_co * S + (long long)(z_) * HW + (long long)yy_ * k.W + xx_;      \
      if (xx_ < k.W) rd[0] = *reinterpret_cast<const float4*>(p_);                                         \
      if (xx_ + 4 < k.W) rd[1] = *reinterpret_cast<const float4*>(p_ + 4);                                 \
      if (xx_ > 0 && xx_ - 1 < k.W) rdl = p_[-1];                                                          \
      if (xx_ + 8 < k.W) rdr = p_[8];                                                                      \
    }                                                                                                      \
  }
  // the three shifted copies: column (co, dx) element x' = dY[x' - dx + 1]  ->  dx 0: +1 .. +8, dx 1: 0 .. 7, dx 2: -1 .. 6
#define FW_LSTORE_D(slot_, own_)                                                                           \
  if (d_act) {                                                                                             \
    const float w_[10] = {rdl, rd[0].x, rd[0].y, rd[0].z, rd[0].w, rd[1].x, rd[1].y, rd[1].z, rd[1].w, rdr}; \
    if ((own_) && d_row >= 1 && d_row <= 8)                                                                \
      bacc += ((w_[1] + w_[2]) + (w_[3] + w_[4])) + ((w_[5] + w_[6]) + (w_[7] + w_[8]));                   \
    _Pragma("unroll") for (int dx_ = 0; dx_ < 3; ++dx_) {                                                  \
      u32x4_t h_, r_;                                                                                      \
      split8_t(&w_[2 - dx_], dscale, h_, r_);                                                              \
      const int u_ = (slot_) * FW_DSLOT + (d_co * 3 + dx_) * FW_DCS + d_row * 4 + d_g;                     \
      Dsh[u_] = h_;                                                                                        \
      Dsh[u_ + FW_NCOL * FW_DCS] = r_;                                                                     \
    }                                                                                                      \
  }

  f32x4_t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // prologue: dY planes zs - 1, zs, zs + 1 -> slots (z + 3) % 3; x plane zs -> buffer 0
  FW_GLOAD_D(zs - 1) FW_LSTORE_D((zs + 2) % 3, false)
  FW_GLOAD_D(zs) FW_LSTORE_D(zs % 3, true)
  FW_GLOAD_D(zs + 1) FW_LSTORE_D((zs + 1) % 3, zs + 1 < ze)
  FW_GLOAD_X(zs) FW_LSTORE_X(0)
  const int bcol = l15 < ncol ? l15 : 0;                    // padding columns read column 0 (their results are dropped)
  for (int z = zs; z < ze; ++z) {
    __syncthreads();                                        // x plane z and dY plane z + 1 are in place
    const bool more = z + 1 < ze;
    if (more) FW_GLOAD_X(z + 1)
    FW_GLOAD_D(z + 2)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = 2 * wid + rr;
      const u32x4_t ah = Xa[l15 * FW_XCS + row * 4 + kg];
      const u32x4_t ar = Xa[(16 + l15) * FW_XCS + row * 4 + kg];
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const int slot = (z - dz + 1 + 3) % 3;
#pragma unroll
        for (int dyy = 0; dyy < 3; ++dyy) {
          const int u = slot * FW_DSLOT + bcol * FW_DCS + (row - dyy + 2) * 4 + kg;
          const u32x4_t bh = Dsh[u], br = Dsh[u + FW_NCOL * FW_DCS];
          f32x4_t a_ = acc[dz * 3 + dyy];
          a_ = mma16t(ah, bh, a_);
          a_ = mma16t(ah, br, a_);
          a_ = mma16t(ar, bh, a_);
          acc[dz * 3 + dyy] = a_;
        }
      }
    }
    __syncthreads();                                        // the readers of x plane z and of dY plane z - 1 are done
    if (more) FW_LSTORE_X(0)
    FW_LSTORE_D((z + 2) % 3, z + 2 < ze)
  }
#undef FW_GLOAD_X
#undef FW_LSTORE_X
#undef FW_GLOAD_D
#undef FW_LSTORE_D

  // the four waves' tiles meet in LDS, in wave order (a fixed summation order: deterministic mode relies on it)
  __syncthreads();
  float* redw = reinterpret_cast<float*>(Dsh);              // [9][16 rows][16 cols]
  for (int w = 0; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i_ = t * 256 + (4 * kg + r) * 16 + l15;
          redw[i_] = (w == 0 ? 0.f : redw[i_]) + acc[t][r];
        }
    }
    __syncthreads();
  }
  for (int i = tid; i < 9 * 16 * ncol; i += 256) {
    const int col = i % ncol, ci = (i / ncol) & 15, t9 = i / (ncol * 16);
    const int co = col / 3, dx = col - 3 * co;
    const int tap = t9 * 3 + dx;                            // (dz * 3 + dy) * 3 + dx
    df_acc(dwt, (long long)tap * (16 * k.Cout) + ci * k.Cout + co, redw[t9 * 256 + ci * 16 + col] * osc, k.fx);
  }
  if (k.db) {
    if (tid < 160) bsum[tid] = d_act ? bacc : 0.f;
    __syncthreads();
    if (tid < k.Cout) {
      float s_ = 0.f;
      for (int j = 0; j < 40; ++j) s_ += bsum[tid * 40 + j];
      df_acc(k.db, tid, s_, k.fx);
    }
  }
}
}  // namespace

// Host side (conv3ds.hip::conv3d_split_wgrad_impl calls this for the 16 -> Cout <= 4 flow head).  dwt: tap-major [27][16][Cout].
int df_conv3d_flow_wgrad_ok(const DfConvGeom* g, const float* x, const float* dy) {
  static DfOptFlag off_o{"DFMIR_CONV3D_NO_FLOW_WGRAD"};     // A/B: the swapped-role tiled kernel
  return !off_o.get() && g->Cin == 16 && g->Cout >= 1 && g->Cout <= 4 && (g->Wi & 3) == 0 && g->Di >= 2 &&
         ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0 &&
         (long long)g->Di * g->Hi * g->Wi >= 1024;
}
static inline int CoutMaxWgs(int Cout) { return Cout <= 3 ? 3 : 2; }     // workgroups per CU (LDS)
int df_conv3d_flow_wgrad_launch(const float* x, const float* x_amax, int x_n, const float* dy, const float* dy_amax, int dy_n,
                                float* dwt, float* db, int N, int D, int H, int W, int Cout, hipStream_t st) {
  FwP k{};
  k.N = N; k.D = D; k.H = H; k.W = W; k.Cout = Cout;
  k.x_n = x_n; k.dy_n = dy_n;
  k.ncy = (H + 7) / 8; k.ncx = (W + 31) / 32;
  k.db = db;
  k.fx = df_det_fx();
  // z segments: one round of resident workgroups in all, at least 8 planes each (a segment's prologue stages three dY planes)
  const long long cols = (long long)N * k.ncy * k.ncx;
  long long ns = ((CoutMaxWgs(Cout)) * (long long)df_cu_count() + cols - 1) / cols;
  if (ns > D / 8) ns = D / 8;
  if (ns < 1) ns = 1;
  k.zlen = (int)((D + ns - 1) / ns);
  k.nseg = (D + k.zlen - 1) / k.zlen;
  const long long items = cols * k.nseg;
  if (items >= (1LL << 31)) return 1;
  if (Cout <= 3) conv3d_flow_wgrad_k<3><<<(unsigned)items, 256, 0, st>>>(x, x_amax, dy, dy_amax, dwt, k);
  else conv3d_flow_wgrad_k<4><<<(unsigned)items, 256, 0, st>>>(x, x_amax, dy, dy_amax, dwt, k);
  DF_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfmir_conv3d_s2c2_ok(const DfConvGeom* g) {
  if (!g || tiny_off()) return 0;
  if (!(g->KD == 3 && g->KH == 3 && g->KW == 3 && g->stride == 2 && g->dil == 1 && g->pd == 1 && g->ph == 1 && g->pw == 1 &&
        g->pad_mode == 0 && g->Cin == 2 && g->Cout == 16))
    return 0;
  if (g->Do != (g->Di - 1) / 2 + 1 || g->Ho != (g->Hi - 1) / 2 + 1 || g->Wo != (g->Wi - 1) / 2 + 1) return 0;
  if ((g->Wi & 3) || (g->Wo & 3) || g->Di < 2) return 0;
  return (long long)2 * g->Di * g->Hi * g->Wi < 0x7FFFFFFFLL ? 1 : 0;
}
static C3s2P s2_params(const DfConvGeom* g) {
  C3s2P k{g->N, g->Di, g->Hi, g->Wi, g->Do, g->Ho, g->Wo, (g->Do + S2_TZ - 1) / S2_TZ, (g->Ho + S2_TY - 1) / S2_TY,
          (g->Wo + S2_TX - 1) / S2_TX, g->act, g->slope, 0};
  k.ntile = (long long)g->N * k.nz * k.ny * k.nx;
  return k;
}
// y = act(conv3x3x3 stride 2 (x [N, 2, D, H, W]) + bias) -> [N, 16, Do, Ho, Wo]; w_tcc [27][2][16] (dfmir_weight_pack mode 0);
// y_amax: NULL or the accumulating range-probe slots of y
extern "C" int dfmir_conv3d_s2c2_fwd(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias, float* y,
                                     float* y_amax, void* stream) {
  DF_ARG_CHECK(g && x && w_tcc && y && dfmir_conv3d_s2c2_ok(g));
  DF_ARG_CHECK((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0);
  const C3s2P k = s2_params(g);
  DF_ARG_CHECK(k.ntile < (1LL << 30));
  conv3d_s2c2_fwd_k<<<(unsigned)(8 * ((k.ntile + 7) / 8)), 128, 0, (hipStream_t)stream>>>(x, w_tcc, bias, y, y_amax, k);
  DF_LAUNCH_CHECK();
  return 0;
}
// dw_tcc [27][2][16] += the weight gradient (accumulates, like dfmir_conv_wgrad)
extern "C" int dfmir_conv3d_s2c2_wgrad(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc, void* stream) {
  DF_ARG_CHECK(g && x && dy && dw_tcc && dfmir_conv3d_s2c2_ok(g));
  DF_ARG_CHECK((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0);
  const C3s2P k = s2_params(g);
  DF_ARG_CHECK(k.ntile < (1LL << 30));
  const unsigned grid = (unsigned)(k.ntile < 512 ? k.ntile : 512);
  conv3d_s2c2_wgrad_k<<<grid, 256, 0, (hipStream_t)stream>>>(x, dy, dw_tcc, k, df_det_fx());
  DF_LAUNCH_CHECK();
  return 0;
}

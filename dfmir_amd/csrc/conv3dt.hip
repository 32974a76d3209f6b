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
  static int v = -1;
  if (v < 0) v = getenv("DFMIR_CONV3D_NO_TINY") ? 1 : 0;
  return v == 1;
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

// Tap stack / tap sum: turn the generator's two 7x7 convolutions (1->64 and 64->1, reflect pad 3,
// models/networks.py:982-983,1022-1024) into 1x1 GEMMs on the matrix cores.
//
//   Cin == 1 :  S[tap][p] = x[refl(p + tap - pad)]              (tapstack)   then Y = W[co][tap] * S
//   Cout <= 4:  Z[tap][q] = sum_ci W[ci][tap] X[ci][q]  (1x1 GEMM) then y[p] = sum_tap Z[tap][refl(p + tap - pad)]  (tapsum)
//
// A 3136-long per-pixel dot product (or a 49-wide outer product) has M = 1 or K = 1 in the direct
// implicit-GEMM view and wastes 31/32 of every MFMA; in this form both GEMMs are 49 x 64 per pixel.
// The four kernels here are pure HBM streams (49 planes in or out per pixel); adjoints gather through
// the pre-images of the reflection, so nothing needs atomics.
#include "common.h"

// padded coordinates rp (relative to the frame origin, in [-pad, n+pad)) that land on index i
__device__ __forceinline__ int pre_images(int i, int n, int pad, int pad_mode, int* rp) {
  int cnt = 0;
  rp[cnt++] = i;
  if (pad_mode == 1) {
    if (i >= 1 && i <= pad) rp[cnt++] = -i;
    if (i <= n - 2 && i >= n - 1 - pad) rp[cnt++] = 2 * (n - 1) - i;
  }
  return cnt;
}
__device__ __forceinline__ bool map_coord(int c, int n, int pad_mode, int& out) {
  if (pad_mode == 1) {
    if (c < 0) c = -c;
    if (c >= n) c = 2 * (n - 1) - c;
    out = c < 0 ? 0 : (c >= n ? n - 1 : c);
    return true;
  }
  out = c;
  return (unsigned)c < (unsigned)n;
}

// S[n][c*T + tap][oy][ox] = x[n][c][map(oy + kh - pad)][map(ox + kw - pad)]
__global__ void tapstack_fwd_k(const float* __restrict__ x, float* __restrict__ s, int N, int C, int H, int W,
                               int K, int pad, int pad_mode) {
  const int T = K * K;
  const long long HW = (long long)H * W, total = (long long)N * C * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % W);
    long long r = i / W;
    const int oy = (int)(r % H);
    const long long nc = r / H;
    const float* xp = x + nc * HW;
    float* sp = s + nc * T * HW + (long long)oy * W + ox;
    for (int kh = 0; kh < K; ++kh) {
      int iy;
      const bool vy = map_coord(oy + kh - pad, H, pad_mode, iy);
      for (int kw = 0; kw < K; ++kw) {
        int ix;
        const bool v = map_coord(ox + kw - pad, W, pad_mode, ix) && vy;
        sp[(long long)(kh * K + kw) * HW] = v ? xp[(long long)iy * W + ix] : 0.f;
      }
    }
  }
}
// dx[n][c][iy][ix] = sum_tap sum_{(ry,rx) in pre(iy) x pre(ix)} dS[n][c*T+tap][ry - kh + pad][rx - kw + pad]
__global__ void tapstack_bwd_k(const float* __restrict__ ds, float* __restrict__ dx, int N, int C, int H, int W,
                               int K, int pad, int pad_mode) {
  const int T = K * K;
  const long long HW = (long long)H * W, total = (long long)N * C * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W);
    long long r = i / W;
    const int iy = (int)(r % H);
    const long long nc = r / H;
    const float* sp = ds + nc * T * HW;
    int ry[3], rx[3];
    const int ny = pre_images(iy, H, pad, pad_mode, ry), nx = pre_images(ix, W, pad, pad_mode, rx);
    float acc = 0.f;
    for (int kh = 0; kh < K; ++kh)
      for (int a = 0; a < ny; ++a) {
        const int oy = ry[a] - kh + pad;
        if ((unsigned)oy >= (unsigned)H) continue;
        for (int kw = 0; kw < K; ++kw)
          for (int b = 0; b < nx; ++b) {
            const int ox = rx[b] - kw + pad;
            if ((unsigned)ox < (unsigned)W) acc += sp[(long long)(kh * K + kw) * HW + (long long)oy * W + ox];
          }
      }
    dx[i] = acc;
  }
}

// y[n][c][oy][ox] = act(bias[c] + sum_tap Z[n][c*T+tap][map(oy + kh - opad)][map(ox + kw - opad)])
// Z frame Hz x Wz, output frame Ho x Wo (Ho = Hz for 'same', Hz + 2*opad - (K-1) in general).
__global__ void tapsum_fwd_k(const float* __restrict__ z, const float* __restrict__ bias, float* __restrict__ y,
                             int N, int C, int Hz, int Wz, int Ho, int Wo, int K, int opad, int pad_mode, int act,
                             float slope) {
  const int T = K * K;
  const long long HWz = (long long)Hz * Wz, HWo = (long long)Ho * Wo, total = (long long)N * C * HWo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    long long r = i / Wo;
    const int oy = (int)(r % Ho);
    const long long nc = r / Ho;
    const int c = (int)(nc % C);
    const float* zp = z + nc * T * HWz;
    float acc = bias ? bias[c] : 0.f;
    for (int kh = 0; kh < K; ++kh) {
      int iy;
      if (!map_coord(oy + kh - opad, Hz, pad_mode, iy)) continue;
      for (int kw = 0; kw < K; ++kw) {
        int ix;
        if (map_coord(ox + kw - opad, Wz, pad_mode, ix)) acc += zp[(long long)(kh * K + kw) * HWz + (long long)iy * Wz + ix];
      }
    }
    if (act == 1) acc = acc > 0.f ? acc : acc * slope;
    else if (act == 2) acc = tanhf(acc);
    y[i] = acc;
  }
}
// dZ[n][c*T+tap][iy][ix] = sum_{(ry,rx) in pre(iy) x pre(ix)} dy[n][c][ry - kh + opad][rx - kw + opad]
__global__ void tapsum_bwd_k(const float* __restrict__ dy, float* __restrict__ dz, int N, int C, int Hz, int Wz,
                             int Ho, int Wo, int K, int opad, int pad_mode) {
  const int T = K * K;
  const long long HWz = (long long)Hz * Wz, HWo = (long long)Ho * Wo, total = (long long)N * C * HWz;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % Wz);
    long long r = i / Wz;
    const int iy = (int)(r % Hz);
    const long long nc = r / Hz;
    const float* gp = dy + nc * HWo;
    float* zp = dz + nc * T * HWz + (long long)iy * Wz + ix;
    int ry[3], rx[3];
    const int ny = pre_images(iy, Hz, opad, pad_mode, ry), nx = pre_images(ix, Wz, opad, pad_mode, rx);
    for (int kh = 0; kh < K; ++kh)
      for (int kw = 0; kw < K; ++kw) {
        float acc = 0.f;
        for (int a = 0; a < ny; ++a) {
          const int oy = ry[a] - kh + opad;
          if ((unsigned)oy >= (unsigned)Ho) continue;
          for (int b = 0; b < nx; ++b) {
            const int ox = rx[b] - kw + opad;
            if ((unsigned)ox < (unsigned)Wo) acc += gp[(long long)oy * Wo + ox];
          }
        }
        zp[(long long)(kh * K + kw) * HWz] = acc;
      }
  }
}

// ---------------------------------------------------------------------------------------------
extern "C" int dfmir_tapstack_fwd(const float* x, float* s, int N, int C, int H, int W, int K, int pad,
                                  int pad_mode, void* stream) {
  DF_ARG_CHECK(x && s && N > 0 && C > 0 && H > 0 && W > 0 && K > 0 && (K & 1) && pad >= 0);
  DF_ARG_CHECK(pad_mode == 0 || (pad < H && pad < W));
  tapstack_fwd_k<<<df_grid((long long)N * C * H * W, 256, 16384), 256, 0, (hipStream_t)stream>>>(x, s, N, C, H, W, K,
                                                                                           pad, pad_mode);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_tapstack_bwd(const float* ds, float* dx, int N, int C, int H, int W, int K, int pad,
                                  int pad_mode, void* stream) {
  DF_ARG_CHECK(ds && dx && N > 0 && C > 0 && H > 0 && W > 0 && K > 0 && (K & 1) && pad >= 0);
  DF_ARG_CHECK(pad_mode == 0 || (pad < H && pad < W));
  tapstack_bwd_k<<<df_grid((long long)N * C * H * W, 256, 16384), 256, 0, (hipStream_t)stream>>>(ds, dx, N, C, H, W, K,
                                                                                           pad, pad_mode);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_tapsum_fwd(const float* z, const float* bias, float* y, int N, int C, int Hz, int Wz, int Ho,
                                int Wo, int K, int opad, int pad_mode, int act, float slope, void* stream) {
  DF_ARG_CHECK(z && y && N > 0 && C > 0 && Hz > 0 && Wz > 0 && Ho > 0 && Wo > 0 && K > 0 && opad >= 0);
  DF_ARG_CHECK(pad_mode == 0 || (opad < Hz && opad < Wz && Ho == Hz && Wo == Wz));
  tapsum_fwd_k<<<df_grid((long long)N * C * Ho * Wo, 256, 16384), 256, 0, (hipStream_t)stream>>>(
      z, bias, y, N, C, Hz, Wz, Ho, Wo, K, opad, pad_mode, act, slope);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_tapsum_bwd(const float* dy, float* dz, int N, int C, int Hz, int Wz, int Ho, int Wo, int K,
                                int opad, int pad_mode, void* stream) {
  DF_ARG_CHECK(dy && dz && N > 0 && C > 0 && Hz > 0 && Wz > 0 && Ho > 0 && Wo > 0 && K > 0 && opad >= 0);
  DF_ARG_CHECK(pad_mode == 0 || (opad < Hz && opad < Wz && Ho == Hz && Wo == Wz));
  tapsum_bwd_k<<<df_grid((long long)N * C * Hz * Wz, 256, 16384), 256, 0, (hipStream_t)stream>>>(
      dy, dz, N, C, Hz, Wz, Ho, Wo, K, opad, pad_mode);
  DF_LAUNCH_CHECK();
  return 0;
}


// ---------------------------------------------------------------------------------------------
// Direct forms of the Cin == 1 stem (7x7, 1 -> Cout <= 64): the tap-stack + 1x1 GEMM above moves 49 planes per
// pixel through HBM (411 MB for 32 images at 256^2, against 8 MB of input); here the 49 taps are the MFMA's K and
// the operand is gathered from an LDS patch of the single input channel.
//   forward : Y[co][p] = b[co] + sum_t W[co][t] xp[p + t]       v_mfma_f32_32x32x2_f32, A = W (regs), B = patch gather
//   wgrad   : dW[co][t] += sum_p dY[co][p] xp[p + t],  db[co] += sum_p dY[co][p]     (K = pixels)
// Bound: the 64-plane output (forward) / dY (wgrad) stream and, equally, the fp32 matrix pipe (13 GFLOP per 32 images).
// ---------------------------------------------------------------------------------------------
typedef float st_f32x16 __attribute__((ext_vector_type(16)));
constexpr int S7_K = 7, S7_T = 49, S7_TH = 8, S7_TW = 32, S7_PW = S7_TW + 6, S7_PH = S7_TH + 6, S7_RW = S7_TH / 2;

__device__ __forceinline__ int s7_src(int iy, int ix, int H, int W, int pad_mode) {
  if (pad_mode == 1) {
    if (iy < 0) iy = -iy;
    if (iy >= H) iy = 2 * (H - 1) - iy;
    if (ix < 0) ix = -ix;
    if (ix >= W) ix = 2 * (W - 1) - ix;
    iy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
    ix = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
    return iy * W + ix;
  }
  return ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? iy * W + ix : -1;
}

// tile = 8 rows x 32 px, 256 threads = 4 waves: wave w -> cout block (w & 1), rows 4 (w >> 1) .. +3  (64 accumulator
// registers per wave: several workgroups per CU overlap their patch loads, MFMAs and stores)
__global__ __launch_bounds__(256, 2) void conv7x7_c1_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int H, int W, int Cout, int pad_mode, int tiles_x) {
  __shared__ float patch[S7_PH * S7_PW];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int n = blockIdx.y, ty0 = (blockIdx.x / tiles_x) * S7_TH, tx0 = (blockIdx.x % tiles_x) * S7_TW;
  const float* xn = x + (long long)n * H * W;
  for (int i = tid; i < S7_PH * S7_PW; i += 256) {
    const int r = i / S7_PW, c = i - r * S7_PW;
    const int o = s7_src(ty0 + r - 3, tx0 + c - 3, H, W, pad_mode);
    patch[i] = o >= 0 ? xn[o] : 0.f;
  }
  const int mb = wid & 1, rg = wid >> 1;
  const int co = mb * 32 + l31;
  // A operand: this lane's weights for k-step s (tap 2s + lhi), kept in registers
  float a[25];
#pragma unroll
  for (int s2 = 0; s2 < 25; ++s2) {
    const int t = 2 * s2 + lhi;
    a[s2] = (t < S7_T && co < Cout) ? w[co * S7_T + t] : 0.f;
  }
  __syncthreads();
  st_f32x16 acc[S7_RW];
#pragma unroll
  for (int j = 0; j < S7_RW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < 25; ++s2) {
    const int t = 2 * s2 + lhi < S7_T ? 2 * s2 + lhi : S7_T - 1;     // tap 49 has a zero weight
    const int off = (t / S7_K) * S7_PW + (t % S7_K) + l31;
#pragma unroll
    for (int j = 0; j < S7_RW; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(patch[(rg * S7_RW + j) * S7_PW + off], a[s2], acc[j], 0, 0, 0);   // rows = pixels, columns = couts
  }
  // rows = the 32 pixels of a tile row, columns = output channels: a lane ends with ONE channel (co) and 4 consecutive pixels
  // per accumulator quad -- 16 16-byte stores per lane instead of 64 4-byte ones (W % 4 == 0; otherwise element by element)
  float* yn = y + (long long)n * Cout * H * W;
  const float bz = (bias && co < Cout) ? bias[co] : 0.f;
  const bool vec = (W & 3) == 0 && ((reinterpret_cast<unsigned long long>(y) & 15) == 0);
  if (co < Cout) {
#pragma unroll
    for (int j = 0; j < S7_RW; ++j) {
      const int oy = ty0 + rg * S7_RW + j;
      if (oy >= H) continue;
      float* row = yn + ((long long)co * H + oy) * W;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ox = tx0 + 8 * q + 4 * lhi;
        if (ox >= W) continue;
        const float v0 = acc[j][4 * q] + bz, v1 = acc[j][4 * q + 1] + bz, v2 = acc[j][4 * q + 2] + bz, v3 = acc[j][4 * q + 3] + bz;
        if (vec) *reinterpret_cast<float4*>(row + ox) = make_float4(v0, v1, v2, v3);
        else {
          row[ox] = v0;
          if (ox + 1 < W) row[ox + 1] = v1;
          if (ox + 2 < W) row[ox + 2] = v2;
          if (ox + 3 < W) row[ox + 3] = v3;
        }
      }
    }
  }
}

// persistent workgroups over 8 x 32 pixel tiles; wave w -> (cout block w & 1, tap block w >> 1), one accumulator;
// K = the tile's 256 pixels, two per MFMA.  dY tile and input patch in LDS.
constexpr int S7W_TH = 8, S7W_PH = S7W_TH + 6;
__global__ __launch_bounds__(256) void conv7x7_c1_wgrad_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dw, float* __restrict__ db,
                                                          int N, int H, int W, int Cout, int pad_mode,
                                                          int tiles_x, int tiles_y, const float* __restrict__ fx) {
  __shared__ float patch[S7W_PH * S7_PW];
  __shared__ float dyt[64 * (S7W_TH * S7_TW + 1)];        // [co][px], row stride 257: conflict-free column reads
  constexpr int DS = S7W_TH * S7_TW + 1;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int mb = wid & 1, tb = wid >> 1;
  const int t = tb * 32 + l31;                             // this lane's tap (B operand column)
  const int tc = t < S7_T ? t : S7_T - 1;
  const int toff = (tc / S7_K) * S7_PW + (tc % S7_K);
  st_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // db rides in the padded tap column 49: its B operand is the constant 1, so acc[.][49] = sum_p dY[co][p]
  const long long HW = (long long)H * W;
  const int ntile = N * tiles_y * tiles_x;
  for (int tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
    const int n = tl / (tiles_y * tiles_x), q = tl - n * tiles_y * tiles_x;
    const int ty0 = (q / tiles_x) * S7W_TH, tx0 = (q % tiles_x) * S7_TW;
    const float* xn = x + (long long)n * HW;
    const float* dyn = dy + (long long)n * Cout * HW;
    __syncthreads();
    for (int i = tid; i < S7W_PH * S7_PW; i += 256) {
      const int r = i / S7_PW, c = i - r * S7_PW;
      const int o = s7_src(ty0 + r - 3, tx0 + c - 3, H, W, pad_mode);
      patch[i] = o >= 0 ? xn[o] : 0.f;
    }
    {
      // 64 channels x 256 pixels as 16 float4 per thread, all loads issued before the first LDS store
      float4 ld[16];
#pragma unroll
      for (int q4 = 0; q4 < 16; ++q4) {
        const int i4 = tid + 256 * q4, c = i4 >> 6, r4 = i4 & 63;
        const int oy = ty0 + (r4 >> 3), ox = tx0 + ((r4 & 7) << 2);
        const float* src = dyn + ((long long)c * H + oy) * W + ox;
        if (c < Cout && oy < H && ox + 3 < W && (W & 3) == 0) ld[q4] = *reinterpret_cast<const float4*>(src);
        else {
          ld[q4].x = (c < Cout && oy < H && ox < W) ? src[0] : 0.f;
          ld[q4].y = (c < Cout && oy < H && ox + 1 < W) ? src[1] : 0.f;
          ld[q4].z = (c < Cout && oy < H && ox + 2 < W) ? src[2] : 0.f;
          ld[q4].w = (c < Cout && oy < H && ox + 3 < W) ? src[3] : 0.f;
        }
      }
#pragma unroll
      for (int q4 = 0; q4 < 16; ++q4) {
        const int i4 = tid + 256 * q4, c = i4 >> 6, px = (i4 & 63) << 2;
        float* d = dyt + c * DS + px;
        d[0] = ld[q4].x; d[1] = ld[q4].y; d[2] = ld[q4].z; d[3] = ld[q4].w;
      }
    }
    __syncthreads();
    // 16 pixel pairs per group: the operands of group g+1 are read from LDS while the MFMAs of group g run
    constexpr int G = 16, NG = S7W_TH * S7_TW / (2 * G);
    float av[2][G], bv[2][G];
    const float* arow = dyt + (mb * 32 + l31) * DS + lhi;
#define S7W_LOAD(set_, g_)                                                                       \
    _Pragma("unroll") for (int e = 0; e < G; ++e) {                                              \
      const int px = 2 * ((g_) * G + e) + lhi;                                                   \
      av[set_][e] = arow[2 * ((g_) * G + e)];                                                    \
      bv[set_][e] = t == S7_T ? 1.f : patch[(px >> 5) * S7_PW + (px & 31) + toff];               \
    }
    S7W_LOAD(0, 0)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) S7W_LOAD((g + 1) & 1, g + 1)
#pragma unroll
      for (int e = 0; e < G; ++e)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][e], bv[g & 1][e], acc, 0, 0, 0);
    }
#undef S7W_LOAD
  }
  if (t <= S7_T) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = mb * 32 + 4 * lhi + (r & 3) + 8 * (r >> 2);
      if (c < Cout) {
        if (t < S7_T) df_acc(dw, c * S7_T + t, acc[r], fx);
        else if (db) df_acc(db, c, acc[r], fx);
      }
    }
  }
}

extern "C" int dfmir_conv7x7_c1_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W,
                                    int Cout, int pad_mode, void* stream) {
  DF_ARG_CHECK(x && w && y && N > 0 && N <= 65535 && H >= 4 && W >= 4 && Cout > 0 && Cout <= 64);
  DF_ARG_CHECK(pad_mode == 0 || pad_mode == 1);
  const int tx = (W + S7_TW - 1) / S7_TW, ty = (H + S7_TH - 1) / S7_TH;
  conv7x7_c1_fwd_k<<<dim3((unsigned)(tx * ty), (unsigned)N), 256, 0, (hipStream_t)stream>>>(x, w, bias, y, H, W, Cout,
                                                                                           pad_mode, tx);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_conv7x7_c1_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int H, int W,
                                      int Cout, int pad_mode, void* stream) {
  DF_ARG_CHECK(x && dy && dw && N > 0 && H >= 4 && W >= 4 && Cout > 0 && Cout <= 64);
  DF_ARG_CHECK(pad_mode == 0 || pad_mode == 1);
  const int tx = (W + S7_TW - 1) / S7_TW, ty = (H + S7W_TH - 1) / S7W_TH;
  const long long ntile = (long long)N * tx * ty;
  DF_ARG_CHECK(ntile < (1LL << 31));
  // persistent workgroups, two per CU (68 KB of LDS each); every one ends with 3 200 atomic adds onto the same
  // gradient, so more workgroups than that only add contention (1024: +70 us)
  const unsigned grid = (unsigned)(ntile < 512 ? ntile : 512);
  conv7x7_c1_wgrad_k<<<grid, 256, 0, (hipStream_t)stream>>>(x, dy, dw, db, N, H, W, Cout, pad_mode, tx, ty, df_det_fx());
  DF_LAUNCH_CHECK();
  return 0;
}

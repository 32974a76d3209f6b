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

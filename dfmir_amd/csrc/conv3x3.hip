// 2-D 3x3 stride-1 convolutions -- the shape that carries ~95 % of the generator's FLOPs -- as
// LDS-resident implicit GEMMs on the fp32 matrix cores, software-pipelined.
//
// The generic kernel (conv.hip) re-gathers the input once per filter tap.  Here a workgroup
// stages, per chunk of 8 input channels, (a) ONE halo patch of the input covering its run of
// output pixels (+1 ring) and (b) the weights of all 9 taps; the 9 taps then read the SAME patch
// at shifted LDS addresses.  The global loads of chunk i+1 are issued into registers BEFORE the
// 144-MFMA phase of chunk i and written to LDS after it, so HBM/L2 latency sits under the matrix
// pipe instead of in front of it (rocprof: the un-pipelined form idled the pipe 36 % of the time).
//
// forward / dgrad :  Y[co][p] = sum_{tap,ci} Wt[tap][ci][co] * X[ci][p + tap]       (conv3x3_mfma_k)
// wgrad           :  dWt[tap][ci][co] += sum_p X[ci][p + tap] * dY[co][p]           (conv3x3_wgrad_k)
//   wgrad's MFMA A operand is read straight out of the halo patch: one 32-row tile = 32 input
//   channels of ONE tap, so the (tap, ci) "im2col" axis is never materialised.
#include "conv3x3_common.h"
#include <stdlib.h>

template <int WM, int WN, int TM, int TN, int XP>
__global__ __launch_bounds__(256) void conv3x3_mfma_k(const float* __restrict__ x,
                                                      const float* __restrict__ wt,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ y, Conv3P k) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, CK = 8;
  constexpr int NS = (XP + 255) / 256;
  constexpr int W4 = 9 * CK * BM / 4;  // float4 elements of one weight chunk
  constexpr int NW = (W4 + 255) / 256;
  static_assert(WM * WN == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float Ws[9 * CK * BM];
  __shared__ float Xs[CK * XP];
  __shared__ float bs[BM];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int HWo = k.Ho * k.Wo, HWi = k.Hi * k.Wi;
  const int n = blockIdx.x / k.tiles_per_img;
  const int t = blockIdx.x - n * k.tiles_per_img;
  const int p0 = t * BN;
  const int pend = (p0 + BN < HWo) ? p0 + BN : HWo;
  const int m0 = blockIdx.y * BM;
  const int y0 = p0 / k.Wo, y1 = (pend - 1) / k.Wo;
  const bool single = (y0 == y1);
  const int x0 = p0 - y0 * k.Wo, x1 = (pend - 1) - y1 * k.Wo;
  const int xoff = single ? x0 : 0;
  const int ncols = single ? (x1 - x0 + 3) : (k.Wo + 2);
  const int nrows = y1 - y0 + 3;
  const int npos = nrows * ncols;

  int goff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int pos = tid + 256 * s;
    int off = -1;
    if (pos < npos) {
      const int r = pos / ncols, c = pos - r * ncols;
      off = halo_offset(y0 - k.pad + r, xoff - k.pad + c, k.Hi, k.Wi, k.pad_mode);
    }
    goff[s] = off;
  }
  if (tid < BM) bs[tid] = (bias && (m0 + tid) < k.Cout) ? bias[m0 + tid] : 0.f;

  int pbase[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int q = p0 + (wn * TN + j) * 32 + l31;
    q = q < pend ? q : pend - 1;
    const int yy = q / k.Wo, xx = q - yy * k.Wo;
    pbase[j] = (yy - y0) * ncols + (xx - xoff);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Buffer descriptors: out-of-range offsets (zero padding, channel / tile overrun) read as 0 in
  // hardware, so the staging loads are branch-free and can all be in flight at once.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rx_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x + (long long)n * k.Cin * HWi), 0, (unsigned)(k.Cin * HWi) * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt), 0, (unsigned)(9 * k.Cin * k.Cout) * 4u, 0x00020000);
  const bool vec4 = (k.Cout & 3) == 0;
  unsigned gbyte[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) gbyte[s] = goff[s] < 0 ? OOB : (unsigned)goff[s] * 4u;
  // weight slot -> (row, 4 output channels); byte offset of chunk 0, advanced by CK*Cout*4 per chunk
  unsigned wbyte[NW];
  bool wok[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int idx4 = tid + 256 * j;
    const int row = idx4 / (BM / 4), c4 = idx4 - row * (BM / 4);
    const int tap = row >> 3, ci = row & 7, co = m0 + c4 * 4;
    wok[j] = idx4 < W4 && co < k.Cout;
    wbyte[j] = (unsigned)((tap * k.Cin + ci) * k.Cout + co) * 4u;
  }
  const unsigned wstep = (unsigned)(CK * k.Cout) * 4u;
  const unsigned xstep = (unsigned)HWi * 4u;

  u32x4 rw[NW];
  unsigned rx[NS][CK];

#define C3_GLOAD(ci0_)                                                                           \
  {                                                                                              \
    const unsigned wadd = (unsigned)((ci0_) / CK) * wstep;                                       \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                             \
      /* rows of this chunk with ci >= Cin lie beyond tap's slab only for the LAST tap; mask */  \
      const bool ok = wok[j] && ((ci0_) + ((tid + 256 * j) / (BM / 4) & 7)) < k.Cin;             \
      const unsigned o = ok ? wbyte[j] + wadd : OOB;                                             \
      if (vec4) {                                                                                \
        rw[j] = __builtin_amdgcn_raw_buffer_load_b128(rw_src, o, 0, 0);                          \
      } else {                                                                                   \
        const int co = m0 + ((tid + 256 * j) % (BM / 4)) * 4;                                    \
        rw[j].x = __builtin_amdgcn_raw_buffer_load_b32(rw_src, o, 0, 0);                         \
        rw[j].y = __builtin_amdgcn_raw_buffer_load_b32(rw_src, (ok && co + 1 < k.Cout) ? o + 4u : OOB, 0, 0);  \
        rw[j].z = __builtin_amdgcn_raw_buffer_load_b32(rw_src, (ok && co + 2 < k.Cout) ? o + 8u : OOB, 0, 0);  \
        rw[j].w = __builtin_amdgcn_raw_buffer_load_b32(rw_src, (ok && co + 3 < k.Cout) ? o + 12u : OOB, 0, 0); \
      }                                                                                          \
    }                                                                                            \
    const unsigned xadd = (unsigned)(ci0_) * xstep;                                              \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                             \
      _Pragma("unroll") for (int c = 0; c < CK; ++c)                                             \
        rx[s][c] = __builtin_amdgcn_raw_buffer_load_b32(rx_src, gbyte[s] + xadd + (unsigned)c * xstep, 0, 0); \
    }                                                                                            \
  }
#define C3_LSTORE()                                                                              \
  {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                             \
      const int idx4 = tid + 256 * j;                                                            \
      if (idx4 < W4) *reinterpret_cast<u32x4*>(&Ws[idx4 * 4]) = rw[j];                           \
    }                                                                                            \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                             \
      const int pos = tid + 256 * s;                                                             \
      if (pos < npos) {                                                                          \
        _Pragma("unroll") for (int c = 0; c < CK; ++c) Xs[c * XP + pos] = __uint_as_float(rx[s][c]); \
      }                                                                                          \
    }                                                                                            \
  }

  C3_GLOAD(0);
  C3_LSTORE();
  __syncthreads();

  for (int ci0 = 0; ci0 < k.Cin; ci0 += CK) {
    const bool more = (ci0 + CK) < k.Cin;
    if (more) C3_GLOAD(ci0 + CK);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = (tap / 3) * ncols + (tap % 3);
#pragma unroll
      for (int kk = 0; kk < CK / 2; ++kk) {
        const int kr = 2 * kk + lhi;
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = Ws[(tap * CK + kr) * BM + (wm * TM + i) * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Xs[kr * XP + pbase[j] + toff];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0);   // rows = pixels, columns = couts
      }
    }
    if (more) {
      __syncthreads();  // every wave is done reading this chunk
      C3_LSTORE();
      __syncthreads();
    }
  }
#undef C3_GLOAD
#undef C3_LSTORE

  // ---- epilogue: rows = 32 consecutive (flat) pixels of the run, columns = output channels: a lane ends with one channel
  // and 4 consecutive pixels per accumulator quad -> 16-byte stores (H*W % 4 == 0 and an aligned base; else per element)
  float* yb = y + (long long)n * k.Cout * HWo;
  const bool vec = (HWo & 3) == 0 && (p0 & 3) == 0 && (reinterpret_cast<unsigned long long>(y) & 15) == 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int c = (wm * TM + i) * 32 + l31;
      const int co = m0 + c;
      if (co >= k.Cout) continue;
      const float bv = bs[c];
      float* row = yb + (long long)co * HWo;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int q = p0 + (wn * TN + j) * 32 + 8 * qd + 4 * lhi;
        if (q >= pend) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[i][j][4 * qd + e] + bv;
          if (k.act == 1) t = t > 0.f ? t : t * k.slope;
          else if (k.act == 2) t = tanhf(t);
          v[e] = t;
        }
        if (vec && q + 3 < pend) *reinterpret_cast<float4*>(row + q) = make_float4(v[0], v[1], v[2], v[3]);
        else {
          row[q] = v[0];
          if (q + 1 < pend) row[q + 1] = v[1];
          if (q + 2 < pend) row[q + 2] = v[2];
          if (q + 3 < pend) row[q + 3] = v[3];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient.  Block tile = (32*WI input channels) x (32*WC output channels) x 9 taps; wave
// (wi, wc) owns 9 accumulator tiles [32 ci x 32 co], one per tap.  Reduction over runs of BP = 32
// pixels; LDS double-buffered, next run's loads in flight during the 144-MFMA phase.
// ---------------------------------------------------------------------------------------------
struct W3P {
  int N, Cin, Cout, H, W, pad_mode;
  int runs_per_img, runs_total, runs_per_block;
  const float* fx;        // deterministic mode (common.h df_acc): dwt holds 64-bit fixed-point sums
};

template <int WI, int WC>
__global__ __launch_bounds__(256) void conv3x3_wgrad_k(const float* __restrict__ x,
                                                       const float* __restrict__ dy,
                                                       float* __restrict__ dwt, W3P k) {
  constexpr int BP = 32, XP = 105;            // 105 % 32 == 9: conflict-free channel-strided reads
  constexpr int CT = 32 * WI, BC = 32 * WC, BCP = BC + 1;
  constexpr int HC = CT / 2;                  // channels per thread: 2 threads share one halo position
  constexpr int ND4 = (BP / 4) * BC / 256;    // float4 of dY per thread
  static_assert(WI * WC == 4, "4 waves");
  static_assert(ND4 >= 1, "dY tile");
  __shared__ float Xs[2][CT * XP];
  __shared__ float Ds[2][BP * BCP];
  __shared__ int ppos[2][BP];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wi = wid / WC, wc = wid % WC;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int HW = k.H * k.W;
  const int ci0 = blockIdx.y * CT, co0 = blockIdx.z * BC;
  const int run_beg = blockIdx.x * k.runs_per_block;
  int run_end = run_beg + k.runs_per_block;
  if (run_end > k.runs_total) run_end = k.runs_total;

  // halo position / channel half owned by this thread (geometry is run-invariant up to its origin)
  const int hpos = tid & 127, hhalf = tid >> 7;
  const bool single_row = k.W >= BP;          // host guarantees W % BP == 0 or BP % W == 0
  const int ncols = single_row ? (BP + 2) : (k.W + 2);
  const int nrows = single_row ? 3 : (BP / k.W + 2);
  const int npos = nrows * ncols;
  const int hr = hpos / ncols, hc = hpos - hr * ncols;
  // dY: lanes along pixels
  const int p4 = tid % (BP / 4), dcr = tid / (BP / 4);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  constexpr unsigned OOB = 0x80000000u;
  unsigned rx[HC];
  u32x4 rd[ND4];
  int rpp = 0;
  const unsigned hw4 = (unsigned)HW * 4u;

#define W3_GLOAD(run_)                                                                           \
  {                                                                                              \
    const int n_ = (run_) / k.runs_per_img;                                                      \
    const int p0_ = ((run_) - n_ * k.runs_per_img) * BP;                                         \
    const int y0_ = p0_ / k.W;                                                                   \
    const int xo_ = single_row ? (p0_ - y0_ * k.W) : 0;                                          \
    const __amdgpu_buffer_rsrc_t xs_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(x + (long long)n_ * k.Cin * HW), 0, (unsigned)(k.Cin * HW) * 4u, 0x00020000); \
    const __amdgpu_buffer_rsrc_t ds_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(dy + (long long)n_ * k.Cout * HW), 0, (unsigned)(k.Cout * HW) * 4u, 0x00020000); \
    int off_ = -1;                                                                               \
    if (hpos < npos) off_ = halo_offset(y0_ - 1 + hr, xo_ - 1 + hc, k.H, k.W, k.pad_mode);       \
    const unsigned xb_ = off_ < 0 ? OOB : (unsigned)off_ * 4u + (unsigned)(ci0 + hhalf * HC) * hw4; \
    _Pragma("unroll") for (int c = 0; c < HC; ++c)                                               \
      rx[c] = __builtin_amdgcn_raw_buffer_load_b32(xs_, xb_ + (unsigned)c * hw4, 0, 0);          \
    const unsigned db_ = (unsigned)(co0 + dcr) * hw4 + (unsigned)(p0_ + p4 * 4) * 4u;            \
    _Pragma("unroll") for (int j = 0; j < ND4; ++j)                                              \
      rd[j] = __builtin_amdgcn_raw_buffer_load_b128(ds_, db_ + (unsigned)(j * (256 / (BP / 4))) * hw4, 0, 0); \
    if (tid < BP) {                                                                              \
      const int q = p0_ + tid;                                                                   \
      const int yy = q / k.W, xx = q - yy * k.W;                                                 \
      rpp = (yy - y0_) * ncols + (xx - xo_);                                                     \
    }                                                                                            \
  }
#define W3_LSTORE(buf_)                                                                          \
  {                                                                                              \
    if (hpos < npos) {                                                                           \
      _Pragma("unroll") for (int c = 0; c < HC; ++c) Xs[buf_][(hhalf * HC + c) * XP + hpos] = __uint_as_float(rx[c]); \
    }                                                                                            \
    _Pragma("unroll") for (int j = 0; j < ND4; ++j) {                                            \
      const int c = dcr + j * (256 / (BP / 4));                                                  \
      Ds[buf_][(p4 * 4 + 0) * BCP + c] = __uint_as_float(rd[j].x);                                               \
      Ds[buf_][(p4 * 4 + 1) * BCP + c] = __uint_as_float(rd[j].y);                                               \
      Ds[buf_][(p4 * 4 + 2) * BCP + c] = __uint_as_float(rd[j].z);                                               \
      Ds[buf_][(p4 * 4 + 3) * BCP + c] = __uint_as_float(rd[j].w);                                               \
    }                                                                                            \
    if (tid < BP) ppos[buf_][tid] = rpp;                                                         \
  }

  if (run_beg < run_end) {
    W3_GLOAD(run_beg);
    W3_LSTORE(0);
  }
  __syncthreads();
  int it = 0;
  for (int run = run_beg; run < run_end; ++run, ++it) {
    const int buf = it & 1;
    const bool more = (run + 1) < run_end;
    if (more) W3_GLOAD(run + 1);
    const float* xrow = &Xs[buf][(wi * 32 + l31) * XP];
#pragma unroll 2
    for (int kk = 0; kk < BP / 2; ++kk) {
      const int kr = 2 * kk + lhi;
      const int pp = ppos[buf][kr];
      const float b = Ds[buf][kr * BCP + wc * 32 + l31];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float a = xrow[pp + (t / 3) * ncols + (t % 3)];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      }
    }
    if (more) W3_LSTORE(buf ^ 1);
    __syncthreads();
  }
#undef W3_GLOAD
#undef W3_LSTORE

  const int co = co0 + wc * 32 + l31;
  if (co < k.Cout) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wi * 32 + 4 * lhi + (r & 3) + 8 * (r >> 2);
        if (ci < k.Cin) df_acc(dwt, ((long long)t * k.Cin + ci) * k.Cout + co, acc[t][r], k.fx);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch (called from conv.hip); return true when the launch was taken.
// ---------------------------------------------------------------------------------------------
bool df_conv3x3_fwd_try(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias,
                        float* y, hipStream_t st, int* rc) {
  if (!(g->KD == 1 && g->KH == 3 && g->KW == 3 && g->Di == 1 && g->Do == 1 && g->stride == 1 && g->dil == 1))
    return false;
  if (g->Cout <= 4 || g->ph != g->pw || g->pd != 0) return false;
  const int p = g->ph;
  if (!(p == 1 || (p == 2 && g->pad_mode == 0))) return false;
  if (g->Ho != g->Hi + 2 * p - 2 || g->Wo != g->Wi + 2 * p - 2) return false;
  const long long HWo = (long long)g->Ho * g->Wo;
  if (HWo >= (1LL << 30) || (long long)g->Hi * g->Wi >= (1LL << 30)) return false;
  Conv3P k{g->N, g->Cin, g->Cout, g->Hi, g->Wi, g->Ho, g->Wo, p, g->pad_mode, g->act, g->slope, 0};
  if (g->Cout > 64) {
    if (worst_npos(g->Wo, (int)HWo, 128) > 400) return false;
    k.tiles_per_img = (int)((HWo + 127) / 128);
    dim3 grid((unsigned)(g->N * k.tiles_per_img), (unsigned)((g->Cout + 127) / 128));
    conv3x3_mfma_k<2, 2, 2, 2, 400><<<grid, 256, 0, st>>>(x, w_tcc, bias, y, k);
  } else if (g->Cout > 32) {
    if (worst_npos(g->Wo, (int)HWo, 256) > 1056) return false;
    k.tiles_per_img = (int)((HWo + 255) / 256);
    dim3 grid((unsigned)(g->N * k.tiles_per_img), 1);
    conv3x3_mfma_k<1, 4, 2, 2, 1056><<<grid, 256, 0, st>>>(x, w_tcc, bias, y, k);
  } else {
    if (worst_npos(g->Wo, (int)HWo, 256) > 1056) return false;
    k.tiles_per_img = (int)((HWo + 255) / 256);
    dim3 grid((unsigned)(g->N * k.tiles_per_img), 1);
    conv3x3_mfma_k<1, 4, 1, 2, 1056><<<grid, 256, 0, st>>>(x, w_tcc, bias, y, k);
  }
  hipError_t e = hipGetLastError();
  *rc = (e == hipSuccess) ? 0 : df_set_error((int)e, __FILE__, __LINE__);
  return true;
}

bool df_conv3x3_wgrad_try(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc,
                          hipStream_t st, int* rc) {
  if (!(g->KD == 1 && g->KH == 3 && g->KW == 3 && g->Di == 1 && g->Do == 1 && g->stride == 1 && g->dil == 1))
    return false;
  if (g->ph != 1 || g->pw != 1 || g->pd != 0 || g->Ho != g->Hi || g->Wo != g->Wi) return false;
  if (g->Cout < 64 || g->Cin < 32) return false;
  constexpr int BP = 32;
  const long long HW = (long long)g->Hi * g->Wi;
  if (HW % BP != 0 || HW >= (1LL << 30)) return false;
  const int W = g->Wi;
  if (!((W % BP == 0) || (W < BP && BP % W == 0))) return false;
  const int npos = (W >= BP) ? 3 * (BP + 2) : (BP / W + 2) * (W + 2);
  if (npos > 105) return false;
  W3P k{g->N, g->Cin, g->Cout, g->Hi, g->Wi, g->pad_mode, (int)(HW / BP), 0, 0, df_det_fx()};
  k.runs_total = g->N * k.runs_per_img;
  const bool wide = g->Cout >= 128;
  const int CT = wide ? 32 : 64, BC = wide ? 128 : 64;
  const unsigned ny = (g->Cin + CT - 1) / CT, nz = (g->Cout + BC - 1) / BC;
  long long want = 512 / ((long long)ny * nz);   // one resident round: 2 workgroups per CU
  if (want < 1) want = 1;
  long long maxs = (k.runs_total + 3) / 4;       // >= 4 runs per block
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  k.runs_per_block = (int)((k.runs_total + want - 1) / want);
  const unsigned nx = (k.runs_total + k.runs_per_block - 1) / k.runs_per_block;
  dim3 grid(nx, ny, nz);
  if (wide) conv3x3_wgrad_k<1, 4><<<grid, 256, 0, st>>>(x, dy, dw_tcc, k);
  else conv3x3_wgrad_k<2, 2><<<grid, 256, 0, st>>>(x, dy, dw_tcc, k);
  hipError_t e = hipGetLastError();
  *rc = (e == hipSuccess) ? 0 : df_set_error((int)e, __FILE__, __LINE__);
  return true;
}


// ---------------------------------------------------------------------------------------------
// Weight gradient of the small-channel 3x3 layers (VoxelMorph 2-D U-Net: Cin <= 48, Cout <= 16, stride 1, pad 1).
// The generic implicit-GEMM wgrad gathers every input value once per tap from L2 and runs 32-wide tiles on 2-16
// output channels (770 us for 34 -> 16 @256^2 x 16 images: 210 MB of tensors).  Here: v_mfma_f32_16x16x4_f32 with
// M = (tap, ci) rows, N = output channels, K = pixels; a workgroup owns 8 x 32 pixel tiles (persistent), the input
// patch of ALL channels (10 x 34 per channel) and the dY tile sit in LDS, wave w owns row blocks w, w+8, w+16, w+24.
// ---------------------------------------------------------------------------------------------
typedef float sw_f32x4 __attribute__((ext_vector_type(4)));
constexpr int SW_TH = 8, SW_TW = 32, SW_PW = 36, SW_PP = (SW_TH + 2) * SW_PW, SW_DS = SW_TH * SW_TW + 1;
constexpr int SW_CIN = 48, SW_COUT = 16;   // (a two-column-block build for Cout <= 32 spilled 77 registers: not used)

// the MFMA phase of one tile for a wave with NR row blocks: per group of 4 k-steps all operands are read first
template <int NR, int NCT>
__device__ __forceinline__ void sw_tile(const float* __restrict__ patch, const float* __restrict__ dyt,
                                        const int (&aoff)[4], int l15, int lk, sw_f32x4 (&acc)[4][NCT]) {
  constexpr int KB = NCT == 1 ? 4 : 2;                        // k-steps per batch (register budget)
#pragma unroll 1
  for (int k4 = 0; k4 < SW_TH * SW_TW / (4 * KB); ++k4) {
    float a[KB][NR], b[KB][NCT];
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int px = 4 * KB * k4 + 4 * u + lk;                 // this lane's pixel (K index) of k-step u
      const int poff = (px >> 5) * SW_PW + (px & 31);
#pragma unroll
      for (int c = 0; c < NCT; ++c) b[u][c] = dyt[(c * 16 + l15) * SW_DS + px];
#pragma unroll
      for (int r = 0; r < NR; ++r) a[u][r] = patch[aoff[r] + poff];
    }
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][r], b[u][c], acc[r][c], 0, 0, 0);
  }
}

template <int NCT>   // 16-wide output-channel blocks (1: Cout <= 16, 2: Cout <= 32)
__global__ __launch_bounds__(512, 1) void conv3x3_small_wgrad_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ dwt, int N, int Cin, int Cout,
                                                                int H, int W, int pad_mode, int tiles_x, int tiles_y,
                                                                const float* __restrict__ fx) {
  __shared__ float patch[SW_CIN * SW_PP];
  __shared__ float dyt[16 * NCT * SW_DS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l15 = lane & 15, lk = lane >> 4;
  const int J = 9 * Cin, nrb = (J + 15) >> 4;
  const int nr = wid < nrb ? (nrb - wid + 7) / 8 : 0;          // row blocks of this wave (wave-uniform)
  // A-operand offsets of this lane's row jj = rb*16 + l15 = tap*Cin + ci: patch[ci][ty][tx]
  int aoff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int jj = (wid + 8 * r) * 16 + l15;
    jj = jj < J ? jj : J - 1;
    const int tap = jj / Cin, ci = jj - tap * Cin;
    aoff[r] = ci * SW_PP + (tap / 3) * SW_PW + (tap % 3);
  }
  sw_f32x4 acc[4][NCT];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[r][c][e] = 0.f;
  const long long HW = (long long)H * W;
  const int ntile = N * tiles_y * tiles_x;
  // staging through registers, one tile ahead: threads 0..339 own one patch position each (halo offset decoded once,
  // one load per channel), the other 172 threads the dY tile; the loads of tile t+1 are in flight during the MFMAs of t
  constexpr int NPOS = (SW_TH + 2) * (SW_TW + 2), NDY = 16 * NCT * SW_TH * SW_TW, DYT = 512 - NPOS;
  constexpr int NPRE = (NDY + DYT - 1) / DYT > SW_CIN ? (NDY + DYT - 1) / DYT : SW_CIN;
  float pre[NPRE];
  const int prr = tid / (SW_TW + 2), pc = tid - prr * (SW_TW + 2);
#define SW_FETCH(tl_)                                                                            \
  {                                                                                              \
    const int n_ = (tl_) / (tiles_y * tiles_x), q_ = (tl_) - n_ * tiles_y * tiles_x;             \
    const int ty0_ = (q_ / tiles_x) * SW_TH, tx0_ = (q_ % tiles_x) * SW_TW;                      \
    /* bounds-checked buffer loads: channel stride in the scalar offset (no per-load address registers); outside */ \
    /* the image / past the last channel the load returns 0 */                                   \
    if (tid < NPOS) {                                                                            \
      const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                      \
          const_cast<float*>(x + (long long)n_ * Cin * HW), 0, (unsigned)(Cin * HW) * 4u, 0x00020000); \
      const int o_ = halo_offset(ty0_ + prr - 1, tx0_ + pc - 1, H, W, pad_mode);                 \
      const unsigned vo_ = o_ >= 0 ? (unsigned)o_ * 4u : 0x80000000u;                            \
      _Pragma("unroll") for (int ci = 0; ci < SW_CIN; ++ci)                                      \
        pre[ci] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_, vo_, (unsigned)(ci * (int)HW) * 4u, 0)); \
    } else {                                                                                     \
      const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                      \
          const_cast<float*>(dy + (long long)n_ * Cout * HW), 0, (unsigned)(Cout * HW) * 4u, 0x00020000); \
      _Pragma("unroll") for (int e = 0; e < (NDY + DYT - 1) / DYT; ++e) {                        \
        const int i_ = tid - NPOS + e * DYT;                                                     \
        const int px_ = i_ & (SW_TH * SW_TW - 1), c_ = i_ >> 8;                                  \
        const int oy_ = ty0_ + (px_ >> 5), ox_ = tx0_ + (px_ & 31);                              \
        const bool ok_ = i_ < NDY && c_ < Cout && oy_ < H && ox_ < W;                            \
        pre[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(                           \
            rs_, ok_ ? (unsigned)((c_ * H + oy_) * W + ox_) * 4u : 0x80000000u, 0, 0));          \
      }                                                                                          \
    }                                                                                            \
  }
#define SW_STORE()                                                                               \
  {                                                                                              \
    if (tid < NPOS) {                                                                            \
      float* pd_ = patch + prr * SW_PW + pc;                                                     \
      _Pragma("unroll") for (int ci = 0; ci < SW_CIN; ++ci) if (ci < Cin) pd_[ci * SW_PP] = pre[ci]; \
    } else {                                                                                     \
      _Pragma("unroll") for (int e = 0; e < (NDY + DYT - 1) / DYT; ++e) {                        \
        const int i_ = tid - NPOS + e * DYT;                                                     \
        if (i_ < NDY) dyt[(i_ >> 8) * SW_DS + (i_ & (SW_TH * SW_TW - 1))] = pre[e];              \
      }                                                                                          \
    }                                                                                            \
  }
  if ((int)blockIdx.x < ntile) {
    SW_FETCH((int)blockIdx.x)
    SW_STORE()
  }
  __syncthreads();
  for (int tl = blockIdx.x; tl < ntile; tl += gridDim.x) {
    const bool more = tl + (int)gridDim.x < ntile;
    if (more) SW_FETCH(tl + (int)gridDim.x)
    if (nr == 4) sw_tile<4, NCT>(patch, dyt, aoff, l15, lk, acc);
    else if (nr == 3) sw_tile<3, NCT>(patch, dyt, aoff, l15, lk, acc);
    else if (nr == 2) sw_tile<2, NCT>(patch, dyt, aoff, l15, lk, acc);
    else if (nr == 1) sw_tile<1, NCT>(patch, dyt, aoff, l15, lk, acc);
    __syncthreads();
    if (more) SW_STORE()
    __syncthreads();
  }
#undef SW_FETCH
#undef SW_STORE
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r >= nr) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int jj = (wid + 8 * r) * 16 + lk * 4 + e;        // = tap*Cin + ci: the row index of dwt[tap][ci][co]
      if (jj < J) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const int co = c * 16 + l15;
          if (co < Cout) df_acc(dwt, (long long)jj * Cout + co, acc[r][c][e], fx);
        }
      }
    }
  }
}

bool df_conv3x3_small_wgrad_try(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc, hipStream_t st,
                                int* rc) {
  static DfOptFlag off_o{"DFMIR_NO_SMALL_WGRAD"};
  const bool off = off_o.get();     // A/B switch
  if (off) return false;
  if (!(g->KD == 1 && g->KH == 3 && g->KW == 3 && g->Di == 1 && g->Do == 1 && g->stride == 1 && g->dil == 1))
    return false;
  if (g->ph != 1 || g->pw != 1 || g->pd != 0 || g->Ho != g->Hi || g->Wo != g->Wi) return false;
  if (g->Cin > SW_CIN || g->Cout > SW_COUT || g->Hi < 2 || g->Wi < 2) return false;
  if ((long long)g->Cin * g->Hi * g->Wi * 4 >= (1LL << 31) || (long long)g->Cout * g->Hi * g->Wi * 4 >= (1LL << 31)) return false;
  const int tx = (g->Wi + SW_TW - 1) / SW_TW, ty = (g->Hi + SW_TH - 1) / SW_TH;
  const long long ntile = (long long)g->N * tx * ty;
  if (ntile >= (1LL << 31) || ntile < 32) return false;                  // tiny layers: the generic kernel is as good
  const unsigned grid = (unsigned)(ntile < 256 ? ntile : 256);
  conv3x3_small_wgrad_k<1><<<grid, 512, 0, st>>>(x, dy, dw_tcc, g->N, g->Cin, g->Cout, g->Hi, g->Wi, g->pad_mode, tx, ty, df_det_fx());
  hipError_t e = hipGetLastError();
  *rc = (e == hipSuccess) ? 0 : df_set_error((int)e, __FILE__, __LINE__);
  return true;
}

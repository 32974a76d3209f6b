// Implicit-GEMM convolutions on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// GEMM view (NCHW / NCDHW, "weights-stationary" orientation):
//     Y[co][p] = sum_k  Wt[k][co] * Xg[k][p],   k = (tap, ci),  p = (n, oz, oy, ox)
// MFMA A operand = weights (rows = output channels), B operand = gathered input pixels (cols =
// pixels), so that one accumulator register holds 32 CONSECUTIVE PIXELS of one channel across
// lanes 0..31 -> every store instruction writes two fully-coalesced 128-byte runs of an NCHW plane.
//
// K is walked tap-major: for each filter tap the input offset / padding predicate of a pixel is
// computed ONCE and reused for a chunk of `kc` (<=16, even) input channels, which are Cin/nchunks
// rounded to even so that odd channel counts (34, 48, 2 ...) waste < 6 % of the MFMA issue slots.
// Tiles go global -> registers -> LDS (double buffered, one barrier per K step); fp32 MFMA is
// 64 cycles per 32x32x2 so the per-element gather arithmetic hides under the matrix pipe.
#include "conv3x3_common.h"
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvK {
  DfConvGeom g;
  int kc, nchunks;
  // DIL2 form (dgrad of a stride-2 conv = conv over the zero-dilated dY): output voxels are enumerated parity class
  // by parity class (class-major: p = (((cls N + n) Dm + mz) Hm + my) Wm + mx, o = 2 m + parity), so that the voxels
  // of one workgroup share the taps that meet non-zero input -- (1 or 2) per axis instead of 3
  int Dm, Hm, Wm, ncx, ncy, ncz;       // sub-grid extents and parity classes per axis (1 or 2)
};

static inline void df_chunking(int Cin, int& kc, int& nchunks) {
  nchunks = (Cin + 15) / 16;
  int per = (Cin + nchunks - 1) / nchunks;
  kc = (per + 1) & ~1;
  if (kc < 2) kc = 2;
}

// ---------------------------------------------------------------------------------------------
// forward / dgrad kernel
// ---------------------------------------------------------------------------------------------
// pixel index -> output voxel; DIL2: class-major enumeration (phantom voxels of odd extents are invalid)
template <bool DIL2>
__device__ __forceinline__ bool conv_pixel(const ConvK& k, long long pp, long long P, int& n, int& oz, int& oy, int& ox,
                                           int& cls) {
  const DfConvGeom& g = k.g;
  cls = 0;
  if (pp >= P) { n = oz = oy = ox = 0; return false; }
  long long t = pp;
  if constexpr (DIL2) {
    const int mx = (int)(t % k.Wm); t /= k.Wm;
    const int my = (int)(t % k.Hm); t /= k.Hm;
    const int mz = (int)(t % k.Dm); t /= k.Dm;
    n = (int)(t % g.N);
    cls = (int)(t / g.N);
    const int px = cls % k.ncx, py = (cls / k.ncx) % k.ncy, pz = cls / (k.ncx * k.ncy);
    ox = k.ncx * mx + px; oy = k.ncy * my + py; oz = k.ncz * mz + pz;
    return ox < g.Wo && oy < g.Ho && oz < g.Do;
  } else {
    ox = (int)(t % g.Wo); t /= g.Wo;
    oy = (int)(t % g.Ho); t /= g.Ho;
    oz = (int)(t % g.Do); n = (int)(t / g.Do);
    return true;
  }
}

template <int WM, int WN, int TM, int TN, bool DIL2 = false>
__global__ __launch_bounds__(256) void conv_mfma_k(const float* __restrict__ x,
                                                   const float* __restrict__ wt,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ y, ConvK k) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 16;
  constexpr int NA = BK * BM / 256, NB = BK * BN / 256;
  constexpr int RSB = 256 / BN;            // B rows covered per pass (BN<=256)
  constexpr int RSA = 256 / BM;            // A rows covered per pass (BM<=256)
  static_assert(WM * WN == 4, "4 waves");
  static_assert(BN <= 256 && BM <= 256, "tile");
  __shared__ float As[2][BK][BM];
  __shared__ float Bs[2][BK][BN];
  __shared__ int taplist[DIL2 ? 128 : 1];
  __shared__ int ntap_s;

  const DfConvGeom& g = k.g;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const long long DHWi = (long long)g.Di * g.Hi * g.Wi;
  const long long DHWo = (long long)g.Do * g.Ho * g.Wo;
  const long long P = DIL2 ? (long long)k.ncx * k.ncy * k.ncz * g.N * k.Dm * k.Hm * k.Wm : (long long)g.N * DHWo;
  const long long p0 = (long long)blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;

  // ---- the pixel this thread gathers for the B tile
  const int bcol = tid % BN, brow0 = tid / BN;
  const long long pp = p0 + bcol;
  int ox, oy, oz, n, cls;
  const bool pv = conv_pixel<DIL2>(k, pp, P, n, oz, oy, ox, cls);
  if constexpr (DIL2) {
    // taps that can meet a non-zero of the dilated input: per axis (parity - pad + t) even.  A workgroup whose voxels
    // span two classes (only at class boundaries) walks every tap.
    if (tid == 0) {
      const long long per_cls = (long long)g.N * k.Dm * k.Hm * k.Wm;
      const long long pl = (p0 + BN - 1 < P ? p0 + BN - 1 : P - 1);
      const int c0 = (int)(p0 / per_cls), c1 = (int)(pl / per_cls);
      const int px = c0 % k.ncx, py = (c0 / k.ncx) % k.ncy, pz = c0 / (k.ncx * k.ncy);
      int nt = 0;
      for (int t = 0; t < g.KD * g.KH * g.KW && nt < 128; ++t) {
        const int kw_ = t % g.KW, kh_ = (t / g.KW) % g.KH, kd_ = t / (g.KW * g.KH);
        const bool ok = c0 != c1 || ((k.ncx == 1 || ((px - g.pw + kw_) & 1) == 0) && (k.ncy == 1 || ((py - g.ph + kh_) & 1) == 0) &&
                                     (k.ncz == 1 || ((pz - g.pd + kd_) & 1) == 0));
        if (ok) taplist[nt++] = t;
      }
      ntap_s = nt;
    }
    __syncthreads();
  }
  // hardware-bounds-checked buffer loads: padding / tile overrun = an out-of-range offset = 0.0,
  // so the gather is branch-free (host guarantees every tensor here is < 2 GiB)
  constexpr unsigned OOB = 0x80000000u;
  const unsigned dhw4 = (unsigned)DHWi * 4u;
  const unsigned nbase = (unsigned)n * (unsigned)g.Cin * dhw4;
  const __amdgpu_buffer_rsrc_t x_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x), 0, (unsigned)((long long)g.N * g.Cin * DHWi * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt), 0, (unsigned)((long long)g.KD * g.KH * g.KW * g.Cin * g.Cout * 4), 0x00020000);
  const int acol = tid % BM, arow0 = tid / BM;
  const int aco = m0 + acol;
  const bool av = aco < g.Cout;

  float ra[NA], rb[NB];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nks = (DIL2 ? ntap_s : g.KD * g.KH * g.KW) * k.nchunks;

#define CONV_GLOAD(ks_)                                                                          \
  {                                                                                              \
    const int tq_ = (ks_) / k.nchunks;                                                           \
    const int tap = DIL2 ? taplist[tq_] : tq_;                                                   \
    const int ci0 = ((ks_) - tq_ * k.nchunks) * k.kc;                                            \
    const int kw_ = tap % g.KW;                                                                  \
    const int t2_ = tap / g.KW;                                                                  \
    const int kh_ = t2_ % g.KH;                                                                  \
    const int kd_ = t2_ / g.KH;                                                                  \
    int iz, iy, ix;                                                                              \
    bool v = pv;                                                                                 \
    v &= df_in_coord(oz, kd_, g.stride, g.pd, g.dil, g.Di, g.pad_mode, iz);                      \
    v &= df_in_coord(oy, kh_, g.stride, g.ph, g.dil, g.Hi, g.pad_mode, iy);                      \
    v &= df_in_coord(ox, kw_, g.stride, g.pw, g.dil, g.Wi, g.pad_mode, ix);                      \
    const unsigned sp = v ? nbase + (unsigned)(((long long)iz * g.Hi + iy) * g.Wi + ix) * 4u : OOB; \
    _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                             \
      const int row = brow0 + RSB * j;                                                           \
      const int ci = ci0 + row;                                                                  \
      const unsigned o = (row < k.kc && ci < g.Cin) ? sp + (unsigned)ci * dhw4 : OOB;            \
      rb[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_src, o, 0, 0));             \
    }                                                                                            \
    _Pragma("unroll") for (int j = 0; j < NA; ++j) {                                             \
      const int row = arow0 + RSA * j;                                                           \
      const int ci = ci0 + row;                                                                  \
      const unsigned o = (av && row < k.kc && ci < g.Cin)                                        \
                             ? (unsigned)((tap * g.Cin + ci) * g.Cout + aco) * 4u : OOB;         \
      ra[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(w_src, o, 0, 0));             \
    }                                                                                            \
  }
#define CONV_LSTORE(buf_)                                                                        \
  {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NB; ++j) Bs[buf_][brow0 + RSB * j][bcol] = rb[j];      \
    _Pragma("unroll") for (int j = 0; j < NA; ++j) As[buf_][arow0 + RSA * j][acol] = ra[j];      \
  }

  CONV_GLOAD(0);
  CONV_LSTORE(0);
  __syncthreads();

  const int kpairs = k.kc >> 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  for (int ks = 0; ks < nks; ++ks) {
    const int buf = ks & 1;
    const bool more = (ks + 1) < nks;
    if (more) CONV_GLOAD(ks + 1);
#pragma unroll 2
    for (int kk = 0; kk < kpairs; ++kk) {
      const int kr = 2 * kk + lhi;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][kr][(wm * TM + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[buf][kr][(wn * TN + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) CONV_LSTORE(buf ^ 1);
    __syncthreads();
  }
#undef CONV_GLOAD
#undef CONV_LSTORE

  // ---- epilogue: D[i = co][j = pixel]; lane holds pixel (lane&31), rows (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long long pq = p0 + (wn * TN + j) * 32 + l31;
    long long nn, so;
    if constexpr (DIL2) {
      int qn, qz, qy, qx, qc;
      if (!conv_pixel<true>(k, pq, P, qn, qz, qy, qx, qc)) continue;
      nn = qn;
      so = ((long long)qz * g.Ho + qy) * g.Wo + qx;
    } else {
      if (pq >= P) continue;
      nn = pq / DHWo;
      so = pq - nn * DHWo;
    }
    float* yb = y + nn * g.Cout * DHWo + so;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int cb = m0 + (wm * TM + i) * 32 + 4 * lhi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cb + (r & 3) + 8 * (r >> 2);
        if (co < g.Cout) {
          float v = acc[i][j][r];
          if (bias) v += bias[co];
          if (g.act == 1) v = v > 0.f ? v : v * g.slope;
          else if (g.act == 2) v = tanhf(v);
          yb[(long long)co * DHWo] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// small-Cout direct kernel (Cout <= 4): 7x7 64->1 (+tanh), flow convs 16->2/3, dgrad into 1-3 ch.
// One thread per output voxel, lanes along x -> coalesced input reads; weights are wave-uniform.
// ---------------------------------------------------------------------------------------------
template <int CO>
__global__ __launch_bounds__(256) void conv_small_k(const float* __restrict__ x,
                                                    const float* __restrict__ wt,
                                                    const float* __restrict__ bias,
                                                    float* __restrict__ y, DfConvGeom g) {
  const long long DHWi = (long long)g.Di * g.Hi * g.Wi;
  const long long DHWo = (long long)g.Do * g.Ho * g.Wo;
  const long long P = (long long)g.N * DHWo;
  const long long pp = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pp >= P) return;
  long long t = pp;
  const int ox = (int)(t % g.Wo); t /= g.Wo;
  const int oy = (int)(t % g.Ho); t /= g.Ho;
  const int oz = (int)(t % g.Do);
  const int n = (int)(t / g.Do);
  const float* xin = x + (long long)n * g.Cin * DHWi;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  int tap = 0;
  for (int kd = 0; kd < g.KD; ++kd) {
    int iz;
    const bool vz = df_in_coord(oz, kd, g.stride, g.pd, g.dil, g.Di, g.pad_mode, iz);
    for (int kh = 0; kh < g.KH; ++kh) {
      int iy;
      const bool vy = df_in_coord(oy, kh, g.stride, g.ph, g.dil, g.Hi, g.pad_mode, iy);
      for (int kw = 0; kw < g.KW; ++kw, ++tap) {
        int ix;
        const bool v = df_in_coord(ox, kw, g.stride, g.pw, g.dil, g.Wi, g.pad_mode, ix) && vy && vz;
        if (v) {
          const float* xp = xin + ((long long)iz * g.Hi + iy) * g.Wi + ix;
          const float* wp = wt + (long long)tap * g.Cin * g.Cout;
          for (int ci = 0; ci < g.Cin; ++ci) {
            const float xv = xp[(long long)ci * DHWi];
#pragma unroll
            for (int c = 0; c < CO; ++c)
              if (c < g.Cout) acc[c] = fmaf(xv, wp[ci * g.Cout + c], acc[c]);
          }
        }
      }
    }
  }
  const long long so = pp - (long long)n * DHWo;
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    if (c < g.Cout) {
      float v = acc[c] + (bias ? bias[c] : 0.f);
      if (g.act == 1) v = v > 0.f ? v : v * g.slope;
      else if (g.act == 2) v = tanhf(v);
      y[((long long)n * g.Cout + c) * DHWo + so] = v;
    }
  }
}

// Few output voxels (the deepest U-Net levels of the 6-level 3-D plugin network: 8^3, 4^3 and 2^3 volumes, forward and the
// dgrads of the stride-2 convs): ONE WORKGROUP PER OUTPUT VOXEL, thread = (output channel co, tap share kp): the taps are
// dealt round-robin to the 256 / COP shares, weights [tap][ci][co] are read coalesced along co, the input value is a
// broadcast, fp32 FMAs; the shares meet in LDS.  conv_mfma_k walks K = taps x Cin = 864 .. 1728 serially in one or two
// workgroups for these shapes (70 us per launch, 0.62 of the 3.6 ms step at 128^3); a thread per output element still walks it
// serially behind its load latencies (49 us); here a thread sees 3-4 taps (8 us).
// (Eight voxels per workgroup for 16^3, sharing each weight value, measured slower than conv_mfma_k there: not kept.)
template <int COP>
__global__ __launch_bounds__(256) void conv_tinyvol_k(const float* __restrict__ x, const float* __restrict__ wt,
                                                      const float* __restrict__ bias, float* __restrict__ y, DfConvGeom g) {
  constexpr int KP = 256 / COP;
  __shared__ float sm[KP][COP];
  const long long DHWi = (long long)g.Di * g.Hi * g.Wi;
  const long long DHWo = (long long)g.Do * g.Ho * g.Wo;
  const int co = threadIdx.x % COP, kp = threadIdx.x / COP;
  long long t = blockIdx.x;
  const long long pp = t;
  const int ox = (int)(t % g.Wo); t /= g.Wo;
  const int oy = (int)(t % g.Ho); t /= g.Ho;
  const int oz = (int)(t % g.Do);
  const int n = (int)(t / g.Do);
  const float* xin = x + (long long)n * g.Cin * DHWi;
  const int T = g.KD * g.KH * g.KW;
  float acc0 = 0.f, acc1 = 0.f;
  if (co < g.Cout) {
    for (int tap = kp; tap < T; tap += KP) {
      const int kw = tap % g.KW, kh = (tap / g.KW) % g.KH, kd = tap / (g.KW * g.KH);
      int iz, iy, ix;
      const bool v = df_in_coord(oz, kd, g.stride, g.pd, g.dil, g.Di, g.pad_mode, iz) &
                     df_in_coord(oy, kh, g.stride, g.ph, g.dil, g.Hi, g.pad_mode, iy) &
                     df_in_coord(ox, kw, g.stride, g.pw, g.dil, g.Wi, g.pad_mode, ix);
      if (!v) continue;
      const float* xp = xin + ((long long)iz * g.Hi + iy) * g.Wi + ix;
      const float* wp = wt + (long long)tap * g.Cin * g.Cout + co;
      int ci = 0;
      for (; ci + 8 <= g.Cin; ci += 8) {
        float xv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { xv[u] = xp[(long long)(ci + u) * DHWi]; wv[u] = wp[(long long)(ci + u) * g.Cout]; }
#pragma unroll
        for (int u = 0; u < 8; u += 2) { acc0 = fmaf(xv[u], wv[u], acc0); acc1 = fmaf(xv[u + 1], wv[u + 1], acc1); }
      }
      for (; ci < g.Cin; ++ci) acc0 = fmaf(xp[(long long)ci * DHWi], wp[(long long)ci * g.Cout], acc0);
    }
  }
  sm[kp][co] = acc0 + acc1;
  __syncthreads();
  if (kp == 0 && co < g.Cout) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < KP; ++j) v += sm[j][co];
    v += bias ? bias[co] : 0.f;
    if (g.act == 1) v = v > 0.f ? v : v * g.slope;
    else if (g.act == 2) v = tanhf(v);
    y[((long long)n * g.Cout + co) * DHWo + (pp - (long long)n * DHWo)] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient:  dWt[j][co] += sum_p Xg[j][p] * dY[co][p],  j = tap*Cin + ci
// MFMA A operand = gathered input (rows = j), B operand = dY (cols = co): lanes 0..31 of one
// accumulator register hold 32 consecutive `co` of one j -> coalesced atomics into [tap][ci][co].
// The pixel reduction is split over blockIdx.x; partial tiles are combined with fp32 atomics.
// ---------------------------------------------------------------------------------------------
template <int WJ, int WC, int TJ, int TC, int BP>
__global__ __launch_bounds__(256) void conv_wgrad_mfma_k(const float* __restrict__ x,
                                                         const float* __restrict__ dy,
                                                         float* __restrict__ dwt, DfConvGeom g,
                                                         long long pchunk, const float* __restrict__ fx) {
  constexpr int BJ = WJ * TJ * 32, BC = WC * TC * 32;
  constexpr int RS = 256 / BP;
  constexpr int NA = BJ / RS, NB = (BC + RS - 1) / RS;
  static_assert(WJ * WC == 4, "4 waves");
  __shared__ float As[2][BP][BJ + 1];
  __shared__ float Bs[2][BP][BC + 1];
  __shared__ int jinfo[BJ];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wj = wid / WC, wc = wid % WC;
  const long long DHWi = (long long)g.Di * g.Hi * g.Wi;
  const long long DHWo = (long long)g.Do * g.Ho * g.Wo;
  const long long P = (long long)g.N * DHWo;
  const int J = g.KD * g.KH * g.KW * g.Cin;
  const int j0 = blockIdx.y * BJ, c0 = blockIdx.z * BC;
  const long long pbeg = (long long)blockIdx.x * pchunk;
  long long pend = pbeg + pchunk;
  if (pend > P) pend = P;

  for (int i = tid; i < BJ; i += 256) {
    const int j = j0 + i;
    int info = -1;
    if (j < J) {
      const int tap = j / g.Cin, ci = j - tap * g.Cin;
      const int kw = tap % g.KW, t2 = tap / g.KW, kh = t2 % g.KH, kd = t2 / g.KH;
      info = (ci << 12) | (kd << 8) | (kh << 4) | kw;
    }
    jinfo[i] = info;
  }
  __syncthreads();

  constexpr unsigned OOB = 0x80000000u;
  const unsigned dhwi4 = (unsigned)DHWi * 4u;
  const __amdgpu_buffer_rsrc_t x_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x), 0, (unsigned)((long long)g.N * g.Cin * DHWi * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t d_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dy), 0, (unsigned)((long long)g.N * g.Cout * DHWo * 4), 0x00020000);
  const int pl = tid % BP, r0 = tid / BP;
  float ra[NA], rb[NB];
  f32x16 acc[TJ][TC];
#pragma unroll
  for (int i = 0; i < TJ; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#define WG_GLOAD(pb_)                                                                            \
  {                                                                                              \
    const long long pq = (pb_) + pl;                                                             \
    const bool pv = pq < pend;                                                                   \
    int ox = 0, oy = 0, oz = 0, n = 0;                                                           \
    if (pv) {                                                                                    \
      long long t = pq;                                                                          \
      ox = (int)(t % g.Wo); t /= g.Wo;                                                           \
      oy = (int)(t % g.Ho); t /= g.Ho;                                                           \
      oz = (int)(t % g.Do); n = (int)(t / g.Do);                                                 \
    }                                                                                            \
    const unsigned xb = (unsigned)n * (unsigned)g.Cin * dhwi4;                                   \
    _Pragma("unroll") for (int jj = 0; jj < NA; ++jj) {                                          \
      const int info = jinfo[r0 + RS * jj];                                                      \
      const int ci = info >> 12, kd = (info >> 8) & 15, kh = (info >> 4) & 15, kw = info & 15;   \
      int iz, iy, ix;                                                                            \
      bool v = pv && info >= 0;                                                                  \
      v &= df_in_coord(oz, kd, g.stride, g.pd, 1, g.Di, g.pad_mode, iz);                         \
      v &= df_in_coord(oy, kh, g.stride, g.ph, 1, g.Hi, g.pad_mode, iy);                         \
      v &= df_in_coord(ox, kw, g.stride, g.pw, 1, g.Wi, g.pad_mode, ix);                         \
      const unsigned o = v ? xb + (unsigned)ci * dhwi4 +                                         \
                                 (unsigned)(((long long)iz * g.Hi + iy) * g.Wi + ix) * 4u : OOB; \
      ra[jj] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_src, o, 0, 0));            \
    }                                                                                            \
    const long long so = pq - (long long)n * DHWo;                                               \
    _Pragma("unroll") for (int jj = 0; jj < NB; ++jj) {                                          \
      const int cr = r0 + RS * jj;                                                               \
      const int co = c0 + cr;                                                                    \
      const unsigned o = (pv && cr < BC && co < g.Cout)                                          \
                             ? (unsigned)(((long long)n * g.Cout + co) * DHWo + so) * 4u : OOB;  \
      rb[jj] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(d_src, o, 0, 0));            \
    }                                                                                            \
  }
#define WG_LSTORE(buf_)                                                                          \
  {                                                                                              \
    _Pragma("unroll") for (int jj = 0; jj < NA; ++jj) As[buf_][pl][r0 + RS * jj] = ra[jj];       \
    _Pragma("unroll") for (int jj = 0; jj < NB; ++jj) {                                          \
      const int cr = r0 + RS * jj;                                                               \
      if (cr < BC) Bs[buf_][pl][cr] = rb[jj];                                                    \
    }                                                                                            \
  }

  if (pbeg < pend) {
    WG_GLOAD(pbeg);
    WG_LSTORE(0);
  }
  __syncthreads();
  const int l31 = lane & 31, lhi = lane >> 5;
  int it = 0;
  for (long long pb = pbeg; pb < pend; pb += BP, ++it) {
    const int buf = it & 1;
    const bool more = (pb + BP) < pend;
    if (more) WG_GLOAD(pb + BP);
#pragma unroll 2
    for (int kk = 0; kk < BP / 2; ++kk) {
      const int kr = 2 * kk + lhi;
      float a[TJ], b[TC];
#pragma unroll
      for (int i = 0; i < TJ; ++i) a[i] = As[buf][kr][(wj * TJ + i) * 32 + l31];
#pragma unroll
      for (int j = 0; j < TC; ++j) b[j] = Bs[buf][kr][(wc * TC + j) * 32 + l31];
#pragma unroll
      for (int i = 0; i < TJ; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) WG_LSTORE(buf ^ 1);
    __syncthreads();
  }
#undef WG_GLOAD
#undef WG_LSTORE

#pragma unroll
  for (int j = 0; j < TC; ++j) {
    const int co = c0 + (wc * TC + j) * 32 + l31;
    if (co >= g.Cout) continue;
#pragma unroll
    for (int i = 0; i < TJ; ++i) {
      const int jb = j0 + (wj * TJ + i) * 32 + 4 * lhi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jr = jb + (r & 3) + 8 * (r >> 2);
        if (jr < J) df_acc(dwt, (long long)jr * g.Cout + co, acc[i][j][r], fx);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// 1x1(x1) convolutions' weight gradient as a streaming NT GEMM (round 6): dW[ci][co] += sum_{n,q} X[n][ci][q] dY[n][co][q].
// No taps, no padding, no gather: both operands are read as 16-byte quads of consecutive voxels (the generic kernel above
// decodes a coordinate and issues a 4-byte load per ELEMENT: 20 TFLOP/s on the 64 -> 49 tap GEMM of the generator's 7x7
// head, 0.66 ms for 13 GFLOP over 0.95 GB).  Workgroup = 64 ci x 64 co (4 waves = 2 x 2 tiles of 32 x 32 on
// v_mfma_f32_32x32x2_f32), K = the voxels of its share of 64-voxel chunks, staged channel-major into LDS (row stride 65:
// a wave's 32 channel rows of one voxel sit in 32 banks) one chunk ahead through registers; split-K over blockIdx.x, the
// partial tiles meet in df_acc (fp32 atomics, or 64-bit fixed point in deterministic mode).
// (The FORWARD / data-gradient counterpart was built too -- persistent workgroups, weights resident, 128-voxel tiles -- and
// measured 1.5x SLOWER than conv_mfma_k on the same GEMMs (0.47 vs 0.30 ms at n = 32, 24 % matrix-pipe busy, 2 TB/s:
// profiles/r06_ab_1x1_fwd.txt); it is not in the tree.)
//   users: the head's tap GEMM (models/networks.py:1022-1023 as tap-sum, csrc/taps.hip) and the stem's input gradient,
//   PatchSampleF's two Linear layers (models/networks.py:587-595).   Requires S % 4 == 0 and 16-byte aligned operands.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1x1_wgrad_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dwt, int N, int Cin, int Cout, long long S,
                                                       long long chunks_per_img, long long chunks_per_block,
                                                       const float* __restrict__ fx) {
  constexpr int BK = 64, LD = BK + 1;
  __shared__ float Xs[64 * LD];
  __shared__ float Ds[64 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5, wm = wid >> 1, wn = wid & 1;
  const int ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
  const long long total = (long long)N * chunks_per_img;
  const long long cbeg = (long long)blockIdx.x * chunks_per_block;
  long long cend = cbeg + chunks_per_block;
  if (cend > total) cend = total;
  // staging: thread -> (row r of the 64-channel tile, 16 voxels of the chunk) = four quads
  const int r = tid >> 2, q16 = (tid & 3) * 16;
  const bool xrow = ci0 + r < Cin, drow = co0 + r < Cout;
  float4 rx[4], rd[4];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#define G1_GLOAD(c_)                                                                                         \
  {                                                                                                          \
    const long long n_ = (c_) / chunks_per_img, q0_ = ((c_) - n_ * chunks_per_img) * BK + q16;               \
    const float* xp_ = x + ((long long)n_ * Cin + ci0 + r) * S + q0_;                                        \
    const float* dp_ = dy + ((long long)n_ * Cout + co0 + r) * S + q0_;                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
      const bool in_ = q0_ + 4 * j < S;                                                                      \
      rx[j] = (xrow && in_) ? *reinterpret_cast<const float4*>(xp_ + 4 * j) : z4;                            \
      rd[j] = (drow && in_) ? *reinterpret_cast<const float4*>(dp_ + 4 * j) : z4;                            \
    }                                                                                                        \
  }
#define G1_LSTORE()                                                                                          \
  {                                                                                                          \
    float* xs_ = &Xs[r * LD + q16];                                                                          \
    float* ds_ = &Ds[r * LD + q16];                                                                          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
      xs_[4 * j] = rx[j].x; xs_[4 * j + 1] = rx[j].y; xs_[4 * j + 2] = rx[j].z; xs_[4 * j + 3] = rx[j].w;    \
      ds_[4 * j] = rd[j].x; ds_[4 * j + 1] = rd[j].y; ds_[4 * j + 2] = rd[j].z; ds_[4 * j + 3] = rd[j].w;    \
    }                                                                                                        \
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (cbeg < cend) G1_GLOAD(cbeg)
  for (long long c = cbeg; c < cend; ++c) {
    __syncthreads();                                        // the previous chunk's readers are done
    G1_LSTORE()
    __syncthreads();
    if (c + 1 < cend) G1_GLOAD(c + 1)
    const float* ap = &Xs[(wm * 32 + l31) * LD + lhi];
    const float* bp = &Ds[(wn * 32 + l31) * LD + lhi];
#pragma unroll 8
    for (int ks = 0; ks < BK / 2; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc, 0, 0, 0);
  }
#undef G1_GLOAD
#undef G1_LSTORE
  const int co = co0 + wn * 32 + l31;
  if (co < Cout && cbeg < cend) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ci = ci0 + wm * 32 + 4 * lhi + (i & 3) + 8 * (i >> 2);
      if (ci < Cin) df_acc(dwt, (long long)ci * Cout + co, acc[i], fx);
    }
  }
}

// ---------------------------------------------------------------------------------------------
__global__ void bias_grad_k(const float* __restrict__ dy, float* __restrict__ db, int N, int C,
                            long long S, int nsplit, const float* __restrict__ fx) {
  // one (n, c) plane segment per workgroup: contiguous float4 stream, one atomic per workgroup
  __shared__ float sm[17];
  const int c = blockIdx.x, n = blockIdx.y, part = blockIdx.z;
  const long long per = (((S + nsplit - 1) / nsplit) + 3) & ~3LL;
  const long long beg = part * per;
  long long end = beg + per;
  if (end > S) end = S;
  const float* p = dy + ((long long)n * C + c) * S;
  float s = 0.f;
  if ((S & 3) == 0) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (long long i = (beg >> 2) + threadIdx.x; i < (end >> 2); i += blockDim.x) {
      const float4 v = p4[i];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (long long i = beg + threadIdx.x; i < end; i += blockDim.x) s += p[i];
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0 && beg < S) df_acc(db, c, s, fx);
}

// part (may be NULL): part[blockIdx.x] <- this block's max |w| -- the range probe of the fp16x2 split, taken
// while the weights pass through anyway (weight_split_k reduces the gridDim.x partials)
__global__ __launch_bounds__(256) void weight_pack_k(const float* __restrict__ w, float* __restrict__ o, int Cout,
                                                     int Cin, int T, int mode, float* __restrict__ part) {
  __shared__ unsigned smax;
  if (threadIdx.x == 0) smax = 0u;
  __syncthreads();
  float am = 0.f;
  const long long total = (long long)Cout * Cin * T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    // i indexes the OUTPUT
    if (mode == 0) {  // o[t][ci][co]
      const int co = (int)(i % Cout);
      long long r = i / Cout;
      const int ci = (int)(r % Cin);
      const int t = (int)(r / Cin);
      o[i] = w[((long long)co * Cin + ci) * T + t];
      am = fmaxf(am, fabsf(o[i]));
    } else {  // o[t][co][ci] = w[co][ci][T-1-t]
      const int ci = (int)(i % Cin);
      long long r = i / Cin;
      const int co = (int)(r % Cout);
      const int t = (int)(r / Cout);
      o[i] = w[((long long)co * Cin + ci) * T + (T - 1 - t)];
      am = fmaxf(am, fabsf(o[i]));
    }
  }
  if (part) publish_block_absmax(am, &smax, part + blockIdx.x);
}
__global__ void weight_unpack_k(const float* __restrict__ gt, float* __restrict__ g, int Cout, int Cin,
                                int T) {
  const long long total = (long long)Cout * Cin * T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    // i indexes g[co][ci][t]
    const int t = (int)(i % T);
    long long r = i / T;
    const int ci = (int)(r % Cin);
    const int co = (int)(r / Cin);
    g[i] = gt[((long long)t * Cin + ci) * Cout + co];
  }
}

// Every weight packing of a step in one launch (job = blockIdx.y): same arithmetic as weight_pack_k
__global__ __launch_bounds__(256) void weight_pack_batch_k(const DfPackJobDev* __restrict__ jobs) {
  const DfPackJobDev jb = jobs[blockIdx.y];
  if ((int)blockIdx.x >= jb.nblk) return;
  __shared__ unsigned smax;
  if (threadIdx.x == 0) smax = 0u;
  __syncthreads();
  float am = 0.f;
  const int Cout = jb.Cout, Cin = jb.Cin, T = jb.T;
  const long long total = (long long)Cout * Cin * T;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)jb.nblk * 256) {
    float v;
    if (jb.mode == 0) {  // o[t][ci][co]
      const int co = (int)(i % Cout);
      long long r = i / Cout;
      const int ci = (int)(r % Cin);
      const int t = (int)(r / Cin);
      v = jb.w[((long long)co * Cin + ci) * T + t];
    } else {             // o[t][co][ci] = w[co][ci][T-1-t]
      const int ci = (int)(i % Cin);
      long long r = i / Cin;
      const int co = (int)(r % Cout);
      const int t = (int)(r / Cout);
      v = jb.w[((long long)co * Cin + ci) * T + (T - 1 - t)];
    }
    jb.o[i] = v;
    am = fmaxf(am, fabsf(v));
  }
  if (jb.part) publish_block_absmax(am, &smax, jb.part + blockIdx.x);
}

// All deferred weight gradients of a step in one launch: job j (blockIdx.y) adds its tap-major accumulator into the
// reference-layout gradient, g[co][ci][t] += gt[t][ci][co], and clears the accumulator for the next step.  T <= 9:
// 32 co x 32 ci tiles through LDS, so that both the accumulator rows (co contiguous) and the gradient rows
// ((ci, t) contiguous) move as full cache lines; larger T (the 7x7 ends, a few KB): element-wise.
__global__ __launch_bounds__(256) void weight_unpack_add_batch_k(const DfUnpackJob* __restrict__ jobs) {
  const DfUnpackJob jb = jobs[blockIdx.y];
  const int Cout = jb.Cout, Cin = jb.Cin, T = jb.T;
  if (T > 9) {
    const long long total = (long long)Cout * Cin * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
      const int t = (int)(i % T);
      long long r = i / T;
      const int ci = (int)(r % Cin);
      const int co = (int)(r / Cin);
      const long long j = ((long long)t * Cin + ci) * Cout + co;
      jb.dst[i] += jb.src[j];
      jb.src[j] = 0.f;
    }
    return;
  }
  __shared__ float tile[32][32 * 9 + 1];
  const int tco = (Cout + 31) >> 5, tci = (Cin + 31) >> 5;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;          // 32 x 8
  for (int tl = blockIdx.x; tl < tco * tci; tl += gridDim.x) {
    const int co0 = (tl % tco) << 5, ci0 = (tl / tco) << 5;
    // rows (t, ci) of the accumulator: 32 consecutive co each
    for (int r = ly; r < T * 32; r += 8) {
      const int t = r >> 5, ci = ci0 + (r & 31), co = co0 + lx;
      float v = 0.f;
      if (ci < Cin && co < Cout) {
        const long long j = ((long long)t * Cin + ci) * Cout + co;
        v = jb.src[j];
        jb.src[j] = 0.f;
      }
      tile[lx][(r & 31) * T + t] = v;
    }
    __syncthreads();
    // rows co of the gradient: (ci0 .. ci0+31, t) = 32 T consecutive floats
    const int ncol = 32 * T;
    for (int e = threadIdx.x; e < 32 * ncol; e += 256) {
      const int c = e / ncol, q = e - c * ncol;
      const int co = co0 + c, ci = ci0 + q / T;
      if (co < Cout && ci < Cin) jb.dst[((long long)co * Cin + ci0) * T + q] += tile[c][q];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
bool df_conv3x3_fwd_try(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias,
                        float* y, hipStream_t st, int* rc);
bool df_conv3x3_wgrad_try(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc,
                          hipStream_t st, int* rc);
bool df_conv3x3_small_wgrad_try(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc, hipStream_t st,
                                int* rc);
bool df_conv3x3_split_fwd_try(const DfConvGeom* g, const float* x, const float* x_amax, int x_n, const float* w_packed,
                              const float* bias, const float* res, const float* ring, int ring_rl, float* y,
                              hipStream_t st, int* rc);
int df_conv3x3_split_res_ok(const DfConvGeom* g);
bool df_conv3x3_split_wgrad_try(const DfConvGeom* g, const float* x, const float* x_amax, int x_n, const float* dy,
                                const float* dy_amax, int dy_n, float* dw_tcc, float* db, hipStream_t st, int* rc,
                                const float* dy_pmax);
int df_weight_split_launch(const float* w_tcc, float* packed, int K, int M, int npart, hipStream_t st);
float* df_weight_probe_slots(float* packed, int K, int M);
int df_absmax_launch(const float* x, long long n, float* out, hipStream_t st, bool zero_first);
bool df_conv3d_fwd_try(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias, float* y,
                       hipStream_t st, int* rc);
bool df_conv3d_wgrad_try(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc, hipStream_t st,
                         int* rc);
static bool use_generic_only() {
  static DfOptInt o{"DFMIR_CONV_GENERIC", 0};     // A/B switch (= 1): force the generic gather kernels
  return o.get() == 1;
}

static int check_geom(const DfConvGeom* g) {
  if (!g) return -1;
  if (g->N <= 0 || g->Cin <= 0 || g->Cout <= 0) return -1;
  if (g->Di <= 0 || g->Hi <= 0 || g->Wi <= 0 || g->Do <= 0 || g->Ho <= 0 || g->Wo <= 0) return -1;
  if (g->KD <= 0 || g->KH <= 0 || g->KW <= 0 || g->KD > 15 || g->KH > 15 || g->KW > 15) return -1;
  if (g->stride < 1 || g->dil < 1) return -1;
  if (g->pad_mode != 0 && g->pad_mode != 1) return -1;
  if (g->pad_mode == 1 && g->dil != 1) return -1;
  return 0;
}

static int conv_fwd_impl(const DfConvGeom* g, const float* x, const float* x_amax, int x_n, const float* w_tcc,
                         const float* bias, float* y, void* stream);
extern "C" int dfmir_conv_fwd(const DfConvGeom* g, const float* x, const float* w_tcc,
                              const float* bias, float* y, void* stream) {
  return conv_fwd_impl(g, x, nullptr, 0, w_tcc, bias, y, stream);
}
extern "C" int dfmir_conv_fwd_scaled(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                     const float* w_tcc, const float* bias, float* y, void* stream) {
  return conv_fwd_impl(g, x, x_amax, x_amax_n, w_tcc, bias, y, stream);
}
extern "C" int dfmir_conv3x3_res_ok(const DfConvGeom* g) {
  return (g && check_geom(g) == 0 && !use_generic_only()) ? df_conv3x3_split_res_ok(g) : 0;
}
extern "C" int dfmir_conv3x3_fwd_scaled_res(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                            const float* w_tcc, const float* bias, const float* res,
                                            const float* ring, int ring_len, float* y, void* stream) {
  DF_ARG_CHECK(g && check_geom(g) == 0 && x && x_amax && x_amax_n > 0 && w_tcc && (res || ring) && y);
  DF_ARG_CHECK(!use_generic_only() && df_conv3x3_split_res_ok(g));
  DF_ARG_CHECK(!ring || (ring_len >= (g->Ho > g->Wo ? g->Ho : g->Wo) + 2 && g->Ho >= 4 && g->Wo >= 4));
  int rc = 0;
  if (!df_conv3x3_split_fwd_try(g, x, x_amax, x_amax_n, w_tcc, bias, res, ring, ring_len, y, (hipStream_t)stream, &rc))
    return df_set_error((int)hipErrorInvalidValue, __FILE__, __LINE__);
  return rc;
}
extern "C" int dfmir_absmax(const float* x, long long n, float* out, void* stream) {
  DF_ARG_CHECK(x && out && n > 0);
  const int rc = df_absmax_launch(x, n, out, (hipStream_t)stream, false);
  if (rc) return df_set_error(rc, __FILE__, __LINE__);
  return 0;
}
static int conv_fwd_impl(const DfConvGeom* g, const float* x, const float* x_amax, int x_n, const float* w_tcc,
                         const float* bias, float* y, void* stream) {
  DF_ARG_CHECK(check_geom(g) == 0 && x && w_tcc && y);
  hipStream_t st = (hipStream_t)stream;
  const long long P = (long long)g->N * g->Do * g->Ho * g->Wo;
  static DfOptFlag tinyvol_o{"DFMIR_NO_TINYVOL"};             // A/B: the deepest levels on conv_mfma_k
  // chosen by the PER-IMAGE volume, 3-D only: the kernel a sample runs on must not depend on the batch it arrives in
  // (tests/test_gpu_models.py::test_batch16_equals_per_sample_runs), and the 2-D deep levels keep their tuned path.
  // Ahead of the fp32-MFMA 3x3x3 kernel too: its 2 x 8 x 16 patches are one or two workgroups at these volumes (0.14-0.27 ms)
  const bool tinyvol = !use_generic_only() && g->Do > 1 && (long long)g->Do * g->Ho * g->Wo <= 512 && P <= 8192 && g->Cout <= 64 &&
                       g->Cout > 4 && !tinyvol_o.get();
  if (!use_generic_only()) {
    int rc = 0;
    if (df_conv3x3_split_fwd_try(g, x, x_amax, x_n, w_tcc, bias, nullptr, nullptr, 0, y, st, &rc)) return rc;
    if (df_conv3x3_fwd_try(g, x, w_tcc, bias, y, st, &rc)) return rc;
    if (!tinyvol && df_conv3d_fwd_try(g, x, w_tcc, bias, y, st, &rc)) return rc;
  }
  if (g->Cout <= 4) {
    const unsigned grid = (unsigned)((P + 255) / 256);
    if (g->Cout == 1) conv_small_k<1><<<grid, 256, 0, st>>>(x, w_tcc, bias, y, *g);
    else if (g->Cout == 2) conv_small_k<2><<<grid, 256, 0, st>>>(x, w_tcc, bias, y, *g);
    else conv_small_k<4><<<grid, 256, 0, st>>>(x, w_tcc, bias, y, *g);
    DF_LAUNCH_CHECK();
    return 0;
  }
  if (tinyvol) {
    if (g->Cout <= 8) conv_tinyvol_k<8><<<(unsigned)P, 256, 0, st>>>(x, w_tcc, bias, y, *g);
    else if (g->Cout <= 16) conv_tinyvol_k<16><<<(unsigned)P, 256, 0, st>>>(x, w_tcc, bias, y, *g);
    else if (g->Cout <= 32) conv_tinyvol_k<32><<<(unsigned)P, 256, 0, st>>>(x, w_tcc, bias, y, *g);
    else conv_tinyvol_k<64><<<(unsigned)P, 256, 0, st>>>(x, w_tcc, bias, y, *g);
    DF_LAUNCH_CHECK();
    return 0;
  }
  // buffer descriptors address < 2 GiB: split the batch when a tensor is larger
  const long long in_img = (long long)g->Cin * g->Di * g->Hi * g->Wi * 4;
  const long long out_img = (long long)g->Cout * g->Do * g->Ho * g->Wo * 4;
  DF_ARG_CHECK(in_img < 0x7FFFFFFFLL && out_img < 0x7FFFFFFFLL);
  DF_ARG_CHECK((long long)g->KD * g->KH * g->KW * g->Cin * g->Cout * 4 < 0x7FFFFFFFLL);
  long long nb = 0x7FFFFFFFLL / (in_img > out_img ? in_img : out_img);
  if (nb > g->N) nb = g->N;
  for (int n0 = 0; n0 < g->N; n0 += (int)nb) {
    ConvK k;
    k.g = *g;
    k.g.N = (g->N - n0 < nb) ? g->N - n0 : (int)nb;
    df_chunking(g->Cin, k.kc, k.nchunks);
    const float* xs = x + (long long)n0 * (in_img / 4);
    float* ys = y + (long long)n0 * (out_img / 4);
    long long Ps = (long long)k.g.N * g->Do * g->Ho * g->Wo;
    // dgrad of a stride-2 conv: parity-class enumeration (taps that only meet the inserted zeros are skipped)
    static DfOptFlag dil2_o{"DFMIR_NO_DIL2"};
  const bool dil2_off = dil2_o.get();
    const bool dil2 = g->dil == 2 && g->stride == 1 && g->pad_mode == 0 && g->KD * g->KH * g->KW <= 128 && g->Cout <= 64 && !dil2_off;
    k.ncx = k.ncy = k.ncz = 1;
    k.Dm = g->Do; k.Hm = g->Ho; k.Wm = g->Wo;
    if (dil2) {
      // an axis has two classes when the input is really dilated along it (2-D convs: Di == 1, KD == 1 stays one class)
      if (g->Wi > 1 || g->KW > 1) { k.ncx = 2; k.Wm = (g->Wo + 1) / 2; }
      if (g->Hi > 1 || g->KH > 1) { k.ncy = 2; k.Hm = (g->Ho + 1) / 2; }
      if (g->Di > 1 || g->KD > 1) { k.ncz = 2; k.Dm = (g->Do + 1) / 2; }
      Ps = (long long)k.ncx * k.ncy * k.ncz * k.g.N * k.Dm * k.Hm * k.Wm;
    }
    // few output voxels (the deep U-Net levels: 8^2 .. 32^2 x 16 images, 10x12x14 voxels): 256-pixel tiles leave most
    // CUs idle while each workgroup walks the whole K loop; 64 x 64 tiles give 4x the workgroups at a quarter of the work
    static DfOptFlag small_o{"DFMIR_NO_SMALL_TILES"};
  const bool small_off = small_o.get();
    const bool small_p = !small_off && (Ps + 255) / 256 < 192 && g->Cout > 8;
    if (dil2) {
      if (small_p) {
        dim3 grid((unsigned)((Ps + 63) / 64), (unsigned)((g->Cout + 63) / 64));
        conv_mfma_k<2, 2, 1, 1, true><<<grid, 256, 0, st>>>(xs, w_tcc, bias, ys, k);
      } else {
        dim3 grid((unsigned)((Ps + 255) / 256), 1);
        if (g->Cout > 32) conv_mfma_k<1, 4, 2, 2, true><<<grid, 256, 0, st>>>(xs, w_tcc, bias, ys, k);
        else conv_mfma_k<1, 4, 1, 2, true><<<grid, 256, 0, st>>>(xs, w_tcc, bias, ys, k);
      }
    } else if (small_p && g->Cout <= 64) {
      dim3 grid((unsigned)((Ps + 63) / 64), (unsigned)((g->Cout + 63) / 64));
      conv_mfma_k<2, 2, 1, 1><<<grid, 256, 0, st>>>(xs, w_tcc, bias, ys, k);
    } else if (g->Cout > 64) {
      const long long big = ((Ps + 127) / 128) * ((g->Cout + 127) / 128);
      static DfOptInt big_o{"DFMIR_GEMM_BIG_MIN", 256};
  const int big_min = big_o.get();
      if (big < big_min) {   // small GEMMs (PatchNCE MLP: 4096 rows x 256): 64x64 tiles fill the 256 CUs
        dim3 grid((unsigned)((Ps + 63) / 64), (unsigned)((g->Cout + 63) / 64));
        conv_mfma_k<2, 2, 1, 1><<<grid, 256, 0, st>>>(xs, w_tcc, bias, ys, k);
      } else {
        dim3 grid((unsigned)((Ps + 127) / 128), (unsigned)((g->Cout + 127) / 128));
        conv_mfma_k<2, 2, 2, 2><<<grid, 256, 0, st>>>(xs, w_tcc, bias, ys, k);
      }
    } else if (g->Cout > 32) {
      dim3 grid((unsigned)((Ps + 255) / 256), 1);
      conv_mfma_k<1, 4, 2, 2><<<grid, 256, 0, st>>>(xs, w_tcc, bias, ys, k);
    } else {
      dim3 grid((unsigned)((Ps + 255) / 256), 1);
      conv_mfma_k<1, 4, 1, 2><<<grid, 256, 0, st>>>(xs, w_tcc, bias, ys, k);
    }
    DF_LAUNCH_CHECK();
  }
  return 0;
}

static int conv_wgrad_impl(const DfConvGeom* g, const float* x, const float* x_amax, int x_n, const float* dy,
                           const float* dy_amax, int dy_n, float* dw_tcc, float* db, void* stream,
                           const float* dy_pmax = nullptr);
extern "C" int dfmir_conv_wgrad_scaled_ch(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                          const float* dy, const float* dy_amax, int dy_amax_n, const float* dy_pmax,
                                          float* dw_tcc, float* db, void* stream) {
  return conv_wgrad_impl(g, x, x_amax, x_amax_n, dy, dy_amax, dy_amax_n, dw_tcc, db, stream, dy_pmax);
}
extern "C" int dfmir_conv_wgrad(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc,
                                void* stream) {
  return conv_wgrad_impl(g, x, nullptr, 0, dy, nullptr, 0, dw_tcc, nullptr, stream);
}
extern "C" int dfmir_conv_wgrad_scaled(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                       const float* dy, const float* dy_amax, int dy_amax_n, float* dw_tcc,
                                       float* db, void* stream) {
  return conv_wgrad_impl(g, x, x_amax, x_amax_n, dy, dy_amax, dy_amax_n, dw_tcc, db, stream);
}
static int bias_grad_launch(const float* dy, float* db, int N, int C, long long S, hipStream_t st);
static int conv_wgrad_impl(const DfConvGeom* g, const float* x, const float* x_amax, int x_n, const float* dy,
                           const float* dy_amax, int dy_n, float* dw_tcc, float* db, void* stream, const float* dy_pmax) {
  DF_ARG_CHECK(check_geom(g) == 0 && x && dy && dw_tcc);
  DF_ARG_CHECK(g->dil == 1 && g->Cin < (1 << 19));
  hipStream_t st = (hipStream_t)stream;
  if (!use_generic_only()) {
    int rc = 0;
    if (df_conv3x3_split_wgrad_try(g, x, x_amax, x_n, dy, dy_amax, dy_n, dw_tcc, db, st, &rc, dy_pmax)) return rc;   // db fused
  }
  if (db) {   // every other kernel: the bias gradient is its own pass over dY
    const int rcb = bias_grad_launch(dy, db, g->N, g->Cout, (long long)g->Do * g->Ho * g->Wo, st);
    if (rcb) return rcb;
  }
  if (!use_generic_only()) {
    int rc = 0;
    if (df_conv3x3_wgrad_try(g, x, dy, dw_tcc, st, &rc)) return rc;
    if (df_conv3x3_small_wgrad_try(g, x, dy, dw_tcc, st, &rc)) return rc;
    if (df_conv3d_wgrad_try(g, x, dy, dw_tcc, st, &rc)) return rc;
  }
  {
    // 1x1x1, stride 1, no padding: the streaming NT GEMM (A/B: DFMIR_NO_1X1_WGRAD=1 = the generic gather kernel)
    static DfOptFlag no1x1_o{"DFMIR_NO_1X1_WGRAD"};
    const long long S1 = (long long)g->Di * g->Hi * g->Wi;
    if (!use_generic_only() && !no1x1_o.get() && g->KD == 1 && g->KH == 1 && g->KW == 1 && g->stride == 1 && g->pd == 0 &&
        g->ph == 0 && g->pw == 0 && g->Do == g->Di && g->Ho == g->Hi && g->Wo == g->Wi && (S1 & 3) == 0 && S1 >= 64 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0 && g->Cin >= 8 && g->Cout >= 8) {
      const long long cpi = (S1 + 63) / 64, total = (long long)g->N * cpi;
      const unsigned ny = (unsigned)((g->Cin + 63) / 64), nz = (unsigned)((g->Cout + 63) / 64);
      long long want = 1024 / ((long long)ny * nz);          // ~4 workgroups per CU in all
      if (want < 1) want = 1;
      long long maxs = (total + 7) / 8;                        // >= 8 chunks per workgroup
      if (maxs < 1) maxs = 1;
      if (want > maxs) want = maxs;
      const long long cpb = (total + want - 1) / want;
      const unsigned nx = (unsigned)((total + cpb - 1) / cpb);
      conv1x1_wgrad_k<<<dim3(nx, ny, nz), 256, 0, st>>>(x, dy, dw_tcc, g->N, g->Cin, g->Cout, S1, cpi, cpb, df_det_fx());
      DF_LAUNCH_CHECK();
      return 0;
    }
  }
  long long P = (long long)g->N * g->Do * g->Ho * g->Wo;
  const int J = g->KD * g->KH * g->KW * g->Cin;
  auto plan = [&](int BJ, int BC, int BP, unsigned& nP, long long& pchunk, dim3& grid) {
    const unsigned nJ = (J + BJ - 1) / BJ, nC = (g->Cout + BC - 1) / BC;
    long long want = 2048 / ((long long)nJ * nC);
    if (want < 1) want = 1;
    long long maxp = (P + (long long)BP * 8 - 1) / ((long long)BP * 8);  // >= 8 steps per block
    if (maxp < 1) maxp = 1;
    if (want > maxp) want = maxp;
    pchunk = (P + want - 1) / want;
    pchunk = ((pchunk + BP - 1) / BP) * BP;
    nP = (unsigned)((P + pchunk - 1) / pchunk);
    grid = dim3(nP, nJ, nC);
  };
  // buffer descriptors address < 2 GiB: split the batch when a tensor is larger (dw accumulates)
  const long long in_img = (long long)g->Cin * g->Di * g->Hi * g->Wi * 4;
  const long long out_img = (long long)g->Cout * g->Do * g->Ho * g->Wo * 4;
  DF_ARG_CHECK(in_img < 0x7FFFFFFFLL && out_img < 0x7FFFFFFFLL);
  long long nb = 0x7FFFFFFFLL / (in_img > out_img ? in_img : out_img);
  if (nb > g->N) nb = g->N;
  const int Nfull = g->N;
  for (int n0 = 0; n0 < Nfull; n0 += (int)nb) {
    DfConvGeom gs = *g;
    gs.N = (Nfull - n0 < nb) ? Nfull - n0 : (int)nb;
    const float* xs = x + (long long)n0 * (in_img / 4);
    const float* ds = dy + (long long)n0 * (out_img / 4);
    P = (long long)gs.N * g->Do * g->Ho * g->Wo;
    unsigned nP;
    long long pchunk;
    dim3 grid;
    if (g->Cout > 64) {
      plan(128, 128, 16, nP, pchunk, grid);
      conv_wgrad_mfma_k<2, 2, 2, 2, 16><<<grid, 256, 0, st>>>(xs, ds, dw_tcc, gs, pchunk, df_det_fx());
    } else if (g->Cout > 32) {
      plan(128, 64, 16, nP, pchunk, grid);
      conv_wgrad_mfma_k<4, 1, 1, 2, 16><<<grid, 256, 0, st>>>(xs, ds, dw_tcc, gs, pchunk, df_det_fx());
    } else {
      plan(128, 32, 32, nP, pchunk, grid);
      conv_wgrad_mfma_k<4, 1, 1, 1, 32><<<grid, 256, 0, st>>>(xs, ds, dw_tcc, gs, pchunk, df_det_fx());
    }
    DF_LAUNCH_CHECK();
  }
  return 0;
}

static int bias_grad_launch(const float* dy, float* db, int N, int C, long long S, hipStream_t st) {
  int nsplit = (int)((S + 65535) / 65536);      // planes larger than 64K elements are split further
  if (nsplit > 64) nsplit = 64;
  if (nsplit < 1) nsplit = 1;
  DF_ARG_CHECK(N <= 65535);
  bias_grad_k<<<dim3(C, N, nsplit), S >= 4096 ? 256 : 64, 0, st>>>(dy, db, N, C, S, nsplit, df_det_fx());
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_bias_grad(const float* dy, float* db, int N, int C, long long S, void* stream) {
  DF_ARG_CHECK(dy && db && N > 0 && C > 0 && S > 0);
  return bias_grad_launch(dy, db, N, C, S, (hipStream_t)stream);
}

extern "C" int dfmir_weight_pack(const float* w, float* w_tcc, int Cout, int Cin, int T, int mode,
                                 void* stream) {
  DF_ARG_CHECK(w && w_tcc && Cout > 0 && Cin > 0 && T > 0 && (mode == 0 || mode == 1));
  const long long total = (long long)Cout * Cin * T;
  const unsigned nblk = df_grid(total, 256, 2048);
  const int K = mode ? Cout : Cin, M = mode ? Cin : Cout;   // K = reduction, M = produced channels
  float* part = (T == 9) ? df_weight_probe_slots(w_tcc, K, M) : nullptr;
  weight_pack_k<<<nblk, 256, 0, (hipStream_t)stream>>>(w, w_tcc, Cout, Cin, T, mode, part);
  DF_LAUNCH_CHECK();
  if (T == 9) {   // split section for the 3x3 kernels (conv3x3s.hip)
    const int rc = df_weight_split_launch(w_tcc, w_tcc, K, M, (int)nblk, (hipStream_t)stream);
    if (rc) return df_set_error(rc, __FILE__, __LINE__);
  }
  return 0;
}
int df_conv3x3_reflect_ring_ok(const DfConvGeom* g);
int df_conv3x3_reflect_ring_launch(const DfConvGeom* g, const float* dy, const float* dy_cols, const float* dy_amax,
                                   int dy_n, const float* wd_packed, float* ring, hipStream_t st);
int df_conv3x3_reflect_ring_len(const DfConvGeom* g);
extern "C" int dfmir_conv3x3_reflect_ring_ok(const DfConvGeom* g) {
  return (g && check_geom(g) == 0) ? df_conv3x3_reflect_ring_ok(g) : 0;
}
extern "C" int dfmir_conv3x3_reflect_ring_len(const DfConvGeom* g) {
  return (g && check_geom(g) == 0 && df_conv3x3_reflect_ring_ok(g)) ? df_conv3x3_reflect_ring_len(g) : 0;
}
extern "C" int dfmir_conv3x3_reflect_ring(const DfConvGeom* g, const float* dy, const float* dy_cols,
                                          const float* dy_amax, int dy_amax_n, const float* wd_packed, float* ring,
                                          void* stream) {
  DF_ARG_CHECK(g && check_geom(g) == 0 && dy && dy_amax && dy_amax_n > 0 && wd_packed && ring);
  DF_ARG_CHECK(df_conv3x3_reflect_ring_ok(g));
  const int rc = df_conv3x3_reflect_ring_launch(g, dy, dy_cols, dy_amax, dy_amax_n, wd_packed, ring, (hipStream_t)stream);
  if (rc) return df_set_error(rc, __FILE__, __LINE__);
  return 0;
}
extern "C" int dfmir_weight_pack_batch(const DfPackJob* jobs_host, int njobs, void* table_dev, int upload,
                                       void* stream) {
  DF_ARG_CHECK(jobs_host && table_dev && njobs > 0 && njobs <= 65535);
  static_assert(sizeof(DfPackJobDev) == 64, "job table stride");
  hipStream_t st = (hipStream_t)stream;
  int max_nblk = 0, max_nsplit = 0;
  std::vector<DfPackJobDev> tab((size_t)njobs);
  for (int i = 0; i < njobs; ++i) {
    const DfPackJob& h = jobs_host[i];
    DF_ARG_CHECK(h.w && h.packed && h.Cout > 0 && h.Cin > 0 && h.T > 0 && (h.mode == 0 || h.mode == 1));
    DfPackJobDev& d = tab[(size_t)i];
    d.w = h.w; d.o = h.packed; d.Cout = h.Cout; d.Cin = h.Cin; d.T = h.T; d.mode = h.mode;
    d.nblk = (int)df_grid((long long)h.Cout * h.Cin * h.T, 256, 2048);
    df_weight_split_fill(&d);
    max_nblk = d.nblk > max_nblk ? d.nblk : max_nblk;
    max_nsplit = d.nsplit > max_nsplit ? d.nsplit : max_nsplit;
  }
  if (upload) {
    // pageable source: the runtime stages it before returning, so `tab` may go out of scope
    hipError_t e = hipMemcpyAsync(table_dev, tab.data(), sizeof(DfPackJobDev) * (size_t)njobs, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return df_set_error((int)e, __FILE__, __LINE__);
  }
  weight_pack_batch_k<<<dim3((unsigned)max_nblk, (unsigned)njobs), 256, 0, st>>>(
      reinterpret_cast<const DfPackJobDev*>(table_dev));
  DF_LAUNCH_CHECK();
  const int rc = df_weight_split_batch_launch(reinterpret_cast<const DfPackJobDev*>(table_dev), njobs, max_nsplit, st);
  if (rc) return df_set_error(rc, __FILE__, __LINE__);
  return 0;
}
extern "C" long long dfmir_weight_pack_floats(int Cout, int Cin, int T) {
  if (Cout <= 0 || Cin <= 0 || T <= 0) return -1;
  // [T][K][M] fp32 (rounded up to 16 B) + the split section when T == 9; one size for both packings
  const long long s0 = df_pack_split_floats(Cin, Cout, T), s1 = df_pack_split_floats(Cout, Cin, T);
  return df_pack_tcc_floats(Cin, Cout, T) + (s0 > s1 ? s0 : s1) + (T == 9 ? 4 + 2048 : 0);   // + scale trailer, probe slots
}
extern "C" int dfmir_weight_unpack_add_batch(const DfUnpackJob* jobs_dev, int njobs, long long max_total,
                                             void* stream) {
  DF_ARG_CHECK(jobs_dev && njobs > 0 && njobs <= 65535 && max_total > 0);
  long long bx = (max_total + 255) / 256;
  if (bx > 64) bx = 64;
  weight_unpack_add_batch_k<<<dim3((unsigned)bx, (unsigned)njobs), 256, 0, (hipStream_t)stream>>>(jobs_dev);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_weight_unpack(const float* g_tcc, float* g, int Cout, int Cin, int T,
                                   void* stream) {
  DF_ARG_CHECK(g_tcc && g && Cout > 0 && Cin > 0 && T > 0);
  const long long total = (long long)Cout * Cin * T;
  weight_unpack_k<<<df_grid(total, 256, 2048), 256, 0, (hipStream_t)stream>>>(g_tcc, g, Cout, Cin, T);
  DF_LAUNCH_CHECK();
  return 0;
}

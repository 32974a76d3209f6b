// The STRIDE-2 3x3x3 convolutions of the VoxelMorph encoder below its first level (torchvoxelmorph/networks.py:66-71,
// 1506-1521: ConvBlock(stride 2): 16 -> 32, 32 -> 32 ... at 1/4, 1/8, 1/16 of the volume), forward / weight gradient / data
// gradient, on v_mfma_f32_16x16x4_f32 (exact fp32 products).
//
// These layers are 0.5 % of the step's FLOPs (2.97 + 0.74 + 0.09 GFLOP at 160 x 192 x 224) and were 9 % of its time: they ran
// on the generic gather kernels (csrc/conv.hip: per-element coordinate arithmetic in the reduction loop, 4-byte gathers,
// 27-60 workgroups on 256 CUs) at 1-37 TFLOP/s -- 0.20 ms forward, 0.35 ms data gradient, 0.30 ms weight gradient.  Here:
//   forward   workgroup = 2 x 8 x 16 output voxels x 16 TMT output channels; per 4-channel chunk the (5 x 17 x 33) input
//             patch is staged once (offsets precomputed: 44 independent loads per thread in flight) and read at
//             position 2 v + tap; weights of the chunk [27][4][couts] beside it; bias, LeakyReLU and the range probe of the
//             result in the epilogue (the consumers are split convolutions: no separate absmax pass).
#include "common.h"

typedef float f32x4_s2 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_s2 __attribute__((ext_vector_type(4)));

namespace {
struct S2mP {
  int N, Cin, Cout, D, H, W, Do, Ho, Wo;
  int act;
  float slope;
  int nz, ny, nx;                // output patches per axis
};
constexpr unsigned S2_OOB = 0x80000000u;

template <int TMT>
__global__ __launch_bounds__(256) void conv3d_s2_fwd_k(const float* __restrict__ x, const float* __restrict__ wt,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ y_amax, S2mP k) {
  constexpr int PZ = 2, PY = 8, PX = 16, HZ = 2 * PZ + 1, HY = 2 * PY + 1, HX = 2 * PX + 1;
  constexpr int XP = HZ * HY * HX;                        // 2805 (odd: the four k groups of a read fall on all 32 banks twice)
  constexpr int CK = 4, BMC = 16 * TMT;
  constexpr int WSTR = (BMC % 32 == 16) ? BMC : BMC + 16; // row stride == 16 (mod 32)
  constexpr int NS = (XP + 255) / 256;                    // 11
  constexpr int W4 = 27 * CK * (BMC / 4);                 // float4 per weight chunk
  constexpr int NW = (W4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Ws[27 * CK * WSTR];
  __shared__ float Xs[CK * XP];
  __shared__ unsigned smax;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lk = lane >> 4;
  if (tid == 0) smax = 0u;
  const int Si = k.D * k.H * k.W, So = k.Do * k.Ho * k.Wo;
  int pid = blockIdx.x;
  const int bx = pid % k.nx; pid /= k.nx;
  const int by = pid % k.ny; pid /= k.ny;
  const int bz = pid % k.nz;
  const int n = pid / k.nz;
  const int z0 = bz * PZ, y0 = by * PY, x0 = bx * PX;      // output coordinates
  const int m0 = blockIdx.y * BMC;

  const __amdgpu_buffer_rsrc_t x_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x + (long long)n * k.Cin * Si), 0, (unsigned)(k.Cin * Si) * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt), 0, (unsigned)(27 * k.Cin * k.Cout) * 4u, 0x00020000);
  const unsigned s4 = (unsigned)Si * 4u;

  unsigned gbyte[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int pos = tid + 256 * s;
    unsigned off = S2_OOB;
    if (pos < XP) {
      const int hx = pos % HX, t = pos / HX, hy = t % HY, hz = t / HY;
      const int gz = 2 * z0 - 1 + hz, gy = 2 * y0 - 1 + hy, gx = 2 * x0 - 1 + hx;
      if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W)
        off = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;
    }
    gbyte[s] = off;
  }
  unsigned wbyte[NW];
  bool wok[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int idx4 = tid + 256 * j;
    const int row = idx4 / (BMC / 4), c4 = idx4 - row * (BMC / 4);
    const int tap = row / CK, ci = row % CK, co = m0 + c4 * 4;
    wok[j] = idx4 < W4 && co < k.Cout;
    wbyte[j] = (unsigned)((tap * k.Cin + ci) * k.Cout + co) * 4u;
  }
  const unsigned wstep = (unsigned)(CK * k.Cout) * 4u;

  int pbase[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wid * 4 + j;                 // x-row of the output patch: (pz, py)
    pbase[j] = ((2 * (r >> 3)) * HY + 2 * (r & 7)) * HX + 2 * l15;
  }

  f32x4_s2 acc[TMT][4];
#pragma unroll
  for (int i = 0; i < TMT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  u32x4_s2 rw[NW];
  unsigned rx[NS][CK];
#define S2F_GLOAD(ci0_)                                                                          \
  {                                                                                              \
    const unsigned wadd = (unsigned)((ci0_) / CK) * wstep;                                       \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                             \
      const int row_ = (tid + 256 * j) / (BMC / 4);                                              \
      const bool ok = wok[j] && ((ci0_) + (row_ % CK)) < k.Cin;                                  \
      rw[j] = __builtin_amdgcn_raw_buffer_load_b128(w_src, ok ? wbyte[j] + wadd : S2_OOB, 0, 0); \
    }                                                                                            \
    const unsigned xadd = (unsigned)(ci0_) * s4;                                                 \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                             \
      _Pragma("unroll") for (int c = 0; c < CK; ++c)                                             \
        rx[s][c] = __builtin_amdgcn_raw_buffer_load_b32(                                         \
            x_src, ((ci0_) + c < k.Cin) ? gbyte[s] : S2_OOB, xadd + (unsigned)c * s4, 0);        \
    }                                                                                            \
  }
#define S2F_LSTORE()                                                                             \
  {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                             \
      const int idx4 = tid + 256 * j;                                                            \
      if (idx4 < W4) {                                                                           \
        const int row = idx4 / (BMC / 4), c4 = idx4 - row * (BMC / 4);                           \
        *reinterpret_cast<u32x4_s2*>(&Ws[row * WSTR + c4 * 4]) = rw[j];                          \
      }                                                                                          \
    }                                                                                            \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                             \
      const int pos = tid + 256 * s;                                                             \
      if (pos < XP) {                                                                            \
        _Pragma("unroll") for (int c = 0; c < CK; ++c) Xs[c * XP + pos] = __uint_as_float(rx[s][c]); \
      }                                                                                          \
    }                                                                                            \
  }

  S2F_GLOAD(0);
  S2F_LSTORE();
  __syncthreads();
  for (int ci0 = 0; ci0 < k.Cin; ci0 += CK) {
    const bool more = (ci0 + CK) < k.Cin;
    if (more) S2F_GLOAD(ci0 + CK);
#pragma unroll 3
    for (int tap = 0; tap < 27; ++tap) {
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      const int toff = (dz * HY + dy) * HX + dx;
      float a[TMT], b[4];
#pragma unroll
      for (int i = 0; i < TMT; ++i) a[i] = Ws[(tap * CK + lk) * WSTR + i * 16 + l15];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Xs[lk * XP + pbase[j] + toff];
#pragma unroll
      for (int i = 0; i < TMT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      __syncthreads();
      S2F_LSTORE();
      __syncthreads();
    }
  }
#undef S2F_GLOAD
#undef S2F_LSTORE

  // ---- epilogue: D[row = (lane >> 4) * 4 + r -> cout][col = lane & 15 -> x]
  const int gx = x0 + l15;
  float pm = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wid * 4 + j;
    const int gz = z0 + (r >> 3), gy = y0 + (r & 7);
    if (gz >= k.Do || gy >= k.Ho || gx >= k.Wo) continue;
    float* yb = y + (long long)n * k.Cout * So + ((long long)gz * k.Ho + gy) * k.Wo + gx;
#pragma unroll
    for (int i = 0; i < TMT; ++i) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = m0 + i * 16 + lk * 4 + rr;
        if (co < k.Cout) {
          float v = acc[i][j][rr] + (bias ? bias[co] : 0.f);
          if (k.act == 1) v = v > 0.f ? v : v * k.slope;
          yb[(long long)co * So] = v;
          pm = fmaxf(pm, fabsf(v));
        }
      }
    }
  }
  if (y_amax) {
    __syncthreads();
    publish_block_absmax_acc(pm, &smax, y_amax);
  }
}

// Below this many output voxels per image (dgrad: voxels of dy) the layers stay where they were: the deepest levels of the
// 6-level U-Net (8^3, 4^3, 2^3) are one or two patches here -- a single workgroup walking all of Cin x 27 taps -- and
// conv_tinyvol_k's one-voxel workgroups win (128^3 step with them on these kernels: 3.16 ms, without: 3.06).
constexpr long long S2_MIN_VOX = 512;
bool s2m_off() {
  static DfOptFlag o{"DFMIR_CONV3D_NO_S2"};
  return o.get();
}
bool s2m_geom_ok(const DfConvGeom* g) {
  return g->KD == 3 && g->KH == 3 && g->KW == 3 && g->stride == 2 && g->dil == 1 && g->pd == 1 && g->ph == 1 && g->pw == 1 &&
         g->pad_mode == 0 && (g->act == 0 || g->act == 1) && g->Di > 1 && g->Do == (g->Di + 1) / 2 && g->Ho == (g->Hi + 1) / 2 &&
         g->Wo == (g->Wi + 1) / 2 && (long long)g->Do * g->Ho * g->Wo > S2_MIN_VOX && g->Cin >= 4 && (g->Cin % 4) == 0 && g->Cout >= 16 && (g->Cout % 16) == 0 && g->Cout <= 64 &&
         (long long)(g->Cin > g->Cout ? g->Cin : g->Cout) * g->Di * g->Hi * g->Wi * 4 < 0x7FFFFFFFLL;
}
}  // namespace

extern "C" int dfmir_conv3d_s2_ok(const DfConvGeom* g) { return (g && !s2m_off() && s2m_geom_ok(g)) ? 1 : 0; }

// y = act(conv3x3x3 stride 2 pad 1 (x, w) + bias); w_tcc = the fp32 tap-major packing [27][Cin][Cout]; y_amax (may be NULL):
// DF_PROBE_SLOTS accumulating range-probe slots of y.
extern "C" int dfmir_conv3d_s2_fwd(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias, float* y,
                                   float* y_amax, void* stream) {
  DF_ARG_CHECK(g && x && w_tcc && y && dfmir_conv3d_s2_ok(g));
  DF_ARG_CHECK((reinterpret_cast<uintptr_t>(w_tcc) & 15) == 0);
  S2mP k{g->N, g->Cin, g->Cout, g->Di, g->Hi, g->Wi, g->Do, g->Ho, g->Wo, g->act, g->slope,
         (g->Do + 1) / 2, (g->Ho + 7) / 8, (g->Wo + 15) / 16};
  const long long patches = (long long)g->N * k.nz * k.ny * k.nx;
  DF_ARG_CHECK(patches < (1LL << 30));
  hipStream_t st = (hipStream_t)stream;
  // few patches: one 16-channel tile per workgroup (more workgroups, a shorter MFMA chain in each)
  const int tmt = (patches >= 256 && g->Cout % 32 == 0) ? 2 : 1;
  dim3 grid((unsigned)patches, (unsigned)(g->Cout / (16 * tmt)));
  if (tmt == 2) conv3d_s2_fwd_k<2><<<grid, 256, 0, st>>>(x, w_tcc, bias, y, y_amax, k);
  else conv3d_s2_fwd_k<1><<<grid, 256, 0, st>>>(x, w_tcc, bias, y, y_amax, k);
  DF_LAUNCH_CHECK();
  return 0;
}

// ================================================================================================
// Weight gradient: dW[tap][ci][co] += sum_o x[ci][2 o + tap - 1] dY[co][o].  MFMA rows = (tap, ci) of a channel group of
// <= 16 input channels (27 cg rows: the whole [27 cg, Cout] tile of the group lives in the workgroup's accumulators, as in
// conv3d_wgrad16_k), columns = output channels, K = the output voxels of 1 x 4 x 16 patches, 4 per MFMA.  Per patch the
// (3 x 9 x 33) input halo of the group and the [64 voxels][Cout] dY tile are staged in LDS (offsets decoded once per
// patch: 56 independent loads per thread); row (tap, ci) reads x at 2 v + tap.  Split-K over blockIdx.x, df_acc at the end.
// ================================================================================================
namespace {
struct S2wP {
  int N, Cin, Cout, D, H, W, Do, Ho, Wo;
  int cg, ngroups;               // channels per group, number of groups (blockIdx.y)
  int ny, nx;                    // patches per axis (nz == Do)
  long long npatch, per_block;
  const float* fx;               // deterministic mode (common.h df_acc)
};

template <int NCT, int RT>
__global__ __launch_bounds__(256) void conv3d_s2_wgrad_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dwt, float* __restrict__ db, S2wP k) {
  constexpr int PY = 4, PX = 16, BP = PY * PX, HY = 2 * PY + 1, HX = 2 * PX + 1;
  constexpr int NPOS = 3 * HY * HX;                 // 891 (odd: see conv3d_s2_fwd_k)
  static_assert(256 / HX + 1 < HY, "one row carry per element step");
  constexpr int DSTR = (NCT == 1) ? 16 : ((16 * NCT) % 32 == 16 ? 16 * NCT : 16 * NCT + 16);
  constexpr int CGMAX = 16;
  constexpr int NXL = (CGMAX * NPOS + 255) / 256;   // halo loads per thread (56)
  constexpr int ND4 = ((BP / 4) * (16 * NCT) + 255) / 256;
  __shared__ float Xs[CGMAX * NPOS];
  __shared__ float Ds[BP * DSTR];
  __shared__ int ppos[BP];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lk = lane >> 4;
  const int Si = k.D * k.H * k.W, So = k.Do * k.Ho * k.Wo;
  const int grp = blockIdx.y;
  const int c0 = grp * k.cg;
  const int cgn = (k.Cin - c0 < k.cg) ? k.Cin - c0 : k.cg;
  const int nrows = 27 * cgn;
  const long long pbeg = (long long)blockIdx.x * k.per_block;
  long long pend = pbeg + k.per_block;
  if (pend > k.npatch) pend = k.npatch;

  int aoff[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int jj = (wid + 4 * r) * 16 + l15;
    int off = 0;
    if (jj < nrows) {
      const int tap = jj / cgn, ci = jj - tap * cgn;
      const int dz = tap / 9, dyy = (tap / 3) % 3, dx = tap % 3;
      off = ci * NPOS + (dz * HY + dyy) * HX + dx;
    }
    aoff[r] = off;
  }
  if (tid < BP) ppos[tid] = (2 * (tid >> 4)) * HX + 2 * (tid & 15);   // output voxel (py, px) -> halo position of its tap (0, 0, 0)

  f32x4_s2 acc[RT][NCT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[r][c][e] = 0.f;

  // bias gradient (db != NULL, channel group 0): wave 0 sums the dY operands it reads anyway (every wave reads them all)
  const bool do_db = db != nullptr && grp == 0 && wid == 0;
  float bsum[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) bsum[c] = 0.f;
  const unsigned si4 = (unsigned)Si * 4u, so4 = (unsigned)So * 4u;
  unsigned rxv[NXL];
  u32x4_s2 rd[ND4];
  const int nelem = cgn * NPOS;

#define S2W_GLOAD(p_)                                                                            \
  {                                                                                              \
    long long q_ = (p_);                                                                         \
    const int bx_ = (int)(q_ % k.nx); q_ /= k.nx;                                                \
    const int by_ = (int)(q_ % k.ny); q_ /= k.ny;                                                \
    const int z_ = (int)(q_ % k.Do);                                                             \
    const int n_ = (int)(q_ / k.Do);                                                             \
    const int y0_ = by_ * PY, x0_ = bx_ * PX;                                                    \
    const __amdgpu_buffer_rsrc_t xs_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(x + ((long long)n_ * k.Cin + c0) * Si), 0, (unsigned)(cgn * Si) * 4u, 0x00020000); \
    const __amdgpu_buffer_rsrc_t ds_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(dy + (long long)n_ * k.Cout * So), 0, (unsigned)(k.Cout * So) * 4u, 0x00020000);   \
    int tq_ = tid;                                                                               \
    asm volatile("" : "+v"(tq_));   /* opaque: keeps the address decode inside the patch loop */  \
    /* element e = tid + 256 i of the group's [c][hz][hy][hx] halo image; its coordinates are ADVANCED (256 = 7 rows of 33 */ \
    /* + 25), not decoded: four divisions per element were 2 200 instructions per thread and patch, more than its MFMAs   */ \
    int hx_ = tq_ % HX, r0_ = tq_ / HX, hy_ = r0_ % HY, r1_ = r0_ / HY, hz_ = r1_ % 3, c_ = r1_ / 3;  \
    _Pragma("unroll") for (int i = 0; i < NXL; ++i) {                                            \
      unsigned o = S2_OOB;                                                                       \
      {                                                                                          \
        const int gz = 2 * z_ - 1 + hz_, gy = 2 * y0_ - 1 + hy_, gx = 2 * x0_ - 1 + hx_;         \
        if (c_ < cgn && (unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W) \
          o = (unsigned)c_ * si4 + (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;                  \
      }                                                                                          \
      rxv[i] = __builtin_amdgcn_raw_buffer_load_b32(xs_, o, 0, 0);                               \
      hx_ += 256 % HX;                                                                           \
      const int cx_ = hx_ >= HX ? 1 : 0;                                                         \
      hx_ -= cx_ * HX;                                                                           \
      hy_ += 256 / HX + cx_;                                                                     \
      if (hy_ >= HY) { hy_ -= HY; ++hz_; }                                                       \
      if (hz_ >= 3) { hz_ -= 3; ++c_; }                                                          \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < ND4; ++i) {                                            \
      const int e = tid + 256 * i;              /* (co, py, x4): x4 fastest */                   \
      const int x4 = e & 3, py = (e >> 2) & 3, co = e >> 4;                                      \
      const int gy = y0_ + py, gx = x0_ + x4 * 4;                                                \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) {    /* (Wo need not be a multiple of 4: element loads) */ \
        unsigned o = S2_OOB;                                                                     \
        if (co < 16 * NCT && co < k.Cout && gy < k.Ho && gx + u < k.Wo)                          \
          o = (unsigned)co * so4 + (unsigned)((z_ * k.Ho + gy) * k.Wo + gx + u) * 4u;            \
        rd[i][u] = __builtin_amdgcn_raw_buffer_load_b32(ds_, o, 0, 0);                           \
      }                                                                                          \
    }                                                                                            \
  }
#define S2W_LSTORE()                                                                             \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < NXL; ++i) {                                            \
      const int e = tid + 256 * i;                                                               \
      if (e < nelem) Xs[e] = __uint_as_float(rxv[i]);                                            \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < ND4; ++i) {                                            \
      const int e = tid + 256 * i;                                                               \
      const int x4 = e & 3, py = (e >> 2) & 3, co = e >> 4;                                      \
      if (co < 16 * NCT) {                                                                       \
        const int p = py * PX + x4 * 4;                                                          \
        Ds[(p + 0) * DSTR + co] = __uint_as_float(rd[i].x);                                      \
        Ds[(p + 1) * DSTR + co] = __uint_as_float(rd[i].y);                                      \
        Ds[(p + 2) * DSTR + co] = __uint_as_float(rd[i].z);                                      \
        Ds[(p + 3) * DSTR + co] = __uint_as_float(rd[i].w);                                      \
      }                                                                                          \
    }                                                                                            \
  }

  if (pbeg < pend) {
    S2W_GLOAD(pbeg);
    S2W_LSTORE();
  }
  __syncthreads();
  for (long long p = pbeg; p < pend; ++p) {
    const bool more = (p + 1) < pend;
    if (more) S2W_GLOAD(p + 1);
#pragma unroll 2
    for (int kq = 0; kq < BP / 4; ++kq) {
      const int kp = 4 * kq + lk;
      const int pp = ppos[kp];
      float b[NCT];
#pragma unroll
      for (int c = 0; c < NCT; ++c) b[c] = Ds[kp * DSTR + c * 16 + l15];
      if (do_db) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) bsum[c] += b[c];
      }
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const float a = Xs[aoff[r] + pp];
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[c], acc[r][c], 0, 0, 0);
      }
    }
    if (more) {
      __syncthreads();
      S2W_LSTORE();
      __syncthreads();
    }
  }
#undef S2W_GLOAD
#undef S2W_LSTORE

  // ---- D[row = lk * 4 + e -> (tap, ci)][col = l15 -> co]
#pragma unroll
  for (int r = 0; r < RT; ++r) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int jj = (wid + 4 * r) * 16 + lk * 4 + e;
      if (jj < nrows) {
        const int tap = jj / cgn, ci = jj - tap * cgn;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const int co = c * 16 + l15;
          if (co < k.Cout) df_acc(dwt, ((long long)tap * k.Cin + c0 + ci) * k.Cout + co, acc[r][c][e], k.fx);
        }
      }
    }
  }
  if (do_db) {                                   // lane (l15 = channel, lk = voxel mod 4): fold the four voxel classes
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      float v = bsum[c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int co = c * 16 + l15;
      if (lk == 0 && co < k.Cout) df_acc(db, co, v, k.fx);
    }
  }
}
}  // namespace

const float* df_det_fx();

// dw_tcc[27][Cin][Cout] += the weight gradient of the stride-2 convolution y = conv(x) for the output gradient dy;
// db (may be NULL) [Cout] += the bias gradient, summed from the dY tiles as they pass (no separate pass over dY).
extern "C" int dfmir_conv3d_s2_wgrad(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc, float* db, void* stream) {
  DF_ARG_CHECK(g && x && dy && dw_tcc && dfmir_conv3d_s2_ok(g));
  S2wP k{};
  k.fx = df_det_fx();
  k.N = g->N; k.Cin = g->Cin; k.Cout = g->Cout; k.D = g->Di; k.H = g->Hi; k.W = g->Wi; k.Do = g->Do; k.Ho = g->Ho; k.Wo = g->Wo;
  k.ngroups = (g->Cin + 15) / 16;
  k.cg = (g->Cin + k.ngroups - 1) / k.ngroups;
  k.ny = (g->Ho + 3) / 4; k.nx = (g->Wo + 15) / 16;
  k.npatch = (long long)g->N * g->Do * k.ny * k.nx;
  long long want = 512 / k.ngroups;                          // two workgroups per CU
  if (want < 1) want = 1;
  static DfOptInt minp_o{"DFMIR_S2W_MINP", 2};               // patches per workgroup at least (every workgroup ends in 27 Cin Cout atomics)
  const long long minp = minp_o.get() > 0 ? minp_o.get() : 2;
  long long maxs = (k.npatch + minp - 1) / minp;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  k.per_block = (k.npatch + want - 1) / want;
  const unsigned nbx = (unsigned)((k.npatch + k.per_block - 1) / k.per_block);
  dim3 grid(nbx, (unsigned)k.ngroups);
  hipStream_t st = (hipStream_t)stream;
  // 27 cg <= 432 rows = 27 row tiles: 7 per wave
  if (g->Cout <= 16) conv3d_s2_wgrad_k<1, 7><<<grid, 256, 0, st>>>(x, dy, dw_tcc, db, k);
  else if (g->Cout <= 32) conv3d_s2_wgrad_k<2, 7><<<grid, 256, 0, st>>>(x, dy, dw_tcc, db, k);
  else if (g->Cout <= 48) conv3d_s2_wgrad_k<3, 7><<<grid, 256, 0, st>>>(x, dy, dw_tcc, db, k);
  else conv3d_s2_wgrad_k<4, 7><<<grid, 256, 0, st>>>(x, dy, dw_tcc, db, k);
  DF_LAUNCH_CHECK();
  return 0;
}

// ================================================================================================
// Data gradient of the stride-2 convolution = the convolution of the ZERO-DILATED output gradient (geometry as
// ConvFn.backward passes it: stride 1, dil 2, pad 1; w_tcc = the dgrad packing [27][Cd][Cx], taps flipped):
//   dx[cx][i] = sum_{t, cd} w[t][cd][cx] dyd[cd][i - 1 + t],   dyd[2 o] = dy[o], dyd[odd] = 0.
// In PARITY CLASSES of i = 2 a + p nothing multiplies an inserted zero: per axis p = 0 meets tap 1 at o = a, p = 1 meets
// tap 0 at o = a and tap 2 at o = a + 1 -- the 27 taps fall into the 8 classes as 1 + 3 x 2 + 3 x 4 + 8 products, each a
// plain small convolution of dy.  Workgroup = 2 x 4 x 16 a-positions (4 x 8 x 32 voxels of dx) x 16 TMT channels; per chunk
// of 8 dy channels the (3 x 5 x 17) dy patch and the chunk's 27 tap matrices are staged; a wave owns two a-rows and all 8
// classes of them (8 accumulator sets), reads each of the 8 shifted dy operands once per chunk and row and uses it for
// every (class, tap) that meets it; the px = 0 / 1 classes of a lane leave as one 8-byte store.
// ================================================================================================
namespace {
struct S2dP {
  int N, Cd, Cx, Dd, Hd, Wd, Dx, Hx, Wx;   // dy [N, Cd, Dd, Hd, Wd] -> dx [N, Cx, Dx, Hx, Wx]
  int naz, nay, nax;                       // a-space tiles per axis
};

template <int TMT>
__global__ __launch_bounds__(256) void conv3d_s2_dgrad_k(const float* __restrict__ dy, const float* __restrict__ wt,
                                                         float* __restrict__ dx, S2dP k) {
  constexpr int AZ = 2, AY = 4, AX = 16, HZ = AZ + 1, HY = AY + 1, HX = AX + 1;
  constexpr int XP = HZ * HY * HX;                        // 255 (odd)
  constexpr int CK = 8, BMC = 16 * TMT;
  constexpr int WSTR = (BMC % 32 == 16) ? BMC : BMC + 16;
  constexpr int W4 = 27 * CK * (BMC / 4);
  constexpr int NW = (W4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Ws[27 * CK * WSTR];
  __shared__ float Xs[CK * XP];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lk = lane >> 4;
  const int Sd = k.Dd * k.Hd * k.Wd, Sx = k.Dx * k.Hx * k.Wx;
  int pid = blockIdx.x;
  const int bx = pid % k.nax; pid /= k.nax;
  const int by = pid % k.nay; pid /= k.nay;
  const int bz = pid % k.naz;
  const int n = pid / k.naz;
  const int az0 = bz * AZ, ay0 = by * AY, ax0 = bx * AX;
  const int m0 = blockIdx.y * BMC;

  const __amdgpu_buffer_rsrc_t d_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dy + (long long)n * k.Cd * Sd), 0, (unsigned)(k.Cd * Sd) * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt), 0, (unsigned)(27 * k.Cd * k.Cx) * 4u, 0x00020000);
  const unsigned s4 = (unsigned)Sd * 4u;

  unsigned gbyte = S2_OOB;
  if (tid < XP) {
    const int hx = tid % HX, t = tid / HX, hy = t % HY, hz = t / HY;
    const int gz = az0 + hz, gy = ay0 + hy, gx = ax0 + hx;
    if (gz < k.Dd && gy < k.Hd && gx < k.Wd) gbyte = (unsigned)((gz * k.Hd + gy) * k.Wd + gx) * 4u;
  }
  unsigned wbyte[NW];
  bool wok[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int idx4 = tid + 256 * j;
    const int row = idx4 / (BMC / 4), c4 = idx4 - row * (BMC / 4);
    const int tap = row / CK, ci = row % CK, co = m0 + c4 * 4;
    wok[j] = idx4 < W4 && co < k.Cx;
    wbyte[j] = (unsigned)((tap * k.Cd + ci) * k.Cx + co) * 4u;
  }
  const unsigned wstep = (unsigned)(CK * k.Cx) * 4u;

  // the wave's two a-rows: r = 2 wid + j -> (az, ay) = (r >> 2, r & 3); operand of shift (sz, sy, sx) at + (sz HY + sy) HX + sx
  int pbase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 2 * wid + j;
    pbase[j] = ((r >> 2) * HY + (r & 3)) * HX + l15;
  }
  f32x4_s2 acc[8][TMT][2];                     // [class pz py px][channel tile][row]
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int i = 0; i < TMT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c][i][j][e] = 0.f;

  u32x4_s2 rw[NW];
  unsigned rx[CK];
#define S2D_GLOAD(c0_)                                                                           \
  {                                                                                              \
    const unsigned wadd = (unsigned)((c0_) / CK) * wstep;                                        \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                             \
      const int row_ = (tid + 256 * j) / (BMC / 4);                                              \
      const bool ok = wok[j] && ((c0_) + (row_ % CK)) < k.Cd;                                    \
      rw[j] = __builtin_amdgcn_raw_buffer_load_b128(w_src, ok ? wbyte[j] + wadd : S2_OOB, 0, 0); \
    }                                                                                            \
    const unsigned xadd = (unsigned)(c0_) * s4;                                                  \
    _Pragma("unroll") for (int c = 0; c < CK; ++c)                                               \
      rx[c] = __builtin_amdgcn_raw_buffer_load_b32(d_src, ((c0_) + c < k.Cd) ? gbyte : S2_OOB, xadd + (unsigned)c * s4, 0); \
  }
#define S2D_LSTORE()                                                                             \
  {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                             \
      const int idx4 = tid + 256 * j;                                                            \
      if (idx4 < W4) {                                                                           \
        const int row = idx4 / (BMC / 4), c4 = idx4 - row * (BMC / 4);                           \
        *reinterpret_cast<u32x4_s2*>(&Ws[row * WSTR + c4 * 4]) = rw[j];                          \
      }                                                                                          \
    }                                                                                            \
    if (tid < XP) {                                                                              \
      _Pragma("unroll") for (int c = 0; c < CK; ++c) Xs[c * XP + tid] = __uint_as_float(rx[c]);  \
    }                                                                                            \
  }

  S2D_GLOAD(0);
  S2D_LSTORE();
  __syncthreads();
  for (int c0 = 0; c0 < k.Cd; c0 += CK) {
    const bool more = (c0 + CK) < k.Cd;
    if (more) S2D_GLOAD(c0 + CK);
#pragma unroll
    for (int kq = 0; kq < CK / 4; ++kq) {
      const int kr = 4 * kq + lk;
      float b[8][2];                           // [shift sz sy sx][row]
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[s][j] = Xs[kr * XP + pbase[j] + (((s >> 2) & 1) * HY + ((s >> 1) & 1)) * HX + (s & 1)];
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
        const int cls = ((tz != 1) << 2) | ((ty != 1) << 1) | (tx != 1);
        const int sh = ((tz == 2) << 2) | ((ty == 2) << 1) | (tx == 2);
        float a[TMT];
#pragma unroll
        for (int i = 0; i < TMT; ++i) a[i] = Ws[(tap * CK + kr) * WSTR + i * 16 + l15];
#pragma unroll
        for (int i = 0; i < TMT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[cls][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[sh][j], acc[cls][i][j], 0, 0, 0);
      }
    }
    if (more) {
      __syncthreads();
      S2D_LSTORE();
      __syncthreads();
    }
  }
#undef S2D_GLOAD
#undef S2D_LSTORE

  // ---- epilogue: D[row = lk * 4 + e -> channel][col = l15 -> ax]; voxel (2 az + pz, 2 ay + py, 2 ax + px)
  const int gx = 2 * (ax0 + l15);
  const bool pair = (k.Wx & 1) == 0 && (reinterpret_cast<uintptr_t>(dx) & 7) == 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 2 * wid + j;
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const int gz = 2 * (az0 + (r >> 2)) + pz, gy = 2 * (ay0 + (r & 3)) + py;
        if (gz >= k.Dx || gy >= k.Hx || gx >= k.Wx) continue;
        float* ob = dx + (long long)n * k.Cx * Sx + ((long long)gz * k.Hx + gy) * k.Wx + gx;
#pragma unroll
        for (int i = 0; i < TMT; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cx = m0 + i * 16 + lk * 4 + e;
            if (cx >= k.Cx) continue;
            const float v0 = acc[(pz << 2) | (py << 1)][i][j][e], v1 = acc[(pz << 2) | (py << 1) | 1][i][j][e];
            float* o = ob + (long long)cx * Sx;
            if (pair) *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
            else {
              o[0] = v0;
              if (gx + 1 < k.Wx) o[1] = v1;
            }
          }
      }
  }
}

bool s2d_geom_ok(const DfConvGeom* g) {
  // the dgrad call of a 3x3x3 stride-2 pad-1 convolution: in = dy, out = dx with Do = the forward input's size
  return g->KD == 3 && g->KH == 3 && g->KW == 3 && g->stride == 1 && g->dil == 2 && g->pd == 1 && g->ph == 1 && g->pw == 1 &&
         g->pad_mode == 0 && g->act == 0 && g->Di > 1 && g->Di == (g->Do + 1) / 2 && g->Hi == (g->Ho + 1) / 2 &&
         g->Wi == (g->Wo + 1) / 2 && (long long)g->Di * g->Hi * g->Wi > S2_MIN_VOX && g->Cin >= 8 && (g->Cin % 4) == 0 && g->Cout >= 16 && (g->Cout % 16) == 0 && g->Cout <= 64 &&
         (long long)(g->Cin > g->Cout ? g->Cin : g->Cout) * g->Do * g->Ho * g->Wo * 4 < 0x7FFFFFFFLL;
}
}  // namespace

extern "C" int dfmir_conv3d_s2_dgrad_ok(const DfConvGeom* g) { return (g && !s2m_off() && s2d_geom_ok(g)) ? 1 : 0; }

// dx = the data gradient of a 3x3x3 stride-2 pad-1 convolution; g = the geometry of the dgrad call (Cin = channels of dy,
// Cout = channels of dx, stride 1, dil 2), w_tcc = the dgrad packing [27][Cin][Cout] (dfmir_weight_pack mode 1).
extern "C" int dfmir_conv3d_s2_dgrad(const DfConvGeom* g, const float* dy, const float* w_tcc, float* dx, void* stream) {
  DF_ARG_CHECK(g && dy && w_tcc && dx && dfmir_conv3d_s2_dgrad_ok(g));
  DF_ARG_CHECK((reinterpret_cast<uintptr_t>(w_tcc) & 15) == 0);
  S2dP k{g->N, g->Cin, g->Cout, g->Di, g->Hi, g->Wi, g->Do, g->Ho, g->Wo, 0, 0, 0};
  k.naz = ((g->Do + 1) / 2 + 1) / 2; k.nay = ((g->Ho + 1) / 2 + 3) / 4; k.nax = ((g->Wo + 1) / 2 + 15) / 16;
  const long long tiles = (long long)g->N * k.naz * k.nay * k.nax;
  DF_ARG_CHECK(tiles < (1LL << 30));
  hipStream_t st = (hipStream_t)stream;
  // one 16-channel tile per workgroup: the 8 class accumulators of two tiles take 400 registers, and the dy patch a second
  // workgroup re-stages is 8 KB
  dim3 grid((unsigned)tiles, (unsigned)(g->Cout / 16));
  conv3d_s2_dgrad_k<1><<<grid, 256, 0, st>>>(dy, w_tcc, dx, k);
  DF_LAUNCH_CHECK();
  return 0;
}

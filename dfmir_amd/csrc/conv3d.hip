// 3-D 3x3x3 stride-1 convolutions of the VoxelMorph U-Net (small channel counts: 16/32/34/48/64 in,
// 16/32/34 out; up to 6.9 M voxels) on v_mfma_f32_16x16x4_f32, LDS-resident and software-pipelined.
//
// 16-row MFMA tiles fit these channel counts (16, 32, 48 exactly; 34 -> 48) where the 32x32 form would
// idle half the array.  forward / dgrad stage a (2+2)x(8+2)x(16+2) halo patch per 8-channel chunk and
// read it at the 27 tap offsets.  wgrad flattens (tap, ci) into the MFMA row axis (27*Cin rows, padded
// to 16 -- 1 % waste at Cin = 34) and reads each row's operand straight from the halo patch at its own
// tap offset; the workgroup keeps the whole [27*Cin, Cout] tile of its channel group in accumulators.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct C3dP {
  int N, Cin, Cout, D, H, W;     // stride 1, pad 1: output frame == input frame
  int act;
  float slope;
  int nz, ny, nx;                // patches per axis
};

// ---------------------------------------------------------------------------------------------
// forward / dgrad.  Patch 2 x 8 x 16 = 256 voxels; wave w owns x-rows 4w..4w+3 of the patch.
// ---------------------------------------------------------------------------------------------
template <int TMT>
__global__ __launch_bounds__(256) void conv3d_mfma16_k(const float* __restrict__ x,
                                                       const float* __restrict__ wt,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ y, C3dP k) {
  constexpr int PZ = 2, PY = 8, PX = 16, HY = PY + 2, HX = PX + 2;
  constexpr int XP = (PZ + 2) * HY * HX;                  // 720; 720 % 32 == 16 -> k-groups hit disjoint banks
  constexpr int CK = 8, BMC = 16 * TMT;
  constexpr int WSTR = (BMC % 32 == 16) ? BMC : BMC + 16; // row stride == 16 (mod 32)
  constexpr int NS = (XP + 255) / 256;                    // 3
  constexpr int W4 = 27 * CK * (BMC / 4);                 // float4 per weight chunk
  constexpr int NW = (W4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Ws[27 * CK * WSTR];
  __shared__ float Xs[CK * XP];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lk = lane >> 4;
  const int S = k.D * k.H * k.W;
  int pid = blockIdx.x;
  const int bx = pid % k.nx; pid /= k.nx;
  const int by = pid % k.ny; pid /= k.ny;
  const int bz = pid % k.nz;
  const int n = pid / k.nz;
  const int z0 = bz * PZ, y0 = by * PY, x0 = bx * PX;
  const int m0 = blockIdx.y * BMC;

  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t x_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x + (long long)n * k.Cin * S), 0, (unsigned)(k.Cin * S) * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt), 0, (unsigned)(27 * k.Cin * k.Cout) * 4u, 0x00020000);
  const unsigned s4 = (unsigned)S * 4u;

  unsigned gbyte[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int pos = tid + 256 * s;
    unsigned off = OOB;
    if (pos < XP) {
      const int hx = pos % HX, t = pos / HX, hy = t % HY, hz = t / HY;
      const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W)
        off = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;
    }
    gbyte[s] = off;
  }
  const bool vec4 = (k.Cout & 3) == 0;
  unsigned wbyte[NW];
  bool wok[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int idx4 = tid + 256 * j;
    const int row = idx4 / (BMC / 4), c4 = idx4 - row * (BMC / 4);
    const int tap = row >> 3, ci = row & 7, co = m0 + c4 * 4;
    wok[j] = idx4 < W4 && co < k.Cout;
    wbyte[j] = (unsigned)((tap * k.Cin + ci) * k.Cout + co) * 4u;
  }
  const unsigned wstep = (unsigned)(CK * k.Cout) * 4u;

  int pbase[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wid * 4 + j;                 // x-row of the patch: (pz, py)
    pbase[j] = ((r >> 3) * HY + (r & 7)) * HX + l15;
  }

  f32x4 acc[TMT][4];
#pragma unroll
  for (int i = 0; i < TMT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  u32x4 rw[NW];
  unsigned rx[NS][CK];

#define C3D_GLOAD(ci0_)                                                                          \
  {                                                                                              \
    const unsigned wadd = (unsigned)((ci0_) / CK) * wstep;                                       \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                             \
      const int row_ = (tid + 256 * j) / (BMC / 4);                                              \
      const bool ok = wok[j] && ((ci0_) + (row_ & 7)) < k.Cin;                                   \
      const unsigned o = ok ? wbyte[j] + wadd : OOB;                                             \
      if (vec4) {                                                                                \
        rw[j] = __builtin_amdgcn_raw_buffer_load_b128(w_src, o, 0, 0);                           \
      } else {                                                                                   \
        const int co = m0 + ((tid + 256 * j) % (BMC / 4)) * 4;                                   \
        rw[j].x = __builtin_amdgcn_raw_buffer_load_b32(w_src, o, 0, 0);                          \
        rw[j].y = __builtin_amdgcn_raw_buffer_load_b32(w_src, (ok && co + 1 < k.Cout) ? o + 4u : OOB, 0, 0);  \
        rw[j].z = __builtin_amdgcn_raw_buffer_load_b32(w_src, (ok && co + 2 < k.Cout) ? o + 8u : OOB, 0, 0);  \
        rw[j].w = __builtin_amdgcn_raw_buffer_load_b32(w_src, (ok && co + 3 < k.Cout) ? o + 12u : OOB, 0, 0); \
      }                                                                                          \
    }                                                                                            \
    const unsigned xadd = (unsigned)(ci0_) * s4;                                                 \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                             \
      _Pragma("unroll") for (int c = 0; c < CK; ++c)                                             \
        rx[s][c] = __builtin_amdgcn_raw_buffer_load_b32(x_src, gbyte[s] + xadd + (unsigned)c * s4, 0, 0); \
    }                                                                                            \
  }
#define C3D_LSTORE()                                                                             \
  {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                             \
      const int idx4 = tid + 256 * j;                                                            \
      if (idx4 < W4) {                                                                           \
        const int row = idx4 / (BMC / 4), c4 = idx4 - row * (BMC / 4);                           \
        *reinterpret_cast<u32x4*>(&Ws[row * WSTR + c4 * 4]) = rw[j];                             \
      }                                                                                          \
    }                                                                                            \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                             \
      const int pos = tid + 256 * s;                                                             \
      if (pos < XP) {                                                                            \
        _Pragma("unroll") for (int c = 0; c < CK; ++c) Xs[c * XP + pos] = __uint_as_float(rx[s][c]); \
      }                                                                                          \
    }                                                                                            \
  }

  C3D_GLOAD(0);
  C3D_LSTORE();
  __syncthreads();

  for (int ci0 = 0; ci0 < k.Cin; ci0 += CK) {
    const bool more = (ci0 + CK) < k.Cin;
    if (more) C3D_GLOAD(ci0 + CK);
#pragma unroll 3
    for (int tap = 0; tap < 27; ++tap) {
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      const int toff = (dz * HY + dy) * HX + dx;
#pragma unroll
      for (int kq = 0; kq < CK / 4; ++kq) {
        const int kr = 4 * kq + lk;
        float a[TMT], b[4];
#pragma unroll
        for (int i = 0; i < TMT; ++i) a[i] = Ws[(tap * CK + kr) * WSTR + i * 16 + l15];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Xs[kr * XP + pbase[j] + toff];
#pragma unroll
        for (int i = 0; i < TMT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      __syncthreads();
      C3D_LSTORE();
      __syncthreads();
    }
  }
#undef C3D_GLOAD
#undef C3D_LSTORE

  // ---- epilogue: D[row = (lane>>4)*4 + r -> cout][col = lane&15 -> x]
  const int gx = x0 + l15;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wid * 4 + j;
    const int gz = z0 + (r >> 3), gy = y0 + (r & 7);
    if (gz >= k.D || gy >= k.H || gx >= k.W) continue;
    float* yb = y + (long long)n * k.Cout * S + ((long long)gz * k.H + gy) * k.W + gx;
#pragma unroll
    for (int i = 0; i < TMT; ++i) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = m0 + i * 16 + lk * 4 + rr;
        if (co < k.Cout) {
          float v = acc[i][j][rr] + (bias ? bias[co] : 0.f);
          if (k.act == 1) v = v > 0.f ? v : v * k.slope;
          else if (k.act == 2) v = tanhf(v);
          yb[(long long)co * S] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad.  Rows = flattened (tap, ci_local) of one channel group (cg channels), cols = Cout (16*NCT),
// reduction over 1 x 4 x 16 voxel patches.  Wave w owns row tiles w, w+4, ...
// ---------------------------------------------------------------------------------------------
struct W3dP {
  int N, Cin, Cout, D, H, W;
  int cg, ngroups;               // channels per group, number of groups (blockIdx.y)
  int ny, nx;                    // patches per axis (nz == D)
  long long npatch, per_block;   // total patches (N*D*ny*nx) and patches per block
  const float* fx;               // deterministic mode (common.h df_acc)
};

template <int NCT, int RT>
__global__ __launch_bounds__(256) void conv3d_wgrad16_k(const float* __restrict__ x,
                                                        const float* __restrict__ dy,
                                                        float* __restrict__ dwt, W3dP k) {
  constexpr int PY = 4, PX = 16, BP = PY * PX, HY = PY + 2, HX = PX + 2;
  constexpr int NPOS = 3 * HY * HX;                 // 324
  constexpr int XP = NPOS;                          // dense [c][pos] image: staged element e lands at Xs[e]
  constexpr int DSTR = (NCT == 1) ? 16 : 16 * NCT + 16;
  constexpr int CGMAX = 32;
  constexpr int NXL = (CGMAX * NPOS + 255) / 256;   // halo loads per thread (41)
  constexpr int ND4 = (BP / 4) * (16 * NCT) / 256 > 0 ? (BP / 4) * (16 * NCT) / 256 : 1;
  __shared__ float Xs[CGMAX * XP];
  __shared__ float Ds[BP * DSTR];
  __shared__ int ppos[BP];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lk = lane >> 4;
  const int S = k.D * k.H * k.W;
  const int grp = blockIdx.y;
  const int c0 = grp * k.cg;
  const int cgn = (k.Cin - c0 < k.cg) ? k.Cin - c0 : k.cg;   // channels in this group
  const int nrows = 27 * cgn;
  const long long pbeg = (long long)blockIdx.x * k.per_block;
  long long pend = pbeg + k.per_block;
  if (pend > k.npatch) pend = k.npatch;

  // per-lane operand offsets of this wave's row tiles
  int aoff[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int jj = (wid + 4 * r) * 16 + l15;
    int off = 0;
    if (jj < nrows) {
      const int tap = jj / cgn, ci = jj - tap * cgn;
      const int dz = tap / 9, dyy = (tap / 3) % 3, dx = tap % 3;
      off = ci * XP + (dz * HY + dyy) * HX + dx;
    }
    aoff[r] = off;
  }
  if (tid < BP) ppos[tid] = ((tid >> 4) * HX) + (tid & 15);   // voxel (py, px) -> patch position (dz=dy=dx=0 corner)

  f32x4 acc[RT][NCT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[r][c][e] = 0.f;

  constexpr unsigned OOB = 0x80000000u;
  const unsigned s4 = (unsigned)S * 4u;
  unsigned rxv[NXL];
  u32x4 rd[ND4];
  const int nelem = cgn * NPOS;

#define W3D_GLOAD(p_)                                                                            \
  {                                                                                              \
    long long q_ = (p_);                                                                         \
    const int bx_ = (int)(q_ % k.nx); q_ /= k.nx;                                                \
    const int by_ = (int)(q_ % k.ny); q_ /= k.ny;                                                \
    const int z_ = (int)(q_ % k.D);                                                              \
    const int n_ = (int)(q_ / k.D);                                                              \
    const int y0_ = by_ * PY, x0_ = bx_ * PX;                                                    \
    const __amdgpu_buffer_rsrc_t xs_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(x + ((long long)n_ * k.Cin + c0) * S), 0, (unsigned)(cgn * S) * 4u, 0x00020000); \
    const __amdgpu_buffer_rsrc_t ds_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(dy + (long long)n_ * k.Cout * S), 0, (unsigned)(k.Cout * S) * 4u, 0x00020000);   \
    int tq_ = tid;                                                                               \
    asm volatile("" : "+v"(tq_));   /* opaque: keeps the address decode inside the patch loop */  \
    _Pragma("unroll") for (int i = 0; i < NXL; ++i) {                                            \
      const int e = tq_ + 256 * i;                                                               \
      unsigned o = OOB;                                                                          \
      if (e < nelem) {                                                                           \
        const int c = e / NPOS, pos = e - c * NPOS;                                              \
        const int hx = pos % HX, t = pos / HX, hy = t % HY, hz = t / HY;                         \
        const int gz = z_ - 1 + hz, gy = y0_ - 1 + hy, gx = x0_ - 1 + hx;                        \
        if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W) \
          o = (unsigned)c * s4 + (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;                    \
      }                                                                                          \
      rxv[i] = __builtin_amdgcn_raw_buffer_load_b32(xs_, o, 0, 0);                               \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < ND4; ++i) {                                            \
      const int e = tid + 256 * i;              /* (co, py, x4): x4 fastest */                   \
      const int x4 = e & 3, py = (e >> 2) & 3, co = e >> 4;                                      \
      const int gy = y0_ + py, gx = x0_ + x4 * 4;                                                \
      unsigned o = OOB;                                                                          \
      if (co < 16 * NCT && co < k.Cout && gy < k.H && gx < k.W)                                  \
        o = (unsigned)co * s4 + (unsigned)((z_ * k.H + gy) * k.W + gx) * 4u;                     \
      rd[i] = __builtin_amdgcn_raw_buffer_load_b128(ds_, o, 0, 0);                               \
    }                                                                                            \
  }
#define W3D_LSTORE()                                                                             \
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
    W3D_GLOAD(pbeg);
    W3D_LSTORE();
  }
  __syncthreads();
  // centre of the halo patch for tap (1,1,1) is (1*HY+1)*HX+1; taps add (dz*HY+dy)*HX+dx to the corner
  for (long long p = pbeg; p < pend; ++p) {
    const bool more = (p + 1) < pend;
    if (more) W3D_GLOAD(p + 1);
#pragma unroll 2
    for (int kq = 0; kq < BP / 4; ++kq) {
      const int kp = 4 * kq + lk;
      const int pp = ppos[kp];
      float b[NCT];
#pragma unroll
      for (int c = 0; c < NCT; ++c) b[c] = Ds[kp * DSTR + c * 16 + l15];
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const float a = Xs[aoff[r] + pp];
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[c], acc[r][c], 0, 0, 0);
      }
    }
    if (more) {
      __syncthreads();
      W3D_LSTORE();
      __syncthreads();
    }
  }
#undef W3D_GLOAD
#undef W3D_LSTORE

  // ---- D[row = lk*4 + e -> (tap, ci)][col = l15 -> co]
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
}

// ---------------------------------------------------------------------------------------------
static bool is_3x3x3_s1_p1(const DfConvGeom* g) {
  return g->KD == 3 && g->KH == 3 && g->KW == 3 && g->stride == 1 && g->dil == 1 && g->pd == 1 && g->ph == 1 &&
         g->pw == 1 && g->pad_mode == 0 && g->Do == g->Di && g->Ho == g->Hi && g->Wo == g->Wi && g->Di > 1;
}

bool df_conv3d_fwd_try(const DfConvGeom* g, const float* x, const float* w_tcc, const float* bias, float* y,
                       hipStream_t st, int* rc) {
  if (!is_3x3x3_s1_p1(g) || (g->Cout <= 4 && g->Cin < 8)) return false;   // 16->3 flow conv: one 16-row tile, 19 % used, still 3x the direct kernel
  const long long S = (long long)g->Di * g->Hi * g->Wi;
  if ((long long)g->Cin * S * 4 >= 0x7FFFFFFFLL || (long long)g->Cout * S * 4 >= 0x7FFFFFFFLL) return false;
  C3dP k{g->N, g->Cin, g->Cout, g->Di, g->Hi, g->Wi, g->act, g->slope, (g->Di + 1) / 2, (g->Hi + 7) / 8, (g->Wi + 15) / 16};
  const long long nb = (long long)g->N * k.nz * k.ny * k.nx;
  if (nb >= (1LL << 31)) return false;
  const int tiles = (g->Cout + 15) / 16;
  if (tiles == 1) conv3d_mfma16_k<1><<<dim3((unsigned)nb, 1), 256, 0, st>>>(x, w_tcc, bias, y, k);
  else if (tiles == 2) conv3d_mfma16_k<2><<<dim3((unsigned)nb, 1), 256, 0, st>>>(x, w_tcc, bias, y, k);
  else if (tiles == 3) conv3d_mfma16_k<3><<<dim3((unsigned)nb, 1), 256, 0, st>>>(x, w_tcc, bias, y, k);
  else conv3d_mfma16_k<4><<<dim3((unsigned)nb, (unsigned)((tiles + 3) / 4)), 256, 0, st>>>(x, w_tcc, bias, y, k);
  hipError_t e = hipGetLastError();
  *rc = (e == hipSuccess) ? 0 : df_set_error((int)e, __FILE__, __LINE__);
  return true;
}

bool df_conv3d_wgrad_try(const DfConvGeom* g, const float* x, const float* dy, float* dw_tcc, hipStream_t st,
                         int* rc) {
  if (!is_3x3x3_s1_p1(g) || g->Cout > 32) return false;   // incl. the 16->3 flow conv (padded to one 16-col tile)
  const long long S = (long long)g->Di * g->Hi * g->Wi;
  if ((long long)g->Cin * S * 4 >= 0x7FFFFFFFLL || (long long)g->Cout * S * 4 >= 0x7FFFFFFFLL) return false;
  if ((g->Wi & 3) != 0) return false;                       // float4 dY loads
  W3dP k{};
  k.fx = df_det_fx();
  k.N = g->N; k.Cin = g->Cin; k.Cout = g->Cout; k.D = g->Di; k.H = g->Hi; k.W = g->Wi;
  k.ngroups = (g->Cin + 31) / 32;
  k.cg = (g->Cin + k.ngroups - 1) / k.ngroups;
  k.ny = (g->Hi + 3) / 4; k.nx = (g->Wi + 15) / 16;
  k.npatch = (long long)g->N * g->Di * k.ny * k.nx;
  long long want = 512 / k.ngroups;
  if (want < 1) want = 1;
  long long maxs = (k.npatch + 3) / 4;
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  k.per_block = (k.npatch + want - 1) / want;
  const unsigned nx = (unsigned)((k.npatch + k.per_block - 1) / k.per_block);
  const int rtiles = (27 * k.cg + 15) / 16;                 // <= 54
  const int rt = (rtiles + 3) / 4;                          // row tiles per wave, <= 14
  dim3 grid(nx, (unsigned)k.ngroups);
  const bool two = g->Cout > 16;
#define W3D_LAUNCH(NCT_, RT_) conv3d_wgrad16_k<NCT_, RT_><<<grid, 256, 0, st>>>(x, dy, dw_tcc, k)
  if (two) {
    if (rt <= 8) W3D_LAUNCH(2, 8); else W3D_LAUNCH(2, 14);
  } else {
    if (rt <= 8) W3D_LAUNCH(1, 8); else W3D_LAUNCH(1, 14);
  }
#undef W3D_LAUNCH
  hipError_t e = hipGetLastError();
  *rc = (e == hipSuccess) ? 0 : df_set_error((int)e, __FILE__, __LINE__);
  return true;
}

// LDS-windowed displacement-field warps (W % 4 == 0, linear mode): the "coalesced HBM gather" form of
// SpatialTransformer.forward / its adjoint (torchvoxelmorph/layers.py:36-48).
//
// A 512-thread workgroup owns a tile of output voxels (3-D: 8 x 8 x 32, 2-D: 64 x 32), 4 x-consecutive
// voxels per thread.  It reads their displacements (one 16-B load per axis), block-reduces the minimum
// sample corner, and copies the src window that starts there (tile + 4 planes/rows, 40 columns) into
// LDS with 16-B hardware-bounds-checked buffer loads -- rows and planes outside the frame read 0.0,
// which is grid_sample's zero padding, so the window path needs no masks at all.  The 8 (4) corner taps
// of every voxel then come from LDS at full ds_read rate instead of as 32 scattered dword gathers per
// lane through the texture path (46 clk per wave instruction, 69 % L2 miss: profiles/r01_warp_pmc.md).
// Because the window follows the block's minimum sample corner it tracks any bulk displacement; a voxel
// whose taps still leave it (the field varies by more than ~3 voxels inside one tile) is redone start
// to finish by the scalar per-voxel routine at the end of the kernel, so every field is handled.
//
// The backward kernel gathers the same window for d(flow), then privatises d(src): each workgroup
// scatter-adds into a zeroed LDS window (ds_add_f32) and flushes the touched cells with one coalesced
// global atomic each -- about 1.3 device-scope atomics per voxel and channel instead of 8.
//
// LDS layout: natural x order, row stride 41 words (== 9 mod 32).  A half-wave is 8 x-quads x 4 rows;
// its lanes read words 4*xq + 41*row + shift, i.e. banks {4i + 9r}: all 32 distinct.
#include "common.h"

typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));

template <int ND> struct WinGeom;
template <> struct WinGeom<3> { static constexpr int TZ = 8, TY = 8, EZ = 12, EY = 12; };
template <> struct WinGeom<2> { static constexpr int TZ = 1, TY = 64, EZ = 1, EY = 68; };
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
static constexpr int WTX = 32;        // tile width (8 lanes x 4 voxels)
static constexpr int WEX = 40;        // window width
static constexpr int WRS = 41;        // LDS row stride (== 9 mod 32: see the bank note above)
static constexpr int WQ = WEX / 4;    // 16-B units per window row
static constexpr int WNT = 512;       // threads per workgroup (8 waves)
static constexpr unsigned WOOB = 0x80000000u;

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}

// Per-thread sample geometry of its 4 x-consecutive voxels; `fast` bit e = every corner of voxel e is
// inside the LDS window.
template <int ND>
struct WinThread {
  int base[4];                 // LDS index of corner (z0,y0,x0)
  float wz1[4], wy1[4], wx1[4];
  unsigned fast;
};

// tile decode with an XCD-contiguous remap: consecutive workgroup ids round-robin over the 8 XCDs, so
// id -> (id % 8) * ceil(T/8) + id / 8 hands every XCD (and its L2) one contiguous run of tiles.
__device__ __forceinline__ bool win_tile(int ntx, int nty, int ntz, int B, int& tx, int& tyb, int& tzb, int& b) {
  const int T = ntx * nty * ntz * B;
  const int per = (T + 7) >> 3;
  int L = (int)blockIdx.x;
  L = (L & 7) * per + (L >> 3);
  if (L >= T) return false;
  tx = L % ntx; L /= ntx;
  tyb = L % nty; L /= nty;
  tzb = L % ntz;
  b = L / ntz;
  return true;
}

template <int ND>
__device__ __forceinline__ void win_prologue(const float* __restrict__ flow, int b, int S, int H, int W, int D,
                                             int z, int y, int x, bool active, bool need_own, WinThread<ND>& th,
                                             int* red, int& oz, int& oy, int& ox, bool neigh = false, int tz0 = 0,
                                             int ty0 = 0, int tx0 = 0) {
  using G = WinGeom<ND>;
  const int sp = (z * H + y) * W + x;
  const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(flow + (long long)b * ND * S), 0, (unsigned)(ND * S) * 4u, 0x00020000);
  wu32x4 f[3];
#pragma unroll
  for (int k = 0; k < ND; ++k)
    f[k] = __builtin_amdgcn_raw_buffer_load_b128(rf, active ? (unsigned)(k * S + sp) * 4u : WOOB, 0, 0);
  int z0[4], y0[4], x0[4];
  int mz = 0x7fffffff, my = 0x7fffffff, mx = 0x7fffffff;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float fz = (ND == 3) ? (float)z + __uint_as_float(f[0][e]) : 0.f;
    const float fy = (float)y + __uint_as_float(f[ND - 2][e]);
    const float fx = (float)(x + e) + __uint_as_float(f[ND - 1][e]);
    const float z0f = floorf(fz), y0f = floorf(fy), x0f = floorf(fx);
    z0[e] = (ND == 3) ? (int)z0f : 0;
    y0[e] = (int)y0f;
    x0[e] = (int)x0f;
    th.wz1[e] = (ND == 3) ? fz - z0f : 0.f;
    th.wy1[e] = fy - y0f;
    th.wx1[e] = fx - x0f;
    if (active) {   // corners below -1 / above size-1 only ever read zeros: clamp them out of the minimum
      if (ND == 3) mz = min(mz, max(-1, min(z0[e], D - 1)));
      my = min(my, max(-1, min(y0[e], H - 1)));
      mx = min(mx, max(-1, min(x0[e], W - 1)));
    }
  }
  if (need_own && active) {   // backward with an identity / self term: keep the tile itself in the window
    if (ND == 3) mz = min(mz, z);
    my = min(my, y); mx = min(mx, x);
  }
  mz = wave_min_i(mz); my = wave_min_i(my); mx = wave_min_i(mx);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w * 3] = mz; red[w * 3 + 1] = my; red[w * 3 + 2] = mx; }
  __syncthreads();
  mz = red[0]; my = red[1]; mx = red[2];
#pragma unroll
  for (int k = 1; k < WNT / 64; ++k) { mz = min(mz, red[3 * k]); my = min(my, red[3 * k + 1]); mx = min(mx, red[3 * k + 2]); }
  if (neigh) {   // owner-gather backward: the window must stay inside the 3x3(x3) tiles around its own tile
    if (ND == 3) mz = max(tz0 - G::TZ, min(mz, tz0 + 2 * G::TZ - G::EZ));
    my = max(ty0 - G::TY, min(my, ty0 + 2 * G::TY - G::EY));
    mx = max(tx0 - WTX, min(mx, tx0 + 2 * WTX - WEX));
  }
  oz = (ND == 3) ? mz : 0;
  oy = my;
  ox = mx & ~3;
  th.fast = 0u;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int lz = z0[e] - oz, ly = y0[e] - oy, lx = x0[e] - ox;
    bool inw = (ND == 2 || (unsigned)lz <= (unsigned)(G::EZ - 2)) && (unsigned)ly <= (unsigned)(G::EY - 2) &&
               (unsigned)lx <= (unsigned)(WEX - 2);
    if (need_own)
      inw = inw && (unsigned)(z - oz) < (unsigned)G::EZ && (unsigned)(y - oy) < (unsigned)G::EY &&
            (unsigned)(x + e - ox) < (unsigned)WEX;
    th.base[e] = inw ? (lz * G::EY + ly) * WRS + lx : 0;
    if (inw && active) th.fast |= 1u << e;
  }
}

// copy the src window at (oz,oy,ox) of one channel plane into LDS (zeros outside the frame)
template <int ND>
__device__ __forceinline__ void win_load(const __amdgpu_buffer_rsrc_t rs, float* __restrict__ win, int D, int H,
                                         int W, int oz, int oy, int ox) {
  using G = WinGeom<ND>;
  constexpr int NU = G::EZ * G::EY * WQ, IT = (NU + WNT - 1) / WNT;
  wu32x4 v[IT];
  int li[IT];
  // unit u = t + i*WNT -> (row r = u / WQ, qx = u % WQ), advanced incrementally (WNT = 51*WQ + 2)
  int qx = (int)threadIdx.x % WQ, r = (int)threadIdx.x / WQ;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int ry = r % G::EY, rz = r / G::EY;
    const int gx = ox + 4 * qx, gy = oy + ry, gz = oz + rz;
    const bool ok = r < G::EZ * G::EY && (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H &&
                    (unsigned)gz < (unsigned)D;
    v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)((gz * H + gy) * W + gx) * 4u : WOOB, 0, 0);
    li[i] = r < G::EZ * G::EY ? r * WRS + 4 * qx : -1;
    qx += WNT % WQ; r += WNT / WQ;
    if (qx >= WQ) { qx -= WQ; r += 1; }
  }
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    if (li[i] >= 0) {
      float* p = win + li[i];
      p[0] = __uint_as_float(v[i][0]); p[1] = __uint_as_float(v[i][1]);
      p[2] = __uint_as_float(v[i][2]); p[3] = __uint_as_float(v[i][3]);
    }
  }
}

// ---- scalar per-voxel routines for the (rare) voxels whose taps leave the window; same arithmetic as
// the one-voxel-per-thread kernels in warp.hip
template <int ND>
struct VoxelTaps {
  int z0, y0, x0;
  float wz1, wy1, wx1;
  bool zv[2], yv[2], xv[2];
};
template <int ND>
__device__ __forceinline__ VoxelTaps<ND> voxel_taps(const float* __restrict__ fb, int S, int sp, int z, int y, int x,
                                                    int D, int H, int W) {
  VoxelTaps<ND> v;
  const float fz = (ND == 3) ? (float)z + fb[sp] : 0.f;
  const float fy = (float)y + fb[(long long)(ND - 2) * S + sp];
  const float fx = (float)x + fb[(long long)(ND - 1) * S + sp];
  const float z0f = floorf(fz), y0f = floorf(fy), x0f = floorf(fx);
  v.z0 = (ND == 3) ? (int)z0f : 0; v.y0 = (int)y0f; v.x0 = (int)x0f;
  v.wz1 = (ND == 3) ? fz - z0f : 0.f; v.wy1 = fy - y0f; v.wx1 = fx - x0f;
  v.zv[0] = (ND == 2) || (unsigned)v.z0 < (unsigned)D; v.zv[1] = (ND == 3) && (unsigned)(v.z0 + 1) < (unsigned)D;
  v.yv[0] = (unsigned)v.y0 < (unsigned)H; v.yv[1] = (unsigned)(v.y0 + 1) < (unsigned)H;
  v.xv[0] = (unsigned)v.x0 < (unsigned)W; v.xv[1] = (unsigned)(v.x0 + 1) < (unsigned)W;
  return v;
}

template <int ND>
__device__ void voxel_fwd_slow(const float* __restrict__ src_b, const float* __restrict__ flow_b,
                               float* __restrict__ out_b, int C, int D, int H, int W, int S, int z, int y, int x,
                               int add_identity) {
  const int sp = (z * H + y) * W + x;
  const VoxelTaps<ND> v = voxel_taps<ND>(flow_b, S, sp, z, y, x, D, H, W);
  const int o = (v.z0 * H + v.y0) * W + v.x0, HW = H * W;
  for (int c = 0; c < C; ++c) {
    const float* sc = src_b + (long long)c * S;
    float acc = 0.f;
#pragma unroll
    for (int kz = ND - 2; kz >= 0; --kz) {
      float pl = 0.f;
#pragma unroll
      for (int ky = 1; ky >= 0; --ky) {
        const bool rv = v.zv[kz] && v.yv[ky];
        const float a0 = (rv && v.xv[0]) ? sc[o + kz * HW + ky * W] : 0.f;
        const float a1 = (rv && v.xv[1]) ? sc[o + kz * HW + ky * W + 1] : 0.f;
        const float row = (1.f - v.wx1) * a0 + v.wx1 * a1;
        pl = ky ? v.wy1 * row : (1.f - v.wy1) * row + pl;
      }
      acc = (ND == 3) ? (kz ? v.wz1 * pl : (1.f - v.wz1) * pl + acc) : pl;
    }
    if (add_identity) acc += sc[sp];
    out_b[(long long)c * S + sp] = acc;
  }
}

template <int ND>
__device__ void voxel_bwd_slow(const float* __restrict__ dout_b, const float* __restrict__ src_b,
                               const float* __restrict__ flow_b, float* __restrict__ dsrc_b,
                               float* __restrict__ dflow_b, int C, int D, int H, int W, int S, int z, int y, int x,
                               int add_identity, int flow_into_src) {
  const int sp = (z * H + y) * W + x;
  const VoxelTaps<ND> v = voxel_taps<ND>(flow_b, S, sp, z, y, x, D, H, W);
  const int o = (v.z0 * H + v.y0) * W + v.x0, HW = H * W;
  const float wz[2] = {1.f - v.wz1, v.wz1}, wy[2] = {1.f - v.wy1, v.wy1}, wx[2] = {1.f - v.wx1, v.wx1};
  float gz = 0.f, gy = 0.f, gx = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = dout_b[(long long)c * S + sp];
    const float* sc = src_b + (long long)c * S;
    float* dc = dsrc_b ? dsrc_b + (long long)c * S : nullptr;
    float cv[2][2][2];
#pragma unroll
    for (int kz = 0; kz < ND - 1; ++kz)
#pragma unroll
      for (int ky = 0; ky < 2; ++ky)
#pragma unroll
        for (int kx = 0; kx < 2; ++kx) {
          const bool ok = v.zv[kz] && v.yv[ky] && v.xv[kx];
          const int off = o + kz * HW + ky * W + kx;
          cv[kz][ky][kx] = ok ? sc[off] : 0.f;
          if (dc && ok) atomicAdd(dc + off, (ND == 3) ? g * wz[kz] * wy[ky] * wx[kx] : g * wy[ky] * wx[kx]);
        }
    if (ND == 3) {
      const float p0 = wy[0] * (wx[0] * cv[0][0][0] + wx[1] * cv[0][0][1]) + wy[1] * (wx[0] * cv[0][1][0] + wx[1] * cv[0][1][1]);
      const float p1 = wy[0] * (wx[0] * cv[1][0][0] + wx[1] * cv[1][0][1]) + wy[1] * (wx[0] * cv[1][1][0] + wx[1] * cv[1][1][1]);
      gz += g * (p1 - p0);
      gy += g * (wz[0] * (wx[0] * (cv[0][1][0] - cv[0][0][0]) + wx[1] * (cv[0][1][1] - cv[0][0][1])) +
                 wz[1] * (wx[0] * (cv[1][1][0] - cv[1][0][0]) + wx[1] * (cv[1][1][1] - cv[1][0][1])));
      gx += g * (wz[0] * (wy[0] * (cv[0][0][1] - cv[0][0][0]) + wy[1] * (cv[0][1][1] - cv[0][1][0])) +
                 wz[1] * (wy[0] * (cv[1][0][1] - cv[1][0][0]) + wy[1] * (cv[1][1][1] - cv[1][1][0])));
    } else {
      gy += g * (wx[0] * (cv[0][1][0] - cv[0][0][0]) + wx[1] * (cv[0][1][1] - cv[0][0][1]));
      gx += g * (wy[0] * (cv[0][0][1] - cv[0][0][0]) + wy[1] * (cv[0][1][1] - cv[0][1][0]));
    }
    if (dc && add_identity) atomicAdd(dc + sp, g);
  }
  if (flow_into_src) {
    if (ND == 3) atomicAdd(dsrc_b + sp, gz);
    atomicAdd(dsrc_b + (long long)(ND - 2) * S + sp, gy);
    atomicAdd(dsrc_b + (long long)(ND - 1) * S + sp, gx);
  } else if (dflow_b) {
    if (ND == 3) dflow_b[sp] = gz;
    dflow_b[(long long)(ND - 2) * S + sp] = gy;
    dflow_b[(long long)(ND - 1) * S + sp] = gx;
  }
}

#define WIN_THREAD_COORDS()                                                            \
  using G = WinGeom<ND>;                                                               \
  const int t = (int)threadIdx.x;                                                      \
  int tx_, tyb_, tzb_, b;                                                              \
  if (!win_tile(ntx, nty, ntz, B, tx_, tyb_, tzb_, b)) return;                         \
  const int x = tx_ * WTX + (t & 7) * 4, y = tyb_ * G::TY + (t >> 3) % G::TY,          \
            z = tzb_ * G::TZ + (t >> 3) / G::TY;                                       \
  const bool active = x < W && y < H && z < D;                                         \
  const int S = D * H * W;                                                             \
  const int sp = (z * H + y) * W + x

// WW_OCC (A/B build switch, scripts/build_ko.sh warp_win WW_OCC 1 2 3): bit 0 = the forward kernel at 8 waves per SIMD
// (64 VGPRs: 4 instead of 3 workgroups per CU), bit 1 = pass 1 of the owner-gather backward at 6 waves per SIMD
#ifndef WW_OCC
#define WW_OCC 1     // measured (r03): forward 28.4 -> 28.2 us; bit 1 made pass 1 37 % slower (32 spilled registers)
#endif
#ifndef WW_KO
#define WW_KO 0      // knock-out builds of pass 1 (timing only): 1 no LDS atomics, 2 no sub-box store, 4 no d(flow) phase, 8 no LDS zero
#endif
#if (WW_OCC & 1)
#define WW_FWD_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define WW_FWD_ATTR
#endif
#if (WW_OCC & 2)
#define WW_OWN_ATTR __attribute__((amdgpu_waves_per_eu(6, 6)))
#else
#define WW_OWN_ATTR
#endif
template <int ND>
__global__ __launch_bounds__(WNT) WW_FWD_ATTR void warp_win_fwd_k(const float* __restrict__ src, const float* __restrict__ flow,
                                                      float* __restrict__ out, int B, int C, int D, int H, int W,
                                                      int add_identity, int ntx, int nty, int ntz) {
  __shared__ float win[WinGeom<ND>::EZ * WinGeom<ND>::EY * WRS];
  __shared__ int red[3 * WNT / 64];
  WIN_THREAD_COORDS();
  WinThread<ND> th;
  int oz, oy, ox;
  win_prologue<ND>(flow, b, S, H, W, D, z, y, x, active, false, th, red, oz, oy, ox);
  for (int c = 0; c < C; ++c) {
    const float* sc = src + ((long long)b * C + c) * S;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc), 0, (unsigned)S * 4u, 0x00020000);
    if (c) __syncthreads();
    win_load<ND>(rs, win, D, H, W, oz, oy, ox);
    __syncthreads();
    if (!active) continue;
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* p = win + th.base[e];
      const float wx1 = th.wx1[e], wy1 = th.wy1[e], wx0 = 1.f - wx1, wy0 = 1.f - wy1;
      float val = wy0 * (wx0 * p[0] + wx1 * p[1]) + wy1 * (wx0 * p[WRS] + wx1 * p[WRS + 1]);
      if (ND == 3) {
        const float* q = p + G::EY * WRS;
        const float p1 = wy0 * (wx0 * q[0] + wx1 * q[1]) + wy1 * (wx0 * q[WRS] + wx1 * q[WRS + 1]);
        val = (1.f - th.wz1[e]) * val + th.wz1[e] * p1;
      }
      r[e] = val;
    }
    if (add_identity) {
      const float4 id = *reinterpret_cast<const float4*>(sc + sp);
      r[0] += id.x; r[1] += id.y; r[2] += id.z; r[3] += id.w;
    }
    *reinterpret_cast<float4*>(out + ((long long)b * C + c) * S + sp) = make_float4(r[0], r[1], r[2], r[3]);
  }
  if (active && th.fast != 15u) {
    const unsigned slow = ~th.fast & 15u;
#pragma unroll 1
    for (int e = 0; e < 4; ++e)
      if ((slow >> e) & 1u)
        voxel_fwd_slow<ND>(src + (long long)b * C * S, flow + (long long)b * ND * S, out + (long long)b * C * S, C, D,
                           H, W, S, z, y, x + e, add_identity);
  }
}

template <int ND>
__global__ __launch_bounds__(WNT) void warp_win_bwd_k(const float* __restrict__ dout, const float* __restrict__ src,
                                                      const float* __restrict__ flow, float* __restrict__ dsrc,
                                                      float* __restrict__ dflow, int B, int C, int D, int H, int W,
                                                      int add_identity, int flow_into_src, int ntx, int nty, int ntz) {
  __shared__ float win[WinGeom<ND>::EZ * WinGeom<ND>::EY * WRS];
  __shared__ int red[3 * WNT / 64];
  WIN_THREAD_COORDS();
  WinThread<ND> th;
  int oz, oy, ox;
  const bool need_own = dsrc && (add_identity || flow_into_src);
  win_prologue<ND>(flow, b, S, H, W, D, z, y, x, active, need_own, th, red, oz, oy, ox);
  // fast voxels only: a slow voxel is handled start to finish by voxel_bwd_slow at the end
  float fm[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) fm[e] = ((th.fast >> e) & 1u) ? 1.f : 0.f;
  // ---- phase A: d(flow) = sum_c dout_c * d(interp)/d(position), corner values from the src window
  float gz[4] = {0.f, 0.f, 0.f, 0.f}, gy[4] = {0.f, 0.f, 0.f, 0.f}, gx[4] = {0.f, 0.f, 0.f, 0.f};
  if ((dflow || flow_into_src) && !(WW_KO & 4)) {
    for (int c = 0; c < C; ++c) {
      const float* sc = src + ((long long)b * C + c) * S;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc), 0, (unsigned)S * 4u, 0x00020000);
      if (c) __syncthreads();
      win_load<ND>(rs, win, D, H, W, oz, oy, ox);
      __syncthreads();
      if (!active) continue;
      const float4 g4 = *reinterpret_cast<const float4*>(dout + ((long long)b * C + c) * S + sp);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* p = win + th.base[e];
        const float g = gg[e] * fm[e];
        const float wx1 = th.wx1[e], wy1 = th.wy1[e], wz1 = th.wz1[e];
        const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
        const float c0 = p[0], c1 = p[1], c2 = p[WRS], c3 = p[WRS + 1];
        if (ND == 3) {
          const float* q = p + G::EY * WRS;
          const float c4 = q[0], c5 = q[1], c6 = q[WRS], c7 = q[WRS + 1];
          const float p0 = wy0 * (wx0 * c0 + wx1 * c1) + wy1 * (wx0 * c2 + wx1 * c3);
          const float p1 = wy0 * (wx0 * c4 + wx1 * c5) + wy1 * (wx0 * c6 + wx1 * c7);
          gz[e] += g * (p1 - p0);
          gy[e] += g * (wz0 * (wx0 * (c2 - c0) + wx1 * (c3 - c1)) + wz1 * (wx0 * (c6 - c4) + wx1 * (c7 - c5)));
          gx[e] += g * (wz0 * (wy0 * (c1 - c0) + wy1 * (c3 - c2)) + wz1 * (wy0 * (c5 - c4) + wy1 * (c7 - c6)));
        } else {
          gy[e] += g * (wx0 * (c2 - c0) + wx1 * (c3 - c1));
          gx[e] += g * (wy0 * (c1 - c0) + wy1 * (c3 - c2));
        }
      }
    }
    if (active && !flow_into_src) {
      float* fb = dflow + (long long)b * ND * S + sp;
      if (ND == 3) *reinterpret_cast<float4*>(fb) = make_float4(gz[0], gz[1], gz[2], gz[3]);
      *reinterpret_cast<float4*>(fb + (long long)(ND - 2) * S) = make_float4(gy[0], gy[1], gy[2], gy[3]);
      *reinterpret_cast<float4*>(fb + (long long)(ND - 1) * S) = make_float4(gx[0], gx[1], gx[2], gx[3]);
    }
  }
  // ---- phase B: d(src): scatter into a zeroed LDS window, then flush the touched cells
  if (dsrc) {
    constexpr int NW = G::EZ * G::EY * WRS;
    const int own = ((z - oz) * G::EY + (y - oy)) * WRS + (x - ox);   // valid for fast voxels when need_own
    for (int c = 0; c < C; ++c) {
      float* dc = dsrc + ((long long)b * C + c) * S;
      __syncthreads();
      for (int u = t; u < NW; u += WNT) win[u] = 0.f;
      __syncthreads();
      if (active && th.fast) {
        const float4 g4 = *reinterpret_cast<const float4*>(dout + ((long long)b * C + c) * S + sp);
        const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!((th.fast >> e) & 1u)) continue;
          float* p = win + th.base[e];
          const float g = gg[e];
          const float wx1 = th.wx1[e], wy1 = th.wy1[e], wz1 = th.wz1[e];
          const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
          if (ND == 3) {
            float* q = p + G::EY * WRS;
            atomicAdd(p, g * wz0 * wy0 * wx0);
            atomicAdd(p + 1, g * wz0 * wy0 * wx1);
            atomicAdd(p + WRS, g * wz0 * wy1 * wx0);
            atomicAdd(p + WRS + 1, g * wz0 * wy1 * wx1);
            atomicAdd(q, g * wz1 * wy0 * wx0);
            atomicAdd(q + 1, g * wz1 * wy0 * wx1);
            atomicAdd(q + WRS, g * wz1 * wy1 * wx0);
            atomicAdd(q + WRS + 1, g * wz1 * wy1 * wx1);
          } else {
            atomicAdd(p, g * wy0 * wx0);
            atomicAdd(p + 1, g * wy0 * wx1);
            atomicAdd(p + WRS, g * wy1 * wx0);
            atomicAdd(p + WRS + 1, g * wy1 * wx1);
          }
          if (need_own) {
            float ov = add_identity ? g : 0.f;
            if (flow_into_src) ov += (ND == 3) ? (c == 0 ? gz[e] : (c == 1 ? gy[e] : gx[e])) : (c == 0 ? gy[e] : gx[e]);
            atomicAdd(win + own + e, ov);
          }
        }
      }
      __syncthreads();
      for (int u = t; u < G::EZ * G::EY * WEX; u += WNT) {
        const int lx = u % WEX, r = u / WEX;
        const int ly = r % G::EY, lz = r / G::EY;
        const float v = win[r * WRS + lx];
        const int fx = ox + lx, fy = oy + ly, fz = oz + lz;
        if (v != 0.f && (unsigned)fx < (unsigned)W && (unsigned)fy < (unsigned)H && (unsigned)fz < (unsigned)D)
          atomicAdd(dc + (fz * H + fy) * W + fx, v);
      }
    }
  }
  if (active && th.fast != 15u) {
    const unsigned slow = ~th.fast & 15u;
#pragma unroll 1
    for (int e = 0; e < 4; ++e)
      if ((slow >> e) & 1u)
        voxel_bwd_slow<ND>(dout + (long long)b * C * S, src + (long long)b * C * S, flow + (long long)b * ND * S,
                           dsrc ? dsrc + (long long)b * C * S : nullptr,
                           dflow ? dflow + (long long)b * ND * S : nullptr, C, D, H, W, S, z, y, x + e, add_identity,
                           flow_into_src);
  }
}

// ------------------------------------------------------------------------------------------------
// Backward without device-scope atomics on d(src), bit-reproducible ("owner gathers"):
//   pass 1  warp_win_bwd_own_k   as warp_win_bwd_k, but the privatised d(src) window is accumulated in 32-bit FIXED
//           POINT (LDS integer atomics are associative: the sum does not depend on the order the lanes arrive in).
//           Scale: 2^(30 - be) with 2^be >= SUM over the workgroup's contributions of |g| -- every voxel spreads its g
//           with weights that sum to 1, so no cell can exceed that sum and the accumulation cannot overflow, whatever
//           the field (collapsing fields included).  Absolute error per contribution <= 2^(be - 31), i.e. <= 2^-19 x
//           the workgroup's MEAN |g| for a 2048-voxel tile (tests bound it against the fp32-atomic path).  The window
//           is converted back and only its TOUCHED SUB-BOX [0,ez) x [0,ey) x [0,ex) (ex a multiple of 4) is written to
//           the tile's scratch slot; origin and extents go to the tile's metadata.  The origin is clamped so that the
//           window stays inside the 3 x 3 (x 3) tiles around its own; voxels whose taps leave it go to the tile's own
//           list (no global counter: nothing has to be zeroed before the launch).  A non-finite contribution makes the
//           whole sub-box of that channel NaN (the reference's grid_sample backward propagates NaN / inf as well).
//   pass 2  warp_win_gather_k    every d(src) cell sums, in fixed tile order, the sub-boxes of its 27 (9) neighbour
//           tiles that cover it and writes the result (no pre-zeroed d(src) needed).
//   pass 3  warp_win_slow_k      the listed voxels (none on registration-like fields), scalar routine + atomics.
// ------------------------------------------------------------------------------------------------
template <int ND> struct WinOwn {
  static constexpr int CELLS = WinGeom<ND>::EZ * WinGeom<ND>::EY * WEX;
  static constexpr int TILE_VOX = WinGeom<ND>::TZ * WinGeom<ND>::TY * WTX;     // 2048: slow-list capacity of a tile
};
static constexpr int WMETA = 8;      // ints of metadata per tile: oz, oy, ox, (ez << 16 | ey << 8 | ex), nslow, -, -, -

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum_x(float v) {       // xor butterfly: every lane ends with the same bits
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int ND>
__global__ __launch_bounds__(WNT) WW_OWN_ATTR void warp_win_bwd_own_k(const float* __restrict__ dout, const float* __restrict__ src,
                                                          const float* __restrict__ flow, float* __restrict__ dflow,
                                                          int B, int C, int D, int H, int W, int add_identity,
                                                          int flow_into_src, int ntx, int nty, int ntz,
                                                          float* __restrict__ scratch, int* __restrict__ meta,
                                                          unsigned* __restrict__ slowlist) {
  using GG = WinGeom<ND>;
  constexpr int NW = GG::EZ * GG::EY * WRS;
  __shared__ float win[NW];                                // phase A: the src window; phase B: int32 fixed point
  __shared__ int red[3 * WNT / 64];
  __shared__ float redf[WNT / 64];
  __shared__ int s_ext[3];
  __shared__ int s_nslow, s_bad;
  int* wini = reinterpret_cast<int*>(win);
  WIN_THREAD_COORDS();
  // linear tile id as win_tile decodes it
  const int tileL = ((b * ntz + tzb_) * nty + tyb_) * ntx + tx_;
  WinThread<ND> th;
  int oz, oy, ox;
  const bool need_own = add_identity || flow_into_src;
  if (t == 0) { s_ext[0] = 0; s_ext[1] = 0; s_ext[2] = 0; s_nslow = 0; }
  // (fetching channel 0's gradient quad here, together with the flow, was measured: 138 -> 152 us -- four more live
  // registers cost more than the round trip saves; knock-outs: without atomics, stores, the d(flow) phase and the LDS
  // zeroing the three passes still take 119 of 138 us: the chain of dependent loads and barriers at 2 workgroups per
  // CU bounds pass 1, not its arithmetic)
  win_prologue<ND>(flow, b, S, H, W, D, z, y, x, active, need_own, th, red, oz, oy, ox, true, tzb_ * G::TZ, tyb_ * G::TY,
                   tx_ * WTX);
  float fm[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) fm[e] = ((th.fast >> e) & 1u) ? 1.f : 0.f;
  // touched sub-box of the window: upper corner of every fast voxel's taps (and its own cell), workgroup maximum
  {
    int uz = -1, uy = -1, ux = -1;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((th.fast >> e) & 1u) {
        const int bse = th.base[e];
        const int lx = bse % WRS, r = bse / WRS;
        uz = max(uz, (ND == 3) ? r / GG::EY + 1 : 0);
        uy = max(uy, r % GG::EY + 1);
        ux = max(ux, lx + 1);
        if (need_own) { uz = max(uz, z - oz); uy = max(uy, y - oy); ux = max(ux, x + e - ox); }
      }
    uz = wave_max_i(uz); uy = wave_max_i(uy); ux = wave_max_i(ux);
    if ((t & 63) == 0 && ux >= 0) { atomicMax(&s_ext[0], uz + 1); atomicMax(&s_ext[1], uy + 1); atomicMax(&s_ext[2], ux + 1); }
  }
  // ---- phase A: d(flow) (identical to warp_win_bwd_k)
  float gz[4] = {0.f, 0.f, 0.f, 0.f}, gy[4] = {0.f, 0.f, 0.f, 0.f}, gx[4] = {0.f, 0.f, 0.f, 0.f};
  if (dflow || flow_into_src) {
    for (int c = 0; c < C; ++c) {
      const float* sc = src + ((long long)b * C + c) * S;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sc), 0, (unsigned)S * 4u, 0x00020000);
      if (c) __syncthreads();
      win_load<ND>(rs, win, D, H, W, oz, oy, ox);
      __syncthreads();
      if (!active) continue;
      const float4 g4 = *reinterpret_cast<const float4*>(dout + ((long long)b * C + c) * S + sp);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* p = win + th.base[e];
        const float g = gg[e] * fm[e];
        const float wx1 = th.wx1[e], wy1 = th.wy1[e], wz1 = th.wz1[e];
        const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
        const float c0 = p[0], c1 = p[1], c2 = p[WRS], c3 = p[WRS + 1];
        if (ND == 3) {
          const float* q = p + G::EY * WRS;
          const float c4 = q[0], c5 = q[1], c6 = q[WRS], c7 = q[WRS + 1];
          const float p0 = wy0 * (wx0 * c0 + wx1 * c1) + wy1 * (wx0 * c2 + wx1 * c3);
          const float p1 = wy0 * (wx0 * c4 + wx1 * c5) + wy1 * (wx0 * c6 + wx1 * c7);
          gz[e] += g * (p1 - p0);
          gy[e] += g * (wz0 * (wx0 * (c2 - c0) + wx1 * (c3 - c1)) + wz1 * (wx0 * (c6 - c4) + wx1 * (c7 - c5)));
          gx[e] += g * (wz0 * (wy0 * (c1 - c0) + wy1 * (c3 - c2)) + wz1 * (wy0 * (c5 - c4) + wy1 * (c7 - c6)));
        } else {
          gy[e] += g * (wx0 * (c2 - c0) + wx1 * (c3 - c1));
          gx[e] += g * (wy0 * (c1 - c0) + wy1 * (c3 - c2));
        }
      }
    }
    if (active && !flow_into_src && dflow) {
      float* fb = dflow + (long long)b * ND * S + sp;
      // slow voxels: their d(flow) is written by pass 3; fast ones here (a float4 store covers both: pass 3 runs later)
      if (ND == 3) *reinterpret_cast<float4*>(fb) = make_float4(gz[0], gz[1], gz[2], gz[3]);
      *reinterpret_cast<float4*>(fb + (long long)(ND - 2) * S) = make_float4(gy[0], gy[1], gy[2], gy[3]);
      *reinterpret_cast<float4*>(fb + (long long)(ND - 1) * S) = make_float4(gx[0], gx[1], gx[2], gx[3]);
    }
  }
  // ---- phase B: d(src) of the fast voxels, fixed point in LDS, store of the touched sub-box
  const int own = ((z - oz) * G::EY + (y - oy)) * WRS + (x - ox);
  int ez = 0, ey = 0, ex4 = 0;
  for (int c = 0; c < C; ++c) {
    float gg[4] = {0.f, 0.f, 0.f, 0.f}, ov[4] = {0.f, 0.f, 0.f, 0.f};
    float m = 0.f;
    bool bad = false;
    if (active && th.fast) {
      const float4 g4 = *reinterpret_cast<const float4*>(dout + ((long long)b * C + c) * S + sp);
      gg[0] = g4.x; gg[1] = g4.y; gg[2] = g4.z; gg[3] = g4.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gg[e] *= fm[e];
        ov[e] = add_identity ? gg[e] : 0.f;
        if (flow_into_src) ov[e] += fm[e] * ((ND == 3) ? (c == 0 ? gz[e] : (c == 1 ? gy[e] : gx[e])) : (c == 0 ? gy[e] : gx[e]));
        const float am = fabsf(gg[e]) + fabsf(ov[e]);
        bad = bad || !(am < 3.0e38f);                       // NaN or inf
        m += am;
      }
    }
    // workgroup SUM of |contribution| -> power-of-two scale; fixed reduction order (bit-reproducible)
    m = wave_sum_x(bad ? 0.f : m);
    __syncthreads();                                        // phase A / previous channel is done with the window
    if ((t & 63) == 0) redf[t >> 6] = m;
    if (t == 0) s_bad = 0;
    if (!(WW_KO & 8)) for (int u = t; u < NW; u += WNT) wini[u] = 0;
    __syncthreads();
    if (bad) s_bad = 1;
    float bm = redf[0];
#pragma unroll
    for (int q = 1; q < WNT / 64; ++q) bm += redf[q];
    int be = (int)((__float_as_uint(bm) >> 23) & 0xffu) - 127 + 1;      // sum < 2^be
    be = bm > 0.f ? (be > 90 ? 90 : (be < -90 ? -90 : be)) : 0;
    const float up = __uint_as_float((unsigned)(30 - be + 127) << 23);        // |cell * up| <= 2^30
    const float dn = __uint_as_float((unsigned)(be - 30 + 127) << 23);
    if (active && th.fast && !bad && !(WW_KO & 1)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!((th.fast >> e) & 1u)) continue;
        int* p = wini + th.base[e];
        const float g = gg[e] * up;
        const float wx1 = th.wx1[e], wy1 = th.wy1[e], wz1 = th.wz1[e];
        const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
#define OWN_ADD(ptr_, val_) atomicAdd(reinterpret_cast<unsigned*>(ptr_), (unsigned)__float2int_rn(val_))
        if (ND == 3) {
          int* q = p + G::EY * WRS;
          OWN_ADD(p, g * wz0 * wy0 * wx0);
          OWN_ADD(p + 1, g * wz0 * wy0 * wx1);
          OWN_ADD(p + WRS, g * wz0 * wy1 * wx0);
          OWN_ADD(p + WRS + 1, g * wz0 * wy1 * wx1);
          OWN_ADD(q, g * wz1 * wy0 * wx0);
          OWN_ADD(q + 1, g * wz1 * wy0 * wx1);
          OWN_ADD(q + WRS, g * wz1 * wy1 * wx0);
          OWN_ADD(q + WRS + 1, g * wz1 * wy1 * wx1);
        } else {
          OWN_ADD(p, g * wy0 * wx0);
          OWN_ADD(p + 1, g * wy0 * wx1);
          OWN_ADD(p + WRS, g * wy1 * wx0);
          OWN_ADD(p + WRS + 1, g * wy1 * wx1);
        }
        if (need_own) OWN_ADD(wini + own + e, ov[e] * up);
#undef OWN_ADD
      }
    }
    __syncthreads();
    ez = s_ext[0]; ey = s_ext[1]; ex4 = (s_ext[2] + 3) & ~3;
    if (ND == 2) ez = ey > 0 ? 1 : 0;
    const bool anybad = s_bad != 0;
    float* slot = scratch + ((long long)tileL * C + c) * WinOwn<ND>::CELLS;
    const int qx = ex4 >> 2, nq = (WW_KO & 2) ? 0 : ez * ey * qx;
    for (int u = t; u < nq; u += WNT) {
      const int q = u % qx, r = u / qx;                    // r = lz * ey + ly within the sub-box
      const int ly = r % ey, lz = r / ey;
      const int* wp = wini + (lz * GG::EY + ly) * WRS + 4 * q;
      float4 v;
      if (anybad) {
        v.x = v.y = v.z = v.w = __uint_as_float(0x7fc00000u);
      } else {
        v.x = (float)wp[0] * dn; v.y = (float)wp[1] * dn; v.z = (float)wp[2] * dn; v.w = (float)wp[3] * dn;
      }
      *reinterpret_cast<float4*>(slot + (lz * GG::EY + ly) * WEX + 4 * q) = v;
    }
  }
  // ---- the voxels whose taps leave the window: listed (per tile) for pass 3
  if (active && th.fast != 15u) {
    const unsigned sl = ~th.fast & 15u;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((sl >> e) & 1u) {
        const int i = atomicAdd(&s_nslow, 1);
        slowlist[(long long)tileL * WinOwn<ND>::TILE_VOX + i] = (unsigned)(b * S + sp + e);
      }
  }
  __syncthreads();
  if (t == 0) {
    int* mt = meta + (long long)tileL * WMETA;
    mt[0] = oz; mt[1] = oy; mt[2] = ox;
    mt[3] = (s_ext[0] << 16) | (s_ext[1] << 8) | ((s_ext[2] + 3) & ~3);
    mt[4] = s_nslow;
  }
}

template <int ND>
__global__ __launch_bounds__(WNT) void warp_win_gather_k(const float* __restrict__ scratch, const int* __restrict__ meta,
                                                         float* __restrict__ dsrc, int B, int C, int D, int H, int W,
                                                         int ntx, int nty, int ntz) {
  // one workgroup per tile, the thread <-> voxel-quad map of pass 1: the 27 (9) neighbour origins / extents are
  // workgroup-uniform, the per-quad coverage test is a few integer compares, only covering sub-boxes are read
  WIN_THREAD_COORDS();
  constexpr int NZ = (ND == 3) ? 3 : 1, NN = NZ * 9;
  // the neighbour tiles' window origins and extents, fetched once per workgroup (one lane each); -1 = no such tile
  __shared__ int org[NN][8];
  if (t < NN) {
    const int iz = t / 9, iy = (t / 3) % 3, ix = t % 3;
    const int nz = tzb_ + (ND == 3 ? iz - 1 : 0), ny = tyb_ + iy - 1, nx = tx_ + ix - 1;
    const bool ok = (unsigned)nz < (unsigned)ntz && (unsigned)ny < (unsigned)nty && (unsigned)nx < (unsigned)ntx;
    const int tl = ok ? ((b * ntz + nz) * nty + ny) * ntx + nx : -1;
    const int* mt = meta + (long long)(ok ? tl : 0) * WMETA;
    const int ext = ok ? mt[3] : 0;
    org[t][0] = ok ? mt[0] : 0;
    org[t][1] = ok ? mt[1] : 0;
    org[t][2] = ok ? mt[2] : 0;
    org[t][3] = tl;
    org[t][4] = (ND == 3) ? (ext >> 16) & 0xff : (ext ? 1 : 0);
    org[t][5] = (ext >> 8) & 0xff;
    org[t][6] = ext & 0xff;
  }
  __syncthreads();
  if (!active) return;
  // Branch-free: a window that does not cover the quad gets an out-of-range buffer offset (reads 0.0 without touching
  // memory), so the loads of 9 neighbours are in flight together instead of one conditional load after the other
  // (the sum keeps its fixed neighbour order: bit-reproducible as before).
  constexpr unsigned OOBG = 0x80000000u;
  unsigned off[NN];
#pragma unroll
  for (int i = 0; i < NN; ++i) {
    const int lz = z - org[i][0], ly = y - org[i][1], lx = x - org[i][2];     // ox is a multiple of 4: whole quad or none
    const bool cov = org[i][3] >= 0 && (unsigned)lz < (unsigned)org[i][4] && (unsigned)ly < (unsigned)org[i][5] &&
                     lx >= 0 && lx + 4 <= org[i][6];
    off[i] = cov ? (unsigned)(((lz * G::EY + ly) * WEX + lx) * 4) : OOBG;
  }
  for (int c = 0; c < C; ++c) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g0 = 0; g0 < NN; g0 += 9) {
      float4 v[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int tl = org[g0 + i][3] < 0 ? 0 : org[g0 + i][3];
        const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(scratch + ((long long)tl * C + c) * WinOwn<ND>::CELLS), 0,
            (unsigned)(WinOwn<ND>::CELLS * 4), 0x00020000);
        const u32x4w q = __builtin_amdgcn_raw_buffer_load_b128(wsrc, off[g0 + i], 0, 0);
        v[i] = make_float4(__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3]));
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) { acc[0] += v[i].x; acc[1] += v[i].y; acc[2] += v[i].z; acc[3] += v[i].w; }
    }
    *reinterpret_cast<float4*>(dsrc + ((long long)b * C + c) * S + sp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

template <int ND>
__global__ __launch_bounds__(256) void warp_win_slow_k(const float* __restrict__ dout, const float* __restrict__ src,
                                                       const float* __restrict__ flow, float* __restrict__ dsrc,
                                                       float* __restrict__ dflow, int B, int C, int D, int H, int W,
                                                       int add_identity, int flow_into_src, const int* __restrict__ meta,
                                                       const unsigned* __restrict__ slowlist, int T) {
  // 4 tiles per workgroup (one wave each): a tile's count is a wave-uniform load, empty tiles cost nothing else
  const int tile = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (tile >= T) return;
  const int n = meta[(long long)tile * WMETA + 4];
  const int S = D * H * W;
  for (int i = threadIdx.x & 63; i < n; i += 64) {
    const unsigned v = slowlist[(long long)tile * WinOwn<ND>::TILE_VOX + i];
    const int b = (int)(v / (unsigned)S), sp = (int)(v - (unsigned)b * (unsigned)S);
    const int x = sp % W, y = (sp / W) % H, z = sp / (W * H);
    voxel_bwd_slow<ND>(dout + (long long)b * C * S, src + (long long)b * C * S, flow + (long long)b * ND * S,
                       dsrc + (long long)b * C * S, dflow ? dflow + (long long)b * ND * S : nullptr, C, D, H, W, S, z, y, x,
                       add_identity, flow_into_src);
  }
}

// ------------------------------------------------------------------------------------------------
// Host side: returns 1 when the windowed kernel took the launch, 0 when the shape is not eligible.
template <int ND>
static bool win_eligible(int B, int C, int D, int H, int W, int& ntx, int& nty, int& ntz, long long& grid) {
  using G = WinGeom<ND>;
  if ((W & 3) != 0) return false;
  const long long S = (long long)D * H * W;
  if (S * ND * 4 >= (1LL << 31)) return false;
  ntx = (W + WTX - 1) / WTX; nty = (H + G::TY - 1) / G::TY; ntz = (D + G::TZ - 1) / G::TZ;
  const long long T = (long long)ntx * nty * ntz * B;
  if (T >= (1LL << 30)) return false;
  grid = ((T + 7) / 8) * 8;
  (void)C;
  return true;
}

int df_warp_win_fwd_try(int nd, const float* src, const float* flow, float* out, int B, int C, int D, int H, int W,
                        int add_identity, hipStream_t st) {
  int ntx, nty, ntz; long long grid;
  if (nd == 3) {
    if (!win_eligible<3>(B, C, D, H, W, ntx, nty, ntz, grid)) return 0;
    warp_win_fwd_k<3><<<(unsigned)grid, WNT, 0, st>>>(src, flow, out, B, C, D, H, W, add_identity, ntx, nty, ntz);
  } else {
    if (!win_eligible<2>(B, C, 1, H, W, ntx, nty, ntz, grid)) return 0;
    warp_win_fwd_k<2><<<(unsigned)grid, WNT, 0, st>>>(src, flow, out, B, C, 1, H, W, add_identity, ntx, nty, ntz);
  }
  return 1;
}

int df_warp_win_bwd_try(int nd, const float* dout, const float* src, const float* flow, float* dsrc, float* dflow,
                        int B, int C, int D, int H, int W, int add_identity, int flow_into_src, hipStream_t st) {
  int ntx, nty, ntz; long long grid;
  if (nd == 3) {
    if (!win_eligible<3>(B, C, D, H, W, ntx, nty, ntz, grid)) return 0;
    warp_win_bwd_k<3><<<(unsigned)grid, WNT, 0, st>>>(dout, src, flow, dsrc, dflow, B, C, D, H, W, add_identity,
                                                      flow_into_src, ntx, nty, ntz);
  } else {
    if (!win_eligible<2>(B, C, 1, H, W, ntx, nty, ntz, grid)) return 0;
    warp_win_bwd_k<2><<<(unsigned)grid, WNT, 0, st>>>(dout, src, flow, dsrc, dflow, B, C, 1, H, W, add_identity,
                                                      flow_into_src, ntx, nty, ntz);
  }
  return 1;
}

// owner-gather backward: scratch = [metadata: WMETA ints per tile][slow lists: TILE_VOX uints per tile][windows: tiles * C * CELLS floats]
template <int ND>
static long long own_ws_floats(int B, int C, int D, int H, int W) {
  int ntx, nty, ntz; long long grid;
  if (!win_eligible<ND>(B, C, D, H, W, ntx, nty, ntz, grid)) return 0;
  const long long T = (long long)ntx * nty * ntz * B;
  long long n = WMETA * T + T * WinOwn<ND>::TILE_VOX;
  n = (n + 3) & ~3LL;
  return n + T * C * WinOwn<ND>::CELLS;
}
long long df_warp_win_bwd_own_ws(int nd, int B, int C, int D, int H, int W) {
  if ((long long)B * D * H * W >= (1LL << 31)) return 0;
  return nd == 3 ? own_ws_floats<3>(B, C, D, H, W) : own_ws_floats<2>(B, C, 1, H, W);
}
template <int ND>
static int own_launch(const float* dout, const float* src, const float* flow, float* dsrc, float* dflow, int B, int C, int D,
                      int H, int W, int add_identity, int flow_into_src, float* ws, hipStream_t st) {
  int ntx, nty, ntz; long long grid;
  if (!win_eligible<ND>(B, C, D, H, W, ntx, nty, ntz, grid)) return 0;
  const long long T = (long long)ntx * nty * ntz * B;
  int* meta = reinterpret_cast<int*>(ws);
  unsigned* slow = reinterpret_cast<unsigned*>(ws) + WMETA * T;
  long long off = WMETA * T + T * WinOwn<ND>::TILE_VOX;
  off = (off + 3) & ~3LL;
  float* windows = ws + off;
  warp_win_bwd_own_k<ND><<<(unsigned)grid, WNT, 0, st>>>(dout, src, flow, dflow, B, C, D, H, W, add_identity, flow_into_src,
                                                         ntx, nty, ntz, windows, meta, slow);
  warp_win_gather_k<ND><<<(unsigned)grid, WNT, 0, st>>>(windows, meta, dsrc, B, C, D, H, W, ntx, nty, ntz);
  warp_win_slow_k<ND><<<(unsigned)((T + 3) / 4), 256, 0, st>>>(dout, src, flow, dsrc, dflow, B, C, D, H, W, add_identity,
                                                              flow_into_src, meta, slow, (int)T);
  return hipGetLastError() == hipSuccess ? 1 : -1;
}
int df_warp_win_bwd_own_try(int nd, const float* dout, const float* src, const float* flow, float* dsrc, float* dflow,
                            int B, int C, int D, int H, int W, int add_identity, int flow_into_src, float* ws,
                            hipStream_t st) {
  return nd == 3 ? own_launch<3>(dout, src, flow, dsrc, dflow, B, C, D, H, W, add_identity, flow_into_src, ws, st)
                 : own_launch<2>(dout, src, flow, dsrc, dflow, B, C, 1, H, W, add_identity, flow_into_src, ws, st);
}

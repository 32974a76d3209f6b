// Scalar losses of the registration step (masked L1, flow smoothness, windowed NCC), a generic
// scaled sum, and the fused Adam update.  All reductions: per-wave shuffle -> per-block LDS ->
// one fp32 atomic per block into a tiny workspace; a 1-thread finaliser forms the scalar on device
// so the host never synchronises.
#include "common.h"
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <string.h>

// ---------------------------------------------------------------------------------- masked L1
__device__ __forceinline__ float l1_mask(const float a, const float b, const unsigned char* mask,
                                         long long i, float thr) {
  if (mask) return mask[i] ? 1.f : 0.f;
  return (a > thr || b > thr) ? 1.f : 0.f;
}
// (1024-thread workgroups, at most 128 of them: the two atomic adds of every workgroup land on one cache line, and 1024 x 2
// of them were 25 of the kernel's 30 us on a 1 M-element image batch)
__global__ __launch_bounds__(1024) void masked_l1_fwd_k(const float* __restrict__ a,
                                                        const float* __restrict__ b,
                                                        const unsigned char* __restrict__ mask, float thr,
                                                        float* __restrict__ ws, long long n) {
  __shared__ float sm[17];
  float s = 0.f, m = 0.f;
  for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n; i += (long long)gridDim.x * 1024) {
    const float av = a[i], bv = b[i];
    const float mk = l1_mask(av, bv, mask, i, thr);
    s += fabsf(av - bv) * mk;
    m += mk;
  }
  s = block_sum(s, sm);
  m = block_sum(m, sm);
  if (threadIdx.x == 0) {
    atomicAdd(&ws[0], s);
    atomicAdd(&ws[1], m);
  }
}
__global__ void masked_l1_fin_k(const float* ws, float* out) {
  out[0] = ws[1] > 0.f ? (1.f / ws[1]) * ws[0] : 0.f;
}
__global__ __launch_bounds__(256) void masked_l1_bwd_k(const float* __restrict__ a,
                                                       const float* __restrict__ b,
                                                       const unsigned char* __restrict__ mask, float thr,
                                                       const float* __restrict__ ws,
                                                       const float* __restrict__ gout,
                                                       float* __restrict__ da, float* __restrict__ db,
                                                       long long n) {
  const float sc = ws[1] > 0.f ? gout[0] / ws[1] : 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float av = a[i], bv = b[i];
    const float mk = l1_mask(av, bv, mask, i, thr);
    const float d = av - bv;
    const float g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * mk * sc;
    if (da) da[i] = g;
    if (db) db[i] = -g;
  }
}

// ------------------------------------------------------------------------------ flow smoothness
// One workgroup = RB consecutive rows (a row = W values at fixed plane, z, y): the row's coordinates are decoded once per
// row with wave-uniform arithmetic, lanes run along x -- no per-element divisions (they used to cost more than the
// memory traffic), every load a contiguous run.  <= 2048 workgroups: each ends in three same-address atomics.
// L1: sum |d| instead of sum d^2 (Grad_Loss / vxm Grad with penalty 'l1': util/losses.py:92-117, torchvoxelmorph/losses.py:102-112)
constexpr int FS_PART = 8, FS_MAXWG = 2048;     // ws: [0..7] the sums, then 3 x FS_MAXWG per-workgroup partials
template <bool L1>
__global__ __launch_bounds__(256) void flow_smooth_fwd_k(const float* __restrict__ f, float* __restrict__ ws,
                                                         long long planes, int D, int H, int W, int RB) {
  __shared__ float sm[17];
  const long long rows = planes * D * H, HW = (long long)H * W;
  float sd = 0.f, sh = 0.f, sw = 0.f;
  const long long row0 = (long long)blockIdx.x * RB;
  int y = (int)(row0 % H), z = (int)((row0 / H) % D);      // advanced incrementally: one division per workgroup
  for (int x = threadIdx.x; x < W; x += 256) {
    int yy = y, zz = z;
#pragma unroll 4
    for (int rr = 0; rr < RB; ++rr) {                       // (independent iterations: 4 rows of loads in flight)
      const long long row = row0 + rr;
      if (row < rows) {
        const float* p = f + row * W;
        const float v = p[x];
        if (x + 1 < W) { const float d = p[x + 1] - v; sw += L1 ? fabsf(d) : d * d; }
        if (yy + 1 < H) { const float d = p[x + W] - v; sh += L1 ? fabsf(d) : d * d; }
        if (zz + 1 < D) { const float d = p[x + HW] - v; sd += L1 ? fabsf(d) : d * d; }
      }
      if (++yy == H) { yy = 0; if (++zz == D) zz = 0; }
    }
  }
  sd = block_sum(sd, sm);
  sh = block_sum(sh, sm);
  sw = block_sum(sw, sm);
  if (threadIdx.x == 0) {          // per-workgroup partials (flow_smooth_fin_k adds them in index order)
    ws[FS_PART + blockIdx.x] = sd;
    ws[FS_PART + FS_MAXWG + blockIdx.x] = sh;
    ws[FS_PART + 2 * FS_MAXWG + blockIdx.x] = sw;
  }
}
// W % 4 == 0: a thread owns 16-B quads (grid-stride), one float4 load each for the quad, its y + 1 and z + 1 neighbours and
// one scalar for x + 4: 4 independent loads per 4 elements instead of 12 dependent-ish scalar ones per row walk
// TPR = threads per row (a power of two >= W / 4): (row, quad) come from shifts and the row's (y, z) from 32-bit divisions --
// the 64-bit `t % Wq`, `row % H`, `row / H % D` of a flat quad index cost more than the loads
template <int TPR>
__global__ __launch_bounds__(1024) void flow_smooth_fwd_v4_k(const float* __restrict__ f, float* __restrict__ ws,
                                                            long long planes, int D, int H, int W) {
  __shared__ float sm[17];
  const int Wq = W >> 2;
  const unsigned nrow = (unsigned)(planes * D * H);
  const long long HW = (long long)H * W;
  float sd = 0.f, sh = 0.f, sw = 0.f;
  const int q = threadIdx.x & (TPR - 1);
  // workgroup ids go round-robin over the 8 XCDs (one L2 each): XCD e walks a CONTIGUOUS eighth of the rows, so the y +- 1 /
  // z +- 1 neighbours of its rows are its own rows (row-interleaved, every row was fetched into three L2s)
  const unsigned per = (nrow + 7u) / 8u, xbase = (blockIdx.x & 7u) * per;
  // 1024-thread workgroups, at most 512 of them: every workgroup ends with three atomic adds onto the SAME three floats, and
  // 2048 x 3 of them serialised in L2 were the whole kernel (102 us whatever the volume)
  for (unsigned r = ((blockIdx.x >> 3) * 1024u + threadIdx.x) / TPR; r < per && xbase + r < nrow && q < Wq;
       r += (gridDim.x >> 3) * (1024u / TPR)) {
    const unsigned row = xbase + r;
    const unsigned rz = row / (unsigned)H;
    const int y = (int)(row - rz * (unsigned)H), z = (int)(rz % (unsigned)D);
    const float* p = f + (long long)row * W + 4 * q;
    const float4 v = *reinterpret_cast<const float4*>(p);
    const bool hy = y + 1 < H, hz = z + 1 < D, hx = q + 1 < Wq;
    float4 vy = v, vz = v;
    if (hy) vy = *reinterpret_cast<const float4*>(p + W);
    if (hz) vz = *reinterpret_cast<const float4*>(p + HW);
    // x + 4 is the next lane's first value (threads of a row are consecutive lanes: TPR <= 64)
    const float nsh = __shfl_down(v.x, 1, 64);
    const float nx = hx ? nsh : v.w;
    float d;
    d = v.y - v.x; sw += d * d; d = v.z - v.y; sw += d * d; d = v.w - v.z; sw += d * d; d = nx - v.w; sw += d * d;
    d = vy.x - v.x; sh += d * d; d = vy.y - v.y; sh += d * d; d = vy.z - v.z; sh += d * d; d = vy.w - v.w; sh += d * d;
    d = vz.x - v.x; sd += d * d; d = vz.y - v.y; sd += d * d; d = vz.z - v.z; sd += d * d; d = vz.w - v.w; sd += d * d;
  }
  sd = block_sum(sd, sm);
  sh = block_sum(sh, sm);
  sw = block_sum(sw, sm);
  if (threadIdx.x == 0) {          // per-workgroup partials (flow_smooth_fin_k adds them in index order)
    ws[FS_PART + blockIdx.x] = sd;
    ws[FS_PART + FS_MAXWG + blockIdx.x] = sh;
    ws[FS_PART + 2 * FS_MAXWG + blockIdx.x] = sw;
  }
}
template <int TPR>
__global__ __launch_bounds__(256) void flow_smooth_bwd_v4_k(const float* __restrict__ f, const float* __restrict__ gout,
                                                            float* __restrict__ df, long long planes, int D, int H,
                                                            int W, float kd, float kh, float kw) {
  const int Wq = W >> 2;
  const unsigned nrow = (unsigned)(planes * D * H);
  const long long HW = (long long)H * W;
  const float g = gout[0];
  const int q = threadIdx.x & (TPR - 1);
  // workgroup ids go round-robin over the 8 XCDs (one L2 each): XCD e walks a CONTIGUOUS eighth of the rows, so the y +- 1 /
  // z +- 1 neighbours of its rows are its own rows (row-interleaved, every row was fetched into three L2s)
  const unsigned per = (nrow + 7u) / 8u, xbase = (blockIdx.x & 7u) * per;
  for (unsigned r = ((blockIdx.x >> 3) * 256u + threadIdx.x) / TPR; r < per && xbase + r < nrow && q < Wq;
       r += (gridDim.x >> 3) * (256u / TPR)) {
    const unsigned row = xbase + r;
    const unsigned rz = row / (unsigned)H;
    const int y = (int)(row - rz * (unsigned)H), z = (int)(rz % (unsigned)D);
    const float* p = f + (long long)row * W + 4 * q;
    const float4 v = *reinterpret_cast<const float4*>(p);
    // a missing neighbour contributes nothing: substitute the centre value (difference 0)
    const float lsh = __shfl_up(v.w, 1, 64), rsh = __shfl_down(v.x, 1, 64);   // the neighbouring quads are the neighbouring lanes
    const float xl = q > 0 ? lsh : v.x, xr = q + 1 < Wq ? rsh : v.w;
    const float4 yu = y > 0 ? *reinterpret_cast<const float4*>(p - W) : v;
    const float4 yd = y + 1 < H ? *reinterpret_cast<const float4*>(p + W) : v;
    const float4 zu = z > 0 ? *reinterpret_cast<const float4*>(p - HW) : v;
    const float4 zd = z + 1 < D ? *reinterpret_cast<const float4*>(p + HW) : v;
    float4 o;
    o.x = g * (kw * ((v.x - xl) - (v.y - v.x)) + kh * ((v.x - yu.x) - (yd.x - v.x)) + kd * ((v.x - zu.x) - (zd.x - v.x)));
    o.y = g * (kw * ((v.y - v.x) - (v.z - v.y)) + kh * ((v.y - yu.y) - (yd.y - v.y)) + kd * ((v.y - zu.y) - (zd.y - v.y)));
    o.z = g * (kw * ((v.z - v.y) - (v.w - v.z)) + kh * ((v.z - yu.z) - (yd.z - v.z)) + kd * ((v.z - zu.z) - (zd.z - v.z)));
    o.w = g * (kw * ((v.w - v.z) - (xr - v.w)) + kh * ((v.w - yu.w) - (yd.w - v.w)) + kd * ((v.w - zu.w) - (zd.w - v.w)));
    *reinterpret_cast<float4*>(df + (long long)row * W + 4 * q) = o;
  }
}
// The same gradient, MARCHING along z (D > 1).  flow_smooth_bwd_v4_k reads five rows per output row; four of them hit in L2,
// but every one is a fill of the CU's vector cache, and the kernel ran at the rate of those fills (413 MB for 165 MB of HBM
// traffic: 0.088 ms; without the neighbour loads: 0.032).  Here a workgroup owns a strip of FS_RY rows of one plane stack and
// walks a segment of z: a thread keeps its two rows' quads of planes z - 1, z, z + 1 in registers (the z neighbours are its
// own earlier loads), the y neighbours come from the other waves through LDS, only the two halo rows of the strip are loaded
// a second time -- 1.25 row fills per output row (x the segment's two halo planes).  Loads are issued two planes ahead.  The
// expression is flow_smooth_bwd_v4_k's, operand for operand: bit-identical results.
constexpr int FS_RY = 8;
struct FsmP { int D, H, W, nstrip, nseg, zlen; long long nitem; float kd, kh, kw; };
__global__ __launch_bounds__(256) void flow_smooth_bwd_march_k(const float* __restrict__ f, const float* __restrict__ gout,
                                                               float* __restrict__ df, FsmP k) {
  typedef float fsv4 __attribute__((ext_vector_type(4)));     // (HIP's float4 struct is rebuilt element by element from a
  typedef unsigned u32x4_fs __attribute__((ext_vector_type(4))); //  buffer load's result: a copy that waits for the load at once)
  __shared__ __attribute__((aligned(16))) fsv4 rows[2][FS_RY + 2][64];    // [step parity][strip row + 1][quad]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int Wq = k.W >> 2;
  const float g = gout[0];
  // item -> (plane stack, z segment, strip): ids with the same residue mod 8 (one XCD) walk a contiguous eighth of the list,
  // strips fastest (neighbouring strips share their halo rows in one L2)
  const long long per = (k.nitem + 7) / 8;
  const long long item = (long long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((long long)(blockIdx.x >> 3) >= per || item >= k.nitem) return;
  const int strip = (int)(item % k.nstrip);
  const int seg = (int)((item / k.nstrip) % k.nseg);
  const long long pc = item / ((long long)k.nstrip * k.nseg);
  const int y0 = strip * FS_RY, zs = seg * k.zlen, ze = (zs + k.zlen < k.D) ? zs + k.zlen : k.D;
  const long long HW = (long long)k.H * k.W;
  const float* fp = f + pc * k.D * HW;
  float* dp = df + pc * k.D * HW;
  const bool act = lane < Wq;
  const int ya = y0 + 2 * wv, yb = ya + 1;                   // this wave's two rows
  const bool oka = act && ya < k.H, okb = act && yb < k.H;
  // halo rows of the strip: wave 0 carries row y0 - 1, wave 3 row y0 + FS_RY
  const int yh = wv == 0 ? y0 - 1 : y0 + FS_RY;
  const bool okh = act && (wv == 0 || wv == 3) && yh >= 0 && yh < k.H;
  // every load unconditional, with an out-of-range offset where there is nothing to fetch (a bounds-checked buffer load returns
  // 0 and moves nothing): under `cond ? *p : 0` each load was a branch with its own s_waitcnt vmcnt(0) -- no prefetch at all
  const __amdgpu_buffer_rsrc_t f_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fp), 0, (unsigned)(k.D * HW * 4), 0x00020000);
  auto fsm_ld = [&](int z_, int y_, bool ok_) __attribute__((always_inline)) {
    const unsigned off = (ok_ && z_ >= 0 && z_ < k.D) ? (unsigned)((z_ * HW + (long long)y_ * k.W + 4 * lane) * 4) : 0x80000000u;
    return __builtin_bit_cast(fsv4, __builtin_amdgcn_raw_buffer_load_b128(f_src, off, 0, 0));
  };
#define FSM_LD(z_, y_, ok_) fsm_ld((z_), (y_), (ok_))
  // four register sets {row a, row b, halo row} hold planes z - 1, z, z + 1 and the one in flight (z + 2); the plane loop is
  // unrolled by four so that the rotation is a renaming (a register move would wait for the load it moves), and the steps are
  // separated by `break` (DESIGN.md hardware fact 11)
  fsv4 Aa = FSM_LD(zs - 1, ya, oka), Ab = FSM_LD(zs - 1, yb, okb), Ah = fsv4{0.f, 0.f, 0.f, 0.f};
  fsv4 Ba = FSM_LD(zs, ya, oka), Bb = FSM_LD(zs, yb, okb), Bh = FSM_LD(zs, yh, okh);
  fsv4 Ca = FSM_LD(zs + 1, ya, oka), Cb = FSM_LD(zs + 1, yb, okb), Ch = FSM_LD(zs + 1, yh, okh);
  fsv4 Ea, Eb, Eh;
  // LDS only between the waves: wait for the LDS writes, not for the global loads / stores in flight (__syncthreads() is a
  // fence over both and drained the prefetch at every plane)
#define FSM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define FSM_STEP(Ua, Ub, Va, Vb, Vh, Da, Db, Na, Nb, Nh, zz_, par_)                                                     \
  {                                                                                                                      \
    const int z = (zz_);                                                                                                 \
    Na = FSM_LD(z + 2, ya, oka && z + 2 <= ze); Nb = FSM_LD(z + 2, yb, okb && z + 2 <= ze);                              \
    Nh = FSM_LD(z + 2, yh, okh && z + 2 <= ze);                                                                          \
    rows[par_][1 + 2 * wv][lane] = Va;                                                                                   \
    rows[par_][2 + 2 * wv][lane] = Vb;                                                                                   \
    if (wv == 0) rows[par_][0][lane] = Vh;                                                                               \
    if (wv == 3) rows[par_][FS_RY + 1][lane] = Vh;                                                                       \
    FSM_BARRIER();                                           /* (the other parity is rewritten two steps later) */       \
    const bool hzu = z > 0, hzd = z + 1 < k.D;                                                                           \
    _Pragma("unroll") for (int r = 0; r < 2; ++r) {                                                                      \
      const int y = r ? yb : ya;                                                                                         \
      if (!(r ? okb : oka)) continue;                                                                                    \
      const fsv4 v = r ? Vb : Va;                                                                                      \
      const fsv4 zu = hzu ? (r ? Ub : Ua) : v, zd = hzd ? (r ? Db : Da) : v;                                           \
      const fsv4 yu = y > 0 ? (r ? Va : rows[par_][2 * wv][lane]) : v;                                                 \
      const fsv4 yd = y + 1 < k.H ? (r ? rows[par_][3 + 2 * wv][lane] : Vb) : v;                                       \
      const float lsh = __shfl_up(v.w, 1, 64), rsh = __shfl_down(v.x, 1, 64);                                            \
      const float xl = lane > 0 ? lsh : v.x, xr = lane + 1 < Wq ? rsh : v.w;                                             \
      fsv4 o;                                                                                                            \
      o.x = g * (k.kw * ((v.x - xl) - (v.y - v.x)) + k.kh * ((v.x - yu.x) - (yd.x - v.x)) + k.kd * ((v.x - zu.x) - (zd.x - v.x))); \
      o.y = g * (k.kw * ((v.y - v.x) - (v.z - v.y)) + k.kh * ((v.y - yu.y) - (yd.y - v.y)) + k.kd * ((v.y - zu.y) - (zd.y - v.y))); \
      o.z = g * (k.kw * ((v.z - v.y) - (v.w - v.z)) + k.kh * ((v.z - yu.z) - (yd.z - v.z)) + k.kd * ((v.z - zu.z) - (zd.z - v.z))); \
      o.w = g * (k.kw * ((v.w - v.z) - (xr - v.w)) + k.kh * ((v.w - yu.w) - (yd.w - v.w)) + k.kd * ((v.w - zu.w) - (zd.w - v.w))); \
      *reinterpret_cast<fsv4*>(dp + (long long)z * HW + (long long)y * k.W + 4 * lane) = o;                            \
    }                                                                                                                    \
  }
  (void)Ah;
  for (int z0 = zs; z0 < ze; z0 += 4) {
    FSM_STEP(Aa, Ab, Ba, Bb, Bh, Ca, Cb, Ea, Eb, Eh, z0, 0)
    if (z0 + 1 >= ze) break;
    FSM_STEP(Ba, Bb, Ca, Cb, Ch, Ea, Eb, Aa, Ab, Ah, z0 + 1, 1)
    if (z0 + 2 >= ze) break;
    FSM_STEP(Ca, Cb, Ea, Eb, Eh, Aa, Ab, Ba, Bb, Bh, z0 + 2, 0)
    if (z0 + 3 >= ze) break;
    FSM_STEP(Ea, Eb, Aa, Ab, Ah, Ba, Bb, Ca, Cb, Ch, z0 + 3, 1)
  }
#undef FSM_STEP
#undef FSM_BARRIER
#undef FSM_LD
}
__global__ __launch_bounds__(256) void flow_smooth_fin_k(float* ws, float* out, float cd, float ch, float cw, float nd, int nwg) {
  // the partials in a FIXED order (thread t adds slots t, t + 256, ...; the block tree is the same every run): the loss is
  // bit-reproducible, and 512 x 3 same-address atomics at the end of the forward kernel were a third of its 46 us
  __shared__ float sm[17];
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nwg; i += 256) {
    a += ws[FS_PART + i]; b += ws[FS_PART + FS_MAXWG + i]; c += ws[FS_PART + 2 * FS_MAXWG + i];
  }
  a = block_sum(a, sm); b = block_sum(b, sm); c = block_sum(c, sm);
  if (threadIdx.x) return;
  ws[0] = a; ws[1] = b; ws[2] = c;
  float s = 0.f;
  if (cd > 0.f) s += ws[0] / cd;
  if (ch > 0.f) s += ws[1] / ch;
  if (cw > 0.f) s += ws[2] / cw;
  out[0] = s / nd;
}
__device__ __forceinline__ float df_sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
template <typename IDX, bool L1 = false>
__global__ __launch_bounds__(256) void flow_smooth_bwd_k(const float* __restrict__ f,
                                                         const float* __restrict__ gout,
                                                         float* __restrict__ df, long long planes, int D,
                                                         int H, int W, float kd, float kh, float kw) {
  const IDX S = (IDX)D * H * W, total = (IDX)planes * S;
  const float g = gout[0];
  const IDX HW = (IDX)H * W;
  for (IDX i = (IDX)blockIdx.x * 256 + threadIdx.x; i < total; i += (IDX)gridDim.x * 256) {
    const int x = (int)(i % (IDX)W);
    IDX r = i / (IDX)W;
    const int y = (int)(r % (IDX)H); r /= (IDX)H;
    const int z = (int)(r % (IDX)D);
    const float v = f[i];
    float acc = 0.f;
    if (W > 1) {
      float t = 0.f;
      if (x > 0) t += L1 ? df_sgn(v - f[i - 1]) : v - f[i - 1];
      if (x + 1 < W) t -= L1 ? df_sgn(f[i + 1] - v) : f[i + 1] - v;
      acc += kw * t;
    }
    if (H > 1) {
      float t = 0.f;
      if (y > 0) t += L1 ? df_sgn(v - f[i - W]) : v - f[i - W];
      if (y + 1 < H) t -= L1 ? df_sgn(f[i + W] - v) : f[i + W] - v;
      acc += kh * t;
    }
    if (D > 1) {
      float t = 0.f;
      if (z > 0) t += L1 ? df_sgn(v - f[i - HW]) : v - f[i - HW];
      if (z + 1 < D) t -= L1 ? df_sgn(f[i + HW] - v) : f[i + HW] - v;
      acc += kd * t;
    }
    df[i] = g * acc;
  }
}

// ------------------------------------------------------------------------------------- NCC
// box sums are separable: W pass fused with the 5 products, then H, then D.
__global__ __launch_bounds__(256) void ncc_prod_boxw_k(const float* __restrict__ I, const float* __restrict__ J,
                                                       float* __restrict__ o, long long N, int W, int r) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int x = N < 0x7FFFFFFFLL ? (int)((unsigned)i % (unsigned)W) : (int)(i % W);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
  for (int d = -r; d <= r; ++d) {
    const int xx = x + d;
    if ((unsigned)xx < (unsigned)W) {
      const float a = I[i + d], b = J[i + d];
      s0 += a; s1 += b; s2 += a * a; s3 += b * b; s4 += a * b;
    }
  }
  o[i] = s0; o[N + i] = s1; o[2 * N + i] = s2; o[3 * N + i] = s3; o[4 * N + i] = s4;
}
// r = 4, W % 4 == 0: 4 consecutive outputs per thread from its own float4 of I / J and the neighbouring lanes' (see
// box_axis_x4_k); the same sums in the same order
__global__ __launch_bounds__(256) void ncc_prod_boxw_x4_k(const float* __restrict__ I, const float* __restrict__ J,
                                                          float* __restrict__ o, long long N, int W) {
  const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = i4 < (N >> 2);
  const long long i = live ? i4 << 2 : 0;
  const int x = (int)(i % W);
  const bool ha = x >= 4, hb = x + 8 <= W;
  float va[12], vb[12];
#define NPB_WINDOW(P_, v_)                                                                        \
  {                                                                                               \
    const float* p = (P_) + i;                                                                    \
    const float4 c = live ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f); \
    float4 a, b;                                                                                  \
    a.x = __shfl_up(c.x, 1, 64); a.y = __shfl_up(c.y, 1, 64); a.z = __shfl_up(c.z, 1, 64); a.w = __shfl_up(c.w, 1, 64); \
    b.x = __shfl_down(c.x, 1, 64); b.y = __shfl_down(c.y, 1, 64); b.z = __shfl_down(c.z, 1, 64); b.w = __shfl_down(c.w, 1, 64); \
    if (lane == 0 && ha && live) a = *reinterpret_cast<const float4*>(p - 4);                     \
    if (lane == 63 && hb && live) b = *reinterpret_cast<const float4*>(p + 4);                    \
    if (!ha) a = make_float4(0.f, 0.f, 0.f, 0.f);                                                 \
    if (!hb) b = make_float4(0.f, 0.f, 0.f, 0.f);                                                 \
    v_[0] = a.x; v_[1] = a.y; v_[2] = a.z; v_[3] = a.w; v_[4] = c.x; v_[5] = c.y; v_[6] = c.z; v_[7] = c.w; \
    v_[8] = b.x; v_[9] = b.y; v_[10] = b.z; v_[11] = b.w;                                         \
  }
  NPB_WINDOW(I, va)
  NPB_WINDOW(J, vb)
#undef NPB_WINDOW
  if (!live) return;
  float s[5][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
    for (int d = 0; d < 9; ++d) {
      const float a = va[j + d], b = vb[j + d];
      s0 += a; s1 += b; s2 += a * a; s3 += b * b; s4 += a * b;
    }
    s[0][j] = s0; s[1][j] = s1; s[2][j] = s2; s[3][j] = s3; s[4][j] = s4;
  }
#pragma unroll
  for (int f = 0; f < 5; ++f) *reinterpret_cast<float4*>(o + (long long)f * N + i) = make_float4(s[f][0], s[f][1], s[f][2], s[f][3]);
}
// W AND H passes of the 5 product fields in one launch (r = 4): a workgroup owns a 32 x 64 tile of one (b, z) plane, stages
// I and J over (32 + 8) x (64 + 8) positions (zero outside the image = the zero padding of the reference's box filter),
// forms the W-window sums of the 5 products for the 40 rows in LDS and sums 9 rows of them per output: the five
// intermediate fields of the W pass never travel (137 MB out + 137 MB back in at 160x192x224).  Plain 9-term sums, as in the
// separate passes (no running differences).
__global__ __launch_bounds__(256) void ncc_prod_boxwh_k(const float* __restrict__ I, const float* __restrict__ J,
                                                        float* __restrict__ o, long long N, int H, int W, int nty, int ntx) {
  constexpr int TY = 32, TX = 64, R = 4, PY = TY + 2 * R, PX = TX + 2 * R;
  // 75 KB of static LDS: above the 64 KB most parts allow, within gfx950's 160 KB per CU (this library builds for gfx950 only)
  static_assert((2 * PY * (PX + 1) + 5 * PY * (TX + 1)) * sizeof(float) <= 160 * 1024, "LDS tile exceeds gfx950's 160 KB");
  __shared__ float sI[PY][PX + 1], sJ[PY][PX + 1];
  __shared__ float sS[5][PY][TX + 1];
  const int tid = threadIdx.x;
  int q = blockIdx.x;
  const int tx = q % ntx; q /= ntx;
  const int ty = q % nty;
  const long long plane = q / nty;
  const int y0 = ty * TY, x0 = tx * TX;
  const float* pI = I + plane * H * W;
  const float* pJ = J + plane * H * W;
  {   // all loads of the thread in flight before the first LDS store (12 x 2 per thread)
    constexpr int NE = (PY * PX + 255) / 256;
    float vi[NE], vj[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + 256 * u, r = e / PX, c = e - r * PX;
      const int y = y0 - R + r, x = x0 - R + c;
      const bool in = e < PY * PX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      vi[u] = in ? pI[(long long)y * W + x] : 0.f;
      vj[u] = in ? pJ[(long long)y * W + x] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + 256 * u, r = e / PX, c = e - r * PX;
      if (e < PY * PX) { sI[r][c] = vi[u]; sJ[r][c] = vj[u]; }
    }
  }
  __syncthreads();
  // W sums: job = (row of 40, 16 consecutive columns): 160 jobs
  if (tid < PY * (TX / 16)) {
    const int r = tid >> 2, c0 = (tid & 3) * 16;
    float a[24], b[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) { a[j] = sI[r][c0 + j]; b[j] = sJ[r][c0 + j]; }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
      for (int d = 0; d < 9; ++d) {
        const float u = a[j + d], v = b[j + d];
        s0 += u; s1 += v; s2 += u * u; s3 += v * v; s4 += u * v;
      }
      sS[0][r][c0 + j] = s0; sS[1][r][c0 + j] = s1; sS[2][r][c0 + j] = s2; sS[3][r][c0 + j] = s3; sS[4][r][c0 + j] = s4;
    }
  }
  __syncthreads();
  // H sums: thread = (column x, 8 consecutive rows)
  const int x = tid & 63, yb = (tid >> 6) * 8;
  if (x0 + x >= W) return;
#pragma unroll
  for (int f = 0; f < 5; ++f) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = sS[f][yb + j][x];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int y = y0 + yb + j;
      if (y < H) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 9; ++d) s += v[j + d];
        o[(long long)f * N + plane * H * W + (long long)y * W + x0 + x] = s;
      }
    }
  }
}
static void ncc_prod_boxw_launch(const float* I, const float* J, float* o, long long N, int W, int r, hipStream_t st);
// out[f][v] = sum_{d} in[f][v + d*stride] over the axis of length `len` (coordinate = (v/stride)%len)
__global__ __launch_bounds__(256) void box_axis_k(const float* __restrict__ in, float* __restrict__ out,
                                                  int nf, long long N, long long stride, int len, int r) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int c = (N < 0x7FFFFFFFLL && stride < 0x7FFFFFFFLL) ? (int)(((unsigned)i / (unsigned)stride) % (unsigned)len)
                                                             : (int)((i / stride) % len);
  for (int f = 0; f < nf; ++f) {
    const float* p = in + f * N + i;
    float s = 0.f;
    for (int d = -r; d <= r; ++d) {
      const int cc = c + d;
      if ((unsigned)cc < (unsigned)len) s += p[d * stride];
    }
    out[f * N + i] = s;
  }
}
// The same sums along an axis with stride > 1 (H, D), marching: a thread owns SEG consecutive outputs of one line and
// keeps the 2R + 1 window values in a register ring -- every input is loaded once (box_axis_k: 2R + 1 times, from
// L1 / L2), consecutive threads hold consecutive lines, so every load / store instruction is contiguous.  Each output
// is the plain sum of its window (no running difference: the variance terms cancel to 1e-3 of the sums).
template <int R, int SEG>
__global__ __launch_bounds__(256) void box_axis_march_k(const float* __restrict__ in, float* __restrict__ out, int nf,
                                                        long long N, long long stride, int len) {
  constexpr int WN = 2 * R + 1;
  static_assert(SEG % WN == 0, "segment = whole turns of the ring");
  const long long nline = N / len;
  const int nseg = (len + SEG - 1) / SEG;
  const long long per_f = nline * nseg;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= per_f * nf) return;
  const int f = (int)(t / per_f);
  const long long u = t - (long long)f * per_f;
  const int sg = (int)(u / nline);
  const long long l = u - (long long)sg * nline;
  const long long outer = l / stride, inner = l - outer * stride;
  const float* p = in + (long long)f * N + outer * len * stride + inner;
  float* q = out + (long long)f * N + outer * len * stride + inner;
  const int c0 = sg * SEG;
  float ring[WN];
#pragma unroll
  for (int j = 0; j < WN - 1; ++j) {
    const int cc = c0 - R + j;
    ring[j] = ((unsigned)cc < (unsigned)len) ? p[(long long)cc * stride] : 0.f;
  }
  for (int k0 = 0; k0 < SEG; k0 += WN) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int c = c0 + k0 + j;
      if (c < len) {
        const int cc = c + R;
        ring[(WN - 1 + j) % WN] = (cc < len) ? p[(long long)cc * stride] : 0.f;
        float sacc = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) sacc += ring[w];
        q[(long long)c * stride] = sacc;
      }
    }
  }
}
// The contiguous axis (stride 1, r = 4, len % 4 == 0): a thread owns 4 consecutive outputs; the 12 values their windows span
// are its own float4 and the neighbouring lanes' (shuffles; the first / last lane of a wave loads them), masked by the
// line's ends -- 1 load instruction per 4 outputs instead of 36 (box_axis_k is bound by the texture path, not by HBM).
__global__ __launch_bounds__(256) void box_axis_x4_k(const float* __restrict__ in, float* __restrict__ out, int nf,
                                                     long long N, int len) {
  const long long i4 = (long long)blockIdx.x * 256 + threadIdx.x;      // float4 index inside one field
  const long long n4 = N >> 2;
  const int lane = threadIdx.x & 63;
  const bool live = i4 < n4;
  const long long i = live ? i4 << 2 : 0;
  const int x = (int)(i % len);
  for (int f = 0; f < nf; ++f) {
    const float* p = in + (long long)f * N + i;
    const float4 c = live ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a, b;                                               // the float4 before / after
    a.x = __shfl_up(c.x, 1, 64); a.y = __shfl_up(c.y, 1, 64); a.z = __shfl_up(c.z, 1, 64); a.w = __shfl_up(c.w, 1, 64);
    b.x = __shfl_down(c.x, 1, 64); b.y = __shfl_down(c.y, 1, 64); b.z = __shfl_down(c.z, 1, 64); b.w = __shfl_down(c.w, 1, 64);
    const bool ha = x >= 4, hb = x + 8 <= len;                 // the neighbours lie on this line
    if (lane == 0 && ha && live) a = *reinterpret_cast<const float4*>(p - 4);
    if (lane == 63 && hb && live) b = *reinterpret_cast<const float4*>(p + 4);
    if (!ha) a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!hb) b = make_float4(0.f, 0.f, 0.f, 0.f);
    const float v[12] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w, b.x, b.y, b.z, b.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                              // output x + j: window v[j] .. v[j + 8], in box_axis_k's order
      float sacc = 0.f;
#pragma unroll
      for (int d = 0; d < 9; ++d) sacc += v[j + d];
      o[j] = sacc;
    }
    if (live) *reinterpret_cast<float4*>(out + (long long)f * N + i) = make_float4(o[0], o[1], o[2], o[3]);
  }
}
static void box_axis_launch(const float* in, float* out, int nf, long long N, long long stride, int len, int r,
                            hipStream_t st) {
  if (r == 4 && stride > 1 && N % len == 0) {
    constexpr int SEG = 36;
    const long long thr = (N / len) * ((len + SEG - 1) / SEG) * nf;
    box_axis_march_k<4, SEG><<<(unsigned)((thr + 255) / 256), 256, 0, st>>>(in, out, nf, N, stride, len);
  } else if (r == 4 && stride == 1 && (len & 3) == 0 && len >= 8 && N % len == 0 &&
             ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && (N & 3) == 0) {
    box_axis_x4_k<<<(unsigned)(((N >> 2) + 255) / 256), 256, 0, st>>>(in, out, nf, N, len);
  } else {
    box_axis_k<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(in, out, nf, N, stride, len, r);
  }
}
static void ncc_prod_boxw_launch(const float* I, const float* J, float* o, long long N, int W, int r, hipStream_t st) {
  if (r == 4 && (W & 3) == 0 && W >= 8 && (N & 3) == 0 && N % W == 0 &&
      ((reinterpret_cast<uintptr_t>(I) | reinterpret_cast<uintptr_t>(J) | reinterpret_cast<uintptr_t>(o)) & 15) == 0)
    ncc_prod_boxw_x4_k<<<(unsigned)(((N >> 2) + 255) / 256), 256, 0, st>>>(I, J, o, N, W);
  else
    ncc_prod_boxw_k<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(I, J, o, N, W, r);
}
struct NccTerms { float cross, Iv, Jv, uI, uJ, den; };
__device__ __forceinline__ NccTerms ncc_terms(const float* s, long long N, long long i, float wn, float eps) {
  const float Is = s[i], Js = s[N + i], I2 = s[2 * N + i], J2 = s[3 * N + i], IJ = s[4 * N + i];
  NccTerms t;
  t.uI = Is / wn;
  t.uJ = Js / wn;
  t.cross = IJ - t.uJ * Is - t.uI * Js + t.uI * t.uJ * wn;
  t.Iv = I2 - 2.f * t.uI * Is + t.uI * t.uI * wn;
  t.Jv = J2 - 2.f * t.uJ * Js + t.uJ * t.uJ * wn;
  t.den = t.Iv * t.Jv + eps;
  return t;
}
// mask (optional, N floats): ws[0] = sum cc * mask, ws[1] = sum mask (util/losses.py:257-261)
// part (optional, 2 * gridDim.x floats): per-workgroup partial sums instead of atomics -- ncc_fin_k adds them in index
// order, so the loss and (through ws[0]) its gradient do not depend on the order in which workgroups finish
__global__ __launch_bounds__(256) void ncc_cc_reduce_k(const float* __restrict__ s, const float* __restrict__ mask,
                                                       float* __restrict__ ws, long long N, float wn, float eps,
                                                       float* __restrict__ part) {
  __shared__ float sm[17];
  float acc = 0.f, msum = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
    const NccTerms t = ncc_terms(s, N, i, wn, eps);
    const float cc = t.cross * t.cross / t.den;
    if (mask) { const float m = mask[i]; acc += cc * m; msum += m; }
    else acc += cc;
  }
  acc = block_sum(acc, sm);
  if (mask) msum = block_sum(msum, sm);
  if (threadIdx.x == 0) {
    if (part) {
      part[blockIdx.x] = acc;
      part[gridDim.x + blockIdx.x] = mask ? msum : 0.f;
    } else {
      atomicAdd(&ws[0], acc);
      if (mask) atomicAdd(&ws[1], msum);
    }
  }
}
// mode 0: -sqrt(S / n) (NCC_Loss.forward, util/losses.py:248-261; 0 when the mask is empty), mode 1: -S / n
// (torchvoxelmorph/losses.py:67); n = sum(mask) with a mask, else the element count
__device__ __forceinline__ float ncc_norm(const float* ws, float n, int masked) { return masked ? ws[1] : n; }
__global__ __launch_bounds__(256) void ncc_fin_k(float* ws, float* out, float n, int mode, int masked, const float* part,
                                                 int nparts) {
  // the partials in a FIXED order: thread t adds part[t], part[t + 256], ... and the block tree is the same every run
  // (one thread walking 2 x 1024 partials took 44 us)
  __shared__ float sm[17];
  if (part) {
    float a = 0.f, m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) { a += part[i]; m += part[nparts + i]; }
    a = block_sum(a, sm);
    m = block_sum(m, sm);
    if (threadIdx.x == 0) { ws[0] = a; ws[1] = m; }
    __syncthreads();
  }
  if (threadIdx.x) return;
  const float ne = ncc_norm(ws, n, masked);
  if (!(ne > 0.f)) { out[0] = 0.f; return; }
  out[0] = mode == 0 ? -sqrtf(ws[0] / ne) : -(ws[0] / ne);
}
// d loss / d cc_i = ncc_kappa(...) * mask_i
__device__ __forceinline__ float ncc_kappa(const float* ws, const float* gout, float n, int mode, int masked) {
  const float ne = ncc_norm(ws, n, masked);
  if (!(ne > 0.f)) return 0.f;
  if (mode == 1) return -gout[0] / ne;
  const float m = ws[0] / ne;
  return m > 0.f ? gout[0] * (-0.5f / sqrtf(m)) / ne : 0.f;
}
// fields: A = dL/dIJsum, Bq = dL/dI2sum, Cq = dL/dIsum
__global__ __launch_bounds__(256) void ncc_fields_k(const float* __restrict__ s, const float* __restrict__ ws,
                                                    const float* __restrict__ gout, float* __restrict__ fld,
                                                    long long N, float wn, float eps, const float* __restrict__ mask,
                                                    int mode) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float kappa = ncc_kappa(ws, gout, (float)N, mode, mask != nullptr);
  if (mask) kappa *= mask[i];
  const NccTerms t = ncc_terms(s, N, i, wn, eps);
  const float A = kappa * (2.f * t.cross / t.den);
  const float Bq = kappa * (-(t.cross * t.cross) * t.Jv / (t.den * t.den));
  const float Cq = -A * t.uJ - 2.f * Bq * t.uI;
  fld[i] = A; fld[N + i] = Bq; fld[2 * N + i] = Cq;
}
// The three gradient fields AND their W and H box passes in one launch (r = 4), tiled like ncc_prod_boxwh_k: the fields are
// evaluated from the 5 saved box sums over (32 + 8) x (64 + 8) positions of a (b, z) plane (zero outside the image), summed
// over 9 columns, then over 9 rows -- the field tensor and its W-summed copy never travel.
__global__ __launch_bounds__(256) void ncc_fields_boxwh_k(const float* __restrict__ s, const float* __restrict__ ws,
                                                          const float* __restrict__ gout, float* __restrict__ o, long long N,
                                                          int H, int W, int nty, int ntx, float wn, float eps,
                                                          const float* __restrict__ mask, int mode) {
  constexpr int TY = 32, TX = 64, R = 4, PY = TY + 2 * R, PX = TX + 2 * R;
  // 66 KB of static LDS (gfx950: 160 KB per CU; see ncc_prod_boxwh_k)
  static_assert((3 * PY * (PX + 1) + 3 * PY * (TX + 1)) * sizeof(float) <= 160 * 1024, "LDS tile exceeds gfx950's 160 KB");
  __shared__ float sF[3][PY][PX + 1];
  __shared__ float sS[3][PY][TX + 1];
  const int tid = threadIdx.x;
  int q = blockIdx.x;
  const int tx = q % ntx; q /= ntx;
  const int ty = q % nty;
  const long long plane = q / nty;
  const int y0 = ty * TY, x0 = tx * TX;
  const float kappa0 = ncc_kappa(ws, gout, (float)N, mode, mask != nullptr);
  {
    constexpr int NE = (PY * PX + 255) / 256;
    float v[NE][5];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + 256 * u, r = e / PX, c = e - r * PX;
      const int y = y0 - R + r, x = x0 - R + c;
      const bool in = e < PY * PX && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
      const long long i = plane * H * W + (long long)y * W + x;
#pragma unroll
      for (int f = 0; f < 5; ++f) v[u][f] = in ? s[(long long)f * N + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + 256 * u, r = e / PX, c = e - r * PX;
      const int y = y0 - R + r, x = x0 - R + c;
      if (e < PY * PX) {
        float A = 0.f, Bq = 0.f, Cq = 0.f;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {      // the same arithmetic as ncc_terms / ncc_fields_k
          const float Is = v[u][0], Js = v[u][1], I2 = v[u][2], J2 = v[u][3], IJ = v[u][4];
          const float uI = Is / wn, uJ = Js / wn;
          const float cross = IJ - uJ * Is - uI * Js + uI * uJ * wn;
          const float Iv = I2 - 2.f * uI * Is + uI * uI * wn;
          const float Jv = J2 - 2.f * uJ * Js + uJ * uJ * wn;
          const float den = Iv * Jv + eps;
          const float kappa = mask ? kappa0 * mask[plane * H * W + (long long)y * W + x] : kappa0;
          A = kappa * (2.f * cross / den);
          Bq = kappa * (-(cross * cross) * Jv / (den * den));
          Cq = -A * uJ - 2.f * Bq * uI;
        }
        sF[0][r][c] = A; sF[1][r][c] = Bq; sF[2][r][c] = Cq;
      }
    }
  }
  __syncthreads();
  if (tid < PY * (TX / 16)) {
    const int r = tid >> 2, c0 = (tid & 3) * 16;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      float a[24];
#pragma unroll
      for (int j = 0; j < 24; ++j) a[j] = sF[f][r][c0 + j];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float t = 0.f;
#pragma unroll
        for (int d = 0; d < 9; ++d) t += a[j + d];
        sS[f][r][c0 + j] = t;
      }
    }
  }
  __syncthreads();
  const int x = tid & 63, yb = (tid >> 6) * 8;
  if (x0 + x >= W) return;
#pragma unroll
  for (int f = 0; f < 3; ++f) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = sS[f][yb + j][x];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int y = y0 + yb + j;
      if (y < H) {
        float t = 0.f;
#pragma unroll
        for (int d = 0; d < 9; ++d) t += v[j + d];
        o[(long long)f * N + plane * H * W + (long long)y * W + x0 + x] = t;
      }
    }
  }
}
// The backward's last two passes in one: the box sums of the three gradient fields along the strided axis (D) as
// box_axis_march_k forms them -- same ring, same order of additions -- and, per output, the combination of ncc_combine_k.
// The three summed fields (82 MB written and read back at 160x192x224) are never stored; results are bit-identical.
template <int R, int SEG>
__global__ __launch_bounds__(256) void ncc_boxd_combine_k(const float* __restrict__ in, const float* __restrict__ I,
                                                          const float* __restrict__ J, float* __restrict__ dI,
                                                          long long N, long long stride, int len) {
  constexpr int WN = 2 * R + 1;
  static_assert(SEG % WN == 0, "segment = whole turns of the ring");
  const long long nline = N / len;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int nseg = (len + SEG - 1) / SEG;
  if (t >= nline * nseg) return;
  const int sg = (int)(t / nline);
  const long long l = t - (long long)sg * nline;
  const long long outer = l / stride, inner = l - outer * stride;
  const long long base = outer * len * stride + inner;
  const float* p = in + base;
  const int c0 = sg * SEG;
  float ring[3][WN];
#pragma unroll
  for (int f = 0; f < 3; ++f)
#pragma unroll
    for (int j = 0; j < WN - 1; ++j) {
      const int cc = c0 - R + j;
      ring[f][j] = ((unsigned)cc < (unsigned)len) ? p[(long long)f * N + (long long)cc * stride] : 0.f;
    }
  for (int k0 = 0; k0 < SEG; k0 += WN) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int c = c0 + k0 + j;
      if (c < len) {
        const int cc = c + R;
        float sacc[3];
#pragma unroll
        for (int f = 0; f < 3; ++f) {
          ring[f][(WN - 1 + j) % WN] = (cc < len) ? p[(long long)f * N + (long long)cc * stride] : 0.f;
          float a = 0.f;
#pragma unroll
          for (int w = 0; w < WN; ++w) a += ring[f][w];
          sacc[f] = a;
        }
        const long long i = base + (long long)c * stride;
        dI[i] = J[i] * sacc[0] + 2.f * I[i] * sacc[1] + sacc[2];
      }
    }
  }
}
__global__ __launch_bounds__(256) void ncc_combine_k(const float* __restrict__ I, const float* __restrict__ J,
                                                     const float* __restrict__ bx, float* __restrict__ dI,
                                                     long long N) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  dI[i] = J[i] * bx[i] + 2.f * I[i] * bx[N + i] + bx[2 * N + i];
}

// ---------------------------------------------------------------------------------- misc
__global__ __launch_bounds__(256) void sum_scaled_k(const float* __restrict__ x, float* __restrict__ out,
                                                    long long n, float scale) {
  __shared__ float sm[17];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += x[i];
  s = block_sum(s, sm);
  if (threadIdx.x == 0) atomicAdd(out, s * scale);
}
__global__ void fill_from_scalar_k(const float* __restrict__ g, float* __restrict__ dx, long long n, float scale) {
  const float v = g[0] * scale;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dx[i] = v;
}
__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, long long n,
                                              float lr, float b1, float b2, float eps, float bc1, float bc2,
                                              float gs) {
  const float step = lr / bc1;
  const float sb2 = sqrtf(bc2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gr = g[i] * gs;
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    const float den = sqrtf(vi) / sb2 + eps;
    p[i] -= step * (mi / den);
  }
}

// ---------------------------------------------------------------------------------------------
extern "C" int dfmir_masked_l1_fwd(const float* a, const float* b, const unsigned char* mask, float thr,
                                   float* ws, float* out, long long n, void* stream) {
  DF_ARG_CHECK(a && b && ws && out && n > 0);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = df_zero_async(ws, 8, st);
  if (e != hipSuccess) return df_set_error((int)e, __FILE__, __LINE__);
  masked_l1_fwd_k<<<df_grid(n, 1024, 128), 1024, 0, st>>>(a, b, mask, thr, ws, n);
  DF_LAUNCH_CHECK();
  masked_l1_fin_k<<<1, 1, 0, st>>>(ws, out);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_masked_l1_bwd(const float* a, const float* b, const unsigned char* mask, float thr,
                                   const float* ws, const float* gout, float* da, float* db, long long n,
                                   void* stream) {
  DF_ARG_CHECK(a && b && ws && gout && n > 0);
  masked_l1_bwd_k<<<df_grid(n, 256, 4096), 256, 0, (hipStream_t)stream>>>(a, b, mask, thr, ws, gout, da, db, n);
  DF_LAUNCH_CHECK();
  return 0;
}
// floats of `ws` (dfmir_flow_smooth_fwd / _fwd_p): the three sums + one slot per workgroup and axis
extern "C" long long dfmir_flow_smooth_ws_floats(void) { return FS_PART + 3LL * FS_MAXWG; }
extern "C" int dfmir_flow_smooth_fwd_p(const float* flow, float* ws, float* out, int B, int C, int D, int H,
                                       int W, int penalty, void* stream) {
  DF_ARG_CHECK(flow && ws && out && B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && (penalty == 1 || penalty == 2));
  const bool l1 = penalty == 1;
  hipStream_t st = (hipStream_t)stream;
  const long long planes = (long long)B * C;
  const long long nrow = planes * D * H;
  unsigned nwg;                                              // workgroups of the pass = partials the finaliser adds (<= FS_MAXWG)
  if (l1) {
    const int rb = (int)((nrow + 2047) / 2048);
    nwg = (unsigned)((nrow + rb - 1) / rb);
    flow_smooth_fwd_k<true><<<nwg, 256, 0, st>>>(flow, ws, planes, D, H, W, rb);
  } else if ((W & 3) == 0 && W <= 256 && nrow < 0x7FFFFFFFLL && (reinterpret_cast<uintptr_t>(flow) & 15) == 0) {
    const int tpr = W <= 64 ? 16 : (W <= 128 ? 32 : 64);
    const unsigned grid = 8 * ((df_grid(nrow * tpr, 8 * 1024, 512) + 7) / 8);      // >= 8 rows per thread: few workgroups
    nwg = grid;
    if (tpr == 16) flow_smooth_fwd_v4_k<16><<<grid, 1024, 0, st>>>(flow, ws, planes, D, H, W);
    else if (tpr == 32) flow_smooth_fwd_v4_k<32><<<grid, 1024, 0, st>>>(flow, ws, planes, D, H, W);
    else flow_smooth_fwd_v4_k<64><<<grid, 1024, 0, st>>>(flow, ws, planes, D, H, W);
  } else {
    const int rb = (int)((nrow + 2047) / 2048);
    nwg = (unsigned)((nrow + rb - 1) / rb);
    flow_smooth_fwd_k<false><<<nwg, 256, 0, st>>>(flow, ws, planes, D, H, W, rb);
  }
  DF_LAUNCH_CHECK();
  DF_ARG_CHECK(nwg <= (unsigned)FS_MAXWG);
  const float cd = (float)((double)planes * (D - 1) * H * W), ch = (float)((double)planes * D * (H - 1) * W),
              cw = (float)((double)planes * D * H * (W - 1));
  const float nd = (D > 1) ? 3.f : 2.f;
  flow_smooth_fin_k<<<1, 256, 0, st>>>(ws, out, cd, ch, cw, nd, (int)nwg);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_flow_smooth_fwd(const float* flow, float* ws, float* out, int B, int C, int D, int H,
                                     int W, void* stream) {
  return dfmir_flow_smooth_fwd_p(flow, ws, out, B, C, D, H, W, 2, stream);
}
static bool fs_no_march() {
  static DfOptFlag o{"DFMIR_SMOOTH_NO_MARCH"};               // A/B: the backward as five row loads per output row
  return o.get();
}
extern "C" int dfmir_flow_smooth_bwd_p(const float* flow, const float* gout, float* dflow, int B, int C, int D,
                                       int H, int W, int penalty, void* stream) {
  DF_ARG_CHECK(flow && gout && dflow && B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && (penalty == 1 || penalty == 2));
  const long long planes = (long long)B * C;
  const double cd = (double)planes * (D - 1) * H * W, ch = (double)planes * D * (H - 1) * W,
               cw = (double)planes * D * H * (W - 1);
  const double nd = (D > 1) ? 3.0 : 2.0;
  const double two = penalty == 1 ? 1.0 : 2.0;          // d|d| = sgn(d), d(d^2) = 2 d
  const float kd = cd > 0 ? (float)(two / (cd * nd)) : 0.f, kh = ch > 0 ? (float)(two / (ch * nd)) : 0.f,
              kw = cw > 0 ? (float)(two / (cw * nd)) : 0.f;
  if (penalty == 1) {
    if (planes * D * H * W < 0x7FFFFFFFLL)
      flow_smooth_bwd_k<unsigned, true><<<df_grid(planes * D * H * W, 256, 1 << 16), 256, 0, (hipStream_t)stream>>>(
          flow, gout, dflow, planes, D, H, W, kd, kh, kw);
    else
      flow_smooth_bwd_k<long long, true><<<df_grid(planes * D * H * W, 256, 4096), 256, 0, (hipStream_t)stream>>>(
          flow, gout, dflow, planes, D, H, W, kd, kh, kw);
  } else if ((W & 3) == 0 && W <= 256 && D >= 8 && H >= FS_RY && !fs_no_march() && (long long)D * H * W * 4 < 0x7FFFFFFFLL &&
             ((reinterpret_cast<uintptr_t>(flow) | reinterpret_cast<uintptr_t>(dflow)) & 15) == 0) {
    FsmP k{D, H, W, (H + FS_RY - 1) / FS_RY, 1, D, 0, kd, kh, kw};
    // z segments: enough items for ~4 workgroups per CU, at least 8 planes each (two halo planes per segment are read again)
    const long long cols = planes * k.nstrip;
    int nseg = (int)((1024 + cols - 1) / cols);              // (768 ... 1024 workgroups measured best: 0.083 / 0.087 / 0.070 / 0.072 / 0.084 ms at 256 / 512 / 768 / 1024 / 1536)
    if (nseg > D / 8) nseg = D / 8;
    if (nseg < 1) nseg = 1;
    k.zlen = (D + nseg - 1) / nseg;
    k.nseg = (D + k.zlen - 1) / k.zlen;
    k.nitem = cols * k.nseg;
    const unsigned grid = (unsigned)(8 * ((k.nitem + 7) / 8));
    flow_smooth_bwd_march_k<<<grid, 256, 0, (hipStream_t)stream>>>(flow, gout, dflow, k);
  } else if ((W & 3) == 0 && W <= 256 && planes * D * H < 0x7FFFFFFFLL &&
      ((reinterpret_cast<uintptr_t>(flow) | reinterpret_cast<uintptr_t>(dflow)) & 15) == 0) {
    const int tpr = W <= 64 ? 16 : (W <= 128 ? 32 : 64);
    const unsigned grid = 8 * ((df_grid(planes * D * H * tpr, 256, 1 << 15) + 7) / 8);
    hipStream_t st = (hipStream_t)stream;
    if (tpr == 16) flow_smooth_bwd_v4_k<16><<<grid, 256, 0, st>>>(flow, gout, dflow, planes, D, H, W, kd, kh, kw);
    else if (tpr == 32) flow_smooth_bwd_v4_k<32><<<grid, 256, 0, st>>>(flow, gout, dflow, planes, D, H, W, kd, kh, kw);
    else flow_smooth_bwd_v4_k<64><<<grid, 256, 0, st>>>(flow, gout, dflow, planes, D, H, W, kd, kh, kw);
  } else if (planes * D * H * W < 0x7FFFFFFFLL)
    flow_smooth_bwd_k<unsigned><<<df_grid(planes * D * H * W, 256, 1 << 16), 256, 0, (hipStream_t)stream>>>(
        flow, gout, dflow, planes, D, H, W, kd, kh, kw);
  else
    flow_smooth_bwd_k<long long><<<df_grid(planes * D * H * W, 256, 4096), 256, 0, (hipStream_t)stream>>>(
        flow, gout, dflow, planes, D, H, W, kd, kh, kw);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_flow_smooth_bwd(const float* flow, const float* gout, float* dflow, int B, int C, int D,
                                     int H, int W, void* stream) {
  return dfmir_flow_smooth_bwd_p(flow, gout, dflow, B, C, D, H, W, 2, stream);
}

// fwd: final 5 box sums are left in `tmp`.
extern "C" int dfmir_ncc_fwd_m(const float* I, const float* J, const float* mask, int mode, float* tmp, float* tmp2,
                               float* ws, float* out, int B, int D, int H, int W, int win, float eps, void* stream) {
  DF_ARG_CHECK(I && J && tmp && tmp2 && ws && out && B > 0 && D > 0 && H > 0 && W > 0 && (win & 1) && (mode == 0 || mode == 1));
  hipStream_t st = (hipStream_t)stream;
  const long long N = (long long)B * D * H * W;
  const int r = win / 2;
  const unsigned grid = (unsigned)((N + 255) / 256);
  hipError_t e = df_zero_async(ws, 8, st);
  if (e != hipSuccess) return df_set_error((int)e, __FILE__, __LINE__);
  float wn;
  static DfOptFlag nofuse_o{"DFMIR_NCC_NO_WH_FUSE"};         // A/B: the W and H passes as separate launches
  if (D > 1) {
    if (r == 4 && !nofuse_o.get()) {
      const int nty = (H + 31) / 32, ntx = (W + 63) / 64;
      ncc_prod_boxwh_k<<<(unsigned)((long long)B * D * nty * ntx), 256, 0, st>>>(I, J, tmp2, N, H, W, nty, ntx);
    } else {
      ncc_prod_boxw_launch(I, J, tmp, N, W, r, st);
      DF_LAUNCH_CHECK();
      box_axis_launch(tmp, tmp2, 5, N, W, H, r, st);
    }
    DF_LAUNCH_CHECK();
    box_axis_launch(tmp2, tmp, 5, N, (long long)H * W, D, r, st);
    DF_LAUNCH_CHECK();
    wn = (float)win * win * win;
  } else {
    ncc_prod_boxw_launch(I, J, tmp2, N, W, r, st);
    DF_LAUNCH_CHECK();
    box_axis_launch(tmp2, tmp, 5, N, W, H, r, st);
    DF_LAUNCH_CHECK();
    wn = (float)win * win;
  }
  // tmp2 is free again after the box passes: its head takes the per-workgroup partials (volumes of >= 2048 / 5 voxels)
  const unsigned ng = df_grid(N, 256, 1024);
  float* part = (5 * N >= 2LL * ng) ? tmp2 : nullptr;
  ncc_cc_reduce_k<<<ng, 256, 0, st>>>(tmp, mask, ws, N, wn, eps, part);
  DF_LAUNCH_CHECK();
  ncc_fin_k<<<1, 256, 0, st>>>(ws, out, (float)N, mode, mask != nullptr, part, (int)ng);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_ncc_fwd(const float* I, const float* J, float* tmp, float* tmp2, float* ws, float* out,
                             int B, int D, int H, int W, int win, float eps, void* stream) {
  return dfmir_ncc_fwd_m(I, J, nullptr, 0, tmp, tmp2, ws, out, B, D, H, W, win, eps, stream);
}
// bwd: sums = the 5N box sums of the forward; tmp, tmp2 = 3N floats of scratch each.
extern "C" int dfmir_ncc_bwd_m(const float* I, const float* J, const float* mask, int mode, const float* sums,
                               float* tmp, float* tmp2, const float* ws, const float* gout, float* dI, int B, int D,
                               int H, int W, int win, float eps, void* stream) {
  DF_ARG_CHECK(I && J && sums && tmp && tmp2 && ws && gout && dI && B > 0 && D > 0 && H > 0 && W > 0 && (win & 1) &&
               (mode == 0 || mode == 1));
  hipStream_t st = (hipStream_t)stream;
  const long long N = (long long)B * D * H * W;
  const int r = win / 2;
  const unsigned grid = (unsigned)((N + 255) / 256);
  const float wn = (D > 1) ? (float)win * win * win : (float)win * win;
  static DfOptFlag nofuse_o{"DFMIR_NCC_NO_WH_FUSE"};         // A/B: fields, W and H passes as separate launches
  if (r == 4 && D > 1 && !nofuse_o.get()) {
    const int nty = (H + 31) / 32, ntx = (W + 63) / 64;
    ncc_fields_boxwh_k<<<(unsigned)((long long)B * D * nty * ntx), 256, 0, st>>>(sums, ws, gout, tmp, N, H, W, nty, ntx, wn, eps, mask, mode);
    DF_LAUNCH_CHECK();
  } else {
    ncc_fields_k<<<grid, 256, 0, st>>>(sums, ws, gout, tmp, N, wn, eps, mask, mode);
    DF_LAUNCH_CHECK();
    box_axis_launch(tmp, tmp2, 3, N, 1, W, r, st);
    DF_LAUNCH_CHECK();
    box_axis_launch(tmp2, tmp, 3, N, W, H, r, st);
    DF_LAUNCH_CHECK();
  }
  const float* fin = tmp;
  static DfOptFlag nofuse_d{"DFMIR_NCC_NO_D_FUSE"};          // A/B: the D pass and the combination as separate launches
  if (D > 1 && r == 4 && (long long)H * W > 1 && !nofuse_d.get()) {
    constexpr int SEG = 36;
    const long long thr = ((long long)B * H * W) * ((D + SEG - 1) / SEG);
    ncc_boxd_combine_k<4, SEG><<<(unsigned)((thr + 255) / 256), 256, 0, st>>>(tmp, I, J, dI, N, (long long)H * W, D);
    DF_LAUNCH_CHECK();
    return 0;
  }
  if (D > 1) {
    box_axis_launch(tmp, tmp2, 3, N, (long long)H * W, D, r, st);
    DF_LAUNCH_CHECK();
    fin = tmp2;
  }
  ncc_combine_k<<<grid, 256, 0, st>>>(I, J, fin, dI, N);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_ncc_bwd(const float* I, const float* J, const float* sums, float* tmp, float* tmp2,
                             const float* ws, const float* gout, float* dI, int B, int D, int H, int W,
                             int win, float eps, void* stream) {
  return dfmir_ncc_bwd_m(I, J, nullptr, 0, sums, tmp, tmp2, ws, gout, dI, B, D, H, W, win, eps, stream);
}
// out = a * b, element-wise (the `prediction * mask` of Grad_Loss.forward, util/losses.py:120-121, and its gradient)
__global__ __launch_bounds__(256) void mul_k(const float* __restrict__ a, const float* __restrict__ b,
                                             float* __restrict__ o, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) o[i] = a[i] * b[i];
}
extern "C" int dfmir_mul(const float* a, const float* b, float* out, long long n, void* stream) {
  DF_ARG_CHECK(a && b && out && n > 0);
  mul_k<<<df_grid(n, 256, 8192), 256, 0, (hipStream_t)stream>>>(a, b, out, n);
  DF_LAUNCH_CHECK();
  return 0;
}
// ---------------------------------------------------------------------------------------------
// Deterministic weight gradients (include/dfmir_hip.h).  fx lives at the head of the caller's scratch; the 8-byte sum
// slots follow at float offset DF_DET_HEAD.
static thread_local const float* tl_det_fx = nullptr;
const float* df_det_fx() { return tl_det_fx; }
constexpr int DF_DET_HEAD = 8;
__global__ __launch_bounds__(256) void det_scale_k(const float* __restrict__ x_amax, int xn, const float* __restrict__ dy_amax,
                                                   int dyn, float count, float* __restrict__ fx) {
  __shared__ float sm[17];
  const float ax = reduce_absmax(x_amax, xn, sm);
  __syncthreads();
  const float ay = reduce_absmax(dy_amax, dyn, sm);
  if (threadIdx.x) return;
  // every sum the kernels form -- dW[.] = sum x * dy over `count` positions, db[.] = sum dy -- is bounded by B
  const float B = count * fmaxf(ax, 1.f) * ay;
  int e = 0;
  if (B > 0.f && B < 3.0e38f) frexpf(B, &e);               // B < 2^e
  int sh = 61 - e;
  sh = sh > 120 ? 120 : (sh < -120 ? -120 : sh);
  fx[0] = ldexpf(1.f, sh);
  fx[1] = ldexpf(1.f, -sh);
}
__global__ __launch_bounds__(256) void det_final_k(const long long* __restrict__ slots, const float* __restrict__ fx,
                                                   float* __restrict__ out, long long n) {
  const double inv = (double)fx[1];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] += (float)((double)slots[i] * inv);
}
extern "C" int dfmir_det_head_floats(void) { return DF_DET_HEAD; }
extern "C" int dfmir_det_begin(float* scratch, long long slots, const float* x, long long nx, const float* x_amax, int x_amax_n,
                               const float* dy, long long ndy, const float* dy_amax, int dy_amax_n, double count,
                               void* stream) {
  DF_ARG_CHECK(scratch && slots > 0 && count > 0 && (x_amax || (x && nx > 0)) && (dy_amax || (dy && ndy > 0)) &&
               (reinterpret_cast<uintptr_t>(scratch) & 15) == 0 && tl_det_fx == nullptr);
  hipStream_t st = (hipStream_t)stream;
  {
    const int rc = dfmir_fill_zero(scratch, DF_DET_HEAD + 2 * slots, stream);
    if (rc) return rc;
  }
  if (!x_amax) {                                           // no range probe came with the operand: measure it
    const int rc = dfmir_absmax(x, nx, scratch + 4, stream);
    if (rc) return rc;
    x_amax = scratch + 4; x_amax_n = 1;
  }
  if (!dy_amax) {
    const int rc = dfmir_absmax(dy, ndy, scratch + 5, stream);
    if (rc) return rc;
    dy_amax = scratch + 5; dy_amax_n = 1;
  }
  det_scale_k<<<1, 256, 0, st>>>(x_amax, x_amax_n, dy_amax, dy_amax_n, (float)count, scratch);
  DF_LAUNCH_CHECK();
  tl_det_fx = scratch;
  return 0;
}
extern "C" int dfmir_det_end(const float* scratch, float* dw_out, long long n_dw, float* db_out, long long n_db, void* stream) {
  tl_det_fx = nullptr;
  DF_ARG_CHECK(scratch && n_dw >= 0 && n_db >= 0 && (n_dw == 0 || dw_out) && (n_db == 0 || db_out));
  hipStream_t st = (hipStream_t)stream;
  const long long* slots = reinterpret_cast<const long long*>(scratch + DF_DET_HEAD);
  if (n_dw) det_final_k<<<df_grid(n_dw, 256, 2048), 256, 0, st>>>(slots, scratch, dw_out, n_dw);
  if (n_db) det_final_k<<<df_grid(n_db, 256, 64), 256, 0, st>>>(slots + n_dw, scratch, db_out, n_db);
  DF_LAUNCH_CHECK();
  return 0;
}
// p[0..n) = 0 as a kernel (see df_zero_async in common.h for why the step never uses hipMemsetAsync)
__global__ __launch_bounds__(256) void fill_zero_k(float* __restrict__ p, long long n) {
  const long long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) p4[i] = z;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = 0.f;
}
extern "C" int dfmir_fill_zero(float* p, long long n, void* stream) {
  DF_ARG_CHECK(p && n > 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0);
  fill_zero_k<<<df_grid((n + 3) / 4, 256, 8192), 256, 0, (hipStream_t)stream>>>(p, n);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_sum_scaled(const float* x, float* out, long long n, float scale, void* stream) {
  DF_ARG_CHECK(x && out && n > 0);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = df_zero_async(out, 1, st);
  if (e != hipSuccess) return df_set_error((int)e, __FILE__, __LINE__);
  sum_scaled_k<<<df_grid(n, 256, 1024), 256, 0, st>>>(x, out, n, scale);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_fill_from_scalar(const float* gout, float* dx, long long n, float scale, void* stream) {
  DF_ARG_CHECK(gout && dx && n > 0);
  fill_from_scalar_k<<<df_grid(n, 256, 4096), 256, 0, (hipStream_t)stream>>>(gout, dx, n, scale);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                               float beta2, float eps, float bc1, float bc2, float grad_scale, void* stream) {
  DF_ARG_CHECK(p && g && m && v && n > 0);
  adam_k<<<df_grid(n, 256, 8192), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2,
                                                                grad_scale);
  DF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
int df_set_error(int code, const char* file, int line) {
  const char* msg = code > 0 ? hipGetErrorString((hipError_t)code) : "invalid argument";
  snprintf(g_err, sizeof(g_err), "dfmir_hip: %s (code %d) at %s:%d", msg, code, file, line);
  return code;
}
extern "C" const char* dfmir_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// Options: one process-global table (name -> value or "explicitly unset") in front of the environment.
namespace {
struct DfOptTable {
  std::mutex mu;
  std::map<std::string, std::pair<bool, std::string>> over;    // name -> (is set, value)
  std::atomic<int> gen{0};
};
DfOptTable& df_opt_table() {
  static DfOptTable* t = new DfOptTable();      // never destroyed: kernels' static caches may outlive exit handlers
  return *t;
}
}  // namespace
bool df_opt_get(const char* name, char* buf, int buf_len) {
  DfOptTable& t = df_opt_table();
  std::lock_guard<std::mutex> lk(t.mu);        // the copy happens under the lock: no pointer into the table escapes
  const char* v = nullptr;
  auto it = t.over.find(name);
  if (it != t.over.end()) v = it->second.first ? it->second.second.c_str() : nullptr;
  else v = getenv(name);
  if (!v) return false;
  if (buf && buf_len > 0) { strncpy(buf, v, (size_t)buf_len - 1); buf[buf_len - 1] = 0; }
  return true;
}
int df_opt_gen() { return df_opt_table().gen.load(std::memory_order_acquire); }
extern "C" int dfmir_set_option(const char* name, const char* value) {
  DF_ARG_CHECK(name && strncmp(name, "DFMIR_", 6) == 0);
  DfOptTable& t = df_opt_table();
  {
    std::lock_guard<std::mutex> lk(t.mu);
    t.over[name] = std::make_pair(value != nullptr, std::string(value ? value : ""));
  }
  t.gen.fetch_add(1, std::memory_order_release);
  return 0;
}
extern "C" int dfmir_get_option(const char* name, char* buf, int buf_len) {
  if (!name) return -1;
  char tmp[512];
  if (!df_opt_get(name, tmp, sizeof(tmp))) return -1;
  const int n = (int)strlen(tmp);
  if (buf && buf_len > 0) { strncpy(buf, tmp, (size_t)buf_len - 1); buf[buf_len - 1] = 0; }
  return n;
}
extern "C" int dfmir_abi_version(void) { return DFMIR_ABI_VERSION; }

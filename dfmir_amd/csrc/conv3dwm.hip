// Weight (and bias) gradient of the full-resolution 32 -> 16 ConvBlock of VoxelMorph's `extras` chain (3x3x3, stride 1,
// padding 1: torchvoxelmorph/networks.py:73-86,1506-1521) as a z-MARCHING kernel with ALL 27 tap matrices resident:
//     dW[t][ci][co] = sum_v x[v + t - 1][ci] dY[v][co] = sum_u x[u][ci] dY[u - (t - 1)][co]
// -- K = the positions u of the 32-channel operand x, read unshifted; the 16-channel operand dY is the one that is shifted by
// the (flipped) tap, so an MFMA column block = one tap x 16 output channels and a 32 x 32 tile = 32 input channels x TWO taps:
// 27 taps = 14 tiles = 224 accumulator registers, which ONE wave holds (256 AGPRs per lane with one wave per SIMD; csrc/
// conv3duw.hip has the register story).  The four waves of a workgroup split the K blocks of a plane (rows 2w, 2w + 1 of the
// 8 x 32 tile) and march along z: every plane of x (8 x 32 voxels x 32 channels, two buffers) and of dY (10 x 34 positions x
// 16 channels, ring of four) is staged ONCE as scaled fp16 pairs -- in-plane halo of the 16-channel operand only (HBM-side
// bytes 1.1 x algorithmic; the tiled conv3d_wgrad_tr_k stages a 3.75 x halo of the 32-channel operand per 8-channel chunk and dY
// twice).  No padded taps' (the plane-pair form of conv3d_wgrad_tr_k issues 36 / 27 of the useful products; here 28 / 27).
// Operands: K-major ds_read_b64_tr_b16 reads of channel-minor images, as in conv3d_wgrad_tr_k / conv3d_upwgrad4_k; the two taps
// of a tile are one row (oy = -1 | 0), one plane (oz = -1 | 0) or two columns apart, with row stride 36 positions and a slot
// stride = 128 (mod 256) bytes so that the two 128-byte windows of a 32-lane read fall on different banks.  Workgroups end
// with atomics straight into the tap-major gradient (each accumulator row IS a (tap, co) row).  The bias gradient is summed
// from the interior dY quads as they are staged.
#include "conv3x3_common.h"
#include <type_traits>

typedef _Float16 f16x8_w __attribute__((ext_vector_type(8)));
typedef short s16x4_w __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_w* lds_tr_ptr_w;
#ifndef WM_KO
#define WM_KO 0      // knock-out builds (timing only): 1 no MFMAs, 2 no staging loads, 4 no conversion + LDS stores, 8 no operand reads
#endif

namespace {

__device__ __forceinline__ int scale_exp_w(float amax) {
  const int be = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  int e = (amax > 0.f) ? 14 - be : 0;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return e;
}
__device__ __forceinline__ float pow2f_w(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
__device__ __forceinline__ void split_pair_w(float x0, float x1, float s, unsigned& h, unsigned& r) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(x1), "v"(s), "v"(h));
}
__device__ __forceinline__ f32x16 mma_w(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_w, a), __builtin_bit_cast(f16x8_w, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint2 tr_read_w(unsigned byte_addr) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_w)(uintptr_t)byte_addr));
}
__device__ __forceinline__ u32x4 tr_pair_w(unsigned a0, unsigned a1) {
  const uint2 u0 = tr_read_w(a0), u1 = tr_read_w(a1);
  return u32x4{u0.x, u0.y, u1.x, u1.y};
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for_w(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for_w<B + 1, E>(f);
  }
}

struct WmP {
  int N, D, H, W;
  int x_n, dy_n;                 // floats of the two range probes
  int ncy, ncx, nseg, zlen;      // 8 x 32 columns per plane, z segments of zlen planes
  int nitems;                    // N * nseg * ncy * ncx
  long long s_tap;               // dwt index = tap * s_tap + ci * 16 + co
  float* db;                     // optional: db[co] += sum of dY
  const float* fx;               // deterministic mode (common.h df_acc): dwt holds 64-bit fixed-point sums, db is NULL
};

constexpr unsigned WM_FSPLIT = 8u * 32u * 64u;      // one split of an x plane: 8 x 32 voxels x 32 channels x fp16 = 16 KB
constexpr unsigned WM_FBUF = 2u * WM_FSPLIT;
constexpr unsigned WM_SROW = 36u * 32u;             // dY: row of 36 positions x 16 channels x fp16 (34 used): = 128 (mod 256)
constexpr unsigned WM_SSPLIT = 10u * WM_SROW;       // 11 520
constexpr unsigned WM_SSLOT = 2u * WM_SSPLIT + 128u;   // = 128 (mod 256): planes one apart fall on the other half of the banks
constexpr unsigned WM_SOFF = 2u * WM_FBUF;          // 65 536
constexpr unsigned WM_LDS = WM_SOFF + 4u * WM_SSLOT;   // 158 208 bytes

// offsets (oz, oy, ox) in -1..1 of the two taps of tile T_ (g = 0: columns 0-15, g = 1: columns 16-31); tile 13's second tap
// repeats the first (its columns are discarded)
__device__ constexpr int wm_oz(int T, int g) { return T < 9 ? T / 3 - 1 : (T < 12 ? (g ? 0 : -1) : 1); }
__device__ constexpr int wm_oy(int T, int g) { return T < 9 ? (g ? 0 : -1) : 1; }
__device__ constexpr int wm_ox(int T, int g) { return T < 9 ? T % 3 - 1 : (T < 12 ? T - 10 : (T == 12 ? (g ? 1 : -1) : 0)); }

__global__ __launch_bounds__(256, 1) void conv3d_wgrad_march_k(const float* __restrict__ x, const float* __restrict__ x_amax,
                                                               const float* __restrict__ dy, const float* __restrict__ dy_amax,
                                                               float* __restrict__ dwt, WmP k) {
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) unsigned char lds[WM_LDS];
  __shared__ float red[17];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  const int ex = scale_exp_w(reduce_absmax(x_amax, k.x_n, red));
  __syncthreads();
  const int ed = scale_exp_w(reduce_absmax(dy_amax, k.dy_n, red));
  const float xscale = pow2f_w(ex), dscale = pow2f_w(ed), osc_x = pow2f_w(-ex), osc_d = pow2f_w(-ed);

  const unsigned HW = (unsigned)(k.H * k.W), S4 = HW * (unsigned)k.D * 4u;

  f32x16 acc[14];
#pragma unroll
  for (int t = 0; t < 14; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- operand addresses (bytes in LDS).  Source role of this lane in its 16-lane group: voxel sj, channel quad sq.
  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const int sj = (lane & 15) >> 2, sq = lane & 3, sg = (lane >> 4) & 1;
  // x (rows = 32 input channels): voxel vx = 8 hi + 4 i + sj of the K block, channel quad 4 sg + sq, unit swizzled by (vx >> 2)
  unsigned fl[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int vx = 8 * hi + 4 * i + sj, cq = 4 * sg + sq;
    fl[i] = lbase + (unsigned)((vx * 4 + ((cq >> 1) ^ ((vx >> 2) & 3))) * 16 + (cq & 1) * 8);
  }
  // dY (columns = tap of the lane group x 16 output channels): position (8 hi + sj) of the K block, + 128 for the second read
  const unsigned sl = lbase + WM_SOFF + (unsigned)((8 * hi + sj) * 32 + sq * 8);

  // ---- staging roles.  x: thread = (channel group = wave, row r of 8, aligned quad q of 8): eight 16-byte loads.  dY: 200 jobs
  // (channel group of 2, row of 10, aligned quad of 10: x0 - 4 + 4 q ..), eight 16-byte loads; elements outside the 34
  // columns of the patch are not stored.
  const int fq = tid & 7, fr = (tid >> 3) & 7;
  const bool sjob = tid < 200;
  // (channel group in the lowest bit: the 8 lanes of a ds_write_b128 group then cover 4 of the 8 possible 16-byte slots of
  // the 256-byte bank window twice instead of 2 of them four times)
  const int js = sjob ? tid : 0, scg = js & 1, srr = js >> 1, shy = srr / 10, sqq = srr - 10 * shy;
  const unsigned fst0 = lbase + (unsigned)(((fr * 32 + 4 * fq) * 4 + (wid ^ (fq & 3))) * 16);     // element e: + e * 64
  unsigned sst[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int hx = 4 * sqq - 3 + e;
    sst[e] = (sjob && hx >= 0 && hx <= 33) ? lbase + WM_SOFF + (unsigned)(shy * (int)WM_SROW + hx * 32 + scg * 16) : OOB;
  }
  const bool sinterior = sjob && shy >= 1 && shy <= 8;      // rows of the tile itself (bias gradient)

  u32x4 rf[8], rs[8];                                       // the x plane / the dY plane in flight
  float bacc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) bacc[c] = 0.f;
  unsigned fbase = OOB, sbase = OOB;
  __amdgpu_buffer_rsrc_t fsrc, ssrc;
  int z0 = 0, z1 = 0;

#define WM_FLOAD(c_, off_)                                                                        \
  if (!(WM_KO & 2)) rf[c_] = __builtin_amdgcn_raw_buffer_load_b128(fsrc, (off_), (unsigned)(wid * 8 + (c_)) * S4, 0);
#define WM_SLOAD(c_, off_)                                                                        \
  if (!(WM_KO & 2)) rs[c_] = __builtin_amdgcn_raw_buffer_load_b128(ssrc, (off_), (unsigned)(c_) * S4, 0);
#define WM_LDS_ST(addr_, v_) *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(addr_) = (v_);
#define WM_CONV_ST(rq_, e_, scale_, addr_, lo_)                                                   \
  if (!(WM_KO & 4)) {                                                                             \
    u32x4 h_, r_;                                                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                            \
      unsigned hh_, rr_;                                                                          \
      split_pair_w(__uint_as_float(rq_[2 * q_][e_]), __uint_as_float(rq_[2 * q_ + 1][e_]), scale_, hh_, rr_); \
      h_[q_] = hh_; r_[q_] = rr_;                                                                 \
    }                                                                                             \
    WM_LDS_ST(addr_, h_) WM_LDS_ST((addr_) + (lo_), r_)                                           \
  }
#define WM_FCONV(e_, buf_) WM_CONV_ST(rf, e_, xscale, fst0 + (unsigned)(buf_) * WM_FBUF + (unsigned)(e_) * 64u, WM_FSPLIT)
#define WM_SCONV(e_, slot_) if (sst[e_] != OOB) { WM_CONV_ST(rs, e_, dscale, sst[e_] + (slot_), WM_SSPLIT) }
  // bias gradient: the interior elements of the quad (columns 1..32 of the patch), planes of this item's own segment only
#define WM_DBSUM(own_)                                                                            \
  if (k.db && sinterior && (own_)) {                                                              \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                            \
      const int hx_ = 4 * sqq - 3 + e_;                                                           \
      if (hx_ >= 1 && hx_ <= 32) {                                                                \
        _Pragma("unroll") for (int c_ = 0; c_ < 8; ++c_) bacc[c_] += __uint_as_float(rs[c_][e_]); \
      }                                                                                           \
    }                                                                                             \
  }
#define WM_FOFFS(P_, ok_) ((fbase != OOB && (ok_)) ? fbase + (unsigned)(P_) * HW * 4u : OOB)
#define WM_SOFFS(P_, ok_) ((sbase != OOB && (ok_)) ? sbase + (unsigned)(P_) * HW * 4u : OOB)

  // One plane step: x plane P (buffer PB_) against the dY planes P - 1, P, P + 1 (ring slots sm_, s0_, sp_).  56 groups =
  // 4 K blocks (rows 2 w, 2 w + 1 x two halves of 16 voxels) x 14 tiles.  Meanwhile: the x plane P + 1 (loaded during the
  // previous step) is converted into buffer PB_ ^ 1 (groups 1-4) and the loads of plane P + 2 follow (groups 5-12); the dY
  // plane P + 2 (loaded during the previous step) is converted into the free slot sn_ (groups 15-18) and the loads of plane
  // P + 3 follow (groups 19-26).
#define WM_STEP(PB_, sm_, s0_, sp_, sn_, foff_, soff_, own_)                                      \
  {                                                                                               \
    const unsigned fb0_ = fl[0] + (PB_) * WM_FBUF, fb1_ = fl[1] + (PB_) * WM_FBUF;                \
    unsigned toff[14];            /* this lane's tap of each tile: slot of its plane + row + column offset */ \
    static_for_w<0, 14>([&](auto tc_) __attribute__((always_inline)) {                            \
      constexpr int T_ = decltype(tc_)::value;                                                    \
      constexpr int z0_ = wm_oz(T_, 0), z1_ = wm_oz(T_, T_ == 13 ? 0 : 1);                        \
      constexpr unsigned c0_ = (unsigned)((1 + wm_oy(T_, 0)) * (int)WM_SROW + (1 + wm_ox(T_, 0)) * 32); \
      constexpr unsigned c1_ = T_ == 13 ? c0_ : (unsigned)((1 + wm_oy(T_, 1)) * (int)WM_SROW + (1 + wm_ox(T_, 1)) * 32); \
      const unsigned a0_ = (z0_ < 0 ? (sm_) : (z0_ == 0 ? (s0_) : (sp_))) + c0_;                  \
      const unsigned a1_ = (z1_ < 0 ? (sm_) : (z1_ == 0 ? (s0_) : (sp_))) + c1_;                  \
      toff[T_] = sl + (sg ? a1_ : a0_);                                                           \
    });                                                                                           \
    unsigned fo_ = OOB, so_ = OOB;                                                                \
    u32x4 F0, F1, S0[2], S1[2];                                                                   \
    auto rds_ = [&](auto gc2_) __attribute__((always_inline)) {   /* dY operand of group g into buffer g & 1 */ \
      constexpr int g2_ = decltype(gc2_)::value;                                                  \
      if constexpr (g2_ < 56) {                                                                   \
        constexpr int kb2_ = g2_ / 14, T2_ = g2_ % 14;                                            \
        const unsigned a_ = toff[T2_] + (unsigned)((2 * wid + (kb2_ >> 1)) * (int)WM_SROW + (kb2_ & 1) * 512); \
        S0[g2_ & 1] = tr_pair_w(a_, a_ + 128u);                                                   \
        S1[g2_ & 1] = tr_pair_w(a_ + WM_SSPLIT, a_ + 128u + WM_SSPLIT);                           \
      }                                                                                           \
    };                                                                                            \
    rds_(std::integral_constant<int, 0>{});                                                       \
    static_for_w<0, 56>([&](auto gc_) __attribute__((always_inline)) {                            \
      constexpr int g_ = decltype(gc_)::value, kb_ = g_ / 14, T_ = g_ % 14;                       \
      if constexpr (T_ == 0) {      /* the x operand of this K block */                           \
        const unsigned fo2_ = (unsigned)(((2 * wid + (kb_ >> 1)) * 32 + (kb_ & 1) * 16) * 64);    \
        if (!(WM_KO & 8) || kb_ == 0) {                                                           \
          F0 = tr_pair_w(fb0_ + fo2_, fb1_ + fo2_);                                               \
          F1 = tr_pair_w(fb0_ + fo2_ + WM_FSPLIT, fb1_ + fo2_ + WM_FSPLIT);                       \
        }                                                                                         \
      }                                                                                           \
      if (!(WM_KO & 8) || g_ == 0) rds_(std::integral_constant<int, g_ + 1>{});                   \
      if constexpr (g_ >= 1 && g_ < 5) { WM_FCONV(g_ - 1, (PB_) ^ 1) }                            \
      if constexpr (g_ == 5) fo_ = (foff_);                                                       \
      if constexpr (g_ >= 5 && g_ < 13) { WM_FLOAD(g_ - 5, fo_) }                                 \
      if constexpr (g_ == 14) { WM_DBSUM(own_) }                                                  \
      if constexpr (g_ >= 15 && g_ < 19) { WM_SCONV(g_ - 15, sn_) }                               \
      if constexpr (g_ == 19) so_ = (soff_);                                                      \
      if constexpr (g_ >= 19 && g_ < 27) { WM_SLOAD(g_ - 19, so_) }                               \
      if (!(WM_KO & 1)) {                                                                         \
        acc[T_] = mma_w(F1, S0[g_ & 1], acc[T_]);                                                 \
        acc[T_] = mma_w(F0, S1[g_ & 1], acc[T_]);                                                 \
        acc[T_] = mma_w(F0, S0[g_ & 1], acc[T_]);                                                 \
      } else {                                                                                    \
        acc[T_][0] += __uint_as_float(F0[0] ^ F1[1] ^ S0[g_ & 1][2] ^ S1[g_ & 1][3]);             \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    });                                                                                           \
  }

  // Items: XCD e = id & 7 owns the e-th eighth of the item list (x fastest, then y), see conv3d_upwgrad4_k
  const int xcd = blockIdx.x & 7, nslot = ((int)gridDim.x + 7 - xcd) >> 3;
  const int per_xcd = (k.nitems + 7) >> 3, it_end = (xcd + 1) * per_xcd < k.nitems ? (xcd + 1) * per_xcd : k.nitems;
  for (int item = xcd * per_xcd + (int)(blockIdx.x >> 3); item < it_end; item += nslot) {
    int q = item;
    const int cx = q % k.ncx; q /= k.ncx;
    const int cy = q % k.ncy; q /= k.ncy;
    const int seg = q % k.nseg;
    const int n = q / k.nseg;
    z0 = seg * k.zlen;
    z1 = z0 + k.zlen < k.D ? z0 + k.zlen : k.D;
    const int y0 = cy * 8, x0 = cx * 32;
    fsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)n * 32 * (S4 >> 2)), 0, 32u * S4, 0x00020000);
    ssrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + (long long)n * 16 * (S4 >> 2)), 0, 16u * S4, 0x00020000);
    {
      const int gy = y0 + fr, gx = x0 + 4 * fq;
      fbase = (gy < k.H && gx < k.W) ? (unsigned)(gy * k.W + gx) * 4u : OOB;
      const int sy = y0 - 1 + shy, sx = x0 - 4 + 4 * sqq;
      sbase = (sjob && (unsigned)sy < (unsigned)k.H && (unsigned)sx < (unsigned)k.W) ? (unsigned)(sy * k.W + sx) * 4u + (unsigned)(scg * 8) * S4 : OOB;
    }
    // ---- prologue: dY planes z0 - 1, z0, z0 + 1 into slots 0, 1, 2 (plane z0 + 2 stays in flight); x plane z0 into buffer 0
    // (plane z0 + 1 stays in flight)
    unsigned t0 = 0, t1 = WM_SSLOT, t2 = 2 * WM_SSLOT, t3 = 3 * WM_SSLOT;
    {
      const unsigned f0 = WM_FOFFS(z0, true);
#pragma unroll
      for (int c = 0; c < 8; ++c) { WM_FLOAD(c, f0) }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const int P = z0 - 1 + p;
        const unsigned s0 = WM_SOFFS(P, P >= 0 && P < k.D);
#pragma unroll
        for (int c = 0; c < 8; ++c) { WM_SLOAD(c, s0) }
        WM_DBSUM(p == 1 || (p == 2 && z0 + 1 < z1))
#pragma unroll
        for (int e = 0; e < 4; ++e) { WM_SCONV(e, (unsigned)p * WM_SSLOT) }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { WM_FCONV(e, 0) }
      const unsigned f1 = WM_FOFFS(z0 + 1, z0 + 1 < z1), s3 = WM_SOFFS(z0 + 2, z0 + 2 < k.D);
#pragma unroll
      for (int c = 0; c < 8; ++c) { WM_FLOAD(c, f1) WM_SLOAD(c, s3) }
    }
    __syncthreads();

    for (int z = z0; z < z1; z += 2) {
      {   // even step: buffer 0; dY plane z + 2 (in rs) -> slot t3, then the loads of plane z + 3
        WM_STEP(0, t0, t1, t2, t3, WM_FOFFS(z + 2, z + 2 < z1), WM_SOFFS(z + 3, z + 3 < k.D), z + 2 < z1)
        __syncthreads();
        const unsigned t = t0; t0 = t1; t1 = t2; t2 = t3; t3 = t;
      }
      if (z + 1 < z1) {   // odd step: buffer 1
        WM_STEP(1, t0, t1, t2, t3, WM_FOFFS(z + 3, z + 3 < z1), WM_SOFFS(z + 4, z + 4 < k.D), z + 3 < z1)
        __syncthreads();
        const unsigned t = t0; t0 = t1; t1 = t2; t2 = t3; t3 = t;
      }
    }
  }
#undef WM_FLOAD
#undef WM_SLOAD
#undef WM_LDS_ST
#undef WM_CONV_ST
#undef WM_FCONV
#undef WM_SCONV
#undef WM_DBSUM
#undef WM_FOFFS
#undef WM_SOFFS
#undef WM_STEP

  // ---- epilogue: acc[T][r] <-> row ci = (r >> 2) * 8 + hi * 4 + (r & 3), column (g, co) = (l31 >> 4, l31 & 15) of tile T;
  // the lane's tap offset (oz, oy, ox) -> weight tap (1 - oz, 1 - oy, 1 - ox)
  {
    const float sc = osc_x * osc_d;
    const int g = l31 >> 4, co = l31 & 15;
    static_for_w<0, 14>([&](auto tc_) __attribute__((always_inline)) {
      constexpr int T = decltype(tc_)::value;
      const int oz = g ? wm_oz(T, 1) : wm_oz(T, 0), oy = g ? wm_oy(T, 1) : wm_oy(T, 0), ox = g ? wm_ox(T, 1) : wm_ox(T, 0);
      const int tap = (1 - oz) * 9 + (1 - oy) * 3 + (1 - ox);
      if (!(T == 13 && g)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = (r >> 2) * 8 + hi * 4 + (r & 3);
          df_acc(dwt, tap * k.s_tap + ci * 16 + co, acc[T][r] * sc, k.fx);
        }
      }
    });
  }
  if (k.db) {   // even threads below 200: channels 0-7, odd ones: channels 8-15
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float v0 = wave_sum((sjob && scg == 0) ? bacc[c] : 0.f), v1 = wave_sum((sjob && scg == 1) ? bacc[c] : 0.f);
      if (lane == 0 && v0 != 0.f) atomicAdd(&k.db[c], v0);
      if (lane == 0 && v1 != 0.f) atomicAdd(&k.db[8 + c], v1);
    }
  }
}

}  // namespace

// Host side (conv3ds.hip::conv3d_split_wgrad_impl calls this for the 32 -> 16 layer).  dwt: tap-major [27][32][16].
int df_conv3d_wgrad_march_launch(const float* x, const float* x_amax, int x_n, const float* dy, const float* dy_amax, int dy_n,
                                 float* dwt, float* db, int N, int D, int H, int W, hipStream_t st) {
  WmP k{};
  k.N = N; k.D = D; k.H = H; k.W = W;
  k.x_n = x_n; k.dy_n = dy_n;
  k.ncy = (H + 7) / 8; k.ncx = (W + 31) / 32;
  k.s_tap = 32LL * 16;
  k.db = db;
  k.fx = df_det_fx();
  const int ncu = df_cu_count();
  const long long cols = (long long)N * k.ncy * k.ncx;
  static DfOptInt nseg_o{"DFMIR_WGRAD_MARCH_NSEG", 0};
  int best = 1;
  long long best_cost = -1;
  for (int s = 1; s <= D && s <= 64; ++s) {
    const int zl = (D + s - 1) / s;
    const int ns = (D + zl - 1) / zl;
    const long long rounds = (cols * ns + ncu - 1) / ncu;
    const long long cost = rounds * (zl + 3);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ns; }
  }
  const long long forced = nseg_o.get();
  if (forced > 0 && forced <= D) best = (int)forced;
  k.zlen = (D + best - 1) / best;
  k.nseg = (D + k.zlen - 1) / k.zlen;
  k.nitems = (int)(cols * k.nseg);
  const unsigned grid = (unsigned)(k.nitems < ncu ? k.nitems : ncu);
  conv3d_wgrad_march_k<<<grid, 256, 0, st>>>(x, x_amax, dy, dy_amax, dwt, k);
  DF_LAUNCH_CHECK();
  return 0;
}

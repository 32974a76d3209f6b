// Weight gradient of the decoder ConvBlocks that consume cat(nearest_up2(a), b) (torchvoxelmorph/networks.py:64,97-100,
// 1506-1521: nn.Upsample(scale 2, nearest) + torch.cat + Conv3d(3, padding 1)) -- the share of the UP-SAMPLED channels, in
// PARITY CLASSES.  The forward kernel (conv3d_up_phase_k, conv3ds.hip) already uses that a 3x3x3 tap of output voxel
// 2V + p over nearest_up2(a) reads a[V + floor((p + d - 1) / 2)]: per axis and parity p the three taps fall on TWO
// low-resolution voxels (p = 0: V - 1 for d = 0, V for d = 1, 2;  p = 1: V for d = 0, 1, V + 1 for d = 2).  The adjoint of
// that statement for the weights:
//     G[p][i] = sum_V  a[V + (p - 1 + i)]  (x)  dY[2V + p]           p, i in {0, 1}^3   (64 matrices of Ca x Cout)
//     dW[d]   = sum of the 8 G[p][i] with d in D(p, i) per axis,      D(0,0) = {0}, D(0,1) = {1, 2}, D(1,0) = {0, 1}, D(1,1) = {2}
// -- 64 products per LOW-resolution voxel instead of 27 per full-resolution one: 8 / 27 of the multiplications of the
// direct form (conv3d_wgrad_tr_k<., ., UPCAT>, which multiplies every duplicated value again), the same sums in another
// order.  Zero padding carries over (a[-1] = a[D/2] = 0 are exactly the padded voxels of the up-sampled tensor).
//
// Kernels.  G does not depend on the position, so a workgroup keeps ALL of it in registers for its whole life and marches
// along z over 4 x 16 low-resolution columns: every plane of `a` (6 x 18 positions x 32 channels, ring of three) and every
// plane of dY (8 x 32 voxels x 32 channels, de-interleaved into its four (py, px) sub-lattices as it is written to LDS,
// two buffers) is staged ONCE, as scaled fp16 pairs (a = (a0 + a1) / s, products a0 b0 + a0 b1 + a1 b0 on
// v_mfma_f32_32x32x16_f16, fp32 accumulate: conv3ds.hip).  K = 16 low-resolution voxels of an x-row; both operands are
// K-major reads of channel-minor images through ds_read_b64_tr_b16 (conv3d_wgrad_tr_k has the lane algebra); an `a` row
// serves the two dY rows it meets (iy = 0 / 1).  One barrier per half-step (= one dY plane).  A workgroup walks several
// (column, z-segment) items and ends with ONE set of atomics into the 72-tile workspace; small kernels fold it into the
// tap-major gradient.
//   conv3d_upwgrad4_k<FUSEB>  THE PRODUCT PATH: 256 threads = 4 waves = the (py, px) classes, one wave per SIMD with 512
//                             registers per lane (16 tiles = the 256 AGPRs); FUSEB: the two skip channels of the top level
//                             and the bias gradient in the same launch (described at the kernel).  Without FUSEB the skip
//                             channels keep the direct kernel (conv3ds.hip::dfmir_conv3d_upwgrad).
//   conv3d_upwgrad_k          the first form, kept as an A/B (DFMIR_UPWGRAD_8WAVE): 512 threads = 8 waves = (py, px) x (iz),
//                             two waves per SIMD, 8 tiles per wave, never fused.  Same speed on the up-sampled share
//                             (profiles/r05_bench_upwgrad.txt): these kernels are bound by the sum of their matrix and staging
//                             work, not by how the waves share it (hardware fact 10).
#include "conv3x3_common.h"
#include <type_traits>

typedef _Float16 f16x8_u __attribute__((ext_vector_type(8)));
typedef short s16x4_u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_u* lds_tr_ptr_u;
#ifndef UW_KO
#define UW_KO 0      // knock-out builds (timing only): 1 no MFMAs, 2 no staging loads, 4 no conversion + LDS stores, 8 no operand reads,
                     // 16 no epilogue atomics (one store per lane instead), 32 dY loads from a cache-resident 4 KB (4-wave kernel)
#endif

namespace {

__device__ __forceinline__ int scale_exp_u(float amax) {
  const int be = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  int e = (amax > 0.f) ? 14 - be : 0;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return e;
}
__device__ __forceinline__ float pow2f_u(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
// (x0, x1) * s -> leading fp16 pair h and residual pair r
__device__ __forceinline__ void split_pair_u(float x0, float x1, float s, unsigned& h, unsigned& r) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(x1), "v"(s), "v"(h));
}
__device__ __forceinline__ f32x16 mma_u(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_u, a), __builtin_bit_cast(f16x8_u, b), c, 0, 0, 0);
}
// the same with the accumulator in VGPRs.  The 16 `a` tiles of conv3d_upwgrad4_k fill the 256 AGPRs; the compiler only
// emits the AGPR form of an MFMA and would swap whole tiles between the two files around every b product (152 v_accvgpr
// moves per half-step).  Hazards the compiler cannot see behind the asm: the tile is only ever read by the next product of
// its own chain (same vDst as SrcC: back-to-back issue is legal) until the epilogue, which waits first.
__device__ __forceinline__ void mma_v(u32x4 a, u32x4 b, f32x16& c) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// 16 lanes x 8 bytes: lane 4 j + q supplies the address of (voxel j, channel quad q); lane 4 q + c receives the c-th
// channel of quad q at voxels j = 0..3 (scripts/ubench/tr_read_probe.hip)
__device__ __forceinline__ uint2 tr_read_u(unsigned byte_addr) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr_u)(uintptr_t)byte_addr));
}
__device__ __forceinline__ u32x4 tr_pair_u(unsigned a0, unsigned a1) {
  const uint2 u0 = tr_read_u(a0), u1 = tr_read_u(a1);
  return u32x4{u0.x, u0.y, u1.x, u1.y};
}
template <int B, int E, class F>
__device__ __forceinline__ void static_for_u(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for_u<B + 1, E>(f);
  }
}

struct UwP {
  int N, Dl, Hl, Wl, Cout;       // a: [N, 32, Dl, Hl, Wl];  dY: [N, Cout, 2 Dl, 2 Hl, 2 Wl]
  int a_n, dy_n;                 // floats of the two range probes
  int ncy, ncx, nseg, zlen;      // 4 x 16 columns per low-resolution plane, z segments of zlen planes
  int nitems;                    // N * nseg * ncy * ncx
  const float* fx;               // deterministic mode (common.h df_acc): the workspace holds 64-bit fixed-point sums
};

constexpr unsigned UW_ASPLIT = 108u * 64u;          // one split of an `a` plane slot: 6 x 18 positions x 32 channels x fp16
constexpr unsigned UW_ASLOT = 2u * UW_ASPLIT;
constexpr unsigned UW_YOFF = 3u * UW_ASLOT;         // 41 472
constexpr unsigned UW_YSPLIT = 16384u;              // 4 classes x 64 voxels x 64 B
constexpr unsigned UW_YBUF = 2u * UW_YSPLIT;
constexpr unsigned UW_LDS = UW_YOFF + 2u * UW_YBUF; // 107 008 bytes

__global__ __launch_bounds__(512, 1) void conv3d_upwgrad_k(const float* __restrict__ a, const float* __restrict__ a_amax,
                                                           const float* __restrict__ dy, const float* __restrict__ dy_amax,
                                                           float* __restrict__ gws, UwP k) {
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) unsigned char lds[UW_LDS];
  __shared__ float red[17];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int px = wid & 1, py = (wid >> 1) & 1, szi = wid >> 2;

  const int ea = scale_exp_u(reduce_absmax(a_amax, k.a_n, red));
  __syncthreads();
  const int ed = scale_exp_u(reduce_absmax(dy_amax, k.dy_n, red));
  const float ascale = pow2f_u(ea), dscale = pow2f_u(ed), osc_a = pow2f_u(-ea), osc_d = pow2f_u(-ed);

  const int D = 2 * k.Dl, H = 2 * k.Hl, W = 2 * k.Wl;
  (void)D;
  const unsigned HW = (unsigned)(H * W), HWl = (unsigned)(k.Hl * k.Wl);
  const unsigned S4 = HW * (unsigned)D * 4u, Sl4 = HWl * (unsigned)k.Dl * 4u;

  f32x16 acc[2][4];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][s][r] = 0.f;

  // ---- operand addresses (bytes in LDS).  Source role of this lane in its 16-lane group: voxel sj, channel quad cq.
  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const int sj = (lane & 15) >> 2, cq = 4 * ((lane >> 4) & 1) + (lane & 3);
  unsigned bl[2], al[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int vx = 8 * hi + 4 * i + sj;
    bl[i] = lbase + UW_YOFF + (unsigned)((py * 2 + px) * 4096 + (vx * 4 + ((cq >> 1) ^ ((vx >> 2) & 3))) * 16 + (cq & 1) * 8);
#pragma unroll
    for (int ix = 0; ix < 2; ++ix) {
      const int hx = vx + px + ix;
      al[i][ix] = lbase + (unsigned)(py * 1152 + (hx * 4 + ((cq >> 1) ^ ((hx >> 2) & 3))) * 16 + (cq & 1) * 8);
    }
  }

  // ---- staging roles (every thread has both).  dY: thread = (channel group cg of 4, row r of 8, x pair xp of 16) of the
  // plane's 8 x 32 voxels: eight 8-byte loads; element e of the pair belongs to class px = e at Vx = xp, the row to
  // py = r & 1 at Vy = r >> 1.  a: thread mod 432 = (channel group, halo position of 6 x 18): eight 4-byte loads (threads
  // 432 .. 511 repeat the first 80 jobs: same data to the same address, no branch in the MFMA stream).
  const int yxp = tid & 15, yr = (tid >> 4) & 7, ycg = tid >> 7;
  const int aj = tid < 432 ? tid : tid - 432, acg = aj / 108, apos = aj - 108 * acg, ahy = apos / 18, ahx = apos - 18 * ahy;
  // store addresses (bytes): dY element e: + e * 4096 (+ buffer);  a: + slot
  const unsigned yst0 = lbase + UW_YOFF + (unsigned)((yr & 1) * 8192 + (yr >> 1) * 1024 + yxp * 64 + ((ycg ^ ((yxp >> 2) & 3)) << 4));
  const unsigned ast0 = lbase + (unsigned)(apos * 64 + ((acg ^ ((ahx >> 2) & 3)) << 4));

  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  u32x2v rq0[8], rq1[8];                                    // dY planes in flight: even / odd half-steps
  unsigned ra[8];                                           // the `a` plane in flight
  unsigned ybase = OOB, abase = OOB;                        // element offsets of this thread's jobs inside a plane
  __amdgpu_buffer_rsrc_t ysrc, asrc;
  int z0 = 0, z1 = 0;

#define UW_YLOAD(rq_, c_, off_)                                                                   \
  if (!(UW_KO & 2)) rq_[c_] = __builtin_amdgcn_raw_buffer_load_b64(ysrc, (off_) == OOB ? OOB : (off_) + (unsigned)(ycg * 8 + (c_)) * S4, 0, 0);
#define UW_ALOAD(ra_, c_, off_)                                                                   \
  if (!(UW_KO & 2)) ra_[c_] = __builtin_amdgcn_raw_buffer_load_b32(asrc, (off_) == OOB ? OOB : (off_) + (unsigned)(acg * 8 + (c_)) * Sl4, 0, 0);
  // convert 8 channels (expression v_(c)) and write the two 16-byte units at addr_ and addr_ + lo_
#define UW_LDS_ST(addr_, v_) *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(addr_) = (v_);
#define UW_CONV_ST(V_, scale_, addr_, lo_)                                                        \
  if (!(UW_KO & 4)) {                                                                             \
    u32x4 h_, r_;                                                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                            \
      unsigned hh_, rr_;                                                                          \
      split_pair_u(__uint_as_float(V_(2 * q_)), __uint_as_float(V_(2 * q_ + 1)), scale_, hh_, rr_); \
      h_[q_] = hh_; r_[q_] = rr_;                                                                 \
    }                                                                                             \
    UW_LDS_ST(addr_, h_) UW_LDS_ST((addr_) + (lo_), r_)                                           \
  }
#define UW_V_RQ0_E0(c_) rq0[c_][0]
#define UW_V_RQ0_E1(c_) rq0[c_][1]
#define UW_V_RQ1_E0(c_) rq1[c_][0]
#define UW_V_RQ1_E1(c_) rq1[c_][1]
#define UW_V_RA(c_) ra[c_]
  // global offsets (bytes) of this thread's jobs in plane P_ of dY (full resolution) / a (low resolution)
#define UW_YOFFS(P_, ok_) ((ybase != OOB && (ok_)) ? (ybase + (unsigned)(P_) * HW) * 4u : OOB)
#define UW_AOFFS(P_, ok_) ((abase != OOB && (ok_)) ? (abase + (unsigned)(P_) * HWl) * 4u : OOB)

  // One half-step h = (z, PZ_): dY plane 2 z + PZ_ (buffer PZ_) against the `a` planes z + PZ_ - 1 + iz (sa_: this wave's
  // slot).  Meanwhile: the loads of dY plane h + 2 go out (register set PZ_), the set that holds plane h + 1 is
  // converted into buffer PZ_ ^ 1; `a` plane z + 2 is loaded during PZ_ = 0 and converted into slot sn_ during PZ_ = 1 --
  // every load has more than a half-step to arrive and the memory pipe always has one plane in flight.
#define UW_HALF(PZ_, sa_, sn_, yoff_, aoff_)                                                      \
  {                                                                                               \
    const unsigned ab_ = (sa_);                                                                   \
    const unsigned bb0_ = bl[0] + (PZ_) * UW_YBUF, bb1_ = bl[1] + (PZ_) * UW_YBUF;                \
    const unsigned yst_ = yst0 + ((PZ_) ^ 1) * UW_YBUF;                                           \
    u32x4 B0[2], B1[2], A0, A1;                                                                   \
    static_for_u<0, 10>([&](auto jc_) __attribute__((always_inline)) {                            \
      constexpr int j_ = decltype(jc_)::value, R_ = j_ >> 1, ix_ = j_ & 1;                        \
      if constexpr (ix_ == 0 && R_ < 4) {                                                         \
        if (!(UW_KO & 8) || R_ == 0) {                                                            \
          B0[R_ & 1] = tr_pair_u(bb0_ + R_ * 1024, bb1_ + R_ * 1024);                             \
          B1[R_ & 1] = tr_pair_u(bb0_ + R_ * 1024 + UW_YSPLIT, bb1_ + R_ * 1024 + UW_YSPLIT);     \
        }                                                                                         \
      }                                                                                           \
      if (!(UW_KO & 8) || j_ == 0) {                                                              \
        A0 = tr_pair_u(al[0][ix_] + ab_ + R_ * 1152, al[1][ix_] + ab_ + R_ * 1152);               \
        A1 = tr_pair_u(al[0][ix_] + ab_ + R_ * 1152 + UW_ASPLIT, al[1][ix_] + ab_ + R_ * 1152 + UW_ASPLIT); \
      }                                                                                           \
      if constexpr (j_ < 4) {                                                                     \
        if constexpr ((PZ_) == 0) { UW_YLOAD(rq0, 2 * j_, yoff_) UW_YLOAD(rq0, 2 * j_ + 1, yoff_) } \
        else { UW_YLOAD(rq1, 2 * j_, yoff_) UW_YLOAD(rq1, 2 * j_ + 1, yoff_) }                    \
      }                                                                                           \
      if constexpr ((PZ_) == 0 && (j_ == 4 || j_ == 5)) {                                         \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) { UW_ALOAD(ra, 4 * (j_ - 4) + c_, aoff_) } \
      }                                                                                           \
      if constexpr (j_ == 6) {                                                                    \
        if constexpr ((PZ_) == 0) { UW_CONV_ST(UW_V_RQ1_E0, dscale, yst_, UW_YSPLIT) }            \
        else { UW_CONV_ST(UW_V_RQ0_E0, dscale, yst_, UW_YSPLIT) }                                 \
      }                                                                                           \
      if constexpr (j_ == 7) {                                                                    \
        if constexpr ((PZ_) == 0) { UW_CONV_ST(UW_V_RQ1_E1, dscale, yst_ + 4096u, UW_YSPLIT) }    \
        else { UW_CONV_ST(UW_V_RQ0_E1, dscale, yst_ + 4096u, UW_YSPLIT) }                         \
      }                                                                                           \
      if constexpr ((PZ_) == 1 && j_ == 8) { UW_CONV_ST(UW_V_RA, ascale, ast0 + (sn_), UW_ASPLIT) } \
      if (!(UW_KO & 1)) {                                                                         \
        if constexpr (R_ < 4) {       /* iy = 0: dY row R */                                      \
          acc[PZ_][ix_] = mma_u(A1, B0[R_ & 1], acc[PZ_][ix_]);                                   \
          acc[PZ_][ix_] = mma_u(A0, B1[R_ & 1], acc[PZ_][ix_]);                                   \
          acc[PZ_][ix_] = mma_u(A0, B0[R_ & 1], acc[PZ_][ix_]);                                   \
        }                                                                                         \
        if constexpr (R_ >= 1) {      /* iy = 1: dY row R - 1 */                                  \
          acc[PZ_][2 + ix_] = mma_u(A1, B0[(R_ - 1) & 1], acc[PZ_][2 + ix_]);                     \
          acc[PZ_][2 + ix_] = mma_u(A0, B1[(R_ - 1) & 1], acc[PZ_][2 + ix_]);                     \
          acc[PZ_][2 + ix_] = mma_u(A0, B0[(R_ - 1) & 1], acc[PZ_][2 + ix_]);                     \
        }                                                                                         \
      } else {                                                                                    \
        acc[PZ_][ix_][0] += __uint_as_float(A0[0] ^ A1[1] ^ B0[R_ & 1][2] ^ B1[R_ & 1][3]);       \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);   /* the compiler otherwise sinks the loads to the conversion behind the last MFMA */ \
    });                                                                                           \
  }

  for (int item = blockIdx.x; item < k.nitems; item += gridDim.x) {
    // ---- item = (n, segment, column): segment-major so that the workgroups of one wave of items share planes in L2
    int q = item;
    const int cx = q % k.ncx; q /= k.ncx;
    const int cy = q % k.ncy; q /= k.ncy;
    const int seg = q % k.nseg;
    const int n = q / k.nseg;
    z0 = seg * k.zlen;
    z1 = z0 + k.zlen < k.Dl ? z0 + k.zlen : k.Dl;
    const int y0 = cy * 4, x0 = cx * 16;
    ysrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + (long long)n * k.Cout * (S4 >> 2)), 0,
                                             (unsigned)k.Cout * S4, 0x00020000);
    asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a + (long long)n * 32 * (Sl4 >> 2)), 0, 32u * Sl4, 0x00020000);
    {
      const int gy = 2 * y0 + yr, gx = 2 * x0 + 2 * yxp;
      ybase = (gy < H && gx < W) ? (unsigned)(gy * W + gx) : OOB;
      const int ay = y0 - 1 + ahy, ax = x0 - 1 + ahx;
      abase = ((unsigned)ay < (unsigned)k.Hl && (unsigned)ax < (unsigned)k.Wl) ? (unsigned)(ay * k.Wl + ax) : OOB;
    }
    // ---- prologue: a planes z0 - 1, z0, z0 + 1 into slots 0, 1, 2; dY plane 2 z0 into buffer 0; plane 2 z0 + 1 stays in
    // flight in set 1 (converted during the first half-step)
    unsigned so0 = 0, so1 = UW_ASLOT, so2 = 2 * UW_ASLOT;
    {
      unsigned pa[3][8];
      const unsigned y0off = UW_YOFFS(2 * z0, true), y1off = UW_YOFFS(2 * z0 + 1, true);
#pragma unroll
      for (int c = 0; c < 8; ++c) { UW_YLOAD(rq0, c, y0off) }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const unsigned ao = UW_AOFFS(z0 - 1 + p, z0 - 1 + p >= 0 && z0 - 1 + p < k.Dl);
#pragma unroll
        for (int c = 0; c < 8; ++c) { UW_ALOAD(pa[p], c, ao) }
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) { UW_YLOAD(rq1, c, y1off) }
      UW_CONV_ST(UW_V_RQ0_E0, dscale, yst0, UW_YSPLIT)
      UW_CONV_ST(UW_V_RQ0_E1, dscale, yst0 + 4096u, UW_YSPLIT)
#define UW_V_PA0(c_) pa[0][c_]
#define UW_V_PA1(c_) pa[1][c_]
#define UW_V_PA2(c_) pa[2][c_]
      UW_CONV_ST(UW_V_PA0, ascale, ast0 + so0, UW_ASPLIT)
      UW_CONV_ST(UW_V_PA1, ascale, ast0 + so1, UW_ASPLIT)
      UW_CONV_ST(UW_V_PA2, ascale, ast0 + so2, UW_ASPLIT)
#undef UW_V_PA0
#undef UW_V_PA1
#undef UW_V_PA2
    }
    __syncthreads();

    for (int z = z0; z < z1; ++z) {
      {   // pz = 0: a planes z - 1 (iz = 0), z (iz = 1)
        const unsigned yoff = UW_YOFFS(2 * z + 2, z + 1 < z1), aoff = UW_AOFFS(z + 2, z + 2 < k.Dl);
        UW_HALF(0, szi ? so1 : so0, 0u, yoff, aoff)
        __syncthreads();
      }
      {   // pz = 1: a planes z (iz = 0), z + 1 (iz = 1); plane z + 2 replaces plane z - 1
        const unsigned yoff = UW_YOFFS(2 * z + 3, z + 1 < z1);
        UW_HALF(1, szi ? so2 : so1, so0, yoff, OOB)
        __syncthreads();
      }
      const unsigned t = so0; so0 = so1; so1 = so2; so2 = t;
    }
  }
#undef UW_YLOAD
#undef UW_ALOAD
#undef UW_LDS_ST
#undef UW_CONV_ST
#undef UW_YOFFS
#undef UW_AOFFS
#undef UW_HALF

  // ---- epilogue: acc[pz][iy * 2 + ix][r] <-> row ci = (r >> 2) * 8 + hi * 4 + (r & 3), column co = l31 of tile
  // (pz, py, px, iz, iy, ix)
  if (l31 < k.Cout && !(UW_KO & 16)) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int tile = ((((p * 2 + py) * 2 + px) * 2 + szi) * 2 + (s >> 1)) * 2 + (s & 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ci = (r >> 2) * 8 + hi * 4 + (r & 3);
          df_acc(gws, tile * 1024 + ci * 32 + l31, acc[p][s][r] * osc_a * osc_d, k.fx);
        }
      }
  }
  if (UW_KO & 16) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[p][s][r];
    gws[tid] = t * osc_a * osc_d;
  }
}

// ------------------------------------------------------------------------------------------------
// The same product with ONE wave per SIMD and the two skip channels of the top level FUSED.  256 threads = 4 waves =
// the (py, px) classes, 512 registers per lane: a wave holds its class's 2 (pz) x 2 (iz) x 4 (iy, ix) tiles (256
// accumulator registers) and, FUSEB, two more tiles for the direct weight gradient of b: rows = (tap, cb) -- the dY operand
// is already in registers, so the skip channels cost 6 MFMAs per 24 and no second pass over dY (the direct kernel spent 0.40
// ms of its 1.43 on them).  b [N, 2, D, H, W] is staged as (channel 0, channel 1) fp16 pairs, 4 bytes per voxel, one
// full-resolution plane (10 x 34 positions) per half-step in a ring of four, TWICE: x-major (X image) and y-major (Y image).
// An 8-byte transposing read then yields 4 operand rows = 2 neighbouring voxels x 2 channels; with the class's parity the
// neighbours along x are taps (px, px + 1) and the remaining x tap dxs = 2 - 2 px is taken as the y pair (py, py + 1) from
// the Y image plus a single (dys = 2 - 2 py) whose partner rows are discarded: 15 quads = 60 rows of 64 for the 54
// (tap, channel) rows, every read 8-byte aligned (the K index V moves the full-resolution position by two voxels, so the
// parity of an address is a property of the class).  conv3d_upwgrad_foldb_k knows the row tables.
// ------------------------------------------------------------------------------------------------
constexpr unsigned UW_BIMG = 1360u;                 // one image of a b plane: 10 x 34 positions x (2 channels x fp16)
constexpr unsigned UW_BSPLIT = 2u * UW_BIMG;        // [X image][Y image]
constexpr unsigned UW_BSLOT = 2u * UW_BSPLIT;       // [leading][residual]
constexpr unsigned UW_BOFF = UW_LDS;                // behind the a ring and the dY buffers
constexpr unsigned UW_LDS4 = UW_BOFF + 4u * UW_BSLOT;   // 128 768 bytes

template <bool FUSEB>
__global__ __launch_bounds__(256, 1) void conv3d_upwgrad4_k(const float* __restrict__ a, const float* __restrict__ a_amax,
                                                            const float* __restrict__ b, const float* __restrict__ dy,
                                                            const float* __restrict__ dy_amax, float* __restrict__ gws, UwP k) {
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) unsigned char lds[FUSEB ? UW_LDS4 + UW_BSPLIT + 16 : UW_LDS];
  __shared__ float red[17];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int px = wid & 1, py = wid >> 1;

  const int ea = scale_exp_u(reduce_absmax(a_amax, k.a_n, red));
  __syncthreads();
  const int ed = scale_exp_u(reduce_absmax(dy_amax, k.dy_n, red));
  const float ascale = pow2f_u(ea), dscale = pow2f_u(ed), osc_a = pow2f_u(-ea), osc_d = pow2f_u(-ed);

  const int D = 2 * k.Dl, H = 2 * k.Hl, W = 2 * k.Wl;
  const unsigned HW = (unsigned)(H * W), HWl = (unsigned)(k.Hl * k.Wl);
  const unsigned S4 = HW * (unsigned)D * 4u, Sl4 = HWl * (unsigned)k.Dl * 4u;

  f32x16 acc[2][2][4], accb[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int z = 0; z < 2; ++z)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[p][z][s][r] = 0.f;
    accb[0][r] = 0.f; accb[1][r] = 0.f;
  }

  // ---- operand addresses (bytes in LDS): conv3d_upwgrad_k
  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  const int sj = (lane & 15) >> 2, cq = 4 * ((lane >> 4) & 1) + (lane & 3);
  unsigned bl[2], al[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int vx = 8 * hi + 4 * i + sj;
    bl[i] = lbase + UW_YOFF + (unsigned)((py * 2 + px) * 4096 + (vx * 4 + ((cq >> 1) ^ ((vx >> 2) & 3))) * 16 + (cq & 1) * 8);
#pragma unroll
    for (int ix = 0; ix < 2; ++ix) {
      const int hx = vx + px + ix;
      al[i][ix] = lbase + (unsigned)(py * 1152 + (hx * 4 + ((cq >> 1) ^ ((hx >> 2) & 3))) * 16 + (cq & 1) * 8);
    }
  }
  // b operand: tile T = rows 32 T .., this lane supplies quad Q = 8 T + 4 (lane group) + (lane & 3) at voxel sj
  unsigned qb[2], qd[2], qstep[2], qa[2] = {0u, 0u};        // first read, distance to the second (4 voxels on), K-block step; qb + ring slot
  int qdz[2];
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    const int Q = 8 * T + 4 * ((lane >> 4) & 1) + (lane & 3);   // quad 15: the constant (1, 0, 0, 0) -> row 28 of tile 1 = sum of dY
    const int dxs = 2 - 2 * px, dys = 2 - 2 * py;
    {
      const int vx = 8 * hi + sj;
      int o;
      if (Q < 9) o = ((py + Q % 3) * 34 + 2 * vx + 2 * px) * 4;                               // x pair (px, px + 1) of (dz, dy)
      else if (Q < 12) o = (int)UW_BIMG + ((2 * vx + px + dxs) * 10 + 2 * py) * 4;           // y pair (py, py + 1) at dx = dxs
      else if (Q < 15) o = ((py + dys) * 34 + 2 * vx + (px ? 0 : 2)) * 4;                    // single (dys, dxs)
      else o = (int)(UW_LDS4 - UW_BOFF);
      qb[T] = lbase + UW_BOFF + (unsigned)o;
      qd[T] = Q == 15 ? 0u : ((Q >= 9 && Q < 12) ? 320u : 32u);   // 4 voxels on: 8 positions of the x-major / y-major image
    }
    qstep[T] = Q == 15 ? 0u : ((Q >= 9 && Q < 12) ? 8u : 272u);   // K-block Vy -> Vy + 1: two full-resolution rows
    qdz[T] = Q < 9 ? Q / 3 : (Q < 12 ? Q - 9 : (Q < 15 ? Q - 12 : 3));
  }
  if (FUSEB && tid < 4) {                                   // the ones quad (leading image) and its zero residual
    *(__attribute__((address_space(3))) unsigned*)(uintptr_t)(lbase + UW_LDS4 + (unsigned)(tid & 1) * 4u + (unsigned)(tid >> 1) * UW_BSPLIT) =
        tid == 0 ? 0x3c00u : 0u;
  }

  // ---- staging roles.  dY: thread = (channel group = wave, row r of 8, aligned quad q of 8): eight 16-byte loads;
  // element e of the quad belongs to class px = e & 1 at Vx = 2 q + (e >> 1), the row to py = r & 1 at Vy = r >> 1.
  // a: 240 jobs (channel group, halo row of 6, position pair of 10: x0 - 2 + 2 p ..), eight 8-byte loads each.  b: 180 jobs
  // (row of 10, position pair of 18: 2 x0 - 2 + 2 p ..) x both channels.  Threads without a job load from out-of-range offsets
  // (no memory access) and store nothing, like the pair elements outside the patch.
  const int yq = tid & 7, yr = (tid >> 3) & 7;
  const bool ajob = tid < 240, bjob = tid < 180;
  const int ja = ajob ? tid : 0, acg = ja / 60, arr = ja - 60 * acg, ahy = arr / 10, apr = arr - 10 * ahy;
  const int jb = bjob ? tid : 0, brw = jb / 18, bpr = jb - 18 * brw;
  const unsigned yst0 = lbase + UW_YOFF + (unsigned)((yr & 1) * 8192 + (yr >> 1) * 1024 + yq * 128 + ((wid ^ ((yq >> 1) & 3)) << 4));
  unsigned ast[2], bst[2], bsy[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int hx = 2 * apr - 1 + e, bc = 2 * bpr - 1 + e;
    ast[e] = (ajob && hx >= 0 && hx <= 17) ? lbase + (unsigned)((ahy * 18 + hx) * 64 + ((acg ^ ((hx >> 2) & 3)) << 4)) : OOB;
    bst[e] = (bjob && bc >= 0 && bc <= 33) ? lbase + UW_BOFF + (unsigned)((brw * 34 + bc) * 4) : OOB;
    bsy[e] = lbase + UW_BOFF + UW_BIMG + (unsigned)((bc * 10 + brw) * 4);
  }

  typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
  u32x4 rq[8];                                              // the dY plane in flight
  u32x2w ra[8];                                             // the `a` plane in flight
  u32x2w rb[2];

  unsigned ybase = OOB, abase = OOB, bbase = OOB;           // byte offsets of this thread's jobs inside plane 0 (a: + its channel group)
  __amdgpu_buffer_rsrc_t ysrc, asrc, bsrc;
  int z0 = 0, z1 = 0;

  // the channel stride rides in the scalar offset (no address arithmetic per load; an out-of-range voffset stays so)
#define UW_YLOAD(c_, off_)                                                                        \
  if (!(UW_KO & 2)) rq[c_] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, (UW_KO & 32) ? (unsigned)(tid * 16) : (off_), (unsigned)(wid * 8 + (c_)) * S4, 0);
#define UW_ALOAD(c_, off_)                                                                        \
  if (!(UW_KO & 2)) ra[c_] = __builtin_amdgcn_raw_buffer_load_b64(asrc, (off_), (unsigned)(c_) * Sl4, 0);
#define UW_BLOAD(off_)                                                                            \
  if (FUSEB && !(UW_KO & 2)) {                                                                    \
    rb[0] = __builtin_amdgcn_raw_buffer_load_b64(bsrc, (off_), 0, 0);                             \
    rb[1] = __builtin_amdgcn_raw_buffer_load_b64(bsrc, (off_), S4, 0);                            \
  }
#define UW_LDS_ST(addr_, v_) *(__attribute__((address_space(3))) u32x4*)(uintptr_t)(addr_) = (v_);
#define UW_LDS_ST4(addr_, v_) *(__attribute__((address_space(3))) unsigned*)(uintptr_t)(addr_) = (v_);
  // convert 8 channels (V_(c)) and write the two 16-byte units
#define UW_CONV_ST(V_, scale_, addr_, lo_)                                                        \
  if (!(UW_KO & 4)) {                                                                             \
    u32x4 h_, r_;                                                                                 \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                            \
      unsigned hh_, rr_;                                                                          \
      split_pair_u(__uint_as_float(V_(2 * q_)), __uint_as_float(V_(2 * q_ + 1)), scale_, hh_, rr_); \
      h_[q_] = hh_; r_[q_] = rr_;                                                                 \
    }                                                                                             \
    UW_LDS_ST(addr_, h_) UW_LDS_ST((addr_) + (lo_), r_)                                           \
  }
#define UW_V_E0(c_) rq[c_][0]
#define UW_V_E1(c_) rq[c_][1]
#define UW_V_E2(c_) rq[c_][2]
#define UW_V_E3(c_) rq[c_][3]
#define UW_V_A0(c_) ra[c_][0]
#define UW_V_A1(c_) ra[c_][1]
#define UW_YADDR(e_, buf_) (yst0 + (unsigned)(buf_) * UW_YBUF + (unsigned)(((e_) & 1) * 4096 + ((e_) >> 1) * 64))
#define UW_YCONV(e_, buf_)                                                                        \
  {                                                                                               \
    if constexpr ((e_) == 0) { UW_CONV_ST(UW_V_E0, dscale, UW_YADDR(0, buf_), UW_YSPLIT) }        \
    else if constexpr ((e_) == 1) { UW_CONV_ST(UW_V_E1, dscale, UW_YADDR(1, buf_), UW_YSPLIT) }   \
    else if constexpr ((e_) == 2) { UW_CONV_ST(UW_V_E2, dscale, UW_YADDR(2, buf_), UW_YSPLIT) }   \
    else { UW_CONV_ST(UW_V_E3, dscale, UW_YADDR(3, buf_), UW_YSPLIT) }                            \
  }
#define UW_ACONV(j_, slot_)                                                                       \
  if (ast[j_] != OOB) {                                                                           \
    if constexpr ((j_) == 0) { UW_CONV_ST(UW_V_A0, ascale, ast[0] + (slot_), UW_ASPLIT) }         \
    else { UW_CONV_ST(UW_V_A1, ascale, ast[1] + (slot_), UW_ASPLIT) }                             \
  }
#define UW_BCONV(slot_)                                                                           \
  if (FUSEB && !(UW_KO & 4)) {                                                                    \
    _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_) {                                            \
      unsigned h_, r_;                                                                            \
      split_pair_u(__uint_as_float(rb[0][e_]), __uint_as_float(rb[1][e_]), ascale, h_, r_);       \
      if (bst[e_] != OOB) {                                                                       \
        UW_LDS_ST4(bst[e_] + (slot_), h_) UW_LDS_ST4(bsy[e_] + (slot_), h_)                       \
        UW_LDS_ST4(bst[e_] + (slot_) + UW_BSPLIT, r_) UW_LDS_ST4(bsy[e_] + (slot_) + UW_BSPLIT, r_) \
      }                                                                                           \
    }                                                                                             \
  }
#define UW_YOFFS(P_, ok_) ((ybase != OOB && (ok_)) ? ybase + (unsigned)(P_) * HW * 4u : OOB)
#define UW_AOFFS(P_, ok_) ((abase != OOB && (ok_)) ? abase + (unsigned)(P_) * HWl * 4u : OOB)
#define UW_BOFFS(P_, ok_) ((bbase != OOB && (ok_)) ? bbase + (unsigned)(P_) * HW * 4u : OOB)

  // One half-step h = (z, PZ_), dY plane P = 2 z + PZ_ (buffer PZ_): 24 groups -- per `a` row R = 0..4 the dY operand
  // of K-block R with the b tiles, then (ix, iz) x the products with rows R (iy = 0) and R - 1 (iy = 1).  Loads and
  // conversions ride in fixed groups: dY plane P + 1 (loaded during the previous half-step) is converted into buffer
  // PZ_ ^ 1 in groups 1-4 and the loads of plane P + 2 follow at once (groups 5-12); `a` plane z + 2 is loaded in groups
  // 13-20 of PZ_ = 0 (8-byte pairs) and converted in groups 13-14 of PZ_ = 1; b plane P + 2: loads in group 2, conversion
  // in group 17 (into the ring's free slot).
#define UW4_HALF(PZ_, sa0_, sa1_, sn_, yoff_, aoff_, boff_, sbf_)               \
  {                                                                                               \
    const unsigned bb0_ = bl[0] + (PZ_) * UW_YBUF, bb1_ = bl[1] + (PZ_) * UW_YBUF;                \
    u32x4 B0[2], B1[2], A0[2], A1[2];                                                             \
    const unsigned qa_[2] = {qa[0], qa[1]};   /* this lane's quads in the ring slots of their dz (set by the previous half-step) */ \
    unsigned yo_ = OOB, ao_ = OOB;                                                                \
    /* operands of group g (see below) into A buffer g & 1: issued one group ahead of their MFMAs */ \
    auto rd_ = [&](auto gc2_) __attribute__((always_inline)) {                                    \
      constexpr int g2_ = decltype(gc2_)::value, R2_ = g2_ < 20 ? g2_ / 5 : 4, k2_ = g2_ < 20 ? g2_ % 5 : g2_ - 19; \
      if constexpr (g2_ < 24) {                                                                   \
        if constexpr (k2_ == 0) {                                                                 \
          B0[R2_ & 1] = tr_pair_u(bb0_ + R2_ * 1024, bb1_ + R2_ * 1024);                          \
          B1[R2_ & 1] = tr_pair_u(bb0_ + R2_ * 1024 + UW_YSPLIT, bb1_ + R2_ * 1024 + UW_YSPLIT);  \
        } else {                                                                                  \
          constexpr int ix2_ = (k2_ - 1) >> 1, iz2_ = (k2_ - 1) & 1;                              \
          const unsigned ab_ = iz2_ ? (sa1_) : (sa0_);                                            \
          A0[g2_ & 1] = tr_pair_u(al[0][ix2_] + ab_ + R2_ * 1152, al[1][ix2_] + ab_ + R2_ * 1152); \
          A1[g2_ & 1] = tr_pair_u(al[0][ix2_] + ab_ + R2_ * 1152 + UW_ASPLIT, al[1][ix2_] + ab_ + R2_ * 1152 + UW_ASPLIT); \
        }                                                                                         \
      }                                                                                           \
    };                                                                                            \
    rd_(std::integral_constant<int, 0>{});                                                        \
    static_for_u<0, 24>([&](auto gc_) __attribute__((always_inline)) {                            \
      constexpr int g_ = decltype(gc_)::value, R_ = g_ < 20 ? g_ / 5 : 4, kk_ = g_ < 20 ? g_ % 5 : g_ - 19; \
      constexpr int ix_ = kk_ ? (kk_ - 1) >> 1 : 0, iz_ = kk_ ? (kk_ - 1) & 1 : 0;                \
      constexpr bool nextb_ = g_ + 1 < 20 && (g_ + 1) % 5 == 0;   /* the next group opens a row: its dY operand replaces */ \
      if constexpr (!nextb_) {                                    /* one that this group's iy = 1 products still read    */ \
        if (!(UW_KO & 8) || g_ == 0) rd_(std::integral_constant<int, g_ + 1>{});                  \
      }                                                                                           \
      /* (the offsets are EXPRESSIONS: evaluated under the MFMAs of the group that first needs them, not in front of the \
         half-step's first product) */                                                            \
      if constexpr (g_ >= 1 && g_ < 5) { UW_YCONV(g_ - 1, (PZ_) ^ 1) }                            \
      if constexpr (g_ == 5) yo_ = (yoff_);                                                       \
      if constexpr (g_ >= 5 && g_ < 13) { UW_YLOAD(g_ - 5, yo_) }                                 \
      if constexpr ((PZ_) == 0 && g_ == 13) ao_ = (aoff_);                                        \
      if constexpr ((PZ_) == 0 && g_ >= 13 && g_ < 21) { UW_ALOAD(g_ - 13, ao_) }                 \
      if constexpr ((PZ_) == 1 && (g_ == 13 || g_ == 14)) { UW_ACONV(g_ - 13, sn_) }              \
      if constexpr (g_ == 2) { UW_BLOAD(boff_) }                                                  \
      if constexpr (g_ == 17) { UW_BCONV(sbf_) }                                                  \
      if constexpr (FUSEB && g_ == 21) {   /* next half-step's quad addresses: its slots are this one's 1, 2, 3 */ \
        _Pragma("unroll") for (int T_ = 0; T_ < 2; ++T_)                                          \
          qa[T_] = qb[T_] + (qdz[T_] == 0 ? sb1 : (qdz[T_] == 1 ? sb2 : (qdz[T_] == 2 ? (sbf_) : 0u))); \
      }                                                                                           \
      if (!(UW_KO & 1)) {                                                                         \
        if constexpr (kk_ == 0) {                                                                 \
          if constexpr (FUSEB) {      /* b tiles: their operands are read here (the A buffer of this group is free) */ \
            _Pragma("unroll") for (int T_ = 0; T_ < 2; ++T_) {                                    \
              const unsigned q0_ = qa_[T_] + R_ * qstep[T_], q1_ = q0_ + qd[T_];                  \
              A0[g_ & 1] = tr_pair_u(q0_, q1_);                                                   \
              A1[g_ & 1] = tr_pair_u(q0_ + UW_BSPLIT, q1_ + UW_BSPLIT);                           \
              mma_v(A1[g_ & 1], B0[R_ & 1], accb[T_]);                                            \
              mma_v(A0[g_ & 1], B1[R_ & 1], accb[T_]);                                            \
              mma_v(A0[g_ & 1], B0[R_ & 1], accb[T_]);                                            \
            }                                                                                     \
          }                                                                                       \
        } else {                                                                                  \
          if constexpr (R_ < 4) {       /* iy = 0: dY row R */                                    \
            acc[PZ_][iz_][ix_] = mma_u(A1[g_ & 1], B0[R_ & 1], acc[PZ_][iz_][ix_]);               \
            acc[PZ_][iz_][ix_] = mma_u(A0[g_ & 1], B1[R_ & 1], acc[PZ_][iz_][ix_]);               \
            acc[PZ_][iz_][ix_] = mma_u(A0[g_ & 1], B0[R_ & 1], acc[PZ_][iz_][ix_]);               \
          }                                                                                       \
          if constexpr (R_ >= 1) {      /* iy = 1: dY row R - 1 */                                \
            acc[PZ_][iz_][2 + ix_] = mma_u(A1[g_ & 1], B0[(R_ - 1) & 1], acc[PZ_][iz_][2 + ix_]); \
            acc[PZ_][iz_][2 + ix_] = mma_u(A0[g_ & 1], B1[(R_ - 1) & 1], acc[PZ_][iz_][2 + ix_]); \
            acc[PZ_][iz_][2 + ix_] = mma_u(A0[g_ & 1], B0[(R_ - 1) & 1], acc[PZ_][iz_][2 + ix_]); \
          }                                                                                       \
        }                                                                                         \
      } else if constexpr (kk_ != 0) {                                                            \
        acc[PZ_][iz_][ix_][0] += __uint_as_float(A0[g_ & 1][0] ^ A1[g_ & 1][1] ^ B0[R_ & 1][2] ^ B1[R_ & 1][3]); \
      }                                                                                           \
      if constexpr (nextb_) {                                                                     \
        if (!(UW_KO & 8)) rd_(std::integral_constant<int, g_ + 1>{});                             \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    });                                                                                           \
  }

  // Items: workgroup ids go round-robin to the 8 XCDs (one L2 each), so XCD e = id & 7 owns the e-th eighth of the item list
  // (x fastest, then y: neighbouring columns, whose `a` and b halos and partly used 128-byte lines overlap) and its
  // workgroups walk it together.
  const int xcd = blockIdx.x & 7, nslot = ((int)gridDim.x + 7 - xcd) >> 3;
  const int per_xcd = (k.nitems + 7) >> 3, it_end = (xcd + 1) * per_xcd < k.nitems ? (xcd + 1) * per_xcd : k.nitems;
  for (int item = xcd * per_xcd + (int)(blockIdx.x >> 3); item < it_end; item += nslot) {
    int q = item;
    const int cx = q % k.ncx; q /= k.ncx;
    const int cy = q % k.ncy; q /= k.ncy;
    const int seg = q % k.nseg;
    const int n = q / k.nseg;
    z0 = seg * k.zlen;
    z1 = z0 + k.zlen < k.Dl ? z0 + k.zlen : k.Dl;
    const int y0 = cy * 4, x0 = cx * 16;
    ysrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + (long long)n * k.Cout * (S4 >> 2)), 0,
                                             (unsigned)k.Cout * S4, 0x00020000);
    asrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a + (long long)n * 32 * (Sl4 >> 2)), 0, 32u * Sl4, 0x00020000);
    if (FUSEB) bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b + (long long)n * 2 * (S4 >> 2)), 0, 2u * S4, 0x00020000);
    {
      const int gy = 2 * y0 + yr, gx = 2 * x0 + 4 * yq;
      ybase = (gy < H && gx < W && wid * 8 < k.Cout) ? (unsigned)(gy * W + gx) * 4u : OOB;   // (the scalar offset is not range-checked)
      const int ay = y0 - 1 + ahy, ax = x0 - 2 + 2 * apr;
      abase = (ajob && (unsigned)ay < (unsigned)k.Hl && (unsigned)ax < (unsigned)k.Wl) ? (unsigned)(ay * k.Wl + ax) * 4u + (unsigned)(acg * 8) * Sl4 : OOB;
      const int by = 2 * y0 - 1 + brw, bx = 2 * x0 - 2 + 2 * bpr;
      bbase = (bjob && (unsigned)by < (unsigned)H && (unsigned)bx < (unsigned)W) ? (unsigned)(by * W + bx) * 4u : OOB;
    }
    // ---- prologue: a planes z0 - 1, z0, z0 + 1 into slots 0, 1, 2; b planes 2 z0 - 1, 2 z0, 2 z0 + 1 into slots 0, 1, 2;
    // dY plane 2 z0 into buffer 0; plane 2 z0 + 1 stays in flight (converted at the head of the first half-step)
    unsigned so0 = 0, so1 = UW_ASLOT, so2 = 2 * UW_ASLOT;
    unsigned sb0 = 0, sb1 = UW_BSLOT, sb2 = 2 * UW_BSLOT, sb3 = 3 * UW_BSLOT;
    {
      // every load of the prologue in flight before the first conversion: one round trip per item, not seven
      const unsigned y0off = UW_YOFFS(2 * z0, true), y1off = UW_YOFFS(2 * z0 + 1, true);
#pragma unroll
      for (int c = 0; c < 8; ++c) { UW_YLOAD(c, y0off) }
      u32x2w pa[3][8], pb[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const int P = z0 - 1 + p;
        const unsigned a0 = UW_AOFFS(P, P >= 0 && P < k.Dl);
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (!(UW_KO & 2)) pa[p][c] = __builtin_amdgcn_raw_buffer_load_b64(asrc, a0, (unsigned)c * Sl4, 0);
        if constexpr (FUSEB) {
          const int Pb = 2 * z0 - 1 + p;
          const unsigned b0 = UW_BOFFS(Pb, Pb >= 0);
          if (!(UW_KO & 2)) {
            pb[p][0] = __builtin_amdgcn_raw_buffer_load_b64(bsrc, b0, 0, 0);
            pb[p][1] = __builtin_amdgcn_raw_buffer_load_b64(bsrc, b0, S4, 0);
          }
        }
      }
#define UW_V_PA(c_) pa[p][c_][e]
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (ast[e] != OOB) { UW_CONV_ST(UW_V_PA, ascale, ast[e] + (unsigned)p * UW_ASLOT, UW_ASPLIT) }
          if (FUSEB && !(UW_KO & 4) && bst[e] != OOB) {
            unsigned h_, r_;
            split_pair_u(__uint_as_float(pb[p][0][e]), __uint_as_float(pb[p][1][e]), ascale, h_, r_);
            const unsigned so_ = (unsigned)p * UW_BSLOT;
            UW_LDS_ST4(bst[e] + so_, h_) UW_LDS_ST4(bsy[e] + so_, h_)
            UW_LDS_ST4(bst[e] + so_ + UW_BSPLIT, r_) UW_LDS_ST4(bsy[e] + so_ + UW_BSPLIT, r_)
          }
        }
#undef UW_V_PA
      UW_YCONV(0, 0) UW_YCONV(1, 0) UW_YCONV(2, 0) UW_YCONV(3, 0)
#pragma unroll
      for (int c = 0; c < 8; ++c) { UW_YLOAD(c, y1off) }
      if constexpr (FUSEB) {
#pragma unroll
        for (int T = 0; T < 2; ++T) qa[T] = qb[T] + (qdz[T] == 0 ? sb0 : (qdz[T] == 1 ? sb1 : (qdz[T] == 2 ? sb2 : 0u)));
      }
    }
    __syncthreads();

    for (int z = z0; z < z1; ++z) {
      {   // pz = 0: P = 2 z; a planes z - 1 (iz = 0), z (iz = 1); b planes P - 1, P, P + 1 in sb0..2, P + 2 -> sb3
        UW4_HALF(0, so0, so1, 0u, UW_YOFFS(2 * z + 2, z + 1 < z1), UW_AOFFS(z + 2, z + 2 < k.Dl), UW_BOFFS(2 * z + 2, 2 * z + 2 < D), sb3)
        __syncthreads();
        const unsigned t = sb0; sb0 = sb1; sb1 = sb2; sb2 = sb3; sb3 = t;
      }
      {   // pz = 1: P = 2 z + 1; a planes z (iz = 0), z + 1 (iz = 1); plane z + 2 replaces plane z - 1
        UW4_HALF(1, so1, so2, so0, UW_YOFFS(2 * z + 3, z + 1 < z1), OOB, UW_BOFFS(2 * z + 3, 2 * z + 3 < D), sb3)
        __syncthreads();
        const unsigned t = sb0; sb0 = sb1; sb1 = sb2; sb2 = sb3; sb3 = t;
      }
      const unsigned t = so0; so0 = so1; so1 = so2; so2 = t;
    }
  }
#undef UW_YLOAD
#undef UW_ALOAD
#undef UW_V_E0
#undef UW_V_E1
#undef UW_V_E2
#undef UW_V_E3
#undef UW_V_A0
#undef UW_V_A1
#undef UW_YADDR
#undef UW_BLOAD
#undef UW_LDS_ST
#undef UW_LDS_ST4
#undef UW_CONV_ST
#undef UW_YCONV
#undef UW_ACONV
#undef UW_BCONV
#undef UW_YOFFS
#undef UW_AOFFS
#undef UW_BOFFS
#undef UW4_HALF

  // ---- epilogue: acc[pz][iz][iy * 2 + ix][r] <-> row ci = (r >> 2) * 8 + hi * 4 + (r & 3), column co = l31 of tile
  // (pz, py, px, iz, iy, ix); the b tiles follow the 64: (class, T) with row = (r >> 2) * 8 + hi * 4 + (r & 3) of tile T
  if constexpr (FUSEB) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last b product's result (mma_v)
  if (l31 < k.Cout && !(UW_KO & 16)) {
    const float sc = osc_a * osc_d;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int iz = 0; iz < 2; ++iz)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int tile = ((((p * 2 + py) * 2 + px) * 2 + iz) * 2 + (s >> 1)) * 2 + (s & 1);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ci = (r >> 2) * 8 + hi * 4 + (r & 3);
            df_acc(gws, tile * 1024 + ci * 32 + l31, acc[p][iz][s][r] * sc, k.fx);
          }
        }
    if constexpr (FUSEB) {
#pragma unroll
      for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r >> 2) * 8 + hi * 4 + (r & 3);          // row 28 of tile 1: sum of dY (ones operand, unscaled)
          df_acc(gws, (64 + (py * 2 + px) * 2 + T) * 1024 + row * 32 + l31, accb[T][r] * ((T == 1 && row == 28) ? osc_d : sc), k.fx);
        }
    }
  }
  if (UW_KO & 16) {
    float t = accb[0][0] + accb[1][0];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int iz = 0; iz < 2; ++iz)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int r = 0; r < 16; ++r) t += acc[p][iz][s][r];
    gws[tid] = t * osc_a * osc_d;
  }
}

// dWb[tap][cb][co] += the rows of the 8 b tiles (class (py, px), T) that hold (tap, cb): see conv3d_upwgrad4_k
__global__ __launch_bounds__(256) void conv3d_upwgrad_foldb_k(const float* __restrict__ gws, float* __restrict__ dwt, int Cout,
                                                               long long s_tap, int total, float* __restrict__ db,
                                                               const float* __restrict__ fx) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total + Cout) return;
  const long long* g64 = reinterpret_cast<const long long*>(gws);   // deterministic mode: integer sums in, integer sum out
  if (idx >= total) {                                       // bias gradient: row 28 of the classes' second tiles
    const int co = idx - total;
    float s = 0.f;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) s += gws[(64 + cls * 2 + 1) * 1024 + 28 * 32 + co];
    if (db && !fx) db[co] += s;                             // (deterministic mode: db comes from dfmir_bias_grad)
    return;
  }
  const int co = idx % Cout, cb = (idx / Cout) & 1, tap = idx / (Cout * 2);
  const int dz = tap / 9, dyy = (tap / 3) % 3, dx = tap % 3;
  float s = 0.f;
  long long s64 = 0;
#pragma unroll
  for (int cls = 0; cls < 4; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    int Q, pos;
    if (dx == px || dx == px + 1) { Q = dz * 3 + dyy; pos = (dx - px) * 2 + cb; }
    else if (dyy == py || dyy == py + 1) { Q = 9 + dz; pos = (dyy - py) * 2 + cb; }
    else { Q = 12 + dz; pos = (px ? 2 : 0) + cb; }
    if (fx) s64 += g64[(64 + cls * 2 + (Q >> 3)) * 1024 + ((Q & 7) * 4 + pos) * 32 + co];
    else s += gws[(64 + cls * 2 + (Q >> 3)) * 1024 + ((Q & 7) * 4 + pos) * 32 + co];
  }
  if (fx) reinterpret_cast<long long*>(dwt)[tap * s_tap + (long long)cb * Cout + co] += s64;
  else dwt[tap * s_tap + (long long)cb * Cout + co] += s;
}

// dW[tap][ci][co] += the 8 tiles of the workspace that contain tap = (dz, dy, dx): per axis d = 0: (p, i) = (0,0), (1,0);
// d = 1: (0,1), (1,0);  d = 2: (0,1), (1,1)
__global__ __launch_bounds__(256) void conv3d_upwgrad_fold_k(const float* __restrict__ gws, float* __restrict__ dwt, int Cout,
                                                              long long s_tap, int total, const float* __restrict__ fx) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long long* g64 = reinterpret_cast<const long long*>(gws);
  long long s64 = 0;
  const int co = idx % Cout, ci = (idx / Cout) & 31, tap = idx / (Cout * 32);
  const int d[3] = {tap / 9, (tap / 3) % 3, tap % 3};
  int pi[3][2];                                             // the two (p * 2 + i) codes of each axis
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    pi[ax][0] = d[ax] == 0 ? 0 : 1;                         // (0,0) | (0,1)
    pi[ax][1] = d[ax] == 2 ? 3 : 2;                         // (1,1) | (1,0)
  }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int cz = pi[0][m >> 2], cy = pi[1][(m >> 1) & 1], cx = pi[2][m & 1];
    const int tile = (((((cz >> 1) * 2 + (cy >> 1)) * 2 + (cx >> 1)) * 2 + (cz & 1)) * 2 + (cy & 1)) * 2 + (cx & 1);
    if (fx) s64 += g64[tile * 1024 + ci * 32 + co];
    else s += gws[tile * 1024 + ci * 32 + co];
  }
  if (fx) reinterpret_cast<long long*>(dwt)[tap * s_tap + (long long)ci * Cout + co] += s64;
  else dwt[tap * s_tap + (long long)ci * Cout + co] += s;
}

}  // namespace

// Host side.  b != NULL (two skip channels): everything in one launch, rows 0 .. 33 of the tap-major gradient
// [27][Ctot][Cout] (s_tap = Ctot * Cout) and db; b == NULL: rows 0 .. 31 (conv3ds.hip::dfmir_conv3d_upwgrad runs the
// direct kernel on the skip channels).  ws: 72 * 1024 floats.
int df_conv3d_upwgrad_launch(const float* a, const float* a_amax, int a_n, const float* b, const float* dy, const float* dy_amax,
                             int dy_n, float* dwt, long long s_tap, float* db, float* ws, int N, int Dl, int Hl, int Wl,
                             int Cout, hipStream_t st) {
  UwP k{};
  k.N = N; k.Dl = Dl; k.Hl = Hl; k.Wl = Wl; k.Cout = Cout;
  k.a_n = a_n; k.dy_n = dy_n;
  k.ncy = (Hl + 3) / 4; k.ncx = (Wl + 15) / 16;
  const int ncu = df_cu_count();
  // z segments: a workgroup (one per CU) walks ceil(items / CUs) items of zlen planes, each with a prologue worth ~2 planes
  const long long cols = (long long)N * k.ncy * k.ncx;
  static DfOptInt nseg_o{"DFMIR_UPWGRAD_NSEG", 0};
  int best = 1;
  long long best_cost = -1;
  for (int s = 1; s <= Dl && s <= 64; ++s) {
    const int zl = (Dl + s - 1) / s;
    const int ns = (Dl + zl - 1) / zl;
    const long long rounds = (cols * ns + ncu - 1) / ncu;
    const long long cost = rounds * (zl + 2);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ns; }
  }
  const long long forced = nseg_o.get();
  if (forced > 0 && forced <= Dl) best = (int)forced;
  k.zlen = (Dl + best - 1) / best;
  k.nseg = (Dl + k.zlen - 1) / k.zlen;
  k.nitems = (int)(cols * k.nseg);
  k.fx = df_det_fx();
  const unsigned grid = (unsigned)(k.nitems < ncu ? k.nitems : ncu);
  if (df_zero_async(ws, (k.fx ? 2 : 1) * 72 * 1024, st) != hipSuccess) return 2;   // a kernel, not a memset node (common.h: graph ordering on ROCm 7.2)
  static DfOptFlag v1_o{"DFMIR_UPWGRAD_8WAVE"};             // A/B: the two-waves-per-SIMD form of the up-sampled share
  if (b) conv3d_upwgrad4_k<true><<<grid, 256, 0, st>>>(a, a_amax, b, dy, dy_amax, ws, k);
  else if (v1_o.get()) conv3d_upwgrad_k<<<grid, 512, 0, st>>>(a, a_amax, dy, dy_amax, ws, k);
  else conv3d_upwgrad4_k<false><<<grid, 256, 0, st>>>(a, a_amax, nullptr, dy, dy_amax, ws, k);
  const int total = 27 * 32 * Cout;
  conv3d_upwgrad_fold_k<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(ws, dwt, Cout, s_tap, total, k.fx);
  // (deterministic mode: dwt counts 8-byte slots, so the skip rows start 32 * Cout SLOTS = 64 * Cout floats further on)
  if (b) conv3d_upwgrad_foldb_k<<<(unsigned)((27 * 2 * Cout + Cout + 255) / 256), 256, 0, st>>>(
      ws, dwt + (k.fx ? 64LL : 32LL) * Cout, Cout, s_tap, 27 * 2 * Cout, db, k.fx);
  DF_LAUNCH_CHECK();
  return 0;
}

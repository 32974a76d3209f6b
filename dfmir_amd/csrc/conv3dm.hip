// 3-D 3x3x3 stride-1 "same" convolution (forward and dgrad) of the FULL-RESOLUTION VoxelMorph layers -- 32 -> 16, 16 -> 16
// and the input gradients 16 <- 16, 32 <- 16 (torchvoxelmorph/networks.py:73-86,1506-1521: the `extras` chain behind the
// decoder) -- as a z-MARCHING kernel: the scaled fp16x2 split form of conv3ds.hip (a = (a0 + a1) / s, a b ~= a0b0 + a0b1 +
// a1b0, fp32 accumulate on the 16-bit matrix pipe), re-tiled for layers whose Cin x Cout <= 512.
//
// These layers are as much an HBM problem as a matrix problem (32 -> 16 at 160 x 192 x 224: 1.3 GB moved for 190 GFLOP), and
// the 4 x 8 x 16 tiles of conv3d_split_k / conv3d_split_m16_k stage a 6 x 10 x 18 halo patch (2.1 x the tile) per 8-channel
// chunk together with a fresh copy of that chunk's weights, behind two barriers.  Here instead:
//   * a workgroup (512 threads, one per CU) owns a 16 x 32 COLUMN of the volume and marches along z over a segment of
//     planes.  Every input plane (18 x 34 positions x ALL input channels, split into fp16 pairs as it is written to LDS) is
//     staged ONCE and contributes to the three output planes z - 1, z, z + 1 through the nine taps of dz = 2, 1, 0: the
//     halo is in-plane only (1.2 x) -- 43 % less staging work (global loads, conversions, LDS stores) per output voxel;
//   * the accumulators of the three open output planes ROLL through the registers (the plane loop is unrolled by three so
//     that the rotation is a renaming); when input plane P has been consumed output plane P - 1 is complete and leaves
//     through the epilogue (bias, LeakyReLU or the LeakyReLU derivative of a dgrad, range probe) -- 16-byte stores that are
//     spread evenly over the kernel's lifetime instead of a burst at the end of every tile;
//   * ALL weights of the layer (<= 61 KB as fp16 pairs) are split by the workgroup itself in its prologue and stay in
//     LDS: no weight staging in the loop, no separate weight-split launch, no workspace;
//   * Cin = 16: two plane slots, plane P + 1 is converted and stored while plane P is consumed (one barrier per plane);
//     Cin = 32: one slot (84 KB) -- the next plane waits in registers and is written between two barriers (~8 % of a plane
//     step: a step is 648 MFMAs per SIMD).
// MFMA forms.  Cout = 16: v_mfma_f32_16x16x32_f16, rows = 16 voxels of a tile row, columns = the output channels,
// K = 32 = one tap x 32 channels (Cin = 32) or two taps x 16 channels (Cin = 16: the nine taps of a dz are walked as
// (dy, dx 0 | dx 1) x 3, (dy 0 | dy 1, dx 2), (dy 2, dx 2 | zero) -- 10 / 9 of the useful products).  Cout = 32 (Cin = 16):
// v_mfma_f32_32x32x16_f16, rows = the 32 voxels of a tile row, K = 16 = one tap.  A wave owns two tile rows x three
// planes; the voxel operands of a tap column are read once per patch row and serve the (dy, dz) products that use them
// (0.26-0.48 LDS operand reads per MFMA).  MFMA rows = voxels, so a lane ends with 4 consecutive voxels of one channel.
//
// The flow head (16 -> 3, torchvoxelmorph/networks.py:1076-1080) in the same march -- "FLOW" form, Cout = 3.  Three
// output channels fill 3 of 16 MFMA columns, and the three open output planes of the march are three separate
// accumulator sets above: here the COLUMNS are (open plane, channel) = 3 x 4 (one zero column per plane), so ONE
// accumulator set serves the three planes and a plane step issues 60 MFMAs per wave instead of 180.  Column group g
// holds the output plane whose index is = g modulo 3 for as long as it is open; the weight operand of a step is the
// tap matrix with its dz blocks rotated to match (three pre-rotated copies in LDS, one per phase of the march), and when
// input plane P has been consumed the lanes of group (P - 1) mod 3 store their plane and clear their columns.  The
// fp32-FMA kernel this replaces (csrc/conv3dt.hip) ran at the vector rate: 0.32 ms at 160 x 192 x 224 for 0.52 GB.
#include "conv3x3_common.h"
#include <type_traits>

typedef _Float16 f16x8_m __attribute__((ext_vector_type(8)));
typedef float f32x4_m __attribute__((ext_vector_type(4)));
#ifndef M3_KO
#define M3_KO 0      // knock-out builds (timing only, scripts/build_ko_march.sh): 1 no MFMAs, 2 no staging loads, 4 no conversion +
#endif               // LDS stores, 8 no epilogue stores, 16 no epilogue at all, 32 no operand reads after the first plane
#define M3_SINK(v_) asm volatile("" ::"v"(v_))
#ifndef M3_ADEPTH
#define M3_ADEPTH 1  // weight operands are read this many (k-step, dz) blocks ahead of their MFMAs
#endif
#ifndef M3_B32_RC
#define M3_B32_RC 1  // 32-column form: MFMA rows = output channels, columns = the 32 voxels of a tile row -> 4-byte epilogue
#endif               // accesses, each a full 128-byte row segment of one channel (0: rows = voxels, 16-byte pieces of 32 channels)
#ifndef M3_PRIO
#define M3_PRIO 0    // 1: the two waves of a SIMD (w, w + 4) alternate s_setprio 1 / 0 block by block; 2: waves 4-7 at 1
#endif
#ifdef M3_TRACE      // timing build: per-wave cycle sums of the phases of one workgroup's plane steps -> dfmir_m3_trace()
__device__ unsigned long long m3_trace[8 * 8];
#define M3_T0() unsigned long long tlast = __builtin_readcyclecounter(); const bool trace_blk = blockIdx.x == 77;
#define M3_T(i_) { if (trace_blk && lane == 0) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); m3_trace[wid * 8 + (i_)] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } }
extern "C" void dfmir_m3_trace(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(m3_trace), sizeof(m3_trace));
  if (reset) { unsigned long long z[64] = {}; hipMemcpyToSymbol(HIP_SYMBOL(m3_trace), z, sizeof(z)); }
}
#else
#define M3_T0()
#define M3_T(i_)
#endif

namespace {

__device__ __forceinline__ int scale_exp_m(float amax) {
  const int be = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  int e = (amax > 0.f) ? 14 - be : 0;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return e;
}
__device__ __forceinline__ float pow2f_m(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
// (x0, x1) * s -> leading fp16 pair h and residual pair r
__device__ __forceinline__ void split_pair_m(float x0, float x1, float s, unsigned& h, unsigned& r) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(x1), "v"(s), "v"(h));
}
__device__ __forceinline__ void split8_m(const float (&v)[8], float s, u32x4& h, u32x4& r) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned hh, rr;
    split_pair_m(v[2 * q], v[2 * q + 1], s, hh, rr);
    h[q] = hh; r[q] = rr;
  }
}
__device__ __forceinline__ f32x4_m mma16m(u32x4 a, u32x4 b, f32x4_m c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_m, a), __builtin_bit_cast(f16x8_m, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma32m(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_m, a), __builtin_bit_cast(f16x8_m, b), c, 0, 0, 0);
}

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E - 1 (the plane step's schedule is a table over its
// MFMA groups; `#pragma unroll` left some of these loops peeled instead of unrolled and the register arrays in scratch)
template <int B, int E, class F>
__device__ __forceinline__ void static_for_m(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for_m<B + 1, E>(f);
  }
}

struct MarchP {
  int N, D, H, W;
  int nty, ntx, nseg, zlen;      // 16 x 32 columns per plane, z segments of zlen planes
  int x_n;                       // floats of the input range probe
  int act;                       // 0 none, 1 LeakyReLU(slope)
  float slope;
  const float* act_src;          // dgrad into the output of a LeakyReLU: multiply by its derivative (conv3ds.hip, av_mode 1)
  float act_slope;
  long long nwork;               // N * nseg * nty * ntx
};

template <int CIN, int COUT>
struct MarchCfg {
  static_assert((CIN == 32 && COUT == 16) || (CIN == 16 && (COUT == 16 || COUT == 32 || COUT == 3)), "layer shapes of the march kernel");
  static constexpr bool B32 = COUT == 32;                       // 32x32x16 MFMAs (one tap per k-step)
  static constexpr bool FLOW = COUT == 3;                       // columns = (open plane, channel): one accumulator set
  static constexpr int NDZ = FLOW ? 1 : 3;                      // weight blocks walked per k-step and plane step
  static constexpr int NO = CIN / 8;                            // channel octets
  static constexpr int NSTEP = (CIN == 32 || B32) ? 9 : 5;      // k-steps per dz
  static constexpr int NH = B32 ? 1 : 2;                        // voxel operands per tile row (32 / 16 voxels each)
  static constexpr int WU = NSTEP * 3 * 2 * 64;                 // weight units: (step, dz, split, lane)
  static constexpr int SLOTS = CIN == 32 ? 1 : 2;
  static constexpr int SR = 19, RS = 34;                        // slot rows (18 + one finite pad row), row stride in units
  static constexpr int OP = ((SR * RS + 15) / 16) * 16;         // octet plane stride: = 0 mod 16 -> conflict-free b128 reads
  static constexpr int SU = 2 * NO * OP;                        // units of a slot: [split][octet][row][col]
  static constexpr int NQ = NO * 18 * 8, NSG = NO * 18 * 2;     // staging jobs: 16-byte quads, halo columns
  static constexpr int NJ = NQ + NSG, NR = (NJ + 511) / 512;    // rounds of jobs per thread
};

// (kind, first patch row) of k-step j: the voxel operands of a kind are read once per patch row q = 0..3 of the wave's two
// tile rows and serve every k-step of that kind
template <int CIN, int COUT> __device__ constexpr int step_kind(int j) { return MarchCfg<CIN, COUT>::NSTEP == 9 ? j / 3 : (j < 3 ? 0 : 1); }
template <int CIN, int COUT> __device__ constexpr int step_qoff(int j) { return MarchCfg<CIN, COUT>::NSTEP == 9 ? j % 3 : (j < 3 ? j : (j == 3 ? 0 : 2)); }
// operand rows (kind * 4 + q, bit mask) whose reads are issued at the head of step j: each row's registers are re-used by
// the next kind as soon as the last k-step that needs the old contents has been issued
template <int CIN, int COUT> __device__ constexpr unsigned step_loads(int j) {
  if (MarchCfg<CIN, COUT>::NSTEP == 9) {
    const int k = j / 3, d = j % 3;
    unsigned m = 0;
    if (d == 0) m |= 1u << (k * 4 + 2);
    if (d == 1) { m |= 1u << (k * 4 + 3); if (k < 2) m |= 1u << ((k + 1) * 4 + 0); }
    if (d == 2 && k < 2) m |= 1u << ((k + 1) * 4 + 1);
    return m;
  }
  return j == 0 ? (1u << 2) : j == 1 ? ((1u << 3) | (1u << 4)) : j == 2 ? (1u << 5) : j == 3 ? ((1u << 6) | (1u << 7)) : 0u;
}

template <int CIN, int COUT, bool ACTG>
__global__ __launch_bounds__(512, 1) void conv3d_march_k(const float* __restrict__ x, const float* __restrict__ x_amax,
                                                         const float* __restrict__ w_tcc, const float* __restrict__ bias,
                                                         float* __restrict__ y, float* __restrict__ y_amax, MarchP k) {
  using C = MarchCfg<CIN, COUT>;
  constexpr bool B32 = C::B32, FLOW = C::FLOW;
  static_assert(!(FLOW && ACTG), "the flow head is never a data gradient");
  constexpr int NDZ = C::NDZ;
  constexpr int NO = C::NO, NSTEP = C::NSTEP, NH = C::NH, WU = C::WU, SLOTS = C::SLOTS, RS = C::RS, OP = C::OP, SU = C::SU;
  constexpr int NQ = C::NQ, NJ = C::NJ, NR = C::NR;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) u32x4 Xs[SLOTS * SU];
  __shared__ __attribute__((aligned(16))) u32x4 Ws[WU];
  __shared__ float red[17];
  __shared__ unsigned smax;
  __shared__ float sbias[32];
  constexpr bool RC = B32 && (M3_B32_RC != 0);             // rows = output channels (see M3_B32_RC)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4, l31 = lane & 31, lhi = lane >> 5;
  if (tid < 32) sbias[tid] = (bias && tid < COUT) ? bias[tid] : 0.f;

  // workgroup -> (image, z segment, column): the dispatcher deals consecutive ids round-robin to the 8 XCDs (one L2
  // each); ids with the same residue walk one contiguous eighth of the work list (x fastest, then y, then segment), so
  // the columns that share halo rows / columns march through the same L2 side by side
  const long long per_xcd = (k.nwork + 7) / 8;
  const long long lin = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((long long)(blockIdx.x >> 3) >= per_xcd || lin >= k.nwork) return;
  long long t_ = lin;
  const int tx = (int)(t_ % k.ntx); t_ /= k.ntx;
  const int ty = (int)(t_ % k.nty); t_ /= k.nty;
  const int seg = (int)(t_ % k.nseg);
  const int n = (int)(t_ / k.nseg);
  const int y0 = ty * 16, x0 = tx * 32;
  const int zs = seg * k.zlen, ze = (zs + k.zlen < k.D) ? zs + k.zlen : k.D;
  const int nst = ze - zs + 2;                               // input planes zs - 1 .. ze
  const int HW = k.H * k.W;
  const long long S = (long long)k.D * HW;
  const unsigned s4 = (unsigned)S * 4u, hw4 = (unsigned)HW * 4u;

  // ---- prologue: zero the plane slots (pad row / pad units must be finite: they meet zero weights), scales, weights
  for (int u = tid; u < SLOTS * SU; u += 512) Xs[u] = u32x4{0u, 0u, 0u, 0u};
  const float amax = reduce_absmax(x_amax, k.x_n, red);
  const int ex = scale_exp_m(amax);
  float wm = 0.f;
  for (int i = tid; i < 27 * CIN * COUT / 4; i += 512) {
    const float4 v = reinterpret_cast<const float4*>(w_tcc)[i];
    wm = fmaxf(wm, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  wm = block_max(wm, red);
  if (!(wm == wm)) wm = __uint_as_float(0x7f800000u);
  const int ew = scale_exp_m(wm);
  const float xscale = pow2f_m(ex), wscale = pow2f_m(ew), osc = pow2f_m(-ex) * pow2f_m(-ew);
  if (tid == 0) smax = 0u;
  // weight units [(step j, dz)][split][lane]: the lane's 8 reduction values of its k group, for its output channel
  for (int u = tid; u < NSTEP * 3 * 64; u += 512) {
    const int ln = u & 63, jd = u >> 6, j = jd / 3, dz = jd % 3;
    int co, tap, c0;                                         // tap < 0: zero unit
    if constexpr (FLOW) {
      // unit (k-step j, phase ph = jd % 3): column group g = the open plane = g (mod 3), which input plane P0 + i with
      // i = ph (mod 3) reaches through tap plane (ph + 1 - g) mod 3
      const int col = ln & 15, g = col >> 2, kgp = ln >> 4, t2 = kgp >> 1;
      co = col & 3;
      c0 = 8 * (kgp & 1);
      int dy, dx;
      if (j < 3) { dy = j; dx = t2; }
      else if (j == 3) { dy = t2; dx = 2; }
      else { dy = 2; dx = 2; }
      const int dzr = (dz + 1 - g + 3) % 3;
      tap = (g == 3 || co == 3 || (j == 4 && t2 == 1)) ? -1 : dzr * 9 + dy * 3 + dx;
      if (tap < 0) co = 0;
    } else if constexpr (B32) {
      co = ln & 31; c0 = 8 * (ln >> 5);
      tap = dz * 9 + (j % 3) * 3 + j / 3;
    } else if constexpr (CIN == 32) {
      co = ln & 15; c0 = 8 * (ln >> 4);
      tap = dz * 9 + (j % 3) * 3 + j / 3;
    } else {
      co = ln & 15;
      const int g = ln >> 4, t2 = g >> 1;
      c0 = 8 * (g & 1);
      int dy, dx;
      if (j < 3) { dy = j; dx = t2; }
      else if (j == 3) { dy = t2; dx = 2; }
      else { dy = 2; dx = 2; }
      tap = (j == 4 && t2 == 1) ? -1 : dz * 9 + dy * 3 + dx;
    }
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = tap >= 0 ? w_tcc[((long long)tap * CIN + c0 + c) * COUT + co] : 0.f;
    u32x4 h, r;
    split8_m(v, wscale, h, r);
    Ws[(jd * 2 + 0) * 64 + ln] = h;
    Ws[(jd * 2 + 1) * 64 + ln] = r;
  }

  // ---- staging jobs of this thread (the same for every plane): job < NQ = the 16-byte quad xq of patch row q, octet o
  // (8 buffer_load_dwordx4, one per channel -> 4 positions x 2 split units), else one halo column position (8 dword loads)
  // FLOW: a halo column is loaded as the aligned quad that contains it (component jhe = 3 left, 0 right; the rest of the
  // quad is a neighbour column's interior: L2 hits) -- ONE kind of load, issued by every lane with an out-of-range offset
  // where there is no job, so that the loads are straight-line code: with the loads under lane- and plane-dependent
  // branches the compiler's s_waitcnt placement assumed the worst path and drained the other register set's loads (the
  // prefetch of two planes ahead) before every conversion
  unsigned jvo[NR];            // byte offset of (channel 8 o, plane 0, row, column) or OOB
  int jpos[NR];                // LDS unit of the (first) position within a split section; < 0: no job
  int jhe[NR];                 // FLOW: the one quad component a halo job stores, -1 = all four
  bool jquad[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int job = tid + 512 * r;
    jvo[r] = OOB; jpos[r] = -1; jquad[r] = FLOW || job < NQ; jhe[r] = -1;
    if (job < NJ) {
      int o, q, col, gx;
      if (job < NQ) {
        // 8 consecutive lanes (one service group of ds_write_b128) = four quads of TWO consecutive rows: their units sit
        // at 4 xq + 34 (q & 1) + e -> 8 distinct 16-byte bank groups (quads 0..7 of one row: 2-way conflicts)
        const int xq = (job & 3) | (((job >> 3) & 1) << 2), q2 = (job >> 4) % 9;
        q = 2 * q2 + ((job >> 2) & 1); o = (job >> 4) / 9; col = 1 + 4 * xq; gx = x0 + 4 * xq;
      }
      else {
        const int s_ = job - NQ, side = s_ & 1; q = (s_ >> 1) % 18; o = (s_ >> 1) / 18; col = side ? 33 : 0; gx = side ? x0 + 32 : x0 - 1;
        if constexpr (FLOW) { jhe[r] = side ? 0 : 3; gx = side ? x0 + 32 : x0 - 4; }
      }
      const int gy = y0 - 1 + q;
      jpos[r] = o * OP + q * RS + col;
      if ((unsigned)gy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W)
        jvo[r] = (unsigned)(gy * k.W + gx) * 4u + (unsigned)(8 * o) * s4;
    }
  }
  const __amdgpu_buffer_rsrc_t x_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x + (long long)n * CIN * S), 0, (unsigned)((long long)CIN * S * 4), 0x00020000);
  // FLOW: TWO register sets (plane index & 1) -- a plane step is 60 MFMAs per wave, a third of the other forms', and with one
  // set (loads issued half a step before their conversion) every step waited out the memory latency: 3.9 us per plane
  constexpr int NRQ = FLOW ? 2 : NR;
  u32x4 rq[NRQ][8];
  bool ko_st = true;                                         // (knock-out builds switch the staging stores off after the prologue)
#define M3_GLOAD(z_) M3_GLOAD_S(z_, 0)
#define M3_GLOAD_S(z_, set_)               /* set_: register set of the FLOW form (its one job round), else 0 */ \
  {                                                                                               \
    const int zz_ = (z_);                                                                         \
    const bool zok_ = (unsigned)zz_ < (unsigned)k.D;                                              \
    const unsigned zb_ = zok_ ? (unsigned)zz_ * hw4 : 0u;                                         \
    _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                              \
      const unsigned vo_ = (zok_ && !((M3_KO & 2) && k.D > 0)) ? jvo[r] : OOB;                    \
      if (jquad[r]) {                                                                             \
        _Pragma("unroll") for (int c = 0; c < 8; ++c)                                             \
          rq[r + (set_)][c] = __builtin_amdgcn_raw_buffer_load_b128(x_src, vo_, zb_ + (unsigned)c * s4, 0); \
      } else if (jpos[r] >= 0) {                                                                  \
        _Pragma("unroll") for (int c = 0; c < 8; ++c)                                             \
          rq[r + (set_)][c][0] = __builtin_amdgcn_raw_buffer_load_b32(x_src, vo_, zb_ + (unsigned)c * s4, 0); \
      }                                                                                           \
    }                                                                                             \
  }
#define M3_LSTORE(sl_) M3_LSTORE_S(sl_, 0)
#define M3_LSTORE_S(sl_, set_)                                                                    \
  if (ko_st) {                                                                                    \
    u32x4* Xd_ = Xs + (sl_) * SU;                                                                 \
    _Pragma("unroll") for (int r = 0; r < NR; ++r) {                                              \
      if (jquad[r]) {                                                                             \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                           \
          float v_[8];                                                                            \
          _Pragma("unroll") for (int c = 0; c < 8; ++c) v_[c] = __uint_as_float(rq[r + (set_)][c][e]); \
          u32x4 h_, r_;                                                                           \
          split8_m(v_, xscale, h_, r_);                                                           \
          if (jpos[r] >= 0 && (jhe[r] < 0 || jhe[r] == e)) {                                      \
            Xd_[jpos[r] + (jhe[r] < 0 ? e : 0)] = h_;                                             \
            Xd_[NO * OP + jpos[r] + (jhe[r] < 0 ? e : 0)] = r_;                                   \
          }                                                                                       \
        }                                                                                         \
      } else if (jpos[r] >= 0) {                                                                  \
        float v_[8];                                                                              \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) v_[c] = __uint_as_float(rq[r + (set_)][c][0]); \
        u32x4 h_, r_;                                                                             \
        split8_m(v_, xscale, h_, r_);                                                             \
        Xd_[jpos[r]] = h_;                                                                        \
        Xd_[NO * OP + jpos[r]] = r_;                                                              \
      }                                                                                           \
    }                                                                                             \
  }

  // ---- MFMA operand addresses.  Voxel operand of (kind, patch row q, half h, split s):
  //   Xs[slot][s * NO * OP + vb[kind] + q * RS + 16 h]
  int vb[3];
  if constexpr (B32) {
#pragma unroll
    for (int d = 0; d < 3; ++d) vb[d] = lhi * OP + (2 * wid) * RS + l31 + d;
  } else if constexpr (CIN == 32) {
#pragma unroll
    for (int d = 0; d < 3; ++d) vb[d] = kg * OP + (2 * wid) * RS + l15 + d;
  } else {
    vb[0] = (kg & 1) * OP + (2 * wid) * RS + l15 + (kg >> 1);            // (dy, dx 0 | dx 1)
    vb[1] = (kg & 1) * OP + (2 * wid + (kg >> 1)) * RS + l15 + 2;        // (dy | dy + 1, dx 2)
    vb[2] = 0;
  }

  // accumulators of the three open output planes: [plane slot][tile row][half]
  using acc_t = typename std::conditional<B32, f32x16, f32x4_m>::type;
  // NSETS = 4 (16 -> 16): the finished plane keeps its set for one more step, during which its epilogue rides between
  // the MFMA groups like the staging atoms (10-13 % of a step was an epilogue with nothing on the matrix pipe); the
  // 32-column form has no registers for a fourth set (96 + 32) and finishes its plane at the end of the step
  constexpr int NSETS = (B32 || CIN == 32 || FLOW) ? 3 : 4;      // (32 input channels: two job rounds of staging registers)
  constexpr int NACC = FLOW ? 1 : NSETS;                  // FLOW: the three open planes are column groups of one set
  acc_t acc[NACC][2][NH];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < (B32 ? 16 : 4); ++i) acc[a][r][h][i] = 0.f;

  // epilogue constants: lane = one output channel, 4 consecutive voxels per accumulator quad
  const int co = FLOW ? (l15 & 3) : (B32 ? l31 : l15);
  const bool col_ok = !FLOW || (l15 < 12 && co < 3);         // FLOW: column 4 g + 3 and group 3 are padding
  const float bv = (bias && co < COUT) ? bias[co] : 0.f;
  const __amdgpu_buffer_rsrc_t y_dst = __builtin_amdgcn_make_buffer_rsrc(
      y + (long long)n * COUT * S, 0, (unsigned)((long long)COUT * S * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>((k.act_src ? k.act_src : y) + (long long)n * COUT * S), 0, (unsigned)((long long)COUT * S * 4), 0x00020000);
  constexpr int NE = B32 ? 8 : 4;                            // 16-byte stores per lane and plane: [row][half or quad]
  unsigned evo[NE];                                          // byte offset at plane 0 (or OOB)
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int r = B32 ? (e >> 2) : (e >> 1);
    const int gy = y0 + 2 * wid + r;
    const int gx = RC ? (x0 + l31) : (B32 ? (x0 + 8 * (e & 3) + 4 * lhi) : (x0 + 16 * (e & 1) + 4 * kg));
    // RC: element i of group e = (row, q) is channel 8 q + 4 lhi + i at voxel x0 + l31 (channel step in the soffset)
    evo[e] = (gy < k.H && gx < k.W && col_ok) ? (unsigned)(gy * k.W + gx) * 4u + (unsigned)(RC ? 8 * (e & 3) + 4 * lhi : co) * s4 : OOB;
  }
  float pm = 0.f;

  // ---- one plane step: consume input plane P (slot sl_) with roles rotated by PH = (P - (zs - 1)) % 3, then finish
  // output plane P - 1
#define M3_BREAD(kind_, q_)                                                                       \
  if (M3_KO_RD) {                                                                                 \
    _Pragma("unroll") for (int h = 0; h < NH; ++h)                                                \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                               \
        Bu[q_][h][s] = Xc[s * NO * OP + vb[kind_] + (q_) * RS + 16 * h];                          \
  }
#define M3_WU(jd_, PH_) (FLOW ? (jd_) * 3 + (PH_) : (jd_))     /* weight unit of block jd_: FLOW keeps one per (k-step, phase) */
#define M3_AREAD(buf_, jd_)                                                                       \
  if (M3_KO_RD) { Aw[buf_][0] = Ws[((jd_) * 2 + 0) * 64 + lane]; Aw[buf_][1] = Ws[((jd_) * 2 + 1) * 64 + lane]; }
#if M3_KO & 32
  u32x4 Bu[4][NH][2], Aw[M3_ADEPTH + 1][2];
  bool ko_rd = true;
#define M3_KO_DECL
#define M3_KO_RD ko_rd
#else
#define M3_KO_DECL u32x4 Bu[4][NH][2], Aw[M3_ADEPTH + 1][2];
#define M3_KO_RD true
#endif
  // Staging work of a plane step is cut into ATOMS that ride between the MFMA groups of the step (group t = 3 x block + product
  // term; a block = the 12 / 6 MFMAs of one (k-step, dz)), so that neither the vector-memory queue (a burst of 16 x 1 KB loads
  // per wave blocked the ISSUE of everything behind it for 17-19 % of a step) nor the conversion sits in front of the MFMAs:
  //   two slots:  t = 0..7    convert + store plane P + 1 (quad position e = t / 2; half of the channel pairs per atom)
  //               t = 9, 11, .. 23   the 8 channel loads of plane P + 2;   t = 10, 12, ..   the activation-source quads
  //   one slot:   t = 0, 2, .. 30   the 16 channel loads of plane P + 1 (two job rounds), converted and stored between the
  //               two barriers (converting in place during the MFMA phase costs a second register set: a position's unit
  //               needs one component of eight 4-register load results);   t = 45, 47, ..   the activation-source quads
  u32x4 cvh, cvr;                                                   // two slots: the unit being converted
#define M3_CONV_HALF(r_, e_, half_, H_, R_)     /* (every lane converts: no predicate, no merge with old register contents); r_ = register set */ \
  {                                                                                               \
    _Pragma("unroll") for (int q = 2 * (half_); q < 2 * (half_) + 2; ++q) {                       \
      unsigned hh_, rr_;                                                                          \
      split_pair_m(__uint_as_float(rq[r_][2 * q][e_]), __uint_as_float(rq[r_][2 * q + 1][e_]), xscale, hh_, rr_); \
      H_[q] = hh_; R_[q] = rr_;                                                                   \
    }                                                                                             \
  }
#define M3_LOAD1(r_, c_, z_) M3_LOAD1S(r_, r_, c_, z_)
#define M3_LOAD1S(s_, r_, c_, z_)          /* job round r_ into register set s_ */                 \
  {                                                                                               \
    const int zz_ = (z_);                                                                         \
    const bool zok_ = (unsigned)zz_ < (unsigned)k.D && !((M3_KO & 2) && k.D > 0);                 \
    const unsigned zb_ = zok_ ? (unsigned)zz_ * hw4 : 0u;                                         \
    const unsigned vo_ = zok_ ? jvo[r_] : OOB;                                                    \
    if (jquad[r_]) rq[s_][c_] = __builtin_amdgcn_raw_buffer_load_b128(x_src, vo_, zb_ + (unsigned)(c_) * s4, 0); \
    else if (jpos[r_] >= 0) rq[s_][c_][0] = __builtin_amdgcn_raw_buffer_load_b32(x_src, vo_, zb_ + (unsigned)(c_) * s4, 0); \
  }
#define M3_AVLOAD(e_, P_)                  /* activation source of the plane whose epilogue comes next */ \
  {                                                                                               \
    const int p_ = (P_) - (NSETS == 4 ? 2 : 1);                                                   \
    const bool pok_ = p_ >= zs && p_ < ze;                                                        \
    if constexpr (RC) {                                                                           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
        av[e_][i] = __builtin_amdgcn_raw_buffer_load_b32(a_src, pok_ ? evo[e_] + (unsigned)p_ * hw4 : OOB, (unsigned)i * s4, 0); \
    } else {                                                                                      \
      av[e_] = __builtin_amdgcn_raw_buffer_load_b128(a_src, pok_ ? evo[e_] + (unsigned)p_ * hw4 : OOB, 0, 0); \
    }                                                                                             \
  }
  // store e of the finished plane p_ (accumulator set a_): bias, LeakyReLU or the folded LeakyReLU derivative, range probe;
  // the set leaves zeroed
#define M3_EPI_ONE(a_, e_, p_)                                                                    \
  {                                                                                               \
    constexpr int eg_ = (a_), ea_ = FLOW ? 0 : eg_, ee_ = (e_);                                   \
    constexpr int r = B32 ? (ee_ >> 2) : (ee_ >> 1), h = B32 ? 0 : (ee_ & 1), q4 = B32 ? 4 * (ee_ & 3) : 0; \
    const int pp_ = (p_);                                                                         \
    const bool mine_ = !FLOW || (l15 >> 2) == eg_;      /* FLOW: the finished plane is column group eg_ */ \
    const bool ok = pp_ >= zs && pp_ < ze && evo[ee_] != OOB && mine_;                            \
    u32x4 out;                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
      float v = acc[ea_][r][h][q4 + i] * osc + (RC ? sbias[(8 * (ee_ & 3) + 4 * lhi + i) & 31] : bv);  \
      if (k.act == 1) v = v > 0.f ? v : v * k.slope;                                              \
      if (ACTG) v = __uint_as_float(av[ACTG ? ee_ : 0][i]) > 0.f ? v : v * k.act_slope;            \
      out[i] = __float_as_uint(v);                                                                \
      pm = fmaxf(pm, ok ? fabsf(v) : 0.f);                                                        \
      acc[ea_][r][h][q4 + i] = mine_ ? 0.f : acc[ea_][r][h][q4 + i];                              \
    }                                                                                             \
    /* plane offset in the VGPR, soffset literal 0: with an SGPR soffset hipcc assumes that a 16-byte store's */ \
    /* data registers may be overwritten by the next VALU instruction -- on gfx950 they may not (dword 0 of   */ \
    /* the last lanes of every 16 was lost once in ~10^4 launches)                                              */ \
    if constexpr (RC) {                                                                           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
        __builtin_amdgcn_raw_buffer_store_b32(out[i], y_dst, (ok && !((M3_KO & 8) && k.D > 0)) ? evo[ee_] + (unsigned)pp_ * hw4 : OOB, (unsigned)i * s4, 0); \
    } else {                                                                                      \
      __builtin_amdgcn_raw_buffer_store_b128(out, y_dst, (ok && !((M3_KO & 8) && k.D > 0)) ? evo[ee_] + (unsigned)pp_ * hw4 : OOB, 0, 0); \
    }                                                                                             \
  }
#define M3_ATOMS(t, PH_, P_, sl_, st_, ld_, SET_)                                                 \
  {                                                                                               \
    if constexpr (SLOTS == 2) {                                                                   \
      if constexpr (t < 8) {                                                                      \
        if ((st_) && ko_st) {                                                                     \
          constexpr int e = t >> 1;                                                               \
          M3_CONV_HALF(SET_, e, t & 1, cvh, cvr)                                                  \
          if constexpr ((t & 1) == 1) {                                                           \
            if (FLOW ? (jpos[0] >= 0 && (jhe[0] < 0 || jhe[0] == e)) : (jpos[0] >= 0 && (jquad[0] || e == 0))) { \
              u32x4* Xd_ = Xs + ((sl_) ^ 1) * SU;                                                 \
              const int pe_ = (FLOW && jhe[0] >= 0) ? 0 : e;                                      \
              Xd_[jpos[0] + pe_] = cvh;                                                           \
              Xd_[NO * OP + jpos[0] + pe_] = cvr;                                                 \
            }                                                                                     \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
      /* FLOW: plane P + 3 into the set whose plane P + 1 was converted in atoms 0..7 (its channel c was last read by atom 6 / 7) */ \
      if constexpr (FLOW) { if constexpr (t >= 7 && t <= 14) { M3_LOAD1S(SET_, 0, t - 7, (ld_) ? (P_) + 3 : -1) } } \
      else if constexpr (t >= 9 && t <= 23 && (t & 1) == 1) { if (ld_) M3_LOAD1(0, (t - 9) >> 1, (P_) + 2) } \
      if constexpr (ACTG && t >= 10 && t < 10 + 2 * NE && (t & 1) == 0) M3_AVLOAD((t - 10) >> 1, P_) \
      if constexpr (NSETS == 4 && t >= 30 && t < 30 + 3 * NE && (t - 30) % 3 == 0) M3_EPI_ONE(((PH_) + 2) % 4, (t - 30) / 3, (P_) - 2) \
    } else {                                                                                      \
      if constexpr (t <= 30 && (t & 1) == 0) { if (st_) M3_LOAD1((t >> 1) >> 3, (t >> 1) & 7, (P_) + 1) } \
      if constexpr (ACTG && t >= 1 && t < 1 + 2 * NE && (t & 1) == 1) M3_AVLOAD((t - 1) >> 1, P_) \
      if constexpr (NSETS == 4 && t >= 33 && t < 33 + 2 * NE && (t & 1) == 1) M3_EPI_ONE(((PH_) + 2) % 4, (t - 33) >> 1, (P_) - 2) \
    }                                                                                             \
  }
#define M3_STEP(PH_, sl_, P_, st_, ld_, SET_)                                                     \
  {                                                                                               \
    const u32x4* Xc = Xs + (sl_) * SU;                                                            \
    M3_KO_DECL                                                                                    \
    u32x4 av[ACTG ? NE : 1];                                                                      \
    M3_BREAD(0, 0) M3_BREAD(0, 1)                                                                 \
    static_for_m<0, M3_ADEPTH>([&](auto dc_) __attribute__((always_inline)) { M3_AREAD(decltype(dc_)::value, M3_WU(decltype(dc_)::value, PH_)) }); \
    static_for_m<0, NSTEP * NDZ * 3>([&](auto tc_) __attribute__((always_inline)) {               \
      constexpr int t = decltype(tc_)::value, jd = t / 3, p = t % 3, j = jd / NDZ, dz = jd % NDZ, cur = jd % (M3_ADEPTH + 1); \
      constexpr unsigned lmask = step_loads<CIN, COUT>(j);                                        \
      constexpr int qo = step_qoff<CIN, COUT>(j);                                                 \
      constexpr int a = FLOW ? 0 : ((PH_) + 1 - dz + NSETS) % NSETS;                              \
      if constexpr (p == 0) {                                                                     \
        if constexpr (M3_PRIO == 1) { if (wid >= 4) __builtin_amdgcn_s_setprio((jd + 1) & 1); else __builtin_amdgcn_s_setprio(jd & 1); } \
        if constexpr (jd + M3_ADEPTH < NSTEP * NDZ) M3_AREAD((jd + M3_ADEPTH) % (M3_ADEPTH + 1), M3_WU(jd + M3_ADEPTH, PH_)) \
        if constexpr (dz == 0) {                                                                  \
          static_for_m<0, 12>([&](auto bc_) __attribute__((always_inline)) {                      \
            constexpr int b = decltype(bc_)::value;                                               \
            if constexpr ((lmask >> b) & 1u) M3_BREAD(b >> 2, b & 3)                              \
          });                                                                                     \
        }                                                                                         \
      }                                                                                           \
      M3_ATOMS(t, PH_, P_, sl_, st_, ld_, SET_)                                                   \
      constexpr int sb = p == 1 ? 1 : 0, sa = p == 0 ? 1 : 0;                                     \
      _Pragma("unroll") for (int r = 0; r < 2; ++r)                                               \
        _Pragma("unroll") for (int h = 0; h < NH; ++h) {                                          \
          if ((M3_KO & 1) && k.D > 0) { M3_SINK(Bu[r + qo][h][sb]); M3_SINK(Aw[cur][sa]); }      \
          else if constexpr (RC) acc[a][r][h] = mma32m(Aw[cur][sa], Bu[r + qo][h][sb], acc[a][r][h]); \
          else if constexpr (B32) acc[a][r][h] = mma32m(Bu[r + qo][h][sb], Aw[cur][sa], acc[a][r][h]); \
          else acc[a][r][h] = mma16m(Bu[r + qo][h][sb], Aw[cur][sa], acc[a][r][h]);              \
        }                                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    });                                                                                           \
    M3_T(2)                                                                                       \
    /* output plane P - 1 is complete; NSETS == 3: its epilogue now, else inside the next step */  \
    if constexpr (NSETS == 3) {                                                                   \
      if (!((M3_KO & 16) && k.D > 0)) {                                                           \
        static_for_m<0, NE>([&](auto ec_) __attribute__((always_inline)) { M3_EPI_ONE(((PH_) + 2) % 3, decltype(ec_)::value, (P_) - 1) }); \
      }                                                                                           \
    }                                                                                             \
    M3_T(3)                                                                                       \
  }
  // one slot: the converted units of the next plane go to the slot between the two barriers
  if (M3_PRIO == 2 && wid >= 4) __builtin_amdgcn_s_setprio(1);
  const int P0 = zs - 1;
  M3_T0()
  M3_GLOAD(P0)
  __syncthreads();                                           // slots zeroed, weights in place
  M3_LSTORE(0)
  if constexpr (FLOW) {                                      // plane i of the segment lives in register set i & 1
    M3_GLOAD_S(nst > 1 ? P0 + 1 : -1, 1)
    M3_GLOAD_S(nst > 2 ? P0 + 2 : -1, 0)
  } else {
    if (SLOTS == 2 && nst > 1) M3_GLOAD(P0 + 1)
  }
  __syncthreads();
  ko_st = !((M3_KO & 4) && k.D > 0);
  M3_T(6)
#if M3_KO & 32
  { const u32x4* Xc = Xs; M3_BREAD(0, 0) M3_BREAD(0, 1) M3_BREAD(0, 2) M3_BREAD(0, 3) M3_AREAD(0, 0) M3_AREAD(1, 1) }
  ko_rd = k.D < 0;
#endif

  // one step of the march with its staging protocol; i = index of the input plane within the segment
#define M3_ITER(PH_, i_) M3_ITER_S(PH_, i_, 0)
#define M3_ITER_S(PH_, i_, SET_)           /* SET_ = (i_ + 1) & 1 in the FLOW form, else 0 */     \
  {                                                                                               \
    const int ii = (i_);                                                                          \
    if constexpr (SLOTS == 2) {                                                                   \
      M3_STEP(PH_, ii & 1, P0 + ii, ii + 1 < nst, ii + (FLOW ? 3 : 2) < nst, SET_)                \
      __syncthreads();                                                                            \
      M3_T(4)                                                                                     \
    } else {                                                                                      \
      M3_STEP(PH_, 0, P0 + ii, ii + 1 < nst, false, 0)                                            \
      if (ii + 1 < nst) {                                                                         \
        __syncthreads();                                                                          \
        M3_T(4)                                                                                   \
        M3_LSTORE(0)                                                                              \
        M3_T(0)                                                                                   \
        __syncthreads();                                                                          \
        M3_T(5)                                                                                   \
      }                                                                                           \
    }                                                                                             \
  }
  if constexpr (FLOW) {
    // (`break`, not `if (i + k < nst) step`: a step that is reachable around its predecessor makes the compiler place the
    // s_waitcnt of its conversion for the path on which the predecessor's loads were never issued)
    for (int i = 0; i < nst; i += 6) {                       // phase x register set: period 6
      M3_ITER_S(0, i, 1)
      if (i + 1 >= nst) break;
      M3_ITER_S(1, i + 1, 0)
      if (i + 2 >= nst) break;
      M3_ITER_S(2, i + 2, 1)
      if (i + 3 >= nst) break;
      M3_ITER_S(0, i + 3, 0)
      if (i + 4 >= nst) break;
      M3_ITER_S(1, i + 4, 1)
      if (i + 5 >= nst) break;
      M3_ITER_S(2, i + 5, 0)
    }
  } else if constexpr (NSETS == 3) {
    for (int i = 0; i < nst; i += 3) {
      M3_ITER(0, i)
      if (i + 1 < nst) M3_ITER(1, i + 1)
      if (i + 2 < nst) M3_ITER(2, i + 2)
    }
  } else {
    for (int i = 0; i < nst; i += 4) {
      M3_ITER(0, i)
      if (i + 1 < nst) M3_ITER(1, i + 1)
      if (i + 2 < nst) M3_ITER(2, i + 2)
      if (i + 3 < nst) M3_ITER(3, i + 3)
    }
    // the last finished plane (input plane P0 + nst - 1 closed output plane ze - 1) still holds its set
    if (!((M3_KO & 16) && k.D > 0)) {
      u32x4 av[ACTG ? NE : 1];
      const int pl = P0 + nst - 2;
      if constexpr (ACTG) {
#pragma unroll
        for (int e = 0; e < NE; ++e)
          av[e] = __builtin_amdgcn_raw_buffer_load_b128(a_src, (pl >= zs && pl < ze && evo[e] != OOB) ? evo[e] + (unsigned)pl * hw4 : OOB, 0, 0);
      }
      switch ((nst + 2) & 3) {                               // set of plane index (nst - 2) relative to P0: (nst - 2 + 4) % 4
        case 0: static_for_m<0, NE>([&](auto ec_) __attribute__((always_inline)) { M3_EPI_ONE(0, decltype(ec_)::value, pl) }); break;
        case 1: static_for_m<0, NE>([&](auto ec_) __attribute__((always_inline)) { M3_EPI_ONE(1, decltype(ec_)::value, pl) }); break;
        case 2: static_for_m<0, NE>([&](auto ec_) __attribute__((always_inline)) { M3_EPI_ONE(2, decltype(ec_)::value, pl) }); break;
        default: static_for_m<0, NE>([&](auto ec_) __attribute__((always_inline)) { M3_EPI_ONE(NSETS == 4 ? 3 : 0, decltype(ec_)::value, pl) }); break;
      }
    }
  }
#undef M3_ITER
#undef M3_ITER_S
#undef M3_STEP
#undef M3_ATOMS
#undef M3_EPI_ONE
#undef M3_AVLOAD
#undef M3_LOAD1
#undef M3_LOAD1S
#undef M3_CONV_HALF
#undef M3_KO_DECL
#undef M3_KO_RD
#undef M3_AREAD
#undef M3_WU
#undef M3_BREAD
#undef M3_LSTORE
#undef M3_LSTORE_S
#undef M3_GLOAD
#undef M3_GLOAD_S
  if (y_amax) {
    __syncthreads();
    publish_block_absmax_acc(pm, &smax, y_amax);
  }
}

bool march_off() {
  static DfOptFlag a{"DFMIR_CONV3D_NO_MARCH"}, b{"DFMIR_CONV3D_FP32"}, c{"DFMIR_CONV_FP32"};
  return a.get() || b.get() || c.get();
}
bool march_geom_ok(const DfConvGeom* g) {
  const bool shape = (g->Cin == 32 && g->Cout == 16) || (g->Cin == 16 && (g->Cout == 16 || g->Cout == 32)) ||
                     (g->Cin == 16 && g->Cout == 3 && g->act == 0);
  return shape && g->KD == 3 && g->KH == 3 && g->KW == 3 && g->stride == 1 && g->dil == 1 && g->pd == 1 && g->ph == 1 &&
         g->pw == 1 && g->pad_mode == 0 && g->Do == g->Di && g->Ho == g->Hi && g->Wo == g->Wi && (g->act == 0 || g->act == 1) &&
         g->Di >= 4 && g->Hi >= 8 && g->Wi >= 16 && (g->Wi % 4) == 0 &&
         (long long)(g->Cin > g->Cout ? g->Cin : g->Cout) * g->Di * g->Hi * g->Wi * 4 < 0x7FFFFFFFLL;
}

}  // namespace

extern "C" int dfmir_conv3d_march_ok(const DfConvGeom* g) { return (g && !march_off() && march_geom_ok(g)) ? 1 : 0; }

extern "C" int dfmir_conv3d_march_fwd(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                      const float* w_tcc, const float* bias, float* y, float* y_amax,
                                      const float* act_src, float act_slope, void* stream) {
  DF_ARG_CHECK(g && x && x_amax && x_amax_n > 0 && w_tcc && y);
  DF_ARG_CHECK(!march_off() && march_geom_ok(g) && (act_src == nullptr || g->act == 0));
  DF_ARG_CHECK(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(act_src) |
                 reinterpret_cast<uintptr_t>(w_tcc)) & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  MarchP k{};
  k.N = g->N; k.D = g->Di; k.H = g->Hi; k.W = g->Wi;
  k.nty = (g->Hi + 15) / 16; k.ntx = (g->Wi + 31) / 32;
  k.x_n = x_amax_n; k.act = g->act; k.slope = g->slope; k.act_src = act_src; k.act_slope = act_slope;
  // z segments: every workgroup pays two halo planes; pick the count that minimises (rounds of 256 CUs) x (planes per
  // workgroup).  DFMIR_MARCH_NSEG=<n> forces it.
  static DfOptInt nseg_o{"DFMIR_MARCH_NSEG", 0};
  const long long cols = (long long)g->N * k.nty * k.ntx;
  int best = 1;
  long long best_cost = -1;
  for (int ns = 1; ns <= g->Di / 2 && ns <= 128; ++ns) {
    const int zl = (g->Di + ns - 1) / ns;
    if ((long long)(ns - 1) * zl >= g->Di) continue;         // an empty last segment
    const long long rounds = (cols * ns + 255) / 256;
    const long long cost = rounds * (zl + 2);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ns; }
  }
  if (nseg_o.get() > 0 && nseg_o.get() <= g->Di) best = nseg_o.get();
  k.nseg = best;
  k.zlen = (g->Di + best - 1) / best;
  k.nseg = (g->Di + k.zlen - 1) / k.zlen;
  k.nwork = cols * k.nseg;
  const unsigned grid = (unsigned)(8 * ((k.nwork + 7) / 8));
#define M3_LAUNCH(CI_, CO_)                                                                       \
  {                                                                                               \
    if (act_src) conv3d_march_k<CI_, CO_, true><<<grid, 512, 0, st>>>(x, x_amax, w_tcc, bias, y, y_amax, k);  \
    else conv3d_march_k<CI_, CO_, false><<<grid, 512, 0, st>>>(x, x_amax, w_tcc, bias, y, y_amax, k);         \
  }
  if (g->Cout == 3) {
    DF_ARG_CHECK(!act_src);
    conv3d_march_k<16, 3, false><<<grid, 512, 0, st>>>(x, x_amax, w_tcc, bias, y, y_amax, k);
  }
  else if (g->Cin == 32) M3_LAUNCH(32, 16)
  else if (g->Cout == 16) M3_LAUNCH(16, 16)
  else M3_LAUNCH(16, 32)
#undef M3_LAUNCH
  DF_LAUNCH_CHECK();
  return 0;
}

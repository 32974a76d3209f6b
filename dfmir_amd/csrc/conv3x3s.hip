// 2-D 3x3 stride-1 convolution (forward, dgrad, wgrad), fp32 in / fp32 out, on the 16-bit matrix cores by
// operand splitting -- the scheme behind fp32-emulating GEMMs.  Two forms share every kernel here (NSP):
//
//   NSP = 3, "bf16x3":  a = a0 + a1 + a2 (bf16, round-to-nearest residuals: 3 x (8+1) bits >= fp32's 24)
//        a*b ~= a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0        6 MFMAs, dropped terms <= 2^-24 |ab|
//        bf16 has fp32's exponent range: no scaling, exact for any input.
//   NSP = 2, "fp16x2":  a = (a0 + a1) / s  with  a0 = fp16(s*a), a1 = fp16(s*a - a0)   (11 + 11 bits, +-2^-23)
//        a*b ~= a0b0 + a0b1 + a1b0                              3 MFMAs, dropped a1b1 <= 2^-22 |ab|
//        fp16 has 5 exponent bits, so each tensor is scaled by a power of two s = 2^(14 - ilogb(max|a|))
//        taken from a device-side max|a| (dfmir_absmax; weights: at pack time).  Elements within 2^-17 of the
//        tensor maximum keep 22 bits; smaller ones degrade gradually (fixed absolute error 2^-40 max|a|):
//        18 bits at 1e-6 of the maximum, 11 bits at 4e-9.
//        The result is rescaled by 2^-(ea+eb) in the epilogue.  Half the matrix-pipe work of bf16x3.
//
// Products are accumulated in fp32 by v_mfma_f32_32x32x16_{bf16,f16}.  Against an fp64 convolution both forms
// land at the fp32-MFMA kernels' error level (scripts/bench_conv.py, tests/test_gpu_ops.py); those kernels
// (conv3x3.hip) remain under DFMIR_CONV_FP32=1, and DFMIR_CONV_SPLIT=bf16x3 selects the unscaled form.
//
// Tiling follows the fp32 kernel (conv3x3.hip): 128 output channels x a run of 128 output pixels per
// group of 4 waves, one LDS halo patch per chunk of 8 input channels shared by all 9 taps, next chunk's
// global loads in flight during the MFMA phase.  What changes is the operand format:
//   * weights arrive pre-split from dfmir_weight_pack (16-B units of 8 input channels, layout
//     [chunk][split][tap][cout]) and are copied to LDS as they are;
//   * the patch is split when it is written to LDS: [split][position] x (8 channels = 16 B), so a
//     lane's MFMA B operand is one ds_read_b128;
//   * K = 16 of one MFMA = 2 taps x 8 channels: lanes 0-31 feed tap 2j, lanes 32-63 tap 2j+1 (their B
//     reads differ by the tap's patch shift).  The ninth tap pairs with a zero operand (10 % idle).
#include "conv3x3_common.h"

// Wave priorities of the ping-pong kernels (s_setprio): the computing group above the converting one.  With equal
// priorities the SIMD arbiter favours the lower wave slots, so group A's convert/store instructions displaced group
// B's MFMAs and B's compute half-steps ran 20 % longer than A's (per-half-step s_memtime trace, -DCS_TRACE).
#ifndef PP_COMPUTE_PRIO
#define PP_COMPUTE_PRIO 1
#endif
#ifndef CS_COMPUTE_PRIO
#define CS_COMPUTE_PRIO 1
#endif
#ifndef CS_STORE_PRIO
#define CS_STORE_PRIO 0
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (RNE), a in the low half
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// one pair of fp32 values -> NSP packed 16-bit pairs (term 0 = leading term)
template <int NSP>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&p)[NSP]) {
  if constexpr (NSP == 3) {
    p[0] = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(p[0] << 16), r1 = x1 - __uint_as_float(p[0] & 0xffff0000u);
    p[1] = pk_bf16(r0, r1);
    p[2] = pk_bf16(r0 - __uint_as_float(p[1] << 16), r1 - __uint_as_float(p[1] & 0xffff0000u));
  } else {
    const f32x2 v = {x0, x1};
    const f16x2 h = __builtin_convertvector(v, f16x2);            // v_cvt_pk_f16_f32 (RNE)
    const f32x2 r = v - __builtin_convertvector(h, f32x2);        // exact
    p[0] = __builtin_bit_cast(unsigned, h);
    p[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  }
}
// The scaled fp16x2 split of a pair in 4 instructions: v_fma_mix{lo,hi}_f16 multiply by the (power-of-two) scale,
// subtract the leading term read straight from its fp16 half, and round to fp16 once -- the same values as
// cvt(x*s), cvt(x*s - float(h)) (x*s and the difference are exact), without the 2 multiplies, 2 conversions back
// and 2 subtractions.  Every VALU instruction of the converting wave costs the computing wave matrix-pipe time.
__device__ __forceinline__ void split_pair_scaled(float x0, float x1, float s, unsigned (&p)[2]) {
  unsigned h, r;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(x1), "v"(s), "v"(h));
  p[0] = h; p[1] = r;
}
template <int NSP>
__device__ __forceinline__ void split_pair_s(float x0, float x1, float s, unsigned (&p)[NSP]) {
  if constexpr (NSP == 2) split_pair_scaled(x0, x1, s, p);
  else split_pair<NSP>(x0, x1, p);                               // bf16x3 is unscaled
}
// 8 fp32 (times the scale s in the fp16x2 form) -> NSP 16-B vectors of 8 halves
template <int NSP>
__device__ __forceinline__ void split8_s(const float v[8], float s, u32x4 (&out)[NSP]) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    unsigned p[NSP];
    split_pair_s<NSP>(v[2 * w], v[2 * w + 1], s, p);
#pragma unroll
    for (int q = 0; q < NSP; ++q) out[q][w] = p[q];
  }
}
// 8 fp32 -> NSP 16-B vectors of 8 halves
template <int NSP>
__device__ __forceinline__ void split8(const float v[8], u32x4 (&out)[NSP]) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    unsigned p[NSP];
    split_pair<NSP>(v[2 * w], v[2 * w + 1], p);
#pragma unroll
    for (int s = 0; s < NSP; ++s) out[s][w] = p[s];
  }
}
template <int NSP>
__device__ __forceinline__ f32x16 mma16(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (NSP == 3)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// products kept, smallest first: (A term, B term)
template <int NSP> struct Prod;
template <> struct Prod<3> { static constexpr int N = 6; static constexpr int A[6] = {2, 0, 1, 1, 0, 0}, B[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct Prod<2> { static constexpr int N = 3; static constexpr int A[3] = {1, 0, 0}, B[3] = {0, 1, 0}; };

// power-of-two scale exponent of a tensor whose max |.| is amax: |a| * 2^e < 2^15 (fp16 max 65504)
__device__ __forceinline__ int scale_exp(float amax) {
  const int be = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  int e = (amax > 0.f) ? 14 - be : 0;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);   // 2^e and 2^-e stay normal fp32 numbers
  return e;
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }

// ---- max |x| of a tensor, folded into a device scalar: out = max(out, max|x|)
__global__ __launch_bounds__(256) void absmax_k(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
  __shared__ float red[8];
  float m = 0.f;
  const long long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  // four independent 16-byte loads per trip, each under its own bound (a 27 MB probe is 3.3 quads per thread: with one
  // load in flight per thread the kernel was three memory latencies long, 18 us)
  const long long st = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (; i < n4; i += 4 * st) {
    const float4 a = x4[i];
    const float4 b = i + st < n4 ? x4[i + st] : z4;
    const float4 c = i + 2 * st < n4 ? x4[i + 2 * st] : z4;
    const float4 d = i + 3 * st < n4 ? x4[i + 3 * st] : z4;
    const float ma = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
    const float mb = fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)));
    const float mc = fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w)));
    const float md = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));
    m = fmaxf(m, fmaxf(fmaxf(ma, mb), fmaxf(mc, md)));
  }
  for (i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (!(m == m)) m = __uint_as_float(0x7f800000u);   // NaN -> +inf so that it survives the integer max
    // non-negative floats order like their bit patterns; the plain read keeps the blocks off one atomic
    if (__float_as_uint(m) > *reinterpret_cast<volatile unsigned*>(out)) atomicMax(out, __float_as_uint(m));
  }
}
int df_absmax_launch(const float* x, long long n, float* out, hipStream_t st, bool zero_first) {
  if (zero_first) {
    hipError_t e = df_zero_async(out, 1, st);
    if (e != hipSuccess) return (int)e;
  }
  long long blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;                        // (8 resident workgroups per CU)
  absmax_k<<<(unsigned)blocks, 256, 0, st>>>(x, n, reinterpret_cast<unsigned*>(out));
  return (int)hipGetLastError();
}

// ---- weight splitting: w_tcc [9][K][M] fp32 -> [ceil(K/8)][NSP][9][M] x 16 B.
// NSP = 2: trailer[4 .. 4+npart) = partial maxima of |w| left by weight_pack_k; trailer[1] <- the scale exponent
// used (as a float), read by the conv kernels
template <int NSP>
__global__ __launch_bounds__(256) void weight_split_k(const float* __restrict__ w_tcc, u32x4* __restrict__ out,
                                                      float* __restrict__ trailer, int npart, int K, int M) {
  __shared__ float red[17];
  const int chunks = (K + 7) >> 3;
  const long long total = (long long)chunks * 9 * M;
  int e = 0;
  if (NSP == 2) {
    e = scale_exp(reduce_absmax(trailer + 4, npart, red));
    if (blockIdx.x == 0 && threadIdx.x == 0) trailer[1] = (float)e;
  }
  const float sc = pow2f(e);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i % M);
    const int tap = (int)((i / M) % 9);
    const int ch = (int)(i / ((long long)9 * M));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = ch * 8 + j;
      v[j] = kk < K ? w_tcc[((long long)tap * K + kk) * M + m] * sc : 0.f;
    }
    u32x4 sp[NSP];
    split8<NSP>(v, sp);
    const long long base = (long long)ch * 9 * NSP * M + (long long)tap * M + m;
#pragma unroll
    for (int s = 0; s < NSP; ++s) out[base + (long long)s * 9 * M] = sp[s];
  }
}

// which split form the library runs with (read once): 0 = none (fp32 MFMA), 2 = fp16x2 (default), 3 = bf16x3
int df_split_mode() {
  static std::atomic<long long> st{-1};                   // (generation << 8) | mode
  const int g = df_opt_gen();
  const long long cur = st.load(std::memory_order_relaxed);
  if (cur >= 0 && (int)(cur >> 8) == g) return (int)(cur & 0xff);
  char s[16];
  const bool has = df_opt_get("DFMIR_CONV_SPLIT", s, sizeof(s));
  const int mode = df_opt_on("DFMIR_CONV_FP32") ? 0 : ((has && s[0] == 'b') ? 3 : 2);
  st.store(((long long)g << 8) | mode, std::memory_order_relaxed);
  return mode;
}

// split section of a packed weight buffer (K = reduction channels, M = produced channels)
static inline float* split_section(const float* packed, int K, int M) {
  return const_cast<float*>(packed) + df_pack_tcc_floats(K, M, 9);
}
static inline float* split_trailer(const float* packed, int K, int M, int nsp) {
  return split_section(packed, K, M) + (long long)((K + 7) / 8) * 9 * nsp * M * 4;
}

// where weight_pack_k leaves its partial maxima (after the 4-float trailer of the active split form)
float* df_weight_probe_slots(float* packed, int K, int M) {
  const int mode = df_split_mode();
  return mode == 2 ? split_trailer(packed, K, M, mode) + 4 : nullptr;
}
int df_weight_split_launch(const float* w_tcc, float* packed, int K, int M, int npart, hipStream_t st) {
  const int mode = df_split_mode();
  if (mode == 0) return 0;
  const long long total = (long long)((K + 7) / 8) * 9 * M;
  u32x4* sec = reinterpret_cast<u32x4*>(split_section(packed, K, M));
  float* tr = split_trailer(packed, K, M, mode);
  if (mode == 2) weight_split_k<2><<<df_grid(total, 256, 2048), 256, 0, st>>>(w_tcc, sec, tr, npart, K, M);
  else weight_split_k<3><<<df_grid(total, 256, 2048), 256, 0, st>>>(w_tcc, sec, tr, npart, K, M);
  return (int)hipGetLastError();
}

// ---- batched form: every packed weight of a step in one launch (job = blockIdx.y)
template <int NSP>
__global__ __launch_bounds__(256) void weight_split_batch_k(const DfPackJobDev* __restrict__ jobs) {
  const DfPackJobDev jb = jobs[blockIdx.y];
  if ((int)blockIdx.x >= jb.nsplit) return;
  __shared__ float red[17];
  const int K = jb.mode ? jb.Cout : jb.Cin, M = jb.mode ? jb.Cin : jb.Cout;
  const int chunks = (K + 7) >> 3;
  const long long total = (long long)chunks * 9 * M;
  int e = 0;
  if (NSP == 2) {
    e = scale_exp(reduce_absmax(jb.trailer + 4, jb.nblk, red));
    if (blockIdx.x == 0 && threadIdx.x == 0) jb.trailer[1] = (float)e;
  }
  const float sc = pow2f(e);
  u32x4* out = reinterpret_cast<u32x4*>(jb.sec);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)jb.nsplit * 256) {
    const int m = (int)(i % M);
    const int tap = (int)((i / M) % 9);
    const int ch = (int)(i / ((long long)9 * M));
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = ch * 8 + j;
      v[j] = kk < K ? jb.o[((long long)tap * K + kk) * M + m] * sc : 0.f;
    }
    u32x4 sp[NSP];
    split8<NSP>(v, sp);
    const long long base = (long long)ch * 9 * NSP * M + (long long)tap * M + m;
#pragma unroll
    for (int s = 0; s < NSP; ++s) out[base + (long long)s * 9 * M] = sp[s];
  }
}
void df_weight_split_fill(DfPackJobDev* j) {
  const int mode = df_split_mode();
  j->part = nullptr; j->sec = nullptr; j->trailer = nullptr; j->nsplit = 0;
  if (j->T != 9 || mode == 0) return;
  const int K = j->mode ? j->Cout : j->Cin, M = j->mode ? j->Cin : j->Cout;
  j->sec = split_section(j->o, K, M);
  j->trailer = split_trailer(j->o, K, M, mode);
  j->part = mode == 2 ? j->trailer + 4 : nullptr;
  j->nsplit = (int)df_grid((long long)((K + 7) / 8) * 9 * M, 256, 2048);
}
int df_weight_split_batch_launch(const DfPackJobDev* jobs_dev, int njobs, int max_nsplit, hipStream_t st) {
  const int mode = df_split_mode();
  if (mode == 0 || max_nsplit <= 0) return 0;
  const dim3 grid((unsigned)max_nsplit, (unsigned)njobs);
  if (mode == 2) weight_split_batch_k<2><<<grid, 256, 0, st>>>(jobs_dev);
  else weight_split_batch_k<3><<<grid, 256, 0, st>>>(jobs_dev);
  return (int)hipGetLastError();
}

// ---- the MFMA phase of one 8-channel chunk for a wave tile of TM x TN 32x32 blocks.
// Ab: weight chunk [split][tap][BM] (16-B units), Xb: halo patch [split][XP]; aoff[pr] / bidx[pr][j] are
// this lane's unit indices for tap pair pr.  Operands of pair pr+1 are read from LDS while the matrix pipe
// works on pair pr (two register sets); the sched_group_barriers pin that interleave -- left to itself the
// scheduler issued each ds_read right before its first use and the pipe idled ~45 % of the phase.
template <int NSP, int BM, int XP, int TM, int TN, int NV>
__device__ __forceinline__ void split_mma_chunk(const u32x4* __restrict__ Ab, const u32x4* __restrict__ Xb,
                                                const int (&aoff)[5], const int (&bidx)[5][TN],
                                                f32x16 (&acc)[TM][TN]) {
  using P = Prod<NSP>;
  u32x4 a[2][TM][NSP], b[2][TN][NSP];
#define SPLIT_LOAD(set_, pr_)                                                                    \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                               \
      _Pragma("unroll") for (int s = 0; s < NSP; ++s) a[set_][i][s] = Ab[s * 9 * BM + aoff[pr_] + i * 32]; \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                               \
      _Pragma("unroll") for (int s = 0; s < NSP; ++s) b[set_][j][s] = Xb[s * XP + bidx[pr_][j]]; \
  }
  SPLIT_LOAD(0, 0)
#pragma unroll
  for (int pr = 0; pr < 5; ++pr) {
    if (pr < 4) SPLIT_LOAD((pr + 1) & 1, pr + 1)
#pragma unroll
    for (int q = 0; q < P::N; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = mma16<NSP>(a[pr & 1][i][P::A[q]], b[pr & 1][j][P::B[q]], acc[i][j]);
  }
#undef SPLIT_LOAD
  // schedule: the NL ds_reads of the next pair ride behind the first MFMAs of a pair; the caller's NV
  // prefetch loads (global -> registers, issued in the same basic block) behind the following ones
  constexpr int NL = NSP * (TM + TN), NM = P::N * TM * TN;
  constexpr int LP = NL < NM ? NL : NM;                   // reads interleaved one per MFMA
  constexpr int VP = ((NV + 3) / 4 < NM - LP) ? (NV + 3) / 4 : NM - LP;
#pragma unroll
  for (int i = 0; i < NL; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
#pragma unroll
    for (int i = 0; i < LP; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (i == LP - 1) {
#pragma unroll
        for (int e = 0; e < NL - LP; ++e) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < NM - LP - VP; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
  }
#pragma unroll
  for (int i = 0; i < NM; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
}

// ---------------------------------------------------------------------------------------------
// The kernel: ping-pong over two wave groups.  A first version with two independent 256-thread workgroups
// per CU left the matrix pipe 64 % busy (profiles/r01_conv3x3s_pmc.md): fair MFMA arbitration makes the
// co-resident workgroups finish their MFMA phases together, then both convert/store with the pipe idle.
// Here ONE 512-thread workgroup holds two groups of 4 waves (one wave of each group per SIMD) that
// alternate by construction: in every half-step one group runs the MFMA phase of a chunk while the other
// converts + stores its next halo patch and half of the next weight chunk, then a barrier swaps the roles.
// The weight chunk is shared by both groups (BM output channels x 2 x 128 pixels per workgroup) and
// double-buffered.
//   half-step h, chunk c = h >> 1:   group A (0) computes on even h, group B (1) on odd h.
//   A prefetches X_A(c+1), W_A-half(c+1) while computing c and stores them at h = 2c+1;
//   B prefetches X_B(c+1), W_B-half(c+2) while computing c and stores them at h = 2c+2.
//   W(c+1) is therefore complete at the end of h = 2c+1, and its buffer was last read (chunk c-1) at
//   h = 2c-1: every hand-over is ordered by the per-half-step barrier.
struct SplitScale {
  const float* x_amax;    // device: x_n partial maxima of |x| over the input tensor (NSP = 2 only)
  int x_n;
  const float* w_trailer; // device: [1] = weight scale exponent (NSP = 2 only)
};

template <int NSP, int XP, int BM>
__global__ __launch_bounds__(512, 1) void conv3x3_split_pp_k(const float* __restrict__ x,
                                                             const u32x4* __restrict__ ws,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ y, Conv3P k, SplitScale sc) {
  // BM = 128: waves 2 x 2, each 64 couts x 64 pixels;  BM = 64: waves 1 x 4, each 64 couts x 32 pixels
  constexpr int BNG = 128, CK = 8, TM = 2, TN = (BM == 128) ? 2 : 1, WN = (BM == 128) ? 2 : 4;
  constexpr int NS = (XP + 255) / 256;
  constexpr int WU = 9 * NSP * BM, WH = WU / 2;   // 16-B units of one weight chunk / of one group's share
  constexpr int NW = (WH + 255) / 256;
  constexpr int ZPOS = XP - 1;                    // a patch position that always holds zeros
  __shared__ __attribute__((aligned(16))) u32x4 As[2][WU];
  __shared__ __attribute__((aligned(16))) u32x4 Xs[2][NSP * XP];
  __shared__ float bs[BM];

  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int HWo = k.Ho * k.Wo, HWi = k.Hi * k.Wi;
  const int n = blockIdx.x / k.tiles_per_img;
  const int t = blockIdx.x - n * k.tiles_per_img;
  const int p0 = t * (2 * BNG) + grp * BNG;
  const bool gvalid = p0 < HWo;               // the last tile of an image may leave group B without pixels
  const int pend = (p0 + BNG < HWo) ? p0 + BNG : HWo;
  const int m0 = blockIdx.y * BM;
  const int y0 = p0 / k.Wo, y1 = (pend - 1) / k.Wo;
  const bool single = (y0 == y1);
  const int x0 = p0 - y0 * k.Wo, x1 = (pend - 1) - y1 * k.Wo;
  const int xoff = single ? x0 : 0;
  const int ncols = single ? (x1 - x0 + 3) : (k.Wo + 2);
  const int nrows = y1 - y0 + 3;
  const int npos = gvalid ? nrows * ncols : 0;
  u32x4* __restrict__ Xg = Xs[grp];

  // scales (NSP = 2): input scaled by 2^ex when it is split, result rescaled by 2^-(ex+ew)
  float xscale = 1.f, oscale = 1.f, oscale2 = 1.f;   // two factors: 2^-(ex+ew) alone can leave fp32's range
  if (NSP == 2) {
    const int ex = scale_exp(reduce_absmax(sc.x_amax, sc.x_n, bs));   // bs: scratch here, bias below
    __syncthreads();
    const int ew = (int)sc.w_trailer[1];
    xscale = pow2f(ex);
    oscale = pow2f(-ex);
    oscale2 = pow2f(-ew);
  }

  constexpr unsigned OOB = 0x80000000u;
  // byte offsets of this thread's patch positions within one 8-channel slab, per channel: constant over
  // the chunk loop (the slab base moves in the scalar buffer descriptor instead), so the prefetch issues
  // no vector ALU work -- on this chip every VALU instruction delays the matrix pipe by its 4 cycles
  // (scripts/ubench/mfma_peak.hip).
  unsigned gvo[NS][CK];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int pos = tid + 256 * s;
    int off = -1;
    if (pos < npos) {
      const int r = pos / ncols, c = pos - r * ncols;
      off = halo_offset(y0 - k.pad + r, xoff - k.pad + c, k.Hi, k.Wi, k.pad_mode);
    }
#pragma unroll
    for (int c = 0; c < CK; ++c) gvo[s][c] = off < 0 ? OOB : (unsigned)(off + c * HWi) * 4u;
  }
  if ((int)threadIdx.x < BM) bs[threadIdx.x] = (bias && (m0 + (int)threadIdx.x) < k.Cout) ? bias[m0 + threadIdx.x] : 0.f;
  if (tid < NSP) Xg[tid * XP + ZPOS] = u32x4{0u, 0u, 0u, 0u};

  int pbase[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int q = p0 + (wn * TN + j) * 32 + l31;
    q = q < pend ? q : pend - 1;
    const int yy = q / k.Wo, xx = q - yy * k.Wo;
    pbase[j] = (yy - y0) * ncols + (xx - xoff);
  }
  // per tap pair: this half-wave's A unit offset and B patch index (tap 8 pairs with the zero position)
  int aoff[5], bidx[5][TN];
#pragma unroll
  for (int pr = 0; pr < 5; ++pr) {
    const int tp = (pr == 4) ? 8 : 2 * pr + lhi;
    aoff[pr] = tp * BM + wm * TM * 32 + l31;
#pragma unroll
    for (int j = 0; j < TN; ++j)
      bidx[pr][j] = ((pr == 4) && lhi) ? ZPOS : pbase[j] + (tp / 3) * ncols + (tp % 3);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* xn = x + (long long)n * k.Cin * HWi;
  const int chunks = (k.Cin + CK - 1) / CK;
  unsigned wbyte[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int loc = tid + 256 * j;
    const int idx = grp * WH + loc;
    const int seg = idx / BM, co = m0 + (idx % BM);
    wbyte[j] = (loc < WH && co < k.Cout) ? (unsigned)(seg * k.Cout + co) * 16u : OOB;
  }
  const int wunits = 9 * NSP * k.Cout;

  u32x4 rw[NW];
  unsigned rx[NS][CK];

  // chunk ch_ = its own descriptor (base advanced, extent = what is left of the tensor, 0 past the end)
#define P3_GLOADW(ch_)                                                                           \
  {                                                                                              \
    const int c_ = (ch_);                                                                        \
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<u32x4*>(ws + (long long)c_ * wunits), 0, c_ < chunks ? (unsigned)wunits * 16u : 0u, 0x00020000); \
    _Pragma("unroll") for (int j = 0; j < NW; ++j)                                               \
      rw[j] = __builtin_amdgcn_raw_buffer_load_b128(rw_, wbyte[j], 0, 0);                        \
  }
#define P3_GLOADX(ch_)                                                                           \
  {                                                                                              \
    const int c0_ = (ch_) * CK;                                                                  \
    const int left_ = k.Cin - c0_;                                                               \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(xn + (long long)c0_ * HWi), 0,                                        \
        left_ > 0 ? (unsigned)((left_ < CK ? left_ : CK) * HWi) * 4u : 0u, 0x00020000);          \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                             \
      _Pragma("unroll") for (int c = 0; c < CK; ++c)                                             \
        rx[s][c] = __builtin_amdgcn_raw_buffer_load_b32(rx_, gvo[s][c], 0, 0);                   \
    }                                                                                            \
  }
#define P3_LSTOREW(buf_)                                                                         \
  {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) {                                             \
      const int loc = tid + 256 * j;                                                             \
      if (loc < WH) As[buf_][grp * WH + loc] = rw[j];                                            \
    }                                                                                            \
  }
#define P3_LSTOREX()                                                                             \
  {                                                                                              \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                             \
      const int pos = tid + 256 * s;                                                             \
      if (pos < npos) {                                                                          \
        float v[8];                                                                              \
        _Pragma("unroll") for (int c = 0; c < CK; ++c)                                           \
          v[c] = __uint_as_float(rx[s][c]);                                                      \
        u32x4 sp[NSP];                                                                           \
        split8_s<NSP>(v, xscale, sp);                                                                      \
        _Pragma("unroll") for (int q = 0; q < NSP; ++q) Xg[q * XP + pos] = sp[q];                \
      }                                                                                          \
    }                                                                                            \
  }

  // prologue: W(0) (each group its half) + own X(0); group B also fetches its half of W(1)
  P3_GLOADW(0);
  P3_GLOADX(0);
  P3_LSTOREW(0);
  P3_LSTOREX();
  if (grp == 1 && chunks > 1) P3_GLOADW(1);
  __syncthreads();

  for (int h = 0; h < 2 * chunks; ++h) {
    const int c = h >> 1;
    if ((h & 1) == grp) {
      // ---- compute half-step: the MFMAs of chunk c with the prefetch loads interleaved.  Chunks past
      // the end lie beyond the buffer descriptors and read 0, so the loads need no guard.
      __builtin_amdgcn_s_setprio(PP_COMPUTE_PRIO);      // the converting group must fit in, not win (see cs kernel)
      if (gvalid) {
        P3_GLOADX(c + 1);
        P3_GLOADW(c + 1 + grp);
        split_mma_chunk<NSP, BM, XP, TM, TN, NW + NS * CK>(As[c & 1], Xg, aoff, bidx, acc);
      } else {
        P3_GLOADW(c + 1 + grp);
      }
    } else {
      // ---- store half-step: what this group prefetched during its last compute half-step
      __builtin_amdgcn_s_setprio(0);
      const int cx = c + 1 - grp;               // A: X(c+1);  B: X(c)  (B's X(0) is already in place)
      if (cx < chunks && h > 0) P3_LSTOREX();
      if (c + 1 < chunks) P3_LSTOREW((c + 1) & 1);
    }
    __syncthreads();
  }
#undef P3_GLOADW
#undef P3_GLOADX
#undef P3_LSTOREW
#undef P3_LSTOREX

  if (!gvalid) return;
  float* yb = y + (long long)n * k.Cout * HWo;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int q = p0 + (wn * TN + j) * 32 + l31;
    if (q >= pend) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int cl = (wm * TM + i) * 32 + 4 * lhi;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cc = cl + (r & 3) + 8 * (r >> 2);
        const int co = m0 + cc;
        if (co < k.Cout) {
          float v = (NSP == 2 ? acc[i][j][r] * oscale * oscale2 : acc[i][j][r]) + bs[cc];
          if (k.act == 1) v = v > 0.f ? v : v * k.slope;
          else if (k.act == 2) v = tanhf(v);
          yb[(long long)co * HWo + q] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fp16x2 forward/dgrad, second form: 2-D pixel tiles, 16-channel chunks, wave groups split by OUTPUT CHANNEL.
//
// The kernel above spends half of every half-step outside the MFMAs (profiles/r01_conv3x3s_pmc.md): the
// ninth tap's padded k-step, the patch split (done once per wave group because each group has its own
// pixels) and a barrier per 8 channels.  Here both groups of the 512-thread workgroup work on the SAME
// 8 x 32 pixel tile and each owns 64 of the 128 output channels:
//   * the halo patch (10 x 34 positions) is shared: each group converts one 8-channel half of a 16-channel
//     chunk, so the split VALU work per MFMA halves;
//   * K = 16 of an MFMA = one tap x 16 channels (lanes 0-31: channels 0-7, lanes 32-63: 8-15): 9 k-steps,
//     no padded tap;
//   * one barrier per 16 channels per group.
// LDS: W_g[split][tap][half][64 couts] per group, single-buffered (a group rewrites its weights in its own
// store half-step); X[buf][split][half][352] double-buffered (both groups read chunk c, one after the
// other, while chunk c+1 is being written).  Schedule (c = h >> 1):
//   A (couts 0-63)  : computes chunk c at h = 2c (prefetching W_A(c+1), X-half-0(c+1)), stores them at 2c+1;
//   B (couts 64-127): stores W_B(c), X-half-1(c+1) at h = 2c, computes chunk c at 2c+1 (prefetching
//                     W_B(c+1), X-half-1(c+2)).
//   X(c+1) is complete at the end of h = 2c+1; its buffer was last read (chunk c-1) at h = 2c-1.
constexpr int CS_TH = 8, CS_TW = 32, CS_PW = CS_TW + 2, CS_XP = 352;   // 10 x 34 = 340 patch positions

// One compute half-step: 9 taps x 12 MFMAs.  Per tap, in one fenced scheduling region: the ds_reads of the next
// tap's operands, a slice of the caller's prefetch (the 9 weight + 16 patch loads of the chunk this group stores
// next), and the MFMAs, interleaved by sched_group_barrier.  Without the fences the scheduler hoisted all 25
// buffer loads in front of the first MFMA and the half-step paid their issue time (16 % of the kernel).
__device__ __forceinline__ void cs_mma_chunk(const u32x4* __restrict__ Ab, const u32x4* __restrict__ Xb, int abase,
                                             const int (&pb)[2], f32x16 (&acc)[4],
                                             const __amdgpu_buffer_rsrc_t rw_, const unsigned (&wb)[9], u32x4 (&rw)[9],
                                             const __amdgpu_buffer_rsrc_t rx_, const unsigned (&gvo)[2][8],
                                             unsigned (&rx)[2][8]) {
  constexpr int SA = 9 * 2 * 64, SX = 2 * CS_XP;          // units per split in W_g / X
  using P = Prod<2>;
  u32x4 a[2][2][2], b[2][2][2];                           // [set][tile][split]
#define CS_LOAD(set_, t_)                                                                        \
  {                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) a[set_][i][s] = Ab[s * SA + (t_) * 128 + abase + i * 32]; \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) b[set_][j][s] = Xb[s * SX + pb[j] + ((t_) / 3) * CS_PW + (t_) % 3]; \
  }
  CS_LOAD(0, 0)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (t < 8) CS_LOAD((t + 1) & 1, t + 1)
    // prefetch slice: loads 3t .. 3t+2 of the flattened list [9 weight units | 2 x 8 patch channels]
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int L = 3 * t + q;
      if (L < 9) rw[L] = __builtin_amdgcn_raw_buffer_load_b128(rw_, wb[L], 0, 0);
      else if (L < 25) rx[(L - 9) >> 3][(L - 9) & 7] = __builtin_amdgcn_raw_buffer_load_b32(rx_, gvo[(L - 9) >> 3][(L - 9) & 7], 0, 0);
    }
#pragma unroll
    for (int q = 0; q < P::N; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[2 * i + j] = mma16<2>(b[t & 1][j][P::B[q]], a[t & 1][i][P::A[q]], acc[2 * i + j]);   // rows = pixels, columns = couts
    const int nds = t < 8 ? 8 : 0, nvm = (3 * t + 3 <= 25) ? 3 : (25 - 3 * t > 0 ? 25 - 3 * t : 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < nds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      else if (i - nds < nvm) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef CS_LOAD
}

// Row-reuse form of the same half-step.  A wave owns 32 output channels x 4 tile rows instead of 64 x 2: the X
// operand of tap (ky, kx) for tile row r is the operand of tap (ky+1, kx) for row r-1, so per kx the wave reads 6
// patch rows once (instead of 3 taps x 4 rows) and one weight block per tap: 54 ds_read_b128 per 108 MFMAs
// instead of 72 -- the LDS operand traffic is what bounds this kernel (profiles/r01_conv3x3s_pmc.md).
// Stage order: kx outer, ky inner; the reads of stage s+1 (row ky+4 and the next weight block, or the next kx's
// first four rows) and a slice of the prefetch ride between the MFMAs of stage s.
// CPG = output channels per wave group (64: tile 8 x 32, 32: tile 16 x 32), XP = patch positions (padded),
// NW / NS = weight units / patch positions a thread prefetches per chunk.
template <int CPG, int XP, int NW, int NS>
__device__ __forceinline__ void cs_mma_chunk_rr(const u32x4* __restrict__ Ab, const u32x4* __restrict__ Xb, int abase,
                                                int xb, f32x16 (&acc)[4],
                                                const __amdgpu_buffer_rsrc_t rw_, const unsigned (&wb)[NW], u32x4 (&rw)[NW],
                                                const __amdgpu_buffer_rsrc_t rx_, const unsigned (&gvo)[NS][8],
                                                unsigned (&rx)[NS][8]) {
  constexpr int SA = 9 * 2 * CPG, SX = 2 * XP;            // units per split in W_g / X
  constexpr int NL = NW + 8 * NS;                         // prefetch loads of a chunk
  using P = Prod<2>;
  u32x4 a[9][2], xr[3][6][2];                             // [stage][split], [kx][patch row][split]
#define CS_LDA(st_, s_) a[st_][s_] = Ab[(s_) * SA + (((st_) % 3) * 3 + (st_) / 3) * (2 * CPG) + abase];
#define CS_LDX(kx_, r_, s_) xr[kx_][r_][s_] = Xb[(s_) * SX + xb + (r_) * CS_PW + (kx_)];
  // operand reads of stage n (N(n)): stage kx*3: weights + rows 0-3 (10 reads), kx*3+1: weights + row 4, kx*3+2:
  // weights + row 5 (4 each); first-needed first (product 0 = A term 1 x B term 0)
#define CS_NEED(n_)                                                                              \
  {                                                                                              \
    constexpr int kx_ = (n_) / 3, ky_ = (n_) % 3;                                                \
    CS_LDA(n_, 1)                                                                                \
    if (ky_ == 0) { _Pragma("unroll") for (int r = 0; r < 4; ++r) CS_LDX(kx_, r, 0) }            \
    else CS_LDX(kx_, ky_ + 3, 0)                                                                 \
    CS_LDA(n_, 0)                                                                                \
    if (ky_ == 0) { _Pragma("unroll") for (int r = 0; r < 4; ++r) CS_LDX(kx_, r, 1) }            \
    else CS_LDX(kx_, ky_ + 3, 1)                                                                 \
  }
  // Issue plan (reads ride between the MFMAs of the stage named first; the LDS needs ~2 stages of lead when the
  // four computing waves of the CU read in lockstep -- with one stage of lead the MFMAs and the reads serialised):
  //   after the barrier: N0 | stage 0: N1 N2 | 1: N3 | 2: N4 | 3: N5 | 4: N6 | 5: N7 | 6: N8 | 7, 8: none
  CS_NEED(0)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int st = 0; st < 9; ++st) {
    const int kx = st / 3, ky = st % 3;
    int nds = 0;
    if (st == 0) { CS_NEED(1) CS_NEED(2) nds = 8; }
    else if (st == 1) { CS_NEED(3) nds = 10; }
    else if (st == 2) { CS_NEED(4) nds = 4; }
    else if (st == 3) { CS_NEED(5) nds = 4; }
    else if (st == 4) { CS_NEED(6) nds = 10; }
    else if (st == 5) { CS_NEED(7) nds = 4; }
    else if (st == 6) { CS_NEED(8) nds = 4; }
    // prefetch slices per stage.  25 loads: 4 2 4 4 2 4 4 1 0; otherwise spread evenly over stages 0-7
    int L0, nvm;
    if (NL == 25) {
      L0 = st == 0 ? 0 : st == 1 ? 4 : st == 2 ? 6 : st == 3 ? 10 : st == 4 ? 14 : st == 5 ? 16 : st == 6 ? 20 : 24;
      nvm = (st == 1 || st == 4) ? 2 : (st == 7 ? 1 : (st == 8 ? 0 : 4));
    } else {
      const int lo = NL / 8, ex = NL % 8;
      L0 = st * lo + (st < ex ? st : ex);
      nvm = st < 8 ? lo + (st < ex ? 1 : 0) : 0;
    }
#pragma unroll
    for (int q = 0; q < (NL + 7) / 8; ++q) {
      const int L = L0 + q;
      if (q < nvm) {
        if (L < NW) rw[L] = __builtin_amdgcn_raw_buffer_load_b128(rw_, wb[L], 0, 0);
        else if (L < NL) rx[(L - NW) >> 3][(L - NW) & 7] = __builtin_amdgcn_raw_buffer_load_b32(rx_, gvo[(L - NW) >> 3][(L - NW) & 7], 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < P::N; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j] = mma16<2>(xr[kx][ky + j][P::B[q]], a[st][P::A[q]], acc[j]);   // rows = pixels, columns = couts
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < nds) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (i >= 12 - nvm) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#undef CS_NEED
#undef CS_LDA
#undef CS_LDX
}

struct ConvCsP {
  int N, Cin, Cout, Hi, Wi, Ho, Wo, pad, pad_mode, act;
  float slope;
  int tiles_x, tiles_y;
  const float* res;       // optional [N][Cout][Ho][Wo]: y = act(conv + bias) + res (a second gradient / residual)
  const float* ring;      // optional [N][4][Cout][ring_rl]: the reflect ring of a dgrad (conv3x3_reflect_ring_k)
  int ring_rl;
  // 1-D grid of 2 x tiles workgroups, the two cout halves of a pixel tile on the SAME XCD, 8 dispatch slots apart:
  // id = 16 g + 8 h + r  ->  tile 8 g + r, half h (workgroup ids go round-robin over the 8 XCDs, one L2 each)
  int xcd_pair;
  // de-phasing of the CUs: workgroup b < 256 of the first wave of workgroups starts b * dephase / 256 ticks (10 ns) late,
  // so that the CUs' epilogues (a 33 MB burst of stores when all 256 run in lockstep) spread over the main loops
  int dephase;
  int vec4;               // Wo % 4 == 0 and y / res 16-byte aligned: the epilogue moves float4
};

#ifdef CS_TRACE
__device__ unsigned g_cs_trace[8 * 32 * 8 + 8];
extern "C" int dfmir_cs_trace_dump(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cs_trace), sizeof(unsigned) * (8 * 32 * 8 + 8));
}
#define TRC(slot_) { if (trace_blk && lane == 0) trc[((grp * 4 + wid) * 32 + (h & 31)) * 8 + (slot_)] = (unsigned)__builtin_readcyclecounter(); }
// per-workgroup wall-clock stamps of the LAST launch (100 MHz counter): [start, after prologue, end of main loop, end], + HW_ID, XCC_ID
__device__ unsigned g_cs_wg[2048 * 6];
extern "C" int dfmir_cs_wg_dump(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cs_wg), sizeof(unsigned) * 2048 * 6);
}
#define WGT(slot_) { if (threadIdx.x == 0 && blockIdx.x < 2048 && blockIdx.y == 0) g_cs_wg[blockIdx.x * 6 + (slot_)] = (unsigned)wall_clock64(); }
#else
#define WGT(slot_)
#define TRC(slot_)
#endif
// <RR, CPG, TH>: <*, 64, 8> = 128 output channels per workgroup on an 8 x 32 tile (RR: row-reuse compute phase);
// <true, 32, 16> = 64 output channels on a 16 x 32 tile (the 128->64 / 64<-128 layers at 256^2): a wave still owns
// 32 couts x 4 tile rows, a group its 32 couts over all 16 rows.
template <bool RR, int CPG, int TH>
__global__ __launch_bounds__(512, 1) void conv3x3_split_cs_k(const float* __restrict__ x, const u32x4* __restrict__ ws,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             ConvCsP k, SplitScale sc) {
  static_assert((CPG == 64 && TH == 8) || (RR && CPG == 32 && TH == 16), "tile forms");
  constexpr int CS_TH = TH;
  constexpr int NSP = 2, XP = (TH == 8 ? CS_XP : 640), NPOS = (CS_TH + 2) * CS_PW;
  constexpr int WUG = NSP * 9 * 2 * CPG;                  // 16-B units of one group's weight chunk (16 channels)
  constexpr int NW = (WUG + 255) / 256;                   // 9 (CPG 64) / 5 (CPG 32, the last one half used) per thread
  constexpr int NS = (NPOS + 255) / 256;                  // 2 / 3 patch positions per thread
  __shared__ __attribute__((aligned(16))) u32x4 Wg[2][WUG];
  __shared__ __attribute__((aligned(16))) u32x4 Xs[2][NSP * 2 * XP];
  __shared__ float bs[2 * CPG];
#ifdef CS_TRACE
  __shared__ unsigned trc[8 * 32 * 8];
  const bool trace_blk = blockIdx.x == 300 && blockIdx.y == 0;
  const unsigned long long trc_t0 = __builtin_readcyclecounter(), trc_r0 = wall_clock64();
#endif

  // wave-uniform by construction; readfirstlane tells the compiler so (otherwise every buffer load whose
  // descriptor depends on the group is wrapped in a waterfall loop)
  const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int HWo = k.Ho * k.Wo, HWi = k.Hi * k.Wi;
  if (k.dephase > 0 && blockIdx.y == 0 && blockIdx.x < 256) {
    const unsigned long long t0 = wall_clock64();
    const unsigned wait = blockIdx.x * (unsigned)k.dephase >> 8;
    while ((unsigned)(wall_clock64() - t0) < wait) __builtin_amdgcn_s_sleep(4);
  }
  WGT(0)
#ifdef CS_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 2048 && blockIdx.y == 0) {
    g_cs_wg[blockIdx.x * 6 + 4] = __builtin_amdgcn_s_getreg(63492);       // HW_REG_HW_ID
    g_cs_wg[blockIdx.x * 6 + 5] = __builtin_amdgcn_s_getreg(63508);       // HW_REG_XCC_ID
  }
#endif
  int bt = blockIdx.x, bm = blockIdx.y;
  if (k.xcd_pair == 1) { bm = (bt >> 3) & 1; bt = ((bt >> 4) << 3) + (bt & 7); }
  else if (k.xcd_pair >= 2) {   // XCD e = id & 7 walks a contiguous eighth of the tiles; xcd_pair - 1 cout slices of a tile back to back
    const int ny = k.xcd_pair - 1, per = (int)(gridDim.x >> 3) / ny, j = bt >> 3;
    bm = j % ny;
    bt = (bt & 7) * per + j / ny;
  }
  const int tx = bt % k.tiles_x; bt /= k.tiles_x;
  const int ty = bt % k.tiles_y;
  const int n = bt / k.tiles_y;
  const int oy0 = ty * CS_TH, ox0 = tx * CS_TW;
  const int m0 = bm * (2 * CPG), m0g = m0 + CPG * grp;

  const int ex = scale_exp(reduce_absmax(sc.x_amax, sc.x_n, bs));     // bs: scratch here, bias below
  __syncthreads();
  const int ew = (int)sc.w_trailer[1];
  const float xscale = pow2f(ex), oscale = pow2f(-ex), oscale2 = pow2f(-ew);

  constexpr unsigned OOB = 0x80000000u;
  // this thread's patch positions: byte offset of channel ch of its 8-channel half, within a 16-channel slab
  unsigned gvo[NS][8];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int pos = tid + 256 * s;
    int off = -1;
    if (pos < NPOS) {
      const int r = pos / CS_PW, c = pos - r * CS_PW;
      off = halo_offset(oy0 - k.pad + r, ox0 - k.pad + c, k.Hi, k.Wi, k.pad_mode);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) gvo[s][c] = off < 0 ? OOB : (unsigned)(off + (8 * grp + c) * HWi) * 4u;
  }
  if (threadIdx.x < 2 * CPG) bs[threadIdx.x] = (bias && (m0 + (int)threadIdx.x) < k.Cout) ? bias[m0 + threadIdx.x] : 0.f;

  // MFMA operand indices.  Plain form: wave w of a group owns 64 couts x tile rows 2w, 2w+1 (32 pixels each);
  // row-reuse form: 32 couts (w & 1) x tile rows 4(w >> 1) .. +3
  int pb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) pb[j] = lhi * XP + (2 * wid + j) * CS_PW + l31;
  const int rowgrp = CPG == 64 ? (wid >> 1) : wid;        // this wave's 4 tile rows
  const int xb = lhi * XP + 4 * rowgrp * CS_PW + l31;
  const int abase = lhi * CPG + l31 + ((RR && CPG == 64) ? 32 * (wid & 1) : 0);

  f32x16 acc[4];                                          // plain: [cout block i][row j] at 2i + j; row-reuse: [row]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const float* xn = x + (long long)n * k.Cin * HWi;
  const int chunks8 = (k.Cin + 7) / 8, chunks = (k.Cin + 15) / 16;
  // weight unit idx = r*64 + co with r = (split*9 + tap)*2 + half; global unit ((2c + half)*NSP*9 + split*9 + tap)*Cout + cout
  unsigned wb[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int idx = tid + 256 * j;
    const int co = m0g + (idx % CPG), r = idx / CPG;
    const int half = r & 1, st = r >> 1;
    wb[j] = (idx < WUG && co < k.Cout) ? (unsigned)((half * NSP * 9 + st) * k.Cout + co) * 16u : OOB;
  }
  const int wunits8 = NSP * 9 * k.Cout;                   // units of one 8-channel chunk in the packed weights

  u32x4 rw[NW];
  unsigned rx[NS][8];
#define CS_GLOADW(c_)                                                                            \
  {                                                                                              \
    const int q_ = 2 * (c_);                                                                     \
    const int left_ = chunks8 - q_;                                                              \
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<u32x4*>(ws + (long long)q_ * wunits8), 0,                                     \
        left_ > 0 ? (unsigned)((left_ < 2 ? left_ : 2) * wunits8) * 16u : 0u, 0x00020000);       \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) rw[j] = __builtin_amdgcn_raw_buffer_load_b128(rw_, wb[j], 0, 0); \
  }
#define CS_GLOADX(c_)                                                                            \
  {                                                                                              \
    const int c0_ = 16 * (c_);                                                                   \
    const int left_ = k.Cin - c0_;                                                               \
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(xn + (long long)c0_ * HWi), 0,                                        \
        left_ > 0 ? (unsigned)((left_ < 16 ? left_ : 16) * HWi) * 4u : 0u, 0x00020000);          \
    _Pragma("unroll") for (int s = 0; s < NS; ++s)                                               \
      _Pragma("unroll") for (int c = 0; c < 8; ++c)                                              \
        rx[s][c] = __builtin_amdgcn_raw_buffer_load_b32(rx_, gvo[s][c], 0, 0);                   \
  }
#define CS_LSTOREW()                                                                             \
  { _Pragma("unroll") for (int j = 0; j < NW; ++j) if (WUG % 256 == 0 || tid + 256 * j < WUG) Wg[grp][tid + 256 * j] = rw[j]; }
#define CS_LSTOREX(buf_)                                                                         \
  {                                                                                              \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                             \
      const int pos = tid + 256 * s;                                                             \
      if (pos < NPOS) {                                                                          \
        float v[8];                                                                              \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) v[c] = __uint_as_float(rx[s][c]);          \
        CS_NORM_PROBE_OPS()                                                                      \
        u32x4 sp[NSP];                                                                           \
        split8_s<NSP>(v, xscale, sp);                                                                      \
        _Pragma("unroll") for (int q = 0; q < NSP; ++q) Xs[buf_][(q * 2 + grp) * XP + pos] = sp[q]; \
      }                                                                                          \
    }                                                                                            \
  }

  // CS_NORM_PROBE (lab builds, scripts/build_var.sh): what an InstanceNorm + ReLU applied while the patch is staged would
  // cost the converting wave group -- one fma and one max per value with run-time operands that happen to be the identity
  // (x * 1 + 0, max with -3e38: results unchanged, instructions real; profiles/r06_cs_norm_probe.txt)
#ifdef CS_NORM_PROBE
  const float np_r = fmaf(oscale2, 0.f, 1.f), np_m = oscale2 * 0.f, np_lo = fmaf(oscale2, 0.f, -3.0e38f);
#define CS_NORM_PROBE_OPS() _Pragma("unroll") for (int c = 0; c < 8; ++c) v[c] = fmaxf(fmaf(v[c], np_r, np_m), np_lo);
#else
#define CS_NORM_PROBE_OPS()
#endif
  // prologue: X(0) (each group its channel half) and W_A(0) in place; B holds W_B(0), X-half-1(1) in registers
  CS_GLOADX(0);
  if (grp == 0) CS_GLOADW(0);
  CS_LSTOREX(0);
  if (grp == 0) CS_LSTOREW();
  if (grp == 1) { CS_GLOADW(0); CS_GLOADX(1); }
  __syncthreads();
  WGT(1)

  for (int h = 0; h < 2 * chunks; ++h) {
    const int c = h >> 1;
    TRC(0)
    if ((h & 1) == grp) {
      // compute chunk c; the prefetch (chunks past the end read zeros) rides behind the MFMAs
      __builtin_amdgcn_s_setprio(CS_COMPUTE_PRIO);
      const int qw_ = 2 * (c + 1), lw_ = chunks8 - qw_;
      const __amdgpu_buffer_rsrc_t rwd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<u32x4*>(ws + (long long)qw_ * wunits8), 0,
          lw_ > 0 ? (unsigned)((lw_ < 2 ? lw_ : 2) * wunits8) * 16u : 0u, 0x00020000);
      const int cx_ = 16 * (c + 1 + grp), lx_ = k.Cin - cx_;
      const __amdgpu_buffer_rsrc_t rxd = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(xn + (long long)cx_ * HWi), 0,
          lx_ > 0 ? (unsigned)((lx_ < 16 ? lx_ : 16) * HWi) * 4u : 0u, 0x00020000);
      if constexpr (RR) cs_mma_chunk_rr<CPG, XP, NW, NS>(Wg[grp], Xs[c & 1], abase, xb, acc, rwd, wb, rw, rxd, gvo, rx);
      else if constexpr (CPG == 64) cs_mma_chunk(Wg[grp], Xs[c & 1], abase, pb, acc, rwd, wb, rw, rxd, gvo, rx);
    } else {
      // store what this group prefetched during its last compute half-step (B at h = 0: the prologue's)
      __builtin_amdgcn_s_setprio(CS_STORE_PRIO);
      {
#ifdef CS_TRACE
      __builtin_amdgcn_s_waitcnt(0x0f70);                 // vmcnt(0): separates the load wait from the convert + store
      TRC(3)
#endif
#ifdef CS_TRACE
      CS_LSTOREW();
      TRC(4)
      if (c + 1 < chunks) CS_LSTOREX((c + 1) & 1);
#else
      if (grp == 0) {
        if (c + 1 < chunks) { CS_LSTOREW(); CS_LSTOREX((c + 1) & 1); }
      } else {
        CS_LSTOREW();
        if (c + 1 < chunks) CS_LSTOREX((c + 1) & 1);
      }
#endif
      }
    }
    TRC(1)
    // the last half-step is group 1's compute of the last chunk: group 0 has nothing left to stage and goes straight to
    // its epilogue (its 64 KB of stores leave beside group 1's MFMAs instead of after them: a CU stores at ~12 B/clk)
#ifndef CS_NO_EARLY_EPI
    if (h + 1 < 2 * chunks)
#endif
    __syncthreads();
    TRC(2)
  }
  WGT(2)
#ifdef CS_TRACE
  if (trace_blk) {
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * 32 * 8; i += 512) g_cs_trace[i] = trc[i];
    if (threadIdx.x == 0) {
      const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
      g_cs_trace[2048] = (unsigned)trc_t0; g_cs_trace[2049] = (unsigned)t1;
      g_cs_trace[2050] = (unsigned)trc_r0; g_cs_trace[2051] = (unsigned)r1;
    }
  }
#endif
#undef CS_GLOADW
#undef CS_GLOADX
#undef CS_LSTOREW
#undef CS_LSTOREX
#undef CS_NORM_PROBE_OPS

  // Epilogue.  The MFMAs ran with rows = the 32 pixels of a tile row and columns = 32 output channels, so a lane holds
  // ONE output channel (l31) and, per accumulator quad q, the 4 consecutive pixels 8q + 4 lhi .. + 3 of each of its 4 tile
  // rows: 16 16-byte stores per lane instead of 64 4-byte ones (the store tail of a workgroup is bound by the number of
  // store instructions: 8.9 us of an 84-us workgroup with dword stores).  vec4 needs Wo % 4 == 0 and 16-byte aligned bases.
  float* yb = y + (long long)n * k.Cout * HWo;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int i = CPG == 32 ? 0 : (RR ? (wid & 1) : (b >> 1));   // 32-cout block of the group's couts
    const int row = RR ? 4 * rowgrp + b : 2 * wid + (b & 1);
    const int oy = oy0 + row;
    const int cc = CPG * grp + i * 32 + l31, co = m0 + cc;
#ifdef CS_KO_EPI
    if (oy >= k.Ho || co >= k.Cout || acc[b][0] != 12345.678f) continue;    // knock-out: no epilogue loads / stores
#else
    if (oy >= k.Ho || co >= k.Cout) continue;
#endif
    const float bv = bs[cc], osc = oscale * oscale2;
    const long long rowoff = (long long)co * HWo + (long long)oy * k.Wo;
    const float* rb = k.res ? k.res + (long long)n * k.Cout * HWo + rowoff : nullptr;
    const float* rg = k.ring ? k.ring + ((long long)n * 4 * k.Cout + co) * k.ring_rl : nullptr;
    const long long ss = (long long)k.Cout * k.ring_rl;          // strip stride: top, bottom, left, right
    const bool row_ring = rg != nullptr && (oy == 1 || oy == k.Ho - 2);
    float4 rv[4];                                                // residual: all loads in flight before the first use
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      rv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int ox = ox0 + 8 * q + 4 * lhi;
      if (rb && ox < k.Wo) {
        if (k.vec4) rv[q] = *reinterpret_cast<const float4*>(rb + ox);
        else {
          rv[q].x = rb[ox];
          if (ox + 1 < k.Wo) rv[q].y = rb[ox + 1];
          if (ox + 2 < k.Wo) rv[q].z = rb[ox + 2];
          if (ox + 3 < k.Wo) rv[q].w = rb[ox + 3];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ox = ox0 + 8 * q + 4 * lhi;
      if (ox >= k.Wo) continue;
      float v[4];
      const float r4[4] = {rv[q].x, rv[q].y, rv[q].z, rv[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = acc[b][4 * q + e] * osc + bv;
        if (k.act == 1) t = t > 0.f ? t : t * k.slope;
        else if (k.act == 2) t = tanhf(t);
        v[e] = t + r4[e];
      }
      if (rg) {                                                  // the reflection folds frame positions onto ring pixels
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int xx = ox + e;
          if (xx < k.Wo) {
            if (row_ring) {
              if (oy == 1) v[e] += rg[xx + 1] + (xx == 1 ? rg[0] : 0.f) + (xx == k.Wo - 2 ? rg[k.Wo + 1] : 0.f);
              if (oy == k.Ho - 2) v[e] += rg[ss + xx + 1] + (xx == 1 ? rg[ss] : 0.f) + (xx == k.Wo - 2 ? rg[ss + k.Wo + 1] : 0.f);
            }
            if (xx == 1) v[e] += rg[2 * ss + oy + 1];
            if (xx == k.Wo - 2) v[e] += rg[3 * ss + oy + 1];
          }
        }
      }
      float* yp = yb + rowoff + ox;
#ifdef CS_NT_STORE
      if (k.vec4) __builtin_nontemporal_store(f32x4v{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4v*>(yp));
#else
      if (k.vec4) *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
#endif
      else {
        yp[0] = v[0];
        if (ox + 1 < k.Wo) yp[1] = v[1];
        if (ox + 2 < k.Wo) yp[2] = v[2];
        if (ox + 3 < k.Wo) yp[3] = v[3];
      }
    }
  }
#ifdef CS_TRACE
  __builtin_amdgcn_s_waitcnt(0x0f70);       // the stores have left the wave
#endif
  WGT(3)
}

// (The one-wave-per-SIMD experiment of round 4, conv3x3_split_w1_k, lives in scripts/ubench/conv3x3_split_w1.inc: a lab
// build with -DDFMIR_BUILD_W1 includes it and DFMIR_CONV_W1=1 selects it; the product library does not carry it.)
#ifdef DFMIR_BUILD_W1
#include "../../scripts/ubench/conv3x3_split_w1.inc"
#endif

// ---------------------------------------------------------------------------------------------
// res != NULL (y = act(conv + bias) + res): only the shared-tile kernel has that epilogue -- df_conv3x3_split_res_ok(g)
// tells whether this geometry takes it; otherwise *rc is an error
static bool cs_plan(const DfConvGeom* g, int* th_out, int* tx_out, int* ty_out) {
  static DfOptFlag nocs_o{"DFMIR_CONV_NO_CS"};
  const bool use_cs = !nocs_o.get();
  static DfOptFlag plain_o{"DFMIR_CONV_CS_PLAIN"};
  const bool rr = !plain_o.get();
  if (!(df_split_mode() == 2 && use_cs && (g->Cout > 64 || rr))) return false;
  const int th = g->Cout > 64 ? 8 : 16;
  const int tx = (g->Wo + CS_TW - 1) / CS_TW, ty = (g->Ho + th - 1) / th;
  const double fill = (double)g->Ho * g->Wo / ((double)tx * CS_TW * ty * th);
  if (!((long long)g->N * tx * ty < (1LL << 31) && fill >= 0.85)) return false;
  *th_out = th; *tx_out = tx; *ty_out = ty;
  return true;
}
static bool split_fwd_geom_ok(const DfConvGeom* g) {
  if (!(g->KD == 1 && g->KH == 3 && g->KW == 3 && g->Di == 1 && g->Do == 1 && g->stride == 1 && g->dil == 1))
    return false;
  if (g->Cout <= 32 || g->Cin < 16 || g->ph != g->pw || g->pd != 0) return false;
  const int p = g->ph;
  if (!(p == 1 || (p == 2 && g->pad_mode == 0))) return false;
  if (g->Ho != g->Hi + 2 * p - 2 || g->Wo != g->Wi + 2 * p - 2) return false;
  const long long HWo = (long long)g->Ho * g->Wo;
  if (HWo >= (1LL << 30) || (long long)g->Hi * g->Wi >= (1LL << 30)) return false;
  if ((long long)((g->Cin + 7) / 8) * 27 * g->Cout * 16 >= (1LL << 31)) return false;
  if (worst_npos(g->Wo, (int)HWo, 128) > 399) return false;
  return true;
}
int df_conv3x3_split_res_ok(const DfConvGeom* g) {
  int th, tx, ty;
  return (df_split_mode() == 2 && split_fwd_geom_ok(g) && cs_plan(g, &th, &tx, &ty)) ? 1 : 0;
}
bool df_conv3x3_split_fwd_try(const DfConvGeom* g, const float* x, const float* x_amax, int x_n, const float* w_packed,
                              const float* bias, const float* res, const float* ring, int ring_rl, float* y,
                              hipStream_t st, int* rc) {
  const int mode = df_split_mode();
  if (mode == 0 || (mode == 2 && !(x_amax && x_n > 0))) return false;
  if (!split_fwd_geom_ok(g)) return false;
  const int p = g->ph;
  const long long HWo = (long long)g->Ho * g->Wo;
  Conv3P k{g->N, g->Cin, g->Cout, g->Hi, g->Wi, g->Ho, g->Wo, p, g->pad_mode, g->act, g->slope, 0};
  const u32x4* ws = reinterpret_cast<const u32x4*>(split_section(w_packed, g->Cin, g->Cout));
  const SplitScale sc{x_amax, x_n, split_trailer(w_packed, g->Cin, g->Cout, mode)};
  static DfOptFlag plain_o{"DFMIR_CONV_CS_PLAIN"};
  const bool rr = !plain_o.get();
  int th = 0, tlx = 0, tly = 0;
  if (mode == 2 && cs_plan(g, &th, &tlx, &tly)) {
    // 8 x 32 tiles (128 couts per workgroup) or 16 x 32 tiles (64 couts).  They fit the forward shapes exactly but
    // waste 37 % on the 66 x 66 padded frames the dgrad of a reflect-padded conv produces (those go through the
    // zero-padded form + ring kernel instead; what still arrives here unfilled stays on the flat-run kernel below)
    ConvCsP kc{g->N, g->Cin, g->Cout, g->Hi, g->Wi, g->Ho, g->Wo, p, g->pad_mode, g->act, g->slope, tlx, tly, res,
               ring, ring_rl, 0, 0, 0};
    kc.vec4 = (g->Wo % 4 == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)res & 15) == 0) ? 1 : 0;
    static DfOptInt dephase_o{"DFMIR_CS_DEPHASE", 0};
    const long long nb = (long long)g->N * tlx * tly;
    if (nb * ((g->Cout + 127) / 128) >= 512) kc.dephase = dephase_o.get();
    if (g->Cout > 64) {
      dim3 grid((unsigned)nb, (unsigned)((g->Cout + 127) / 128));
      // the two cout halves of a pixel tile on the same XCD (one L2): measured 83.5 -> 83.0 ms per 2-D step, issued
      // fraction 0.432 -> 0.438 (DFMIR_CS_XCD_PAIR=0 restores the 2-D grid; =2: contiguous tile runs per XCD)
      // =2 (default): every XCD walks a contiguous run of tiles (neighbouring tiles share their halo rows in that L2 too):
      // 83.8 -> 82.3 ms, issued 0.432 -> 0.447
      static DfOptInt xcd_o{"DFMIR_CS_XCD_PAIR", 2};
  const int xcd_pair = xcd_o.get();
      if (xcd_pair && grid.y <= 2 && (nb & 7) == 0) {
        kc.xcd_pair = xcd_pair == 1 ? (grid.y == 2 ? 1 : 0) : (int)grid.y + 1;
        if (kc.xcd_pair) grid = dim3((unsigned)(grid.y * nb), 1u);
      }
      // (A persistent form -- one workgroup per CU walking its tiles in one chunk stream, a group's epilogue beside the other
      // group's compute half-step -- was built and measured in round 4: bit-identical, 16 % SLOWER (0.404 vs 0.348 ms at
      // n = 32).  The epilogue is a 33 MB burst of stores issued by all 256 CUs in lockstep, i.e. HBM-write-bound wherever
      // it is placed, and the extra state cost the main loop its registers)
#ifdef DFMIR_BUILD_W1
      static DfOptFlag w1_o{"DFMIR_CONV_W1"};
      if (w1_o.get() && rr && kc.vec4 && g->act != 2 && (g->Cin % 16) == 0 && (g->Ho % 16) == 0 && (g->Wo % 32) == 0) {
        // experiment: one wave per SIMD, 16 x 32 x 128 tiles (conv3x3_split_w1_k)
        ConvCsP kw = kc;
        kw.tiles_y = g->Ho / 16;
        const long long nbw = (long long)g->N * kw.tiles_x * kw.tiles_y;
        const unsigned gyw = (unsigned)((g->Cout + 127) / 128);
        kw.xcd_pair = 0;
        dim3 gw((unsigned)nbw, gyw);
        if ((nbw & 7) == 0 && gyw <= 2) { kw.xcd_pair = (int)gyw + 1; gw = dim3((unsigned)(gyw * nbw), 1u); }
        conv3x3_split_w1_k<0><<<gw, 256, 0, st>>>(x, ws, bias, y, kw, sc);
      } else
#endif
      if (rr) conv3x3_split_cs_k<true, 64, 8><<<grid, 512, 0, st>>>(x, ws, bias, y, kc, sc);
      else conv3x3_split_cs_k<false, 64, 8><<<grid, 512, 0, st>>>(x, ws, bias, y, kc, sc);
    } else {
      static DfOptInt xcd1_o{"DFMIR_CS_XCD_PAIR", 2};
    const int xcd1 = xcd1_o.get();
      if (xcd1 >= 2 && (nb & 7) == 0) kc.xcd_pair = 2;          // one cout slice: contiguous tile runs per XCD
      conv3x3_split_cs_k<true, 32, 16><<<dim3((unsigned)nb, 1u), 512, 0, st>>>(x, ws, bias, y, kc, sc);
    }
    hipError_t e = hipGetLastError();
    *rc = (e == hipSuccess) ? 0 : df_set_error((int)e, __FILE__, __LINE__);
    return true;
  }
  if (res || ring) { *rc = df_set_error((int)hipErrorInvalidValue, __FILE__, __LINE__); return true; }
  k.tiles_per_img = (int)((HWo + 255) / 256);
  const bool big = g->Cout > 64;
  dim3 grid((unsigned)(g->N * k.tiles_per_img), big ? (unsigned)((g->Cout + 127) / 128) : 1u);
  if (mode == 2) {
    if (big) conv3x3_split_pp_k<2, 400, 128><<<grid, 512, 0, st>>>(x, ws, bias, y, k, sc);
    else conv3x3_split_pp_k<2, 400, 64><<<grid, 512, 0, st>>>(x, ws, bias, y, k, sc);
  } else {
    if (big) conv3x3_split_pp_k<3, 400, 128><<<grid, 512, 0, st>>>(x, ws, bias, y, k, sc);
    else conv3x3_split_pp_k<3, 400, 64><<<grid, 512, 0, st>>>(x, ws, bias, y, k, sc);
  }
  hipError_t e = hipGetLastError();
  *rc = (e == hipSuccess) ? 0 : df_set_error((int)e, __FILE__, __LINE__);
  return true;
}

// =================================================================================================
// Weight gradient on the 16-bit pipe:   dWt[tap][ci][co] += sum_{n,p} X[n][ci][p + tap] * dY[n][co][p]
//
// The reduction index is the PIXEL, so an MFMA lane must hold 8 consecutive pixels of one channel (16 B
// of halves).  dY needs no shift; X is needed at the 9 tap shifts, and a +-1 pixel shift of a 16-B unit is
// a 2-byte misalignment that ds_read_b128 / ds_write_b128 only serve at ~1/4 rate
// (scripts/ubench/lds_unaligned.hip).  The shift is therefore applied once, in registers, when the patch
// is converted: a thread loads its 8 pixels plus the pixel on either side, splits the 10 values, and stores
// three aligned units per split -- patch columns [-1..6], [0..7], [1..8] of its group -- so that all nine
// taps read aligned units (dy is a whole row = a different LDS row).  The neighbour pixels of the edge
// groups are the conv padding (reflected or zero), so there is no separate halo path.
//
// Workgroup = 512 threads = 8 waves = (2 tiles of 32 input channels) x (4 tiles of 32 output channels),
// each wave holding the 9 tap accumulators [32 ci x 32 co].  One "run" = 2 image rows x 16 pixels
// (2 MFMA K-steps, patch = 4 rows x 18 columns); global loads of the next run are in flight during the
// MFMA phase, conversion + LDS stores follow it.  The products of a tap chain on one accumulator
// (a dependent MFMA chain issues at full rate, scripts/ubench/mfma_peak.hip), so only one tap's operands
// are live and the next tap's are read from LDS meanwhile.  Runs are split over blockIdx.x (split-K, fp32
// atomics at the end, as conv3x3_wgrad_k).
//   LDS  Xc[split][dx][row 0..3][half][ci 64], Dy[split][kstep][half][co 128] : 16-B units of 8 pixels
struct WS3P {
  int N, Cin, Cout, H, W, pad_mode;
  int runs_per_row, runs_per_img, runs_total, runs_per_block;
  const float* x_amax;    // NSP = 2: partial maxima of |x| (x_n of them) and |dy| (dy_n)
  const float* dy_amax;
  int x_n, dy_n;
  float* db;              // optional: bias gradient db[co] += sum_{n,p} dY, taken from the dY units as they pass
  const float* dy_pmax;   // optional (split2 kernel): max |dY| per (n, co) plane, [N][Cout] -> one scale per output channel
  // split2 kernel with the operands' ROLES SWAPPED (x := dY, dy := X; zero padding only): dW[t][ci][co] =
  // sum_q dY[co][q] X[ci][q + t - 1] is what the kernel computes for tap 8 - t with rows = co and columns = ci, so the
  // epilogue stores tap 8 - t transposed.  Used for 64 output channels under > 64 input channels (the 128 -> 64 decoder
  // layer at 256^2): its natural tile is 64 x 128 (ci x co), which a 64-channel dY half fills; the other way round the
  // layer fills it completely.
  int swap;
  float* dbx;             // swapped roles: the bias gradient is the pixel sum of the kernel's X operand (= dY), interior rows of a run
  const float* fx;        // deterministic mode (common.h df_acc): dwt is then an array of 64-bit fixed-point sums; db / dbx are NULL
};

// BC = 128: 8 waves = 2 ci tiles x 4 co tiles, each wave both k-steps of a run.
// BC = 64 : 8 waves = 2 ci tiles x 2 co tiles x 2 k-steps (the two k-step waves of a tile both add their
//           partial sums atomically, like the split-K workgroups do).
template <int NSP, int BC>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_split_k(const float* __restrict__ x,
                                                                const float* __restrict__ dy,
                                                                float* __restrict__ dwt, WS3P k) {
  using P = Prod<NSP>;
  constexpr int CT = 64, NWC = BC / 32, KS = 4 / NWC, NSTEP = 18 / KS;
  constexpr int XSLAB = 2 * CT;                      // units of one (split, dx, row) slab: [half][ci]
  __shared__ __attribute__((aligned(16))) u32x4 Xc[NSP * 3 * 4 * XSLAB];
  __shared__ __attribute__((aligned(16))) u32x4 Dy[NSP * 2 * 2 * BC];
  __shared__ float bsum[BC];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool want_db = k.db != nullptr && blockIdx.y == 0;   // one ci-tile column of workgroups sees every dY once
  if (tid < BC) bsum[tid] = 0.f;
  float bacc = 0.f;
  const int wc = wid % NWC, wi = (wid / NWC) & 1, kh = wid / (2 * NWC);   // kh: this wave's k-step when KS == 2
  const int l31 = lane & 31, lhi = lane >> 5;
  const int HW = k.H * k.W;
  const int ci0 = blockIdx.y * CT, co0 = blockIdx.z * BC;
  const int run_beg = blockIdx.x * k.runs_per_block;
  int run_end = run_beg + k.runs_per_block;
  if (run_end > k.runs_total) run_end = k.runs_total;

  float xscale = 1.f, dscale = 1.f, oscale = 1.f, oscale2 = 1.f;
  if (NSP == 2) {
    __shared__ float red[17];
    const int ex = scale_exp(reduce_absmax(k.x_amax, k.x_n, red));
    __syncthreads();
    const int ed = scale_exp(reduce_absmax(k.dy_amax, k.dy_n, red));
    xscale = pow2f(ex); dscale = pow2f(ed); oscale = pow2f(-ex); oscale2 = pow2f(-ed);
  }

  // loader roles (512 threads): X group (patch row xr 0..3, half xu 0..1, channel xc 0..63) and
  // dY group (k-step dk, half du, channel dc 0..127)
  const int xc = tid & 63, xu = (tid >> 6) & 1, xr = tid >> 7;
  const int dc = tid % BC, du = (tid / BC) & 1, dk = (tid / (2 * BC)) & 1;
  const bool dload = tid < 4 * BC;                   // BC = 64: 256 dY groups for 512 threads
  const unsigned hw4 = (unsigned)HW * 4u;
  constexpr unsigned OOB = 0x80000000u;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  u32x4 rxa, rxb, rda, rdb;   // 8 px of X, 8 px of dY
  unsigned rxl, rxr;          // the pixel left / right of the X group

#define WS_GLOAD(run_)                                                                           \
  {                                                                                              \
    const int n_ = (run_) / k.runs_per_img;                                                      \
    const int q_ = (run_) - n_ * k.runs_per_img;                                                 \
    const int yp_ = q_ / k.runs_per_row, xs_ = q_ - yp_ * k.runs_per_row;                        \
    const int y0_ = 2 * yp_, x0_ = 16 * xs_ + 8 * xu;                                            \
    const __amdgpu_buffer_rsrc_t bx_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(x + (long long)n_ * k.Cin * HW), 0, (unsigned)(k.Cin * HW) * 4u, 0x00020000); \
    const __amdgpu_buffer_rsrc_t bd_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(dy + (long long)n_ * k.Cout * HW), 0, (unsigned)(k.Cout * HW) * 4u, 0x00020000); \
    const bool cok_ = ci0 + xc < k.Cin;                                                          \
    const unsigned cb_ = (unsigned)(ci0 + xc) * hw4;                                             \
    /* patch row -> image row (reflected, or outside = zeros); the group itself is always inside the row, */ \
    /* only its left / right neighbour can be padding */                                         \
    int ry_ = y0_ - 1 + xr;                                                                      \
    bool rok_ = (unsigned)ry_ < (unsigned)k.H;                                                   \
    if (k.pad_mode == 1) { ry_ = ry_ < 0 ? -ry_ : (ry_ >= k.H ? 2 * (k.H - 1) - ry_ : ry_); rok_ = true; } \
    const int rb_ = ry_ * k.W;                                                                   \
    const bool lin_ = x0_ > 0, rin_ = x0_ + 8 < k.W;                                             \
    const int ol_ = rb_ + (lin_ ? x0_ - 1 : 1), or_ = rb_ + (rin_ ? x0_ + 8 : k.W - 2);          \
    const bool lok_ = rok_ && cok_ && (lin_ || k.pad_mode == 1), rrok_ = rok_ && cok_ && (rin_ || k.pad_mode == 1); \
    const unsigned xb_ = (rok_ && cok_) ? cb_ + (unsigned)(rb_ + x0_) * 4u : OOB;                \
    rxa = __builtin_amdgcn_raw_buffer_load_b128(bx_, xb_, 0, 0);                                 \
    rxb = __builtin_amdgcn_raw_buffer_load_b128(bx_, xb_ == OOB ? OOB : xb_ + 16u, 0, 0);        \
    rxl = __builtin_amdgcn_raw_buffer_load_b32(bx_, lok_ ? cb_ + (unsigned)ol_ * 4u : OOB, 0, 0); \
    rxr = __builtin_amdgcn_raw_buffer_load_b32(bx_, rrok_ ? cb_ + (unsigned)or_ * 4u : OOB, 0, 0); \
    const unsigned db_ = (co0 + dc >= k.Cout || !dload) ? OOB                                    \
        : (unsigned)(co0 + dc) * hw4 + (unsigned)((y0_ + dk) * k.W + 16 * xs_ + 8 * du) * 4u;    \
    rda = __builtin_amdgcn_raw_buffer_load_b128(bd_, db_, 0, 0);                                 \
    rdb = __builtin_amdgcn_raw_buffer_load_b128(bd_, db_ == OOB ? OOB : db_ + 16u, 0, 0);        \
  }
  // X: r[0] = left neighbour, r[1..8] = the group, r[9] = right neighbour.  Pairs (0,1)(2,3)(4,5)(6,7)(8,9)
  // make the units dx=0 (cols -1..6) and dx=2 (cols 1..8); pairs (1,2)..(7,8) make dx=1.  The residual of a
  // pixel does not depend on which pair it was rounded in, so the second pairing only costs its conversions.
#define WS_LSTORE()                                                                              \
  {                                                                                              \
    float r[10];                                                                                 \
    r[0] = __uint_as_float(rxl); r[9] = __uint_as_float(rxr);                                    \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) { r[1 + e] = __uint_as_float(rxa[e]); r[5 + e] = __uint_as_float(rxb[e]); } \
    unsigned pa[5][NSP], pb[4][NSP];                                                             \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) split_pair_s<NSP>(r[2 * i], r[2 * i + 1], xscale, pa[i]); \
    if (NSP == 2) {            /* the odd pairing is the even one shifted by a half: v_alignbit */ \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
        _Pragma("unroll") for (int s = 0; s < NSP; ++s) pb[i][s] = __builtin_amdgcn_alignbit(pa[i + 1][s], pa[i][s], 16); \
    } else {                                                                                     \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) split_pair<NSP>(r[2 * i + 1], r[2 * i + 2], pb[i]); \
    }                                                                                            \
    _Pragma("unroll") for (int s = 0; s < NSP; ++s) {                                            \
      u32x4* dst = Xc + ((s * 3 * 4 + xr) * 2 + xu) * CT + xc;                                   \
      dst[0] = u32x4{pa[0][s], pa[1][s], pa[2][s], pa[3][s]};                                    \
      dst[4 * XSLAB] = u32x4{pb[0][s], pb[1][s], pb[2][s], pb[3][s]};                            \
      dst[8 * XSLAB] = u32x4{pa[1][s], pa[2][s], pa[3][s], pa[4][s]};                            \
    }                                                                                            \
    float v[8];                                                                                  \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(rda[e]); v[4 + e] = __uint_as_float(rdb[e]); } \
    bacc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));   /* this thread's dY channel is fixed */ \
    u32x4 sp[NSP];                                                                               \
    split8_s<NSP>(v, dscale, sp);                                                                \
    if (dload) { _Pragma("unroll") for (int s = 0; s < NSP; ++s) Dy[((s * 2 + dk) * 2 + du) * BC + dc] = sp[s]; } \
  }

  if (run_beg < run_end) {
    WS_GLOAD(run_beg);
    WS_LSTORE();
  }
  __syncthreads();

  // operand unit indices of this lane: A = Xc[((s*3 + dx)*4 + row)*2 + lhi][ci], B = Dy[(s*2 + ks)*2 + lhi][co]
  const int abase = lhi * CT + wi * 32 + l31 + (KS == 2 ? kh * XSLAB : 0);
  const int bbase = lhi * BC + wc * 32 + l31 + (KS == 2 ? kh * 2 * BC : 0);
  for (int run = run_beg; run < run_end; ++run) {
    const bool more = (run + 1) < run_end;
    if (more) WS_GLOAD(run + 1);
    u32x4 b[NSP], a[2][NSP];
#define WS_LOADA(set_, step_)                                                                    \
  _Pragma("unroll") for (int s = 0; s < NSP; ++s)                                                \
    a[set_][s] = Xc[((s * 3 + ((step_) % 3)) * 4 + (step_) / 9 + ((step_) % 9) / 3) * XSLAB + abase];
    // NSTEP steps = (k-step ks, tap): step = ks*9 + ty*3 + dx (KS == 2: this wave's k-step sits in abase /
    // bbase and step = tap); operands of step+1 are read during step
    WS_LOADA(0, 0)
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      if (step % 9 == 0) {
#pragma unroll
        for (int s = 0; s < NSP; ++s) b[s] = Dy[(s * 2 + step / 9) * 2 * BC + bbase];
      }
      if (step + 1 < NSTEP) WS_LOADA((step + 1) & 1, step + 1)
#pragma unroll
      for (int q = 0; q < P::N; ++q)
        acc[step % 9] = mma16<NSP>(a[step & 1][P::A[q]], b[P::B[q]], acc[step % 9]);
    }
#undef WS_LOADA
#pragma unroll
    for (int i = 0; i < 2 * NSP; ++i) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      const int nr = (step + 1 < NSTEP ? NSP : 0) + ((step + 1) % 9 == 0 && step + 1 < NSTEP ? NSP : 0);
#pragma unroll
      for (int i = 0; i < P::N; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < nr) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i == P::N - 1) {
#pragma unroll
          for (int e = P::N; e < nr; ++e) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
    if (more) {
      __syncthreads();
      WS_LSTORE();
      __syncthreads();
    }
  }
#undef WS_GLOAD
#undef WS_LSTORE

  if (want_db) {
    if (dload) atomicAdd(&bsum[dc], bacc);
    __syncthreads();
    if (tid < BC && co0 + tid < k.Cout) atomicAdd(&k.db[co0 + tid], bsum[tid]);
  }
  const int co = co0 + wc * 32 + l31;
  if (co < k.Cout) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wi * 32 + 4 * lhi + (r & 3) + 8 * (r >> 2);
        if (ci < k.Cin)
          df_acc(dwt, ((long long)t * k.Cin + ci) * k.Cout + co, NSP == 2 ? acc[t][r] * oscale * oscale2 : acc[t][r], k.fx);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Input gradient of conv3x3(reflection_pad1(x)): the ring.
// dX = fold(G), G = the full correlation of dY with the flipped weights on the (H+2) x (W+2) padded frame.  The
// interior of G folds onto itself: that part is the ordinary zero-padded dgrad ("same" size), which the shared-tile
// kernel computes at 100 % tile fill (on the 66 x 66 frame it fills 63 %, and the flat-run kernel used instead is
// 20 % slower per pixel).  What remains is the one-pixel ring of G -- frame rows 0 and H+1, columns 0 and W+1 --
// which reflection folds onto rows 1 / H-2 and columns 1 / W-2.  On the ring only one tap row (or column) meets
// non-zero dY, so each of the four strips is a 3-tap 1-D convolution of ONE line of dY:
//     G[0][c]   = sum_j Wd[(2, j)] dY[row 0  ][c + j - 2]      G[r][0]   = sum_j Wd[(j, 2)] dY[col 0  ][r + j - 2]
//     G[H+1][c] = sum_j Wd[(0, j)] dY[row H-1][c + j - 2]      G[r][W+1] = sum_j Wd[(j, 0)] dY[col W-1][r + j - 2]
// (Wd = dgrad packing, taps already flipped; c over 0..W+1, r over 1..H: the corners belong to the row strips.)
// 2 % of the interior's FLOPs.  Workgroup = (strip, image, 128 produced channels): the line [K][L] is split into
// LDS once, 8 waves = 4 channel blocks x 2 K-halves, weights straight from the packed split section (L2-resident),
// results to a compact buffer ring[n][strip][m][RL]; the interior kernel's epilogue adds ring[.][c] to the pixels the
// reflection folds frame position c onto (rows 1 / H-2, columns 1 / W-2; the frame's corners through the row strips).
struct RingP {
  int N, K, M, H, W;          // K = reduction channels (the conv's Cout), M = produced channels (its Cin)
  int RL;                     // row length of the ring buffer (>= max(H, W) + 2)
  const float* cols;          // optional [N*K][2][H]: first / last column of dY, left by its producer (instnorm_bwd)
};
constexpr int RING_KG = 32;                          // 8-channel groups of the reduction held in LDS (K <= 256)

// RP = line positions in LDS: 2 zeros + the line + zeros (72: lines up to 64 -> two workgroups per CU; 100: up to 94)
template <int RP>
__global__ __launch_bounds__(512) void conv3x3_reflect_ring_k(const float* __restrict__ dy, const u32x4* __restrict__ ws,
                                                              float* __restrict__ ring, RingP k, SplitScale sc) {
  constexpr int NSP = 2;
  using P = Prod<2>;
  __shared__ __attribute__((aligned(16))) u32x4 Ls[NSP * RING_KG * RP];
  __shared__ float red[17];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int l31 = lane & 31, lhi = lane >> 5;
  const int strip = blockIdx.x & 3, n = blockIdx.x >> 2;
  const int m0 = blockIdx.y * 128;
  const bool rowstrip = strip < 2;
  const int L = rowstrip ? k.W : k.H;                    // line length
  const int HW = k.H * k.W;
  // the line of dY: element i of channel kk at dyl[kk * HW + i * istr]
  const float* dyl = dy + (long long)n * k.K * HW +
                     (strip == 0 ? 0 : strip == 1 ? (k.H - 1) * k.W : strip == 2 ? 0 : k.W - 1);
  const int istr = rowstrip ? 1 : k.W;

  const int ex = scale_exp(reduce_absmax(sc.x_amax, sc.x_n, red));
  const int ew = (int)sc.w_trailer[1];
  const float xscale = pow2f(ex), oscale = pow2f(-ex), oscale2 = pow2f(-ew);

  const int kg_n = (k.K + 7) >> 3;                       // 8-channel groups of the reduction
  // zero padding positions (0, 1 and L+2 ..) of every group; groups past K entirely
  {
    const int npad = RP - L;
    for (int it = tid; it < NSP * RING_KG * npad; it += 512) {
      const int q = it % npad, g = it / npad;
      Ls[g * RP + (q < 2 ? q : L + q)] = u32x4{0u, 0u, 0u, 0u};
    }
    const int kz = RING_KG - kg_n;
    for (int it = tid; it < NSP * kz * L; it += 512) {
      const int i = it % L, g = it / L, s2 = g / kz, kg = kg_n + g % kz;
      Ls[(s2 * RING_KG + kg) * RP + i + 2] = u32x4{0u, 0u, 0u, 0u};
    }
  }
  for (int it = tid; it < kg_n * L; it += 512) {
    const int i = it % L, kg = it / L;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kk = kg * 8 + e;
      if (!rowstrip && k.cols) v[e] = kk < k.K ? k.cols[(((long long)n * k.K + kk) * 2 + (strip - 2)) * k.H + i] : 0.f;
      else v[e] = kk < k.K ? dyl[(long long)kk * HW + (long long)i * istr] : 0.f;
    }
    u32x4 sp[NSP];
    split8_s<NSP>(v, xscale, sp);
#pragma unroll
    for (int s = 0; s < NSP; ++s) Ls[(s * RING_KG + kg) * RP + i + 2] = sp[s];
  }
  __syncthreads();

  const int mb = wid & 3, kh = wid >> 2;
  const int m = m0 + mb * 32 + l31;                      // this lane's weight column
  const int nb_n = (L + 2 + 31) >> 5;                    // 32-pixel blocks of the strip (<= 3)
  f32x16 acc[3];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  const int chunks = (k.K + 15) >> 4;
  const int tap0 = strip == 0 ? 6 : strip == 1 ? 0 : strip == 2 ? 2 : 0, tapd = rowstrip ? 1 : 3;
  // weights of a chunk: 3 taps x 2 split terms, straight from the packed split section; two chunks ahead (a chunk's
  // 27 MFMAs are shorter than an L2 round trip)
  u32x4 a[3][NSP], an[3][NSP], an2[3][NSP];
#define RING_LOADA(dst_, c_)                                                                     \
  {                                                                                              \
    const int kg_ = 2 * (c_) + lhi;                                                              \
    const bool ok_ = (c_) < chunks && kg_ < kg_n && m < k.M;                                     \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                \
      _Pragma("unroll") for (int s = 0; s < NSP; ++s)                                            \
        dst_[j][s] = ok_ ? ws[((long long)kg_ * NSP * 9 + s * 9 + tap0 + j * tapd) * k.M + m] : u32x4{0u, 0u, 0u, 0u}; \
  }
  RING_LOADA(a, kh)
  RING_LOADA(an, kh + 2)
  for (int c = kh; c < chunks; c += 2) {
    RING_LOADA(an2, c + 4)
    const int kg = 2 * c + lhi;                          // this half-wave's 8-channel group (< RING_KG)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        if (b < nb_n) {
          int pos = b * 32 + l31 + j;
          pos = pos < RP ? pos : RP - 1;                 // lanes past the strip: any zero unit
          u32x4 x2[NSP];
#pragma unroll
          for (int s = 0; s < NSP; ++s) x2[s] = Ls[(s * RING_KG + kg) * RP + pos];
#pragma unroll
          for (int q = 0; q < P::N; ++q) acc[b] = mma16<NSP>(a[j][P::A[q]], x2[P::B[q]], acc[b]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int s = 0; s < NSP; ++s) { a[j][s] = an[j][s]; an[j][s] = an2[j][s]; }
  }
#undef RING_LOADA

  // the two K halves meet in LDS (the line is no longer needed); the strip goes to the compact ring buffer
  // ring[n][strip][m][RL] (index = frame coordinate along the strip), which the interior kernel's epilogue folds in
  __syncthreads();
  float* xch = reinterpret_cast<float*>(Ls);             // [mb][b][r][64 lanes]
  if (kh == 1) {
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[((mb * 3 + b) * 16 + r) * 64 + lane] = acc[b][r];
  }
  __syncthreads();
  if (kh == 1) return;
  float* rg = ring + ((long long)(n * 4 + strip) * k.M) * k.RL;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    if (b >= nb_n) continue;
    const int c = b * 32 + l31;                          // position along the strip, frame coordinates
    if (c > L + 1) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mm = m0 + mb * 32 + 4 * lhi + (r & 3) + 8 * (r >> 2);
      if (mm < k.M)
        rg[(long long)mm * k.RL + c] = (acc[b][r] + xch[((mb * 3 + b) * 16 + r) * 64 + lane]) * oscale * oscale2;
    }
  }
}

// 1 if the ring kernel takes this (forward) geometry: fp16x2 mode, 3x3 stride 1 reflect pad 1
int df_conv3x3_reflect_ring_ok(const DfConvGeom* g) {
  if (df_split_mode() != 2) return 0;
  if (!(g->KD == 1 && g->KH == 3 && g->KW == 3 && g->Di == 1 && g->Do == 1 && g->stride == 1 && g->dil == 1)) return 0;
  if (g->pad_mode != 1 || g->ph != 1 || g->pw != 1 || g->pd != 0) return 0;
  if (g->Ho != g->Hi || g->Wo != g->Wi || g->Hi < 4 || g->Wi < 4 || g->Hi > 94 || g->Wi > 94) return 0;
  if (g->Cout > 8 * RING_KG || g->Cout < 16 || g->Cin <= 32) return 0;   // K in LDS; M > 32: the split dgrad kernels' range
  return 1;
}
int df_conv3x3_reflect_ring_len(const DfConvGeom* g) { return ((g->Hi > g->Wi ? g->Hi : g->Wi) + 2 + 3) & ~3; }
int df_conv3x3_reflect_ring_launch(const DfConvGeom* g, const float* dy, const float* dy_cols, const float* dy_amax,
                                   int dy_n, const float* wd_packed, float* ring, hipStream_t st) {
  // wd_packed: the dgrad packing (K = Cout reduction, M = Cin produced)
  const RingP k{g->N, g->Cout, g->Cin, g->Hi, g->Wi, df_conv3x3_reflect_ring_len(g), dy_cols};
  const u32x4* ws = reinterpret_cast<const u32x4*>(split_section(wd_packed, g->Cout, g->Cin));
  const SplitScale sc{dy_amax, dy_n, split_trailer(wd_packed, g->Cout, g->Cin, 2)};
  const dim3 grid((unsigned)(4 * g->N), (unsigned)((g->Cin + 127) / 128));
  if (g->Hi <= 64 && g->Wi <= 64) conv3x3_reflect_ring_k<72><<<grid, 512, 0, st>>>(dy, ws, ring, k, sc);
  else conv3x3_reflect_ring_k<100><<<grid, 512, 0, st>>>(dy, ws, ring, k, sc);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// fp16x2, 128 output channels: the same tiling as conv3x3_wgrad_split_k<2,128> with the three things its knock-out
// timings asked for (profiles/r01_conv3x3s_pmc.md: the non-matrix work alone was 70 % of that kernel):
//   * LDS tiles double-buffered, ONE barrier per run: the conversion + LDS stores of run r+1 happen during the MFMA
//     phase of run r, and the two wave groups are staggered -- waves 0-3 convert first and then compute, waves 4-7
//     compute first and then convert -- so each SIMD always has one wave on the matrix pipe (a wave of each group);
//   * an operand unit Xc[dx][row] serves both k-steps (row = ks + ty): 12 distinct units per run instead of 18
//     reads, i.e. 28 ds_read_b128 per 54 MFMAs instead of 40;
//   * operands are read one unit (3 or 6 MFMAs) ahead; two or three units ahead measured the same.
#ifdef W2_TRACE
// trace build (scripts/build_ko.sh conv3x3s W2_TRACE 1): cycle stamps of one workgroup's first 64 runs, per wave:
// [0] iteration start, [1] after its first phase (group 0: convert + store, group 1: MFMAs), [2] after the second, [3] after the barrier,
// [4] in the convert phase once the global loads of the run have arrived (an explicit vmcnt(0))
__device__ unsigned g_w2_trace[8 * 64 * 8];
extern "C" int dfmir_w2_trace_dump(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w2_trace), sizeof(g_w2_trace));
}
#define W2T(slot_) { if (trace_blk && lane == 0 && run - run_beg < 64) g_w2_trace[(wid * 64 + (run - run_beg)) * 8 + (slot_)] = (unsigned)__builtin_readcyclecounter(); }
#define W2T_VM() { __builtin_amdgcn_s_waitcnt(0x0f70); W2T(4) }
#else
#define W2T(slot_)
#define W2T_VM()
#endif
#ifndef W2_ALT
#define W2_ALT 0
#endif
#ifndef W2_PRIO_C
#define W2_PRIO_C 0       // s_setprio of the convert + store phase
#endif
#ifndef W2_PRIO_G0
#define W2_PRIO_G0 1      // s_setprio of the MFMA phase, wave group 0 (converts first, then computes)
#endif
#ifndef W2_PRIO_G1
// ... wave group 1 (computes first, then converts).  ABOVE group 0's: with equal priorities the arbiter favours the older
// waves 0-3 whenever both waves of a SIMD have MFMAs ready, group 1 then finishes its MFMAs last and its conversion
// phase runs with nobody on the matrix pipe (0.440 -> 0.395 ms on 256 -> 256 @64^2, n = 32; profiles/r04_wgrad_prio.txt)
#define W2_PRIO_G1 2
#endif
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_split2_k(const float* __restrict__ x,
                                                                 const float* __restrict__ dy,
                                                                 float* __restrict__ dwt, WS3P k) {
  constexpr int NSP = 2, BC = 128, CT = 64;
  using P = Prod<2>;
#ifndef W2_CTPAD
#define W2_CTPAD 0
#endif
  constexpr int CTP = CT + W2_CTPAD;                 // stride of a half inside a slab
  constexpr int XSLAB = 2 * CTP;                     // units of one (split, dx, row) slab: [half][ci]
  constexpr int XCU = NSP * 3 * 4 * XSLAB, DYU = NSP * 2 * 2 * BC;
  __shared__ __attribute__((aligned(16))) u32x4 Xc[2 * XCU];
  __shared__ __attribute__((aligned(16))) u32x4 Dy[2 * DYU];
  __shared__ float bsum[BC];
  __shared__ float red[17];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool want_db = k.db != nullptr && blockIdx.y == 0;
  if (tid < BC) bsum[tid] = 0.f;
  float bacc = 0.f, baccx = 0.f;
  const int wc = wid & 3, wi = wid >> 2;             // wi is also the stagger group
  const int l31 = lane & 31, lhi = lane >> 5;
  const int HW = k.H * k.W;
  const int ci0 = blockIdx.y * CT, co0 = blockIdx.z * BC;
  const int run_beg = blockIdx.x * k.runs_per_block;
  int run_end = run_beg + k.runs_per_block;
  if (run_end > k.runs_total) run_end = k.runs_total;

  const int ex = scale_exp(reduce_absmax(k.x_amax, k.x_n, red));
  __syncthreads();
  const int ed = scale_exp(reduce_absmax(k.dy_amax, k.dy_n, red));
  // dY is scaled per OUTPUT CHANNEL when the per-plane maxima are known: the scale is uniform along the MFMA K (pixels
  // of one channel), so a channel whose gradient is 1e-6 of the tensor's largest keeps its 22 bits; the column's
  // factor 2^-ed[co] goes into the epilogue.  Without them: one scale for the tensor.
  __shared__ int edc[BC];
  if (tid < BC) {
    int e = ed;
    if (k.dy_pmax) {
      float m = 0.f;
      if (co0 + tid < k.Cout)
        for (int n = 0; n < k.N; ++n) m = fmaxf(m, k.dy_pmax[(long long)n * k.Cout + co0 + tid]);
      e = scale_exp(m);
    }
    edc[tid] = e;
  }
  __syncthreads();

  // loader roles: X group (patch row xr 0..3, half xu, channel xc 0..63), dY group (k-step dk, half du, channel dc)
  const int xc = tid & 63, xu = (tid >> 6) & 1, xr = tid >> 7;
  const int dc = tid & (BC - 1), du = (tid >> 7) & 1, dk = tid >> 8;
  const bool xin = k.dbx != nullptr && blockIdx.z == 0 && (xr == 1 || xr == 2);     // rows of a run that are not halo
  const float xscale = pow2f(ex), dscale = pow2f(edc[dc]), oscale = pow2f(-ex), oscale2 = pow2f(-edc[wc * 32 + l31]);
  const unsigned hw4 = (unsigned)HW * 4u;
  constexpr unsigned OOB = 0x80000000u;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  u32x4 rxa, rxb, rda, rdb;   // 8 px of X, 8 px of dY
  unsigned rxl, rxr;          // the pixel left / right of the X group

  // past the last run of this workgroup the descriptors are empty (loads return zeros, stores are harmless)
#ifndef W2_KO
#define W2_KO 0           // knock-out builds (timing experiments): 1 no global loads, 2 no LDS stores, 4 no conversion, 8 no MFMA phase
#endif
#define W2_GLOAD(run_)                                                                           \
  if (!(W2_KO & 1) || k.N < 0) {                                                                 \
    const bool live_ = (run_) < run_end;                                                         \
    const int n_ = live_ ? (run_) / k.runs_per_img : 0;                                          \
    const int q_ = live_ ? (run_) - n_ * k.runs_per_img : 0;                                     \
    const int yp_ = q_ / k.runs_per_row, xs_ = q_ - yp_ * k.runs_per_row;                        \
    const int y0_ = 2 * yp_, x0_ = 16 * xs_ + 8 * xu;                                            \
    const __amdgpu_buffer_rsrc_t bx_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(x + (long long)n_ * k.Cin * HW), 0, live_ ? (unsigned)(k.Cin * HW) * 4u : 0u, 0x00020000); \
    const __amdgpu_buffer_rsrc_t bd_ = __builtin_amdgcn_make_buffer_rsrc(                        \
        const_cast<float*>(dy + (long long)n_ * k.Cout * HW), 0, live_ ? (unsigned)(k.Cout * HW) * 4u : 0u, 0x00020000); \
    const bool cok_ = ci0 + xc < k.Cin;                                                          \
    const unsigned cb_ = (unsigned)(ci0 + xc) * hw4;                                             \
    int ry_ = y0_ - 1 + xr;                                                                      \
    bool rok_ = (unsigned)ry_ < (unsigned)k.H;                                                   \
    if (k.pad_mode == 1) { ry_ = ry_ < 0 ? -ry_ : (ry_ >= k.H ? 2 * (k.H - 1) - ry_ : ry_); rok_ = true; } \
    const int rb_ = ry_ * k.W;                                                                   \
    const bool lin_ = x0_ > 0, rin_ = x0_ + 8 < k.W;                                             \
    const int ol_ = rb_ + (lin_ ? x0_ - 1 : 1), or_ = rb_ + (rin_ ? x0_ + 8 : k.W - 2);          \
    const bool lok_ = rok_ && cok_ && (lin_ || k.pad_mode == 1), rrok_ = rok_ && cok_ && (rin_ || k.pad_mode == 1); \
    const unsigned xb_ = (rok_ && cok_) ? cb_ + (unsigned)(rb_ + x0_) * 4u : OOB;                \
    rxa = __builtin_amdgcn_raw_buffer_load_b128(bx_, xb_, 0, 0);                                 \
    rxb = __builtin_amdgcn_raw_buffer_load_b128(bx_, xb_ == OOB ? OOB : xb_ + 16u, 0, 0);        \
    rxl = __builtin_amdgcn_raw_buffer_load_b32(bx_, lok_ ? cb_ + (unsigned)ol_ * 4u : OOB, 0, 0); \
    rxr = __builtin_amdgcn_raw_buffer_load_b32(bx_, rrok_ ? cb_ + (unsigned)or_ * 4u : OOB, 0, 0); \
    const unsigned db_ = (co0 + dc >= k.Cout) ? OOB                                              \
        : (unsigned)(co0 + dc) * hw4 + (unsigned)((y0_ + dk) * k.W + 16 * xs_ + 8 * du) * 4u;    \
    rda = __builtin_amdgcn_raw_buffer_load_b128(bd_, db_, 0, 0);                                 \
    rdb = __builtin_amdgcn_raw_buffer_load_b128(bd_, db_ == OOB ? OOB : db_ + 16u, 0, 0);        \
  }
  // W2_NORM_PROBE (lab builds): the cost of an InstanceNorm + ReLU applied while the X operand is converted -- (x - m) * r and a
  // max per value with run-time operands that happen to be the identity (profiles/r06_cs_norm_probe.txt)
#ifdef W2_NORM_PROBE
  const float np_r = fmaf(oscale, 0.f, 1.f), np_m = oscale * 0.f, np_lo = fmaf(oscale, 0.f, -3.0e38f);
#define W2_NORM_PROBE_OPS() _Pragma("unroll") for (int i = 0; i < 10; ++i) r[i] = fmaxf((r[i] - np_m) * np_r, np_lo);
#else
#define W2_NORM_PROBE_OPS()
#endif
  // X: r[0] = left neighbour, r[1..8] = the group, r[9] = right neighbour; pairs (0,1)..(8,9) make the units
  // dx=0 (cols -1..6) and dx=2 (cols 1..8), the odd pairing dx=1 is the even one shifted by a half
#define W2_LSTORE(buf_)                                                                          \
  {                                                                                              \
    float r[10];                                                                                 \
    r[0] = __uint_as_float(rxl); r[9] = __uint_as_float(rxr);                                    \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) { r[1 + e] = __uint_as_float(rxa[e]); r[5 + e] = __uint_as_float(rxb[e]); } \
    W2_NORM_PROBE_OPS()                                                                          \
    baccx += xin ? ((r[1] + r[2]) + (r[3] + r[4])) + ((r[5] + r[6]) + (r[7] + r[8])) : 0.f;      \
    unsigned pa[5][NSP], pb[4][NSP];                                                             \
    if (W2_KO & 4) { _Pragma("unroll") for (int i = 0; i < 5; ++i) { pa[i][0] = __float_as_uint(r[2 * i]); pa[i][1] = __float_as_uint(r[2 * i + 1]); } } \
    else _Pragma("unroll") for (int i = 0; i < 5; ++i) split_pair_scaled(r[2 * i], r[2 * i + 1], xscale, pa[i]); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                \
      _Pragma("unroll") for (int s = 0; s < NSP; ++s) pb[i][s] = __builtin_amdgcn_alignbit(pa[i + 1][s], pa[i][s], 16); \
    if (!(W2_KO & 2) || k.N < 0) _Pragma("unroll") for (int s = 0; s < NSP; ++s) {                 \
      u32x4* dst = Xc + (buf_) * XCU + (s * 3 * 4 + xr) * XSLAB + xu * CTP + xc;                    \
      dst[0] = u32x4{pa[0][s], pa[1][s], pa[2][s], pa[3][s]};                                    \
      dst[4 * XSLAB] = u32x4{pb[0][s], pb[1][s], pb[2][s], pb[3][s]};                            \
      dst[8 * XSLAB] = u32x4{pa[1][s], pa[2][s], pa[3][s], pa[4][s]};                            \
    }                                                                                            \
    float v[8];                                                                                  \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(rda[e]); v[4 + e] = __uint_as_float(rdb[e]); } \
    bacc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));                   \
    u32x4 sp[NSP];                                                                               \
    if (W2_KO & 4) { sp[0] = rda; sp[1] = rdb; } else                                            \
    split8_s<NSP>(v, dscale, sp);                                                                \
    if (!(W2_KO & 2) || k.N < 0) _Pragma("unroll") for (int s = 0; s < NSP; ++s) Dy[(buf_) * DYU + ((s * 2 + dk) * 2 + du) * BC + dc] = sp[s]; \
  }

  // operand unit indices of this lane: A = Xc[((s*3 + dx)*4 + row)*2 + lhi][ci], B = Dy[(s*2 + ks)*2 + lhi][co]
  const int abase = lhi * CTP + wi * 32 + l31;
  const int bbase = lhi * BC + wc * 32 + l31;
  // the 12 operand units of a run in an order that never puts two 3-MFMA units (rows 0, 3) next to each other
  //   unit u -> (dx, row);  MFMAs of a unit: k-steps ks with 0 <= row - ks <= 2, tap = (row - ks)*3 + dx
#ifndef W2_LEAD
#define W2_LEAD 1      // operand units read ahead (1-3 measured equal; 1 needs the fewest registers)
#endif
#ifdef W2_NOPRIO
#define W2_PRIO(p_)
#else
#define W2_PRIO(p_) __builtin_amdgcn_s_setprio(p_)
#endif
#define W2_UDX(u_) ((u_) / 4)
#define W2_UROW(u_) ((u_) < 4 ? (u_) : ((u_) % 4 == 0 ? 1 : ((u_) % 4 == 1 ? 0 : (u_) % 4)))
#define W2_LOADA(set_, u_)                                                                       \
  _Pragma("unroll") for (int s = 0; s < NSP; ++s)                                                \
    a[set_][s] = Xb[((s * 3 + W2_UDX(u_)) * 4 + W2_UROW(u_)) * XSLAB + abase];
#define W2_MMA_PHASE(buf_)                                                                       \
  {                                                                                              \
    const u32x4* Xb = Xc + (buf_) * XCU;                                                         \
    const u32x4* Db = Dy + (buf_) * DYU;                                                         \
    u32x4 b[2][NSP], a[W2_LEAD + 1][NSP];                                                        \
    _Pragma("unroll") for (int s = 0; s < NSP; ++s) b[0][s] = Db[(s * 2 + 0) * 2 * BC + bbase];  \
    W2_LOADA(0, 0)                                                                               \
    _Pragma("unroll") for (int s = 0; s < NSP; ++s) b[1][s] = Db[(s * 2 + 1) * 2 * BC + bbase];  \
    if (W2_LEAD > 1) W2_LOADA(1, 1)                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    _Pragma("unroll") for (int u = 0; u < 12; ++u) {                                             \
      if (u + W2_LEAD < 12) W2_LOADA((u + W2_LEAD) % (W2_LEAD + 1), u + W2_LEAD)                 \
      const int dx_ = W2_UDX(u), row_ = W2_UROW(u);                                              \
      /* W2_ALT: the two taps a unit feeds take turns (no two consecutive MFMAs on one accumulator) */ \
      _Pragma("unroll") for (int qo = 0; qo < (W2_ALT ? P::N : 1); ++qo)                         \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                         \
        const int ty_ = row_ - ks;                                                               \
        if (ty_ >= 0 && ty_ <= 2) {                                                              \
          _Pragma("unroll") for (int q = (W2_ALT ? qo : 0); q < (W2_ALT ? qo + 1 : P::N); ++q)   \
            acc[ty_ * 3 + dx_] = mma16<NSP>(a[u % (W2_LEAD + 1)][P::A[q]], b[ks][P::B[q]], acc[ty_ * 3 + dx_]); \
        }                                                                                        \
      }                                                                                          \
      const int nm_ = (row_ == 0 || row_ == 3) ? 3 : 6;                                          \
      _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                            \
        if (i < nm_) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
        if (i < NSP && u + W2_LEAD < 12) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      \
      }                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
  }

  if (run_beg < run_end) {
    W2_GLOAD(run_beg);
    W2_LSTORE(0);
    W2_GLOAD(run_beg + 1);
  }
  __syncthreads();

  // two copies of the run loop rather than a branch inside one: each group's loop gets its own register allocation
  // (a branch in the body spilled 270 registers); both execute the same number of barriers
#ifdef W2_TRACE
  const bool trace_blk = blockIdx.x == 5 && blockIdx.y == 1 && blockIdx.z == 0;
#endif
  if (wi == 0) {
    for (int run = run_beg; run < run_end; ++run) {
      const int buf = (run - run_beg) & 1;
      W2T(0)
      W2_PRIO(W2_PRIO_C);
      W2T_VM()
      W2_LSTORE(buf ^ 1);
      W2_GLOAD(run + 2);
      __builtin_amdgcn_sched_barrier(0);
      W2T(1)
      W2_PRIO(W2_PRIO_G0);
      if (!(W2_KO & 8) || k.N < 0) W2_MMA_PHASE(buf);
      W2T(2)
      __syncthreads();
      W2T(3)
    }
  } else {
    for (int run = run_beg; run < run_end; ++run) {
      const int buf = (run - run_beg) & 1;
      W2T(0)
      W2_PRIO(W2_PRIO_G1);
      if (!(W2_KO & 8) || k.N < 0) W2_MMA_PHASE(buf);
      __builtin_amdgcn_sched_barrier(0);
      W2T(1)
      W2_PRIO(W2_PRIO_C);
      W2T_VM()
      W2_LSTORE(buf ^ 1);
      W2_GLOAD(run + 2);
      W2T(2)
      __syncthreads();
      W2T(3)
    }
  }
#undef W2_GLOAD
#undef W2_LSTORE
#undef W2_LOADA
#undef W2_MMA_PHASE
#undef W2_UDX
#undef W2_UROW

  if (want_db) {
    atomicAdd(&bsum[dc], bacc);
    __syncthreads();
    if (tid < BC && co0 + tid < k.Cout) atomicAdd(&k.db[co0 + tid], bsum[tid]);
  }
  if (k.dbx != nullptr && blockIdx.z == 0) {
    atomicAdd(&bsum[xc], baccx);
    __syncthreads();
    if (tid < CT && ci0 + tid < k.Cin) atomicAdd(&k.dbx[ci0 + tid], bsum[tid]);
  }
  if (k.swap) {
    // transposed store: the real layout is [8 - t][co][ci] with ci (this kernel's rows) fastest.  Each wave turns its
    // 32 x 32 tile around through LDS (the operand buffers are free now) so that a half-wave adds to 32 consecutive
    // floats -- 19 M lane-scattered atomics on 74 K addresses cost 0.4 ms at the 128 -> 64 layer
    __syncthreads();
    float* T = reinterpret_cast<float*>(Xc) + wid * (32 * 33);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) T[(4 * lhi + (r & 3) + 8 * (r >> 2)) * 33 + l31] = acc[t][r] * oscale * oscale2;
      __syncthreads();
      const int ci = ci0 + wi * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = 4 * lhi + (r & 3) + 8 * (r >> 2);
        const int co = co0 + wc * 32 + col;
        const float v = T[l31 * 33 + col];
        if (ci < k.Cin && co < k.Cout) df_acc(dwt, ((long long)(8 - t) * k.Cout + co) * k.Cin + ci, v, k.fx);
      }
      __syncthreads();
    }
    return;
  }
  const int co = co0 + wc * 32 + l31;
  if (co < k.Cout) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wi * 32 + 4 * lhi + (r & 3) + 8 * (r >> 2);
        if (ci < k.Cin) df_acc(dwt, ((long long)t * k.Cin + ci) * k.Cout + co, acc[t][r] * oscale * oscale2, k.fx);
      }
    }
  }
}

// (Round 4 built and measured two restructurings of this kernel, both bit-compatible and both SLOWER, so neither is kept
// -- profiles/r04_wgrad_ring_ab.txt, r04_wgrad_trace_ko.txt: (1) a row-ring form: runs walk down a 16-pixel column strip,
// converted X rows live in a ring of six LDS row slots so that a run stages two new rows instead of four, wave group 0
// stages X and group 1 stages dY in 64-byte segments (16 instead of 64 cache lines per load instruction): 0.391 ms against
// 0.379 ms; (2) the same with the conversion cut into slices between the MFMAs of the same wave: 0.413 ms.  The untraced
// knock-outs say why: with loads, conversion AND LDS stores removed this kernel still takes 0.329 of its 0.386 ms -- the
// MFMA phases themselves (operand reads one unit ahead, one barrier per 2 x 54 MFMAs, the atomic tail) are what is left.)
// does df_conv3x3_split_wgrad_try run this geometry with swapped roles (and therefore leave db to the caller)?
bool df_conv3x3_split_wgrad_swaps(const DfConvGeom* g) {
  static DfOptFlag ns_o{"DFMIR_WGRAD_NO_SWAP"}, v1_o{"DFMIR_WGRAD_V1"};
  const bool off = ns_o.get() || v1_o.get();
  return !off && df_split_mode() == 2 && g->KD == 1 && g->KH == 3 && g->KW == 3 && g->Di == 1 && g->Do == 1 && g->stride == 1 &&
         g->dil == 1 && g->ph == 1 && g->pw == 1 && g->pd == 0 && g->Ho == g->Hi && g->Wo == g->Wi && g->pad_mode == 0 &&
         g->Cout == 64 && g->Cin > 64;
}
bool df_conv3x3_split_wgrad_try(const DfConvGeom* g, const float* x, const float* x_amax, int x_n, const float* dy,
                                const float* dy_amax, int dy_n, float* dw_tcc, float* db, hipStream_t st, int* rc,
                                const float* dy_pmax) {
  const int mode = df_split_mode();
  if (mode == 0 || (mode == 2 && !(x_amax && dy_amax && x_n > 0 && dy_n > 0))) return false;
  if (!(g->KD == 1 && g->KH == 3 && g->KW == 3 && g->Di == 1 && g->Do == 1 && g->stride == 1 && g->dil == 1))
    return false;
  if (g->ph != 1 || g->pw != 1 || g->pd != 0 || g->Ho != g->Hi || g->Wo != g->Wi) return false;
  if (g->Cout < 64 || g->Cin < 64) return false;
  const bool wide = g->Cout > 64;
  if ((g->Hi & 1) || (g->Wi & 15) || g->Hi < 2) return false;
  const long long HW = (long long)g->Hi * g->Wi;
  if (HW * g->Cin * 4 >= (1LL << 31) || HW * g->Cout * 4 >= (1LL << 31)) return false;
  const bool swap = mode == 2 && x_amax && dy_amax && df_conv3x3_split_wgrad_swaps(g);
  WS3P k{g->N, g->Cin, g->Cout, g->Hi, g->Wi, g->pad_mode, g->Wi / 16, 0, 0, 0, x_amax, dy_amax, x_n, dy_n, db, dy_pmax, 0, nullptr, df_det_fx()};
  if (swap) {
    k.Cin = g->Cout; k.Cout = g->Cin;
    k.x_amax = dy_amax; k.dy_amax = x_amax; k.x_n = dy_n; k.dy_n = x_n;
    k.db = nullptr; k.dbx = db; k.dy_pmax = nullptr; k.swap = 1;
  }
  k.runs_per_img = (g->Hi / 2) * k.runs_per_row;
  const long long total = (long long)g->N * k.runs_per_img;
  if (total >= (1LL << 30)) return false;
  k.runs_total = (int)total;
  const unsigned ny = (k.Cin + 63) / 64, nz = (wide || swap) ? (k.Cout + 127) / 128 : 1;
  long long want = 256 / ((long long)ny * nz);    // one resident round: 1 workgroup per CU
  if (want < 1) want = 1;
  long long maxs = (k.runs_total + 7) / 8;        // >= 8 runs per block
  if (maxs < 1) maxs = 1;
  if (want > maxs) want = maxs;
  k.runs_per_block = (int)((k.runs_total + want - 1) / want);
  const unsigned nx = (k.runs_total + k.runs_per_block - 1) / k.runs_per_block;
  const dim3 grid(nx, ny, nz);
  if (mode == 2) {
    static DfOptFlag v1_o{"DFMIR_WGRAD_V1"};
    const bool v1 = v1_o.get();      // A/B: the single-buffered kernel
    if (swap) conv3x3_wgrad_split2_k<<<grid, 512, 0, st>>>(dy, x, dw_tcc, k);
    else if (wide && !v1) conv3x3_wgrad_split2_k<<<grid, 512, 0, st>>>(x, dy, dw_tcc, k);
    else if (wide) conv3x3_wgrad_split_k<2, 128><<<grid, 512, 0, st>>>(x, dy, dw_tcc, k);
    else conv3x3_wgrad_split_k<2, 64><<<grid, 512, 0, st>>>(x, dy, dw_tcc, k);
  } else {
    if (wide) conv3x3_wgrad_split_k<3, 128><<<grid, 512, 0, st>>>(x, dy, dw_tcc, k);
    else conv3x3_wgrad_split_k<3, 64><<<grid, 512, 0, st>>>(x, dy, dw_tcc, k);
  }
  hipError_t e = hipGetLastError();
  *rc = (e == hipSuccess) ? 0 : df_set_error((int)e, __FILE__, __LINE__);
  return true;
}

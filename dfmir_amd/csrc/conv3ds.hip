// 3-D 3x3x3 stride-1 "same" convolution (forward and dgrad) of the VoxelMorph U-Net on the 16-bit matrix cores by
// operand splitting -- the scaled fp16x2 form of conv3x3s.hip (a = (a0 + a1) / s, a*b ~= a0b0 + a0b1 + a1b0, fp32
// accumulate) applied to the small-channel, huge-volume layers (16..64 channels, up to 6.9 M voxels) whose fp32-MFMA
// kernels (conv3d.hip) top out at ~90 TFLOP/s.
//
// Workgroup = 256 threads = 4 waves; output tile 4 x 8 x 16 voxels x 32 output channels; wave w owns z-plane w of the
// tile (128 voxels = 4 MFMA column tiles of 2 rows x 16).  Per chunk of 8 input channels:
//   * the (4+2) x (8+2) x (16+2) halo patch is split when it is written to LDS: Xs[split][position] x (8 channels x
//     fp16 = 16 B), zero padding by hardware-bounds-checked buffer loads;
//   * the chunk's weights arrive pre-split from conv3d_wsplit_k, [chunk][split][28 taps][32 couts] x 16 B, and are
//     copied to LDS as they are;
//   * K = 16 of one v_mfma_f32_32x32x16_f16 = 2 taps x 8 channels: lanes 0-31 feed tap 2t, lanes 32-63 tap 2t+1
//     (their B reads differ by the tap's patch offset); tap 27 pairs with zero weights (3.6 % idle).
// The next chunk's global loads are issued before the 14 x 12 MFMAs of the current one and converted / stored after it
// (one LDS buffer, two barriers per chunk).  More than 32 output channels = more workgroups along grid.y.
// The epilogue rescales by 2^-(ex+ew), adds the bias, applies LeakyReLU and leaves the range probe of its OUTPUT
// (64 accumulating slots, as the InstanceNorm kernels do) so that the next layer needs no absmax pass.
//
// PAIR form (<= 16 output channels: half of the 32 MFMA rows would be padding): the rows are (plane p, cout) -- rows
// 0-15 compute z-plane 2q, rows 16-31 plane 2q+1 of the SAME 16 output channels.  Both planes read the four input
// planes 2q-1 .. 2q+2, so K runs over 36 taps' = (dz' in 0..3, dy, dx) and the weight rows of plane p hold
// W[dz' - p] (zero where dz' - p is outside 0..2): 75 % of the issued products are useful instead of 50 %.  Wave w
// owns plane pair w & 1 and the y-half w >> 1 of the tile (2 column tiles), 18 k-steps x 6 MFMAs per chunk.
#include "conv3x3_common.h"
#ifdef C3S_TRACE      // timing build: per-phase s_memtime sums of every wave 0 -> dfmir_c3s_trace()
__device__ unsigned long long c3s_trace[16];
#define C3S_T(i_) { if (tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&c3s_trace[i_], t_ - tlast); tlast = t_; } }
#define C3S_T0() unsigned long long tlast = __builtin_readcyclecounter();
extern "C" void dfmir_c3s_trace(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(c3s_trace), sizeof(c3s_trace));
  if (reset) { unsigned long long z[16] = {}; hipMemcpyToSymbol(HIP_SYMBOL(c3s_trace), z, sizeof(z)); }
}
#else
#define C3S_T(i_)
#define C3S_T0()
#endif
#ifndef C3S_KO
#define C3S_KO 0     // knock-out builds for timing: 1 = no MFMAs, 2 = no prefetch loads, 4 = no convert + LDS store
#endif

typedef _Float16 f16x8_3 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int scale_exp3(float amax) {
  const int be = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127;
  int e = (amax > 0.f) ? 14 - be : 0;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return e;
}
__device__ __forceinline__ float pow2f3(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }

// (x0, x1) * s -> leading fp16 pair h and residual pair r (see split_pair_scaled in conv3x3s.hip)
__device__ __forceinline__ void split_pair3(float x0, float x1, float s, unsigned& h, unsigned& r) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(x1), "v"(s), "v"(h));
}
__device__ __forceinline__ f32x16 mma3(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_3, a), __builtin_bit_cast(f16x8_3, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Weight split: w_tcc [27][K][M] fp32 (tap-major packing of dfmir_weight_pack, either mode) ->
// ws[mtile][chunk][split][28][32] x (8 reduction channels x fp16), scaled by 2^ew with ew from max|w|;
// trailer (4 floats after the units): [0] = ew as an int.  The layers are tiny (<= 110 K weights): every workgroup
// reduces the maximum itself (L2-resident) and packs its share of the units.
// ------------------------------------------------------------------------------------------------
// pair != 0 (M <= 16): 36 taps' per chunk, row (p, co) = p * 16 + co holds W[dz' - p] (see the PAIR form above).
// Ktot / koff: the layer's reduction channels are rows koff .. koff + K - 1 of a packing with Ktot rows per tap (the skip
// channels of a concatenated input, conv3d_up_phase_k below); Ktot == K, koff == 0 for a whole layer.
// max |w| over rows koff .. koff + K - 1 of every tap of a [27][Ktot][M] packing: per tap K * M consecutive floats, read as
// float4 where the slab is 16-byte aligned (one division per element and 4-byte loads made this reduction -- which every
// workgroup of a weight-split job repeats -- the longest part of conv3d_wsplit_batch_k)
__device__ __forceinline__ float wslab_absmax(const float* __restrict__ w, int K, int Ktot, int koff, int M) {
  float m = 0.f;
  const int slab = K * M;
  if ((slab & 3) == 0 && ((Ktot * M) & 3) == 0 && ((koff * M) & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
    const int s4 = slab >> 2;
    for (int i = threadIdx.x; i < 27 * s4; i += 1024) {
      const int tapi = i / s4, r4 = i - tapi * s4;
      const float4 v = *reinterpret_cast<const float4*>(w + ((long long)tapi * Ktot + koff) * M + 4 * r4);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  } else {
    for (int i = threadIdx.x; i < 27 * slab; i += 1024) {
      const int tapi = i / slab, rem = i - tapi * slab;
      m = fmaxf(m, fabsf(w[((long long)tapi * Ktot + koff) * M + rem]));
    }
  }
  return m;
}
__device__ __forceinline__ void conv3d_wsplit_body(const float* __restrict__ w, u32x4* __restrict__ ws, int K, int M,
                                                   float* __restrict__ trailer, int pair, int Ktot, int koff, float* sm) {
  float m = wslab_absmax(w, K, Ktot, koff, M);
  m = block_max(m, sm);
  if (!(m == m)) m = __uint_as_float(0x7f800000u);
  const int ew = scale_exp3(m);
  const float s = pow2f3(ew);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<int*>(trailer)[0] = ew;
  const int nchunk = (K + 7) / 8, nmt = pair ? 1 : (M + 31) / 32;
  if (pair == 2) {
    // M16 form (conv3d_split_m16_k): [chunk][split][28 taps][16 couts] x (8 channels x fp16), no padding rows
    for (int u = blockIdx.x * 1024 + threadIdx.x; u < nchunk * 28 * 16; u += gridDim.x * 1024) {
      const int co = u & 15, tap = (u >> 4) % 28, ch = (u >> 4) / 28;
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int kk = ch * 8 + c;
        v[c] = (tap < 27 && kk < K && co < M) ? w[((long long)tap * Ktot + koff + kk) * M + co] : 0.f;
      }
      u32x4 h, r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned hh, rr;
        split_pair3(v[2 * q], v[2 * q + 1], s, hh, rr);
        h[q] = hh; r[q] = rr;
      }
      ws[(long long)ch * 896 + tap * 16 + co] = h;
      ws[(long long)ch * 896 + 448 + tap * 16 + co] = r;
    }
    return;
  }
  const int NT = pair ? 36 : 28;
  const int units = nmt * nchunk * NT * 32;                 // one unit = both splits of (mtile, chunk, tap, row)
  for (int u = blockIdx.x * 1024 + threadIdx.x; u < units; u += gridDim.x * 1024) {
    const int co = u & 31;
    int t = u >> 5;
    int tap = t % NT; t /= NT;
    const int ch = t % nchunk, mt = t / nchunk;
    int mo = mt * 32 + co;
    bool tok = tap < 27;
    if (pair) {
      const int dz = tap / 9 - (co >> 4);
      tok = (unsigned)dz < 3u;
      tap = dz * 9 + tap % 9;
      mo = co & 15;
    }
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int kk = ch * 8 + c;
      v[c] = (tok && kk < K && mo < M) ? w[((long long)tap * Ktot + koff + kk) * M + mo] : 0.f;
    }
    u32x4 h, r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned hh, rr;
      split_pair3(v[2 * q], v[2 * q + 1], s, hh, rr);
      h[q] = hh; r[q] = rr;
    }
    const int tu = (u >> 5) % NT;
    const long long base = (((long long)mt * nchunk + ch) * 2) * (NT * 32);
    ws[base + tu * 32 + co] = h;
    ws[base + NT * 32 + tu * 32 + co] = r;
  }
}
__global__ __launch_bounds__(1024) void conv3d_wsplit_k(const float* __restrict__ w, u32x4* __restrict__ ws, int K,
                                                        int M, float* __restrict__ trailer, int pair, int Ktot, int koff) {
  __shared__ float sm[17];
  conv3d_wsplit_body(w, ws, K, M, trailer, pair, Ktot, koff, sm);
}

// ------------------------------------------------------------------------------------------------
struct C3sP {
  int N, Cin, Cout, D, H, W;
  int act;
  float slope;
  int nz, ny, nx;                // tiles per axis
  int nchunk;
  int x_n;                       // floats of the input range probe
  int cout_used;                 // output channels to compute (<= Cout; the rest of y is left untouched)
  long long ntile;               // N * nz * ny * nx
  // dgrad whose result is the gradient w.r.t. the OUTPUT of a LeakyReLU (act_src = that output, same shape as y):
  // the epilogue multiplies by the activation's derivative, i.e. it writes the gradient w.r.t. the pre-activation,
  // and the separate act_bwd pass over the tensor (read dy, read y, write: 3 transfers of up to 0.9 GB) disappears
  const float* act_src;
  float act_slope;
  // av_mode 2: act_src is an ADDEND of the output's shape (may be y itself): y = act(conv + bias + addend) -- the skip
  // channels' share of a convolution over cat(nearest_up2(a), b) whose up-sampled share conv3d_up_phase_k left in y
  int av_mode;                   // 0 none, 1 activation derivative of act_src, 2 addend
};

// TT > 1 (multi-tile form): a workgroup computes TT tiles stacked along y with ONE staging of each chunk's weights -- the
// pre-split weights (29 / 37 KB per chunk) were re-read from L2 for every tile, as much as the patch itself, and in the
// plane-pair form (108 MFMAs per chunk instead of 168) that put the texture path at ~70 % and the matrix pipe at 40 %.
template <bool PAIR, bool VEC, int TT>
__global__ __launch_bounds__(256, 2) void conv3d_split_k(const float* __restrict__ x, const float* __restrict__ x_amax,
                                                      const u32x4* __restrict__ wsp, const float* __restrict__ w_trailer,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      float* __restrict__ y_amax, C3sP k) {
  constexpr int TZ = 4, TY = 8, TX = 16, HY = TY + 2, HX = TX + 2;
  constexpr int XP = (TZ + 2) * HY * HX;                  // 1080 positions
  constexpr int NS = (XP + 255) / 256;                    // 5 position slots per thread
  constexpr int NT = PAIR ? 36 : 28;                      // taps (taps') per chunk in the weight units
  constexpr int WU = 2 * NT * 32;                         // 1792 / 2304 16-B units of one chunk's weights
  constexpr int NW = WU / 256;                            // 7 / 9
  constexpr int NJ = PAIR ? 2 : 4;                        // column tiles (2 rows x 16 voxels) per wave
  constexpr unsigned OOB = 0x80000000u;
  __shared__ u32x4 Xs[2 * XP];
  __shared__ u32x4 Ws[WU];
  __shared__ float red[17];
  __shared__ unsigned smax;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const long long S = (long long)k.D * k.H * k.W;
  C3S_T0()
  // workgroup -> tile: the dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs (one L2 each), so ids
  // with the same residue get one contiguous eighth of the tiles; within it x runs fastest, then z, then y: the
  // z-halo (2 of 6 planes) of a tile is the previous x-row's data, still in that XCD's L2.
  // PERSISTENT: the J = gridDim.x / 8 workgroups of an XCD walk its eighth together (iteration i: tiles i J .. i J + J - 1),
  // and the prefetch of a tile's last phase already fetches the first chunk of the workgroup's NEXT tile, so only the
  // first tile of a workgroup pays the exposed prologue (a per-phase trace had it at 10-23 % of a one-tile workgroup).
  const long long per_xcd = (k.ntile + 7) / 8;
  const int J = (int)(gridDim.x >> 3);
  const long long t_first = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  long long t_lim = (long long)((blockIdx.x & 7) + 1) * per_xcd;
  if (t_lim > k.ntile) t_lim = k.ntile;
  if (t_first >= t_lim) return;
  const int niter = (int)((t_lim - t_first + J - 1) / J);
  int n, z0, y0, x0;                                        // the tile being LOADED (one phase ahead of the compute)
#define C3S_DECODE(t_)                                                                            \
  {                                                                                               \
    long long pid_ = (t_);                                                                        \
    const int bx_ = (int)(pid_ % k.nx); pid_ /= k.nx;                                             \
    const int bz_ = (int)(pid_ % k.nz); pid_ /= k.nz;                                             \
    const int by_ = (int)(pid_ % k.ny);                                                           \
    n = (int)(pid_ / k.ny);                                                                       \
    z0 = bz_ * TZ; y0 = by_ * TY * TT; x0 = bx_ * TX;      /* (k.ny counts groups of TT tiles) */  \
  }
  C3S_DECODE(t_first)
  const int mt = blockIdx.y;
#ifdef C3S_STAGGER
  // the second workgroup of a CU starts half a phase late, so that the pair does not stage / compute in lockstep
  if ((blockIdx.x >> 3) & 32) { for (int i_ = 0; i_ < C3S_STAGGER; ++i_) __builtin_amdgcn_s_sleep(64); }
#endif

  // scales: input scaled by 2^ex when it is split, result rescaled by 2^-ex * 2^-ew
  const float amax = reduce_absmax(x_amax, k.x_n, red);
  const int ex = scale_exp3(amax);
  const int ew = reinterpret_cast<const int*>(w_trailer)[0];
  const float xscale = pow2f3(ex), oscale = pow2f3(-ex), oscale2 = pow2f3(-ew);
  if (tid == 0) smax = 0u;

  __amdgpu_buffer_rsrc_t x_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x + (long long)n * k.Cin * S), 0, (unsigned)((long long)k.Cin * S * 4), 0x00020000);
  const unsigned s4 = (unsigned)S * 4u;
  // Patch loads.  VEC (W % 4 == 0): thread t < 240 owns the 16-B quad q = t & 3 of halo row t >> 2 (rows = 6 planes x
  // 10 y; the quad covers patch columns 1 + 4q .. 4 + 4q, i.e. x0 + 4q ..) in all 8 channels of the chunk -- 8
  // buffer_load_dwordx4, converted to 4 LDS units -- and thread t < 120 additionally the left / right halo column
  // (t & 1) of row t >> 1 (8 buffer_load_dword -> 1 unit).  Otherwise: 5 patch positions per thread, 8 dword loads each.
  constexpr int NPART = VEC ? 8 : NS;                     // load parts of a chunk (spread over the k-steps)
  unsigned gbyte[NS];
  unsigned gq = OOB, gh = OOB;
  int posq = -1, posh = -1;
  // global offsets of this thread's patch loads for tile t_ of the group (rows y0 + t_ * TY ..)
#define C3S_OFFS(t_)                                                                              \
  {                                                                                               \
    const int yt0_ = y0 + (t_) * TY;                                                              \
    if constexpr (VEC) {                                                                          \
      gq = OOB; gh = OOB;                                                                         \
      if (tid < 240) {                                                                            \
        const int row = tid >> 2, q = tid & 3;                                                    \
        const int hz = row / HY, hy = row % HY;                                                   \
        const int gz = z0 - 1 + hz, gy = yt0_ - 1 + hy, gx = x0 + 4 * q;                          \
        posq = row * HX + 1 + 4 * q;                                                              \
        if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && gx < k.W)             \
          gq = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;                                       \
      }                                                                                           \
      if (tid < 120) {                                                                            \
        const int row = tid >> 1, side = tid & 1;                                                 \
        const int hz = row / HY, hy = row % HY;                                                   \
        const int gz = z0 - 1 + hz, gy = yt0_ - 1 + hy, gx = side ? x0 + TX : x0 - 1;             \
        posh = row * HX + (side ? HX - 1 : 0);                                                    \
        if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W) \
          gh = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;                                       \
      }                                                                                           \
    } else {                                                                                      \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                            \
        const int pos = tid + 256 * s;                                                            \
        unsigned off = OOB;                                                                       \
        if (pos < XP) {                                                                           \
          const int hx = pos % HX, t = pos / HX, hy = t % HY, hz = t / HY;                        \
          const int gz = z0 - 1 + hz, gy = yt0_ - 1 + hy, gx = x0 - 1 + hx;                       \
          if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W) \
            off = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;                                    \
        }                                                                                         \
        gbyte[s] = off;                                                                           \
      }                                                                                           \
    }                                                                                             \
  }
  C3S_OFFS(0)
  const u32x4* wchunk = wsp + (long long)mt * k.nchunk * WU;

  // this lane's B positions: column tile j = rows 2j, 2j+1 of plane wid; voxel (row, x) = (2j + (l31 >> 4), lx) with
  // lx = l31 & 15 in the even row and (l31 - 2) & 15 in the odd one: with the row stride of 18 units that rotation
  // puts the 16 lanes a ds_read_b128 serves together ({0-3,12-15,20-27}, {4-11,16-19,28-31}) on 16 distinct bank quads
  // (PAIR: planes 2 (wid & 1) + p, rows 4 (wid >> 1) + 2j, 2j+1)
  const int wz = PAIR ? 2 * (wid & 1) : wid, wy = PAIR ? 4 * (wid >> 1) : 0;
  int pbase[NJ];
  const int lx = (l31 - 2 * (l31 >> 4)) & 15;
#pragma unroll
  for (int j = 0; j < NJ; ++j) pbase[j] = (wz * HY + wy + 2 * j + (l31 >> 4)) * HX + lx;

  f32x16 acc[TT][NJ];

  float rx[NS][8];                                         // !VEC: [slot][channel];  VEC: rx[0][c] = halo column
  u32x4 rq[8];                                             // VEC: the quad of channel c
  u32x4 rw[NW];

  // Global loads of a chunk, in NS + 1 parts that the compute loop spreads over its first k-steps.  No branches: a
  // channel past Cin (padding of the last chunk, or the chunk after the last) is past the descriptor's range and
  // reads as zero without touching memory; the same holds for the weight units past the last chunk.
  const __amdgpu_buffer_rsrc_t w_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(wsp + (long long)mt * k.nchunk * WU), 0, (unsigned)(k.nchunk * WU * 16), 0x00020000);
#define C3S_GLOAD_X(ch_, s_)                                                                      \
  {                                                                                               \
    const unsigned cbase = (unsigned)((ch_) * 8) * s4;                                            \
    if constexpr (VEC) {                                                                          \
      const unsigned co_ = cbase + (unsigned)(s_) * s4;                                           \
      rq[s_] = __builtin_amdgcn_raw_buffer_load_b128(x_src, gq == OOB ? OOB : gq + co_, 0, 0);    \
      rx[0][s_] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_src, gh == OOB ? OOB : gh + co_, 0, 0)); \
    } else {                                                                                      \
      _Pragma("unroll") for (int c = 0; c < 8; ++c)                                               \
        rx[s_][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(                         \
            x_src, gbyte[s_] == OOB ? OOB : gbyte[s_] + cbase + (unsigned)c * s4, 0, 0));         \
    }                                                                                             \
  }
#define C3S_GLOAD_W(ch_)                                                                          \
  _Pragma("unroll") for (int j = 0; j < NW; ++j)                                                  \
    rw[j] = __builtin_amdgcn_raw_buffer_load_b128(w_src, (unsigned)(((ch_) * WU + tid + 256 * j) * 16), 0, 0);
#define C3S_SPLIT8(v_, h_, r_)                                                                    \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                 \
    unsigned hh, rr;                                                                              \
    split_pair3(v_[2 * q], v_[2 * q + 1], xscale, hh, rr);                                        \
    h_[q] = hh; r_[q] = rr;                                                                       \
  }
#define C3S_LSTORE()                                                                              \
  {                                                                                               \
    if constexpr (VEC) {                                                                          \
      if (posq >= 0) {                                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                           \
          float v[8];                                                                             \
          _Pragma("unroll") for (int c = 0; c < 8; ++c) v[c] = __uint_as_float(rq[c][e]);         \
          u32x4 h, r;                                                                             \
          C3S_SPLIT8(v, h, r)                                                                     \
          Xs[posq + e] = h;                                                                       \
          Xs[XP + posq + e] = r;                                                                  \
        }                                                                                         \
      }                                                                                           \
      if (posh >= 0) {                                                                            \
        u32x4 h, r;                                                                               \
        C3S_SPLIT8(rx[0], h, r)                                                                   \
        Xs[posh] = h;                                                                             \
        Xs[XP + posh] = r;                                                                        \
      }                                                                                           \
    } else {                                                                                      \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                            \
        const int pos = tid + 256 * s;                                                            \
        if (pos < XP) {                                                                           \
          u32x4 h, r;                                                                             \
          C3S_SPLIT8(rx[s], h, r)                                                                 \
          Xs[pos] = h;                                                                            \
          Xs[XP + pos] = r;                                                                       \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
  }
#define C3S_LSTORE_W() _Pragma("unroll") for (int j = 0; j < NW; ++j) Ws[tid + 256 * j] = rw[j];
  // operands of k-step tp_ (taps 2tp, 2tp+1; tap 27 of the 28-tap form has zero weights, any valid offset) -> set b_
#define C3S_OPLOAD(b_, tp_)                                                                       \
  {                                                                                               \
    const int t0 = 2 * (tp_), t1 = (PAIR || (2 * (tp_) + 1) < 27) ? 2 * (tp_) + 1 : 0;            \
    const int o0 = ((t0 / 9) * HY + (t0 / 3) % 3) * HX + t0 % 3;                                  \
    const int o1 = ((t1 / 9) * HY + (t1 / 3) % 3) * HX + t1 % 3;                                  \
    const int toff = hi ? o1 : o0;                                                                \
    const int tap = 2 * (tp_) + hi;                                                               \
    A0[b_] = Ws[tap * 32 + l31];                                                                  \
    A1[b_] = Ws[NT * 32 + tap * 32 + l31];                                                        \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                              \
      B0[b_][j] = Xs[pbase[j] + toff];                                                            \
      B1[b_][j] = Xs[XP + pbase[j] + toff];                                                       \
    }                                                                                             \
  }

  u32x4 A0[2], A1[2], B0[2][NJ], B1[2][NJ];
#pragma unroll
  for (int s = 0; s < NPART; ++s) C3S_GLOAD_X(0, s);
  C3S_GLOAD_W(0);
  C3S_T(0)
  C3S_LSTORE();
  C3S_LSTORE_W();
  C3S_T(1)
  __syncthreads();
  C3S_T(2)

  float pm = 0.f;
  int cn = n, cz0 = z0, cy0 = y0, cx0 = x0;                 // the tile being computed
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
  // ONE flat loop over (tile, chunk) -- nested loops made the register allocator keep a second copy of the accumulators
  int it = 0, ch = 0;
  for (int q = 0; q < niter * k.nchunk; ++q) {
  {
#pragma unroll
   for (int tt = 0; tt < TT; ++tt) {
    // the phase after this one: the next tile of the group on the same chunk, tile 0 of the next chunk (+ its weights),
    // or chunk 0 of this workgroup's next tile
    const bool tile_end = (ch + 1 == k.nchunk) && (tt + 1 == TT);
    const bool more = !tile_end || (it + 1 < niter);
    const int nch = tile_end ? 0 : ((tt + 1 < TT) ? ch : ch + 1);
    const int wch = tile_end ? 0 : ch + 1;
    if (tile_end && more) {
      C3S_DECODE(t_first + (long long)(it + 1) * J)
      x_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)n * k.Cin * S), 0,
                                                (unsigned)((long long)k.Cin * S * 4), 0x00020000);
    }
    if (TT > 1 || tile_end) C3S_OFFS((tt + 1) % TT)
    C3S_OPLOAD(0, 0);
    // one k-step: the next step's 2 + 2 NJ operand reads and one part of the next chunk's global loads are pinned
    // between this step's 3 NJ MFMAs (one MFMA, one LDS read, one buffer load, ...), so the wave never waits on
    // a read it has just issued
#pragma unroll
    for (int tp = 0; tp < NT / 2; ++tp) {
      const int cur = tp & 1;
      if (!(C3S_KO & 2)) {
#ifdef C3S_EARLY
        if (2 * tp < NPART) { C3S_GLOAD_X(nch, 2 * tp); if (2 * tp + 1 < NPART) C3S_GLOAD_X(nch, 2 * tp + 1); }
        if (2 * tp == NPART && tt + 1 == TT) C3S_GLOAD_W(wch);
#else
        if (tp < NPART) C3S_GLOAD_X(nch, tp);
        if (tp == NPART && tt + 1 == TT) C3S_GLOAD_W(wch);
#endif
      }
      if (tp + 1 < NT / 2) C3S_OPLOAD(cur ^ 1, tp + 1);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#if (C3S_KO & 1)
        acc[tt][j][0] += __uint_as_float(B0[cur][j][0] ^ B1[cur][j][1] ^ A0[cur][0] ^ A1[cur][1]);
#else
        acc[tt][j] = mma3(A1[cur], B0[cur][j], acc[tt][j]);
        acc[tt][j] = mma3(A0[cur], B1[cur][j], acc[tt][j]);
        acc[tt][j] = mma3(A0[cur], B0[cur][j], acc[tt][j]);
#endif
      }
#pragma unroll
      for (int i = 0; i < 3 * NJ; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
      }
#ifdef C3S_FENCE
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    C3S_T(3)
    if (more) {
      __syncthreads();
      C3S_T(4)
      if (!(C3S_KO & 4)) {
        C3S_LSTORE();
        if (tt + 1 == TT) { C3S_LSTORE_W(); }
      }
      C3S_T(5)
      __syncthreads();
      C3S_T(6)
    }
   }
  }
  const bool tile_done = (ch + 1 == k.nchunk);
  if (!tile_done) { ++ch; continue; }
#undef C3S_GLOAD_X
#undef C3S_GLOAD_W
#undef C3S_OPLOAD
#undef C3S_LSTORE
#undef C3S_LSTORE_W
#undef C3S_OFFS
#undef C3S_DECODE
#undef C3S_SPLIT8

  // ---- epilogue: acc[j][r] <-> row = (r>>2)*8 + hi*4 + (r&3) = cout - mt*32 (PAIR: plane * 16 + cout),
  //      voxel (wz [+ plane], wy + 2j + (l31>>4), l31&15)
  // Stores go through a buffer descriptor with 32-bit offsets (voxel part per lane, channel part uniform) and the 16
  // bias values of this lane are fetched BEFORE the first store: loads and stores share vmcnt on gfx9, so a bias load
  // between stores would wait for every store issued before it.
  __builtin_amdgcn_sched_barrier(0);
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r >> 2) * 8 + hi * 4 + (r & 3);
    const int co = PAIR ? (row & 15) : mt * 32 + row;
    bv[r] = (bias && co < k.cout_used) ? bias[co] : 0.f;
  }
  const __amdgpu_buffer_rsrc_t y_dst = __builtin_amdgcn_make_buffer_rsrc(
      y + (long long)cn * k.Cout * S, 0, (unsigned)((long long)k.Cout * S * 4), 0x00020000);
  const unsigned plane4 = (unsigned)(k.H * k.W) * 4u;
  const float osc = oscale * oscale2;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
  const int yt0 = cy0 + tt * TY;
  // activation-derivative source values one column tile AHEAD of the stores: loads and stores share vmcnt (in order), so a
  // load issued after a batch of stores would wait for all of them; issued before, it only lets them stay in flight
  float av[2][16];
  const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>((k.act_src ? k.act_src : y) + (long long)cn * k.Cout * S), 0,
      (unsigned)((long long)k.Cout * S * 4), 0x00020000);
#define C3S_AVLOAD(set_, j_)                                                                      \
  if (k.act_src) {                                                                                \
    const int gy = yt0 + wy + 2 * (j_) + (l31 >> 4), gx = cx0 + lx;                               \
    const bool vok = gy < k.H && gx < k.W;                                                        \
    const unsigned vo = (unsigned)(((cz0 + wz) * k.H + gy) * k.W + gx) * 4u + (unsigned)(hi * 4) * s4; \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                              \
      const int rowu = (r >> 2) * 8 + (r & 3);                                                    \
      const int cou = PAIR ? (rowu & 15) : mt * 32 + rowu;                                        \
      const int pz = PAIR ? (rowu >> 4) : 0;                                                      \
      const bool ok = vok && (cou + hi * 4) < k.cout_used && (cz0 + wz + pz) < k.D;               \
      av[set_][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(a_src, ok ? vo + (unsigned)pz * plane4 : OOB, \
                                                                            (unsigned)cou * s4, 0)); \
    }                                                                                             \
  }
  C3S_AVLOAD(0, 0)
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if (j + 1 < NJ) C3S_AVLOAD((j + 1) & 1, j + 1)
    const int gy = yt0 + wy + 2 * j + (l31 >> 4), gx = cx0 + lx;
    const bool vok = gy < k.H && gx < k.W;
    // lane part of the offset: voxel + the hi * 4 channels (PAIR: rows 16.. are plane 1 -> hi never changes the plane)
    const unsigned vo = (unsigned)(((cz0 + wz) * k.H + gy) * k.W + gx) * 4u + (unsigned)(hi * 4) * s4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rowu = (r >> 2) * 8 + (r & 3);                       // row without the hi * 4 part
      const int cou = PAIR ? (rowu & 15) : mt * 32 + rowu;
      const int pz = PAIR ? (rowu >> 4) : 0;
      const bool ok = vok && (cou + hi * 4) < k.cout_used && (cz0 + wz + pz) < k.D;
      float v = acc[tt][j][r] * osc + bv[r];
      if (k.av_mode == 2) v += av[j & 1][r];
      if (k.act == 1) v = v > 0.f ? v : v * k.slope;
      else if (k.act == 2) v = tanhf(v);
      if (k.av_mode == 1) v = av[j & 1][r] > 0.f ? v : v * k.act_slope;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), y_dst, ok ? vo + (unsigned)pz * plane4 : OOB,
                                            (unsigned)cou * s4, 0);
      pm = fmaxf(pm, ok ? fabsf(v) : 0.f);
    }
  }
#undef C3S_AVLOAD
  }
  // next tile of this workgroup: fresh accumulators, the coordinates the prefetch already moved to
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
  cn = n; cz0 = z0; cy0 = y0; cx0 = x0;
  ch = 0; ++it;
  }   // (tile, chunk) phases of this workgroup
  C3S_T(7)
  if (y_amax) {
    __syncthreads();
    publish_block_absmax_acc(pm, &smax, y_amax);
  }
  C3S_T(8)
}

// ------------------------------------------------------------------------------------------------
// M16 form (<= 16 output channels): the same tiling and staging, products on v_mfma_f32_16x16x32_f16 -- 16 rows =
// the output channels, 16 columns = one x-row of the tile, K = 32 = 4 taps x 8 channels (7 k-steps per chunk, tap 27
// padding).  No padding ROWS: the plane-pair form issues 36/27 of the useful products (rows (plane, cout) against 4
// input planes), this form 28/27 -- on a package that sits on its power cap, fewer issued products is time.
// Wave w = z-plane w of the tile; column tile j = tile row j (8 per wave); lane = (kg = lane >> 4, x = lane & 15):
// A = Ws[(4 ks + kg) * 16 + x-as-cout], B = patch position of (plane, row j, x) + offset of tap 4 ks + kg.
typedef float f32x4_3 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_3 mma16(u32x4 a, u32x4 b, f32x4_3 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_3, a), __builtin_bit_cast(f16x8_3, b), c, 0, 0, 0);
}
template <bool VEC, int TT>
__global__ __launch_bounds__(256, 2) void conv3d_split_m16_k(const float* __restrict__ x, const float* __restrict__ x_amax,
                                                          const u32x4* __restrict__ wsp, const float* __restrict__ w_trailer,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ y_amax, C3sP k) {
  constexpr int TZ = 4, TY = 8, TX = 16, HY = TY + 2, HX = TX + 2;
  constexpr int XP = (TZ + 2) * HY * HX;                  // 1080 positions
  constexpr int NS = (XP + 255) / 256;
  constexpr int WU = 2 * 28 * 16;                         // 896 16-B units of one chunk's weights
  constexpr int NW = (WU + 255) / 256;                    // 4 (the last partly)
  constexpr int NJ = 8;                                   // column tiles (tile rows) per wave
  constexpr int NKS = 7;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ u32x4 Xs[2 * XP];
  __shared__ u32x4 Ws[WU];
  __shared__ float red[17];
  __shared__ unsigned smax;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  const long long S = (long long)k.D * k.H * k.W;
  const long long per_xcd = (k.ntile + 7) / 8;
  const int J = (int)(gridDim.x >> 3);
  const long long t_first = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  long long t_lim = (long long)((blockIdx.x & 7) + 1) * per_xcd;
  if (t_lim > k.ntile) t_lim = k.ntile;
  if (t_first >= t_lim) return;
  const int niter = (int)((t_lim - t_first + J - 1) / J);
  int n, z0, y0, x0;
#define C3M_DECODE(t_)                                                                            \
  {                                                                                               \
    long long pid_ = (t_);                                                                        \
    const int bx_ = (int)(pid_ % k.nx); pid_ /= k.nx;                                             \
    const int bz_ = (int)(pid_ % k.nz); pid_ /= k.nz;                                             \
    const int by_ = (int)(pid_ % k.ny);                                                           \
    n = (int)(pid_ / k.ny);                                                                       \
    z0 = bz_ * TZ; y0 = by_ * TY * TT; x0 = bx_ * TX;                                             \
  }
  C3M_DECODE(t_first)

  const float amax = reduce_absmax(x_amax, k.x_n, red);
  const int ex = scale_exp3(amax);
  const int ew = reinterpret_cast<const int*>(w_trailer)[0];
  const float xscale = pow2f3(ex), osc = pow2f3(-ex) * pow2f3(-ew);
  if (tid == 0) smax = 0u;

  __amdgpu_buffer_rsrc_t x_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x + (long long)n * k.Cin * S), 0, (unsigned)((long long)k.Cin * S * 4), 0x00020000);
  const unsigned s4 = (unsigned)S * 4u;
  constexpr int NPART = VEC ? 8 : NS;
  unsigned gbyte[NS];
  unsigned gq = OOB, gh = OOB;
  int posq = -1, posh = -1;
#define C3M_OFFS(t_)                                                                              \
  {                                                                                               \
    const int yt0_ = y0 + (t_) * TY;                                                              \
    if constexpr (VEC) {                                                                          \
      gq = OOB; gh = OOB;                                                                         \
      if (tid < 240) {                                                                            \
        const int row = tid >> 2, q = tid & 3;                                                    \
        const int hz = row / HY, hy = row % HY;                                                   \
        const int gz = z0 - 1 + hz, gy = yt0_ - 1 + hy, gx = x0 + 4 * q;                          \
        posq = row * HX + 1 + 4 * q;                                                              \
        if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && gx < k.W)             \
          gq = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;                                       \
      }                                                                                           \
      if (tid < 120) {                                                                            \
        const int row = tid >> 1, side = tid & 1;                                                 \
        const int hz = row / HY, hy = row % HY;                                                   \
        const int gz = z0 - 1 + hz, gy = yt0_ - 1 + hy, gx = side ? x0 + TX : x0 - 1;             \
        posh = row * HX + (side ? HX - 1 : 0);                                                    \
        if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W) \
          gh = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;                                       \
      }                                                                                           \
    } else {                                                                                      \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                            \
        const int pos = tid + 256 * s;                                                            \
        unsigned off = OOB;                                                                       \
        if (pos < XP) {                                                                           \
          const int hx = pos % HX, t = pos / HX, hy = t % HY, hz = t / HY;                        \
          const int gz = z0 - 1 + hz, gy = yt0_ - 1 + hy, gx = x0 - 1 + hx;                       \
          if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W) \
            off = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;                                    \
        }                                                                                         \
        gbyte[s] = off;                                                                           \
      }                                                                                           \
    }                                                                                             \
  }
  C3M_OFFS(0)

  // this lane's B positions (plane wid, row j, column l15) and the patch offsets of its taps 4 ks + kg
  // ROW-REUSE k-steps (round 4).  The 27 taps are walked as two sets of four (dz, dx) pairs -- S0 = pairs 0-3, S1 = pairs 4-7,
  // pair p = (dz = p / 3, dx = p % 3), k group kg <-> pair 4 S + kg -- times the three dy, plus one mixed k-step for pair 8
  // (dz 2, dx 2: k group kg <-> dy = kg, group 3 = the zero tap 27).  Within a set the B operand of (patch row r) serves
  // tile row j = r - dy for dy = 0, 1, 2: 10 row reads feed the 24 (row, dy) products of a set instead of 24 reads --
  // 28 B reads per chunk instead of 56 (the LDS, not the matrix pipe, was what this kernel waited for: 75 % LDS cycles
  // at 100 % pipe, 39 % pipe busy measured).  Same 7 x 8 x 3 MFMAs per chunk; the order of the taps in the sum changes.
  const int prow0 = (wid * HY) * HX + l15;                    // patch row r of this wave's plane: prow0 + r * HX
  int soff[2], tb[2];
#pragma unroll
  for (int S = 0; S < 2; ++S) {
    const int pr = 4 * S + kg;
    soff[S] = (pr / 3) * HY * HX + pr % 3;
    tb[S] = (pr / 3) * 9 + pr % 3;                            // tap of (pair, dy) = tb + 3 dy
  }
  const int moff = 2 * HY * HX + (kg < 3 ? kg : 2) * HX + 2;   // mixed k-step: (dz 2, dy kg, dx 2); group 3 reads any valid unit
  const int mtap = kg < 3 ? 20 + 3 * kg : 27;

  f32x4_3 acc[TT][NJ];
  float rx[NS][8];
  u32x4 rq[8];
  u32x4 rw[NW];
  const __amdgpu_buffer_rsrc_t w_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(wsp), 0, (unsigned)(k.nchunk * WU * 16), 0x00020000);
#define C3M_GLOAD_X(ch_, s_)                                                                      \
  {                                                                                               \
    const unsigned cbase = (unsigned)((ch_) * 8) * s4;                                            \
    if constexpr (VEC) {                                                                          \
      const unsigned co_ = cbase + (unsigned)(s_) * s4;                                           \
      rq[s_] = __builtin_amdgcn_raw_buffer_load_b128(x_src, gq == OOB ? OOB : gq + co_, 0, 0);    \
      rx[0][s_] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_src, gh == OOB ? OOB : gh + co_, 0, 0)); \
    } else {                                                                                      \
      _Pragma("unroll") for (int c = 0; c < 8; ++c)                                               \
        rx[s_][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(                         \
            x_src, gbyte[s_] == OOB ? OOB : gbyte[s_] + cbase + (unsigned)c * s4, 0, 0));         \
    }                                                                                             \
  }
#define C3M_GLOAD_W(ch_)                                                                          \
  _Pragma("unroll") for (int j = 0; j < NW; ++j)                                                  \
    rw[j] = __builtin_amdgcn_raw_buffer_load_b128(w_src, (tid + 256 * j) < WU ? (unsigned)(((ch_) * WU + tid + 256 * j) * 16) : OOB, 0, 0);
#define C3M_SPLIT8(v_, h_, r_)                                                                    \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                 \
    unsigned hh, rr;                                                                              \
    split_pair3(v_[2 * q], v_[2 * q + 1], xscale, hh, rr);                                        \
    h_[q] = hh; r_[q] = rr;                                                                       \
  }
#define C3M_LSTORE()                                                                              \
  {                                                                                               \
    if constexpr (VEC) {                                                                          \
      if (posq >= 0) {                                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                           \
          float v[8];                                                                             \
          _Pragma("unroll") for (int c = 0; c < 8; ++c) v[c] = __uint_as_float(rq[c][e]);         \
          u32x4 h, r;                                                                             \
          C3M_SPLIT8(v, h, r)                                                                     \
          Xs[posq + e] = h;                                                                       \
          Xs[XP + posq + e] = r;                                                                  \
        }                                                                                         \
      }                                                                                           \
      if (posh >= 0) {                                                                            \
        u32x4 h, r;                                                                               \
        C3M_SPLIT8(rx[0], h, r)                                                                   \
        Xs[posh] = h;                                                                             \
        Xs[XP + posh] = r;                                                                        \
      }                                                                                           \
    } else {                                                                                      \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                            \
        const int pos = tid + 256 * s;                                                            \
        if (pos < XP) {                                                                           \
          u32x4 h, r;                                                                             \
          C3M_SPLIT8(rx[s], h, r)                                                                 \
          Xs[pos] = h;                                                                            \
          Xs[XP + pos] = r;                                                                       \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
  }
#define C3M_LSTORE_W() _Pragma("unroll") for (int j = 0; j < NW; ++j) if (tid + 256 * j < WU) Ws[tid + 256 * j] = rw[j];
#ifndef C3M_KO
#define C3M_KO 0          // knock-out builds (timing only): 1 no global X loads, 2 no conversion + LDS stores, 4 no MFMAs, 8 no epilogue stores
#endif
#define C3M_PART(p_) { if ((p_) < NPART && (!(C3M_KO & 1) || k.D < 0)) C3M_GLOAD_X(nch, p_); }
#define C3M_MMA3(acc_, ah_, al_, bh_, bl_)                                                        \
  { if (!(C3M_KO & 4) || k.D < 0) { acc_ = mma16(bh_, al_, acc_); acc_ = mma16(bl_, ah_, acc_); acc_ = mma16(bh_, ah_, acc_); } }   /* rows = the 16 voxels of a tile row, columns = output channels */
#pragma unroll
  for (int s = 0; s < NPART; ++s) C3M_GLOAD_X(0, s);
  C3M_GLOAD_W(0);
  C3M_LSTORE();
  C3M_LSTORE_W();
  __syncthreads();

  float pm = 0.f;
  int cn = n, cz0 = z0, cy0 = y0, cx0 = x0;
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[t][j] = f32x4_3{0.f, 0.f, 0.f, 0.f};
  int it = 0, ch = 0;
  for (int q = 0; q < niter * k.nchunk; ++q) {
#pragma unroll
   for (int tt = 0; tt < TT; ++tt) {
    const bool tile_end = (ch + 1 == k.nchunk) && (tt + 1 == TT);
    const bool more = !tile_end || (it + 1 < niter);
    const int nch = tile_end ? 0 : ((tt + 1 < TT) ? ch : ch + 1);
    const int wch = tile_end ? 0 : ch + 1;
    if (tile_end && more) {
      C3M_DECODE(t_first + (long long)(it + 1) * J)
      x_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)n * k.Cin * S), 0,
                                                (unsigned)((long long)k.Cin * S * 4), 0x00020000);
    }
    if (TT > 1 || tile_end) C3M_OFFS((tt + 1) % TT)
#pragma unroll
    for (int S = 0; S < 2; ++S) {
      u32x4 Ah[3], Al[3], Bh[2], Bl[2];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        Ah[dy] = Ws[(tb[S] + 3 * dy) * 16 + l15];
        Al[dy] = Ws[448 + (tb[S] + 3 * dy) * 16 + l15];
      }
      Bh[0] = Xs[prow0 + soff[S]];
      Bl[0] = Xs[XP + prow0 + soff[S]];
#pragma unroll
      for (int r = 0; r < NJ + 2; ++r) {
        if (r + 1 < NJ + 2) {
          Bh[(r + 1) & 1] = Xs[prow0 + (r + 1) * HX + soff[S]];
          Bl[(r + 1) & 1] = Xs[XP + prow0 + (r + 1) * HX + soff[S]];
        }
        if ((r & 1) == 0 && r < 8) C3M_PART(4 * S + (r >> 1))
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int j = r - dy;
          if (j >= 0 && j < NJ) C3M_MMA3(acc[tt][j], Ah[dy], Al[dy], Bh[r & 1], Bl[r & 1])
        }
      }
    }
    {
      const u32x4 Ah = Ws[mtap * 16 + l15], Al = Ws[448 + mtap * 16 + l15];
      u32x4 Bh[2], Bl[2];
      Bh[0] = Xs[prow0 + moff];
      Bl[0] = Xs[XP + prow0 + moff];
      if (tt + 1 == TT) C3M_GLOAD_W(wch);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (j + 1 < NJ) {
          Bh[(j + 1) & 1] = Xs[prow0 + (j + 1) * HX + moff];
          Bl[(j + 1) & 1] = Xs[XP + prow0 + (j + 1) * HX + moff];
        }
        C3M_MMA3(acc[tt][j], Ah, Al, Bh[j & 1], Bl[j & 1])
      }
    }
    if (more) {
      __syncthreads();
      if (!(C3M_KO & 2) || k.D < 0) C3M_LSTORE();
      if (tt + 1 == TT) { C3M_LSTORE_W(); }
      __syncthreads();
    }
   }
   const bool tile_done = (ch + 1 == k.nchunk);
   if (!tile_done) { ++ch; continue; }
   // ---- epilogue: the MFMAs ran with rows = voxels, columns = output channels: acc[tt][j][i] <-> cout l15, voxel
   // (cz0 + wid, yt0 + j, cx0 + 4 kg + i) -- one 16-byte store (and one 16-byte load of the activation source) per tile
   // row and lane instead of four 4-byte ones (VEC: W % 4 == 0; otherwise element by element)
   __builtin_amdgcn_sched_barrier(0);
   const bool cok = l15 < k.cout_used;
   const float bv = (bias && cok) ? bias[l15] : 0.f;
   const __amdgpu_buffer_rsrc_t y_dst = __builtin_amdgcn_make_buffer_rsrc(
       y + (long long)cn * k.Cout * S, 0, (unsigned)((long long)k.Cout * S * 4), 0x00020000);
   const __amdgpu_buffer_rsrc_t a_src = __builtin_amdgcn_make_buffer_rsrc(
       const_cast<float*>((k.act_src ? k.act_src : y) + (long long)cn * k.Cout * S), 0,
       (unsigned)((long long)k.Cout * S * 4), 0x00020000);
#pragma unroll
   for (int tt = 0; tt < TT; ++tt) {
     const int yt0 = cy0 + tt * TY;
     const int gz = cz0 + wid, gx = cx0 + 4 * kg;
     const unsigned cbyte = (unsigned)l15 * s4;
     u32x4 av[2];
#define C3M_AVLOAD(set_, j_)                                                                      \
     if (k.act_src) {                                                                             \
       const int gy_ = yt0 + (j_);                                                                \
       const bool vok_ = cok && gz < k.D && gy_ < k.H;                                            \
       const unsigned vo_ = (unsigned)((gz * k.H + gy_) * k.W + gx) * 4u + cbyte;                 \
       if constexpr (VEC) av[set_] = __builtin_amdgcn_raw_buffer_load_b128(a_src, (vok_ && gx < k.W) ? vo_ : OOB, 0, 0); \
       else {                                                                                     \
         _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
           av[set_][i] = __builtin_amdgcn_raw_buffer_load_b32(a_src, (vok_ && gx + i < k.W) ? vo_ + 4u * i : OOB, 0, 0); \
       }                                                                                          \
     }
     C3M_AVLOAD(0, 0)
#pragma unroll
     for (int j = 0; j < NJ; ++j) {
       if (j + 1 < NJ) C3M_AVLOAD((j + 1) & 1, j + 1)
       const int gy = yt0 + j;
       const bool vok = cok && gz < k.D && gy < k.H;
       const unsigned vo = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u + cbyte;
       u32x4 out;
#pragma unroll
       for (int i = 0; i < 4; ++i) {
         const bool ok = vok && gx + i < k.W;
         const float a = __uint_as_float(av[j & 1][i]);
         float v = acc[tt][j][i] * osc + bv;
         if (k.av_mode == 2) v += a;
         if (k.act == 1) v = v > 0.f ? v : v * k.slope;
         else if (k.act == 2) v = tanhf(v);
         if (k.av_mode == 1) v = a > 0.f ? v : v * k.act_slope;
         out[i] = __float_as_uint(v);
         pm = fmaxf(pm, ok ? fabsf(v) : 0.f);
       }
       if (!(C3M_KO & 8) || k.D < 0) {
         if constexpr (VEC) __builtin_amdgcn_raw_buffer_store_b128(out, y_dst, (vok && gx < k.W) ? vo : OOB, 0, 0);
         else {
#pragma unroll
           for (int i = 0; i < 4; ++i)
             __builtin_amdgcn_raw_buffer_store_b32(out[i], y_dst, (vok && gx + i < k.W) ? vo + 4u * i : OOB, 0, 0);
         }
       }
     }
#undef C3M_AVLOAD
   }
#pragma unroll
   for (int t = 0; t < TT; ++t)
#pragma unroll
     for (int j = 0; j < NJ; ++j) acc[t][j] = f32x4_3{0.f, 0.f, 0.f, 0.f};
   cn = n; cz0 = z0; cy0 = y0; cx0 = x0;
   ch = 0; ++it;
  }
#undef C3M_DECODE
#undef C3M_OFFS
#undef C3M_GLOAD_X
#undef C3M_GLOAD_W
#undef C3M_SPLIT8
#undef C3M_LSTORE
#undef C3M_LSTORE_W
#undef C3M_PART
#undef C3M_MMA3
  if (y_amax) {
    __syncthreads();
    publish_block_absmax_acc(pm, &smax, y_amax);
  }
}

// ------------------------------------------------------------------------------------------------
static bool split3d_off() {
  static DfOptFlag a{"DFMIR_CONV3D_FP32"}, b{"DFMIR_CONV_FP32"};
  return a.get() || b.get();
}
static bool pair3d_off() {
  static DfOptFlag o{"DFMIR_CONV3D_NO_PAIR"};
  return o.get();
}
static bool m16_off() {       // A/B switch: <= 16 output channels through the plane-pair form instead of the 16-row MFMA form
  static DfOptFlag o{"DFMIR_CONV3D_NO_M16"};
  return o.get();
}
static bool split3d_geom_ok(const DfConvGeom* g) {
  return g->KD == 3 && g->KH == 3 && g->KW == 3 && g->stride == 1 && g->dil == 1 && g->pd == 1 && g->ph == 1 &&
         g->pw == 1 && g->pad_mode == 0 && g->Do == g->Di && g->Ho == g->Hi && g->Wo == g->Wi && g->Di > 1 &&
         g->Cin >= 1 && g->Cout >= 1 && (long long)(g->Cin > g->Cout ? g->Cin : g->Cout) * g->Di * g->Hi * g->Wi * 4 < 0x7FFFFFFFLL;   // buffer offsets, OOB marker
}
extern "C" int dfmir_conv3d_split_ok(const DfConvGeom* g) { return (g && !split3d_off() && split3d_geom_ok(g)) ? 1 : 0; }
extern "C" long long dfmir_conv3d_split_ws_floats(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return -1;
  return (long long)((Cout + 31) / 32) * ((Cin + 7) / 8) * 2 * 36 * 32 * 4 + 4;     // 36: the PAIR form's taps'
}
static int conv3d_split_fwd_impl(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                 const float* w_tcc, float* ws, const float* bias, float* y, float* y_amax,
                                 int cout_used, void* stream, const float* act_src = nullptr, float act_slope = 0.f,
                                 int av_mode = 0, int w_ktot = 0, int w_koff = 0);
extern "C" int dfmir_conv3d_split_fwd_actgrad(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                              const float* w_tcc, float* ws, const float* bias, float* y, float* y_amax,
                                              int cout_used, const float* act_src, float act_slope, void* stream) {
  DF_ARG_CHECK(g && cout_used > 0 && cout_used <= g->Cout && act_src && g->act == 0);
  return conv3d_split_fwd_impl(g, x, x_amax, x_amax_n, w_tcc, ws, bias, y, y_amax, cout_used, stream, act_src, act_slope, 1);
}
extern "C" int dfmir_conv3d_split_fwd(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                      const float* w_tcc, float* ws, const float* bias, float* y, float* y_amax,
                                      void* stream) {
  return conv3d_split_fwd_impl(g, x, x_amax, x_amax_n, w_tcc, ws, bias, y, y_amax, g ? g->Cout : 0, stream);
}
extern "C" int dfmir_conv3d_split_fwd_sub(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                          const float* w_tcc, float* ws, const float* bias, float* y, float* y_amax,
                                          int cout_used, void* stream) {
  DF_ARG_CHECK(g && cout_used > 0 && cout_used <= g->Cout);
  return conv3d_split_fwd_impl(g, x, x_amax, x_amax_n, w_tcc, ws, bias, y, y_amax, cout_used, stream);
}
static int conv3d_split_fwd_impl(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                 const float* w_tcc, float* ws, const float* bias, float* y, float* y_amax,
                                 int cout_used, void* stream, const float* act_src, float act_slope, int av_mode,
                                 int w_ktot, int w_koff) {
  // w_tcc == NULL: ws already holds the split of this layer's weights for this cout_used (a host that keeps ws per
  // (layer, mode) and re-splits only after an optimizer step saves the conv3d_wsplit_k launch of every call)
  DF_ARG_CHECK(g && x && x_amax && x_amax_n > 0 && ws && y);
  DF_ARG_CHECK(!split3d_off() && split3d_geom_ok(g) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = (g->Cin + 7) / 8, nmt = (g->Cout + 31) / 32;
  float* trailer = ws + dfmir_conv3d_split_ws_floats(g->Cin, g->Cout) - 4;          // after the largest unit layout
  const bool pair = cout_used <= 16 && !pair3d_off();
  const bool m16 = pair && !m16_off();
  if (w_tcc) {
    // one unit per thread where possible: every workgroup re-reduces max|w| itself (L2-resident), the packing is what
    // parallelises (8 workgroups took 25 us on the 64 -> 64 layers, a latency chain of strided loads)
    static DfOptInt ws_o{"DFMIR_WSPLIT_WGS", 32};
    const int ws_wgs = ws_o.get();
    const long long units = (long long)(pair ? 1 : nmt) * nchunk * (pair ? 36 : 28) * 32;
    long long nwg = (units + 1023) / 1024;
    if (nwg > ws_wgs) nwg = ws_wgs;
    if (nwg < 1) nwg = 1;
    conv3d_wsplit_k<<<(unsigned)nwg, 1024, 0, st>>>(w_tcc, reinterpret_cast<u32x4*>(ws), g->Cin, pair ? cout_used : g->Cout, trailer,
                                                    m16 ? 2 : (pair ? 1 : 0), w_ktot > 0 ? w_ktot : g->Cin, w_koff);
    DF_LAUNCH_CHECK();
  }
  C3sP k{g->N, g->Cin, g->Cout, g->Di, g->Hi, g->Wi, g->act, g->slope, (g->Di + 3) / 4, (g->Hi + 7) / 8, (g->Wi + 15) / 16,
         nchunk, x_amax_n, cout_used, 0, act_src, act_slope, act_src ? av_mode : 0};
  // vec: 16-byte loads of the patch rows and (16-row form) 16-byte stores of the result / loads of the activation source
  static DfOptFlag novec_o{"DFMIR_CONV3D_NO_VEC"};
  const bool vec = (g->Wi % 4) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                                         reinterpret_cast<uintptr_t>(act_src)) & 15) == 0 && !novec_o.get();
  // plane-pair form: three y-stacked tiles per workgroup share one staging of each chunk's weights
  static DfOptFlag multi_o{"DFMIR_CONV3D_NO_MULTI"};
  const bool multi_off = multi_o.get();
  // (the 16-row form runs one tile per workgroup: 143 registers = three workgroups per CU beat the shared weight staging
  // of the three-tile form, 32->16: 0.85 -> 0.80 ms, 16->16: 0.55 -> 0.48 ms)
  const int tt = (pair && !m16 && vec && !multi_off && k.ny >= 3) ? 3 : 1;
  k.ny = (k.ny + tt - 1) / tt;
  k.ntile = (long long)g->N * k.nz * k.ny * k.nx;
  // One tile (group) per workgroup by default.  DFMIR_CONV3D_WGS=<n> caps the workgroups (512 = two per CU = fully
  // persistent: each then walks several tiles with the next tile's first chunk prefetched under the last phase of the
  // current one).  Measured equal within the box-to-box noise on every layer shape (the kernel is bound by the
  // package power, not by the exposed prologue), and the dispatcher balances one-tile workgroups better.
  const unsigned gy = pair ? 1u : (unsigned)((cout_used + 31) / 32);
  static DfOptInt wg_o{"DFMIR_CONV3D_WGS", 1 << 30};
  const int wg_cap = wg_o.get();
  long long nb = 8 * ((k.ntile + 7) / 8);
  const long long cap = 8 * (((long long)wg_cap / gy + 7) / 8);
  if (nb > cap) nb = cap;
  const dim3 grid((unsigned)nb, gy);
  const u32x4* wsu = reinterpret_cast<const u32x4*>(ws);
  if (m16 && vec) conv3d_split_m16_k<true, 1><<<grid, 256, 0, st>>>(x, x_amax, wsu, trailer, bias, y, y_amax, k);
  else if (m16) conv3d_split_m16_k<false, 1><<<grid, 256, 0, st>>>(x, x_amax, wsu, trailer, bias, y, y_amax, k);
  else if (pair && vec && tt == 3) conv3d_split_k<true, true, 3><<<grid, 256, 0, st>>>(x, x_amax, wsu, trailer, bias, y, y_amax, k);
  else if (pair && vec) conv3d_split_k<true, true, 1><<<grid, 256, 0, st>>>(x, x_amax, wsu, trailer, bias, y, y_amax, k);
  else if (pair) conv3d_split_k<true, false, 1><<<grid, 256, 0, st>>>(x, x_amax, wsu, trailer, bias, y, y_amax, k);
  else if (vec) conv3d_split_k<false, true, 1><<<grid, 256, 0, st>>>(x, x_amax, wsu, trailer, bias, y, y_amax, k);
  else conv3d_split_k<false, false, 1><<<grid, 256, 0, st>>>(x, x_amax, wsu, trailer, bias, y, y_amax, k);
  DF_LAUNCH_CHECK();
  return 0;
}

// ================================================================================================
// conv(cat(nearest_up2(a), b)) WITHOUT the up-sampled tensor and without multiplying by duplicated values
// (torchvoxelmorph/networks.py:64,97-100: nn.Upsample(scale 2, nearest) + torch.cat feeding the next ConvBlock).
//
// A 3x3x3 tap of an output voxel 2V + p (p = its parity per axis) over nearest_up2(a) reads a[V + o] with
// o = floor((p + d - 1) / 2) in {p - 1, p}: the 27 taps fall on 2 x 2 x 2 low-resolution voxels, so the up-sampled
// share of the convolution is, per parity class p, an 8-tap convolution of `a` with the SUMMED weights
//     Weff[p][t] = sum of w[d] over the d that land on low-resolution offset p - 1 + t      (t in {0, 1} per axis)
//     p = 0:  t = 0 <- {d = 0},     t = 1 <- {d = 1, 2};      p = 1:  t = 0 <- {d = 0, 1},  t = 1 <- {d = 2}
// -- 8 / 27 of the products (the same sums in a different order: fp32 round-off only).  conv3d_up_phase_k computes
// that share: workgroup = one 4 x 8 x 16 tile of LOW-resolution voxels x 32 output channels x one (pz, py) pair
// (grid.z), both px classes (two accumulator sets, 8 k-steps of K = 2 x-taps x 8 channels per chunk); the halo patch
// of `a` is staged exactly as conv3d_split_k stages its input (same geometry: offsets -1 .. +1), the output voxels
// (2z + pz, 2y + py, 2x .. 2x + 1) leave as 8-byte stores (a half-wave row = 128 contiguous bytes).  The result is the
// PARTIAL sum; the skip channels b follow as an ordinary conv3d_split_k launch whose epilogue adds it (av_mode 2),
// the bias, the activation and the range probe.
// ================================================================================================
// w_tcc [27][Ktot][M] (forward packing); the up-sampled channels are rows 0 .. Ka - 1.
// ws[pzy 4][mtile][chunk][split 2][16 units u = px*8 + tz*4 + ty*2 + tx][32] x (8 channels x fp16); trailer[0] = ew.
__device__ __forceinline__ float up_weff(const float* __restrict__ w, int Ktot, int M, int kk, int mo, int pz, int py,
                                         int px, int tz, int ty, int tx) {
  // taps d of axis with parity p that land on slot t:  lo = (p == 0) ? t : 2 t,  n = (t == p) ? 1 : 2 ... spelled out:
  const int z0 = pz ? (tz ? 2 : 0) : (tz ? 1 : 0), zn = (pz == tz) ? 1 : 2;
  const int y0 = py ? (ty ? 2 : 0) : (ty ? 1 : 0), yn = (py == ty) ? 1 : 2;
  const int x0 = px ? (tx ? 2 : 0) : (tx ? 1 : 0), xn = (px == tx) ? 1 : 2;
  float s = 0.f;
  for (int a = 0; a < zn; ++a)
    for (int b = 0; b < yn; ++b)
      for (int c = 0; c < xn; ++c)
        s += w[((long long)(((z0 + a) * 3 + (y0 + b)) * 3 + (x0 + c)) * Ktot + kk) * M + mo];
  return s;
}
// Cb in {1, 2} skip channels (the network's input images at the top level): a second section after the phase units,
// [mtile][split 2][10 tap rows t = dz*3 + dy (9 = padding)][32] x (4 x-taps (3 = padding) x 2 channels x fp16) with its own
// scale trailer[1] -- the skip share then runs INSIDE conv3d_up_phase_k (one MFMA k-step = 2 tap rows x 4 x 2).
__device__ __forceinline__ void conv3d_up_wsplit_body(const float* __restrict__ w, u32x4* __restrict__ ws, int Ka, int Ktot,
                                                      int M, float* __restrict__ trailer, int Cb, float* sm) {
  // scale from a BOUND of the effective weights: an effective weight sums <= 8 taps, so |Weff| <= 8 max|w| (evaluating
  // all 64 Ka M sums in every workgroup just for their maximum cost 0.2 ms on the 64-channel levels).  A loose scale
  // only moves the fp16 pairs' exponent window: elements above 2^-14 of the bound keep their 22 bits
  float m = wslab_absmax(w, Ka, Ktot, 0, M);
  m = 8.f * block_max(m, sm);
  if (!(m == m)) m = __uint_as_float(0x7f800000u);
  const int ew = scale_exp3(m);
  const float s = pow2f3(ew);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<int*>(trailer)[0] = ew;
  const int nchunk = (Ka + 7) / 8, nmt = (M + 31) / 32;
  const int units = 4 * nmt * nchunk * 16 * 32;
  for (int u = blockIdx.x * 1024 + threadIdx.x; u < units; u += gridDim.x * 1024) {
    const int co = u & 31;
    int t = u >> 5;
    const int slot = t & 15; t >>= 4;
    const int ch = t % nchunk; t /= nchunk;
    const int mt = t % nmt, pzy = t / nmt;
    const int mo = mt * 32 + co;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int kk = ch * 8 + c;
      v[c] = (kk < Ka && mo < M) ? up_weff(w, Ktot, M, kk, mo, pzy >> 1, pzy & 1, slot >> 3, (slot >> 2) & 1, (slot >> 1) & 1, slot & 1)
                                 : 0.f;
    }
    u32x4 h, r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned hh, rr;
      split_pair3(v[2 * q], v[2 * q + 1], s, hh, rr);
      h[q] = hh; r[q] = rr;
    }
    const long long base = ((((long long)pzy * nmt + mt) * nchunk + ch) * 2) * 512;
    ws[base + slot * 32 + co] = h;
    ws[base + 512 + slot * 32 + co] = r;
  }
  if (Cb <= 0) return;
  __syncthreads();
  float ms = 0.f;
  for (int i = threadIdx.x; i < 27 * Cb * M; i += 1024) {
    const int mo = i % M, c = (i / M) % Cb, tap = i / (M * Cb);
    ms = fmaxf(ms, fabsf(w[((long long)tap * Ktot + Ka + c) * M + mo]));
  }
  ms = block_max(ms, sm);
  if (!(ms == ms)) ms = __uint_as_float(0x7f800000u);
  const int es = scale_exp3(ms);
  const float ss = pow2f3(es);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<int*>(trailer)[1] = es;
  u32x4* wsk = ws + 2LL * units;                          // after the phase units (h and r of every (slot, cout))
  for (int u = blockIdx.x * 1024 + threadIdx.x; u < nmt * 10 * 32; u += gridDim.x * 1024) {
    const int co = u & 31, t = (u >> 5) % 10, mt = (u >> 5) / 10;
    const int mo = mt * 32 + co;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int dx = i >> 1, c = i & 1;
      v[i] = (t < 9 && dx < 3 && c < Cb && mo < M) ? w[((long long)(t * 3 + dx) * Ktot + Ka + c) * M + mo] : 0.f;
    }
    u32x4 h, r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned hh, rr;
      split_pair3(v[2 * q], v[2 * q + 1], ss, hh, rr);
      h[q] = hh; r[q] = rr;
    }
    wsk[(long long)mt * 640 + t * 32 + co] = h;
    wsk[(long long)mt * 640 + 320 + t * 32 + co] = r;
  }
}
__global__ __launch_bounds__(1024) void conv3d_up_wsplit_k(const float* __restrict__ w, u32x4* __restrict__ ws, int Ka,
                                                           int Ktot, int M, float* __restrict__ trailer, int Cb) {
  __shared__ float sm[17];
  conv3d_up_wsplit_body(w, ws, Ka, Ktot, M, trailer, Cb, sm);
}

struct C3uP {
  int N, Ca, Cout, D, H, W;      // D, H, W: the LOW-resolution volume (a); y is [N, Cout, 2D, 2H, 2W]
  int nz, ny, nx, nchunk, x_n;
  long long ntile;
  // SKIP2 form: the <= 2 skip channels b [N, Cb, 2D, 2H, 2W] inside the same launch, then bias, activation, range probe
  const float* b;
  const float* b_amax;
  int b_n, Cb;
  const float* bias;
  int act;
  float slope;
  float* y_amax;
};

// SKIP2 (Cb <= 2: the two input images of the top level): after the up-sampled channels the workgroup stages the skip
// patch it needs -- full-resolution planes 2 z0 + pz - 1 .. + 8, rows 2 y0 + py - 1 .. + 16, columns 2 x0 - 1 .. 2 x0 + 33
// as [split][9][17][40] x (2 channels x fp16) -- and runs 2 px x 5 k-steps with K = 2 tap rows (dz, dy) x 4 x-taps x 2
// channels (operand = 4 consecutive 4-byte LDS words); the accumulators are first moved to the skip products' scale by
// an exact power of two.  No partial sum ever goes to memory.
template <bool SKIP2>
__global__ __launch_bounds__(256, 2) void conv3d_up_phase_k(const float* __restrict__ x, const float* __restrict__ x_amax,
                                                         const u32x4* __restrict__ wsp, const float* __restrict__ w_trailer,
                                                         float* __restrict__ y, C3uP k) {
  constexpr int TZ = 4, TY = 8, TX = 16, HY = TY + 2, HX = TX + 2;
  constexpr int XP = (TZ + 2) * HY * HX;                  // 1080 positions
  constexpr int WU = 2 * 16 * 32;                         // 1024 16-B units of one chunk's weights (2 px x 8 slots)
  constexpr int NW = WU / 256;                            // 4
  constexpr int NJ = 4;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int SRS = 40, SROWS = 9 * 17, SPW = SROWS * SRS;       // skip patch: row stride (words), rows, words per split
  constexpr int XSU = SKIP2 ? ((2 * SPW + 3) / 4 > 2 * XP ? (2 * SPW + 3) / 4 : 2 * XP) : 2 * XP;
  __shared__ u32x4 Xs[XSU];
  __shared__ u32x4 Ws[WU];
  __shared__ float red[17];
  __shared__ unsigned smax;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const long long S = (long long)k.D * k.H * k.W;
  // tiles: XCD-contiguous eighths, x fastest, then z, then y (as conv3d_split_k); one tile per workgroup
  // The four (pz, py) instances of a tile sit in neighbouring workgroup ids of the SAME XCD (round 5): as the outermost grid
  // dimension they were four passes over the volume, each re-reading the patch of `a` from memory (fetch 887 MB for 165 MB
  // of inputs).
  const long long per_xcd = (k.ntile + 7) / 8;
  const unsigned rid = blockIdx.x >> 3;
  const long long tile = (long long)(blockIdx.x & 7) * per_xcd + (rid >> 2);
  if ((long long)(rid >> 2) >= per_xcd || tile >= k.ntile) return;
  int n, z0, y0, x0;
  {
    long long pid = tile;
    const int bx = (int)(pid % k.nx); pid /= k.nx;
    const int bz = (int)(pid % k.nz); pid /= k.nz;
    const int by = (int)(pid % k.ny);
    n = (int)(pid / k.ny);
    z0 = bz * TZ; y0 = by * TY; x0 = bx * TX;
  }
  const int mt = blockIdx.y, pzy = (int)(rid & 3), pz = pzy >> 1, py = pzy & 1;

  const float amax = reduce_absmax(x_amax, k.x_n, red);
  const int ex = scale_exp3(amax);
  const int ew = reinterpret_cast<const int*>(w_trailer)[0];
  const float xscale = pow2f3(ex), osc = pow2f3(-ex) * pow2f3(-ew);

  const __amdgpu_buffer_rsrc_t x_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x + (long long)n * k.Ca * S), 0, (unsigned)((long long)k.Ca * S * 4), 0x00020000);
  const unsigned s4 = (unsigned)S * 4u;
  // patch loads (the VEC form of conv3d_split_k): thread t < 240 owns quad q = t & 3 of halo row t >> 2 in all 8 channels
  // of the chunk, thread t < 120 additionally the left / right halo column of row t >> 1
  unsigned gq = OOB, gh = OOB;
  int posq = -1, posh = -1;
  if (tid < 240) {
    const int row = tid >> 2, q = tid & 3;
    const int hz = row / HY, hy = row % HY;
    const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 + 4 * q;
    posq = row * HX + 1 + 4 * q;
    if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && gx < k.W)
      gq = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;
  }
  if (tid < 120) {
    const int row = tid >> 1, side = tid & 1;
    const int hz = row / HY, hy = row % HY;
    const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = side ? x0 + TX : x0 - 1;
    posh = row * HX + (side ? HX - 1 : 0);
    if ((unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W)
      gh = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u;
  }
  const __amdgpu_buffer_rsrc_t w_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<u32x4*>(wsp + ((long long)pzy * gridDim.y + mt) * k.nchunk * WU), 0, (unsigned)(k.nchunk * WU * 16), 0x00020000);

  // B positions of this lane: column tile j = rows 2j, 2j + 1 of plane wid (see conv3d_split_k for the lane rotation);
  // tap slot (tz, ty, tx) of parity (pz, py, px) sits at patch offset ((pz + tz) HY + (py + ty)) HX + (px + tx)
  const int lx = (l31 - 2 * (l31 >> 4)) & 15;
  int pbase[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) pbase[j] = ((wid + pz) * HY + 2 * j + (l31 >> 4) + py) * HX + lx + hi;   // + tx = hi

  f32x16 acc[2][NJ];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][j][r] = 0.f;

  float rh[8];
  u32x4 stg[8 + NW];                                       // rq = stg[0..7] (the patch quads), rw = stg[8..]; the skip phase reuses them
#define rq stg
#define rw (stg + 8)
#define C3U_GLOAD_X(ch_, s_)                                                                      \
  {                                                                                               \
    const unsigned co_ = (unsigned)((ch_) * 8 + (s_)) * s4;                                       \
    rq[s_] = __builtin_amdgcn_raw_buffer_load_b128(x_src, gq == OOB ? OOB : gq + co_, 0, 0);      \
    rh[s_] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(x_src, gh == OOB ? OOB : gh + co_, 0, 0)); \
  }
#define C3U_GLOAD_W(ch_)                                                                          \
  _Pragma("unroll") for (int j = 0; j < NW; ++j)                                                  \
    rw[j] = __builtin_amdgcn_raw_buffer_load_b128(w_src, (unsigned)(((ch_) * WU + tid + 256 * j) * 16), 0, 0);
#define C3U_SPLIT8(v_, h_, r_)                                                                    \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                 \
    unsigned hh, rr;                                                                              \
    split_pair3(v_[2 * q], v_[2 * q + 1], xscale, hh, rr);                                        \
    h_[q] = hh; r_[q] = rr;                                                                       \
  }
#define C3U_LSTORE()                                                                              \
  {                                                                                               \
    if (posq >= 0) {                                                                              \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                             \
        float v[8];                                                                               \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) v[c] = __uint_as_float(rq[c][e]);           \
        u32x4 h, r;                                                                               \
        C3U_SPLIT8(v, h, r)                                                                       \
        Xs[posq + e] = h;                                                                         \
        Xs[XP + posq + e] = r;                                                                    \
      }                                                                                           \
    }                                                                                             \
    if (posh >= 0) {                                                                              \
      u32x4 h, r;                                                                                 \
      C3U_SPLIT8(rh, h, r)                                                                        \
      Xs[posh] = h;                                                                               \
      Xs[XP + posh] = r;                                                                          \
    }                                                                                             \
    _Pragma("unroll") for (int j = 0; j < NW; ++j) Ws[tid + 256 * j] = rw[j];                     \
  }
  // operands of k-step tp_ = px * 4 + tz * 2 + ty (tx = this half-wave): unit u = 2 tp_ + hi
#define C3U_OPLOAD(b_, tp_)                                                                       \
  {                                                                                               \
    const int toff = ((((tp_) >> 1) & 1) * HY + ((tp_) & 1)) * HX + ((tp_) >> 2);                 \
    const int u = 2 * (tp_) + hi;                                                                 \
    A0[b_] = Ws[u * 32 + l31];                                                                    \
    A1[b_] = Ws[512 + u * 32 + l31];                                                              \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                              \
      B0[b_][j] = Xs[pbase[j] + toff];                                                            \
      B1[b_][j] = Xs[XP + pbase[j] + toff];                                                       \
    }                                                                                             \
  }
  // SKIP2: the skip patch's global loads ride in the k-steps of the LAST up-sampled chunk (the patch registers of the
  // chunk pipeline are free by then); its conversion + LDS stores follow the barrier after that chunk
  constexpr int SNTASK = SROWS * 9, SNIT = (SNTASK + 255) / 256;
  static_assert(2 * SNIT <= 8 + NW, "the skip patch's quads live in the chunk pipeline's staging registers");
  u32x4 srw[SKIP2 ? 3 : 1];
  const int Dfs = 2 * k.D, Hfs = 2 * k.H, Wfs = 2 * k.W;
  const long long Sfs = (long long)Dfs * Hfs * Wfs;
  const __amdgpu_buffer_rsrc_t b_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>((SKIP2 ? k.b : x) + (SKIP2 ? (long long)n * k.Cb * Sfs : 0)), 0,
      SKIP2 ? (unsigned)((long long)k.Cb * Sfs * 4) : 0u, 0x00020000);
  const unsigned sb4 = (unsigned)Sfs * 4u;
  const int bz0 = 2 * z0 + pz - 1, by0 = 2 * y0 + py - 1, bx0 = 2 * x0;      // patch word 4 + i of a row = column bx0 + i
  const u32x4* wsk = wsp + 4LL * gridDim.y * k.nchunk * WU + (long long)mt * 640;
#define C3U_SKIP_GLOAD(i_)                                                                        \
  if constexpr (SKIP2) {                                                                          \
    if ((i_) < SNIT) {                                                                            \
      const int task = tid + 256 * (i_);                                                          \
      const int row = task / 9, q = task - 9 * row;                                               \
      const int gz = bz0 + row / 17, gy = by0 + row % 17, gx = bx0 + 4 * q;                       \
      const bool ok = task < SNTASK && (unsigned)gz < (unsigned)Dfs && (unsigned)gy < (unsigned)Hfs && gx < Wfs; \
      const unsigned off = ok ? (unsigned)((gz * Hfs + gy) * Wfs + gx) * 4u : OOB;                \
      stg[(i_) < SNIT ? (i_) : 0] = __builtin_amdgcn_raw_buffer_load_b128(b_src, off, 0, 0);      \
      stg[(i_) < SNIT ? SNIT + (i_) : 0] = __builtin_amdgcn_raw_buffer_load_b128(b_src, (ok && k.Cb > 1) ? off + sb4 : OOB, 0, 0); \
    } else if ((i_) == SNIT) {                                                                    \
      if (tid < SROWS) {                                                                          \
        const int gz = bz0 + tid / 17, gy = by0 + tid % 17, gx = bx0 - 1;                         \
        const bool ok = (unsigned)gz < (unsigned)Dfs && (unsigned)gy < (unsigned)Hfs && gx >= 0;  \
        const unsigned off = ok ? (unsigned)((gz * Hfs + gy) * Wfs + gx) * 4u : OOB;              \
        rh[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_src, off, 0, 0));          \
        rh[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_src, (ok && k.Cb > 1) ? off + sb4 : OOB, 0, 0)); \
      }                                                                                           \
    }                                                                                             \
  }
#ifndef C3U_SKIP_PREFETCH
#define C3U_SKIP_PREFETCH 0   // 1: the skip patch's loads ride in the last chunk's k-steps -- measured SLOWER (0.74 -> 0.82 ms:
#endif                        //    22 registers spill although the staging registers are shared); 0: loaded after that chunk
#ifndef C3U_KO
#define C3U_KO 0     // knock-out builds (timing only): 1 no epilogue stores, 2 no skip-channel phase, 4 no conversion + LDS stores, 8 no MFMAs
#endif
#ifndef C3U_SB
#define C3U_SB 1     // 1: operands of a k-step are read at its start (single register set: no scratch spills; the other
#endif               //    wave of the SIMD covers the LDS latency); 0: one step ahead (31 spilled registers)
  u32x4 A0[2], A1[2], B0[2][NJ], B1[2][NJ];
#pragma unroll
  for (int s = 0; s < 8; ++s) C3U_GLOAD_X(0, s);
  C3U_GLOAD_W(0);
  C3U_LSTORE();
  __syncthreads();

  for (int ch = 0; ch < k.nchunk; ++ch) {
    const bool more = ch + 1 < k.nchunk;
    if (!C3U_SB) C3U_OPLOAD(0, 0);
#pragma unroll
    for (int tp = 0; tp < 8; ++tp) {
      const int cur = C3U_SB ? 0 : (tp & 1);
      if (C3U_SB) C3U_OPLOAD(0, tp);
      if (more) {                                             // next chunk's loads, one channel per k-step (uniform branch)
        C3U_GLOAD_X(ch + 1, tp);
        if (tp == 7) C3U_GLOAD_W(ch + 1);
      } else if (C3U_SKIP_PREFETCH) {
        C3U_SKIP_GLOAD(tp)                                    // (SNIT = 6 task slots, then the halo column)
      }
      if (!C3U_SB && tp + 1 < 8) C3U_OPLOAD(cur ^ 1, tp + 1);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if ((C3U_KO & 8) && k.D > 0) { acc[tp >> 2][j][0] += __uint_as_float(A1[cur][0] ^ B0[cur][j][1] ^ A0[cur][2] ^ B1[cur][j][3]); continue; }
        acc[tp >> 2][j] = mma3(A1[cur], B0[cur][j], acc[tp >> 2][j]);
        acc[tp >> 2][j] = mma3(A0[cur], B1[cur][j], acc[tp >> 2][j]);
        acc[tp >> 2][j] = mma3(A0[cur], B0[cur][j], acc[tp >> 2][j]);
      }
#pragma unroll
      for (int i = 0; i < 3 * NJ; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
      }
    }
    if (more) {
      __syncthreads();
      if (!((C3U_KO & 4) && k.D > 0)) C3U_LSTORE();
      __syncthreads();
    }
  }
#undef C3U_GLOAD_X
#undef C3U_GLOAD_W
#undef rq
#undef rw
#undef C3U_SPLIT8
#undef C3U_LSTORE
#undef C3U_OPLOAD

  float osc_f = osc;
  if (SKIP2 && !((C3U_KO & 2) && k.D > 0)) {
    // ---- the skip channels: accumulators to the skip products' scale, stage the patch + weights, 2 x 5 k-steps
    __syncthreads();                                       // every wave is done with Xs / Ws
    const float bmax = reduce_absmax(k.b_amax, k.b_n, red);
    const int eb = scale_exp3(bmax);
    const int es = reinterpret_cast<const int*>(w_trailer)[1];
    const float bscale = pow2f3(eb);
    {
      int d = (eb + es) - (ex + ew);
      d = d > 120 ? 120 : (d < -120 ? -120 : d);
      const float rs = pow2f3(d);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[p][j][r] *= rs;
      osc_f = pow2f3(-eb) * pow2f3(-es);
    }
    unsigned* Xw = reinterpret_cast<unsigned*>(Xs);
    constexpr int NTASK = SNTASK, NIT = SNIT;
    if (!C3U_SKIP_PREFETCH) {
#pragma unroll
      for (int i = 0; i <= SNIT; ++i) C3U_SKIP_GLOAD(i)
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) srw[j] = (tid + 256 * j < 640) ? wsk[tid + 256 * j] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int task = tid + 256 * i;
      if (task < NTASK) {
        const int row = task / 9, q = task - 9 * row;
        u32x4 h, r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned hh, rr;
          split_pair3(__uint_as_float(stg[i][e]), __uint_as_float(stg[SNIT + i][e]), bscale, hh, rr);
          h[e] = hh; r[e] = rr;
        }
        *reinterpret_cast<u32x4*>(Xw + row * SRS + 4 + 4 * q) = h;
        *reinterpret_cast<u32x4*>(Xw + SPW + row * SRS + 4 + 4 * q) = r;
      }
    }
    if (tid < SROWS) {
      unsigned hh, rr;
      split_pair3(rh[0], rh[1], bscale, hh, rr);
      Xw[tid * SRS + 3] = hh;
      Xw[SPW + tid * SRS + 3] = rr;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (tid + 256 * j < 640) Ws[tid + 256 * j] = srw[j];
    __syncthreads();
    // lane's word offset of (plane 2 vz, row 2 vy, column word 3 + 2 vx): + (dz * 17 + dy) * SRS + px
    int sbase[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) sbase[j] = ((2 * wid) * 17 + 2 * (2 * j + (l31 >> 4))) * SRS + 3 + 2 * lx;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int t = (2 * ks + hi) < 9 ? 2 * ks + hi : 8;            // tap row 9 is padding (zero weights): any valid address
      const int toff = ((t / 3) * 17 + t % 3) * SRS;
      const u32x4 a0 = Ws[(2 * ks + hi) * 32 + l31], a1 = Ws[320 + (2 * ks + hi) * 32 + l31];
#pragma unroll
      for (int px = 0; px < 2; ++px) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const unsigned* p0 = Xw + sbase[j] + toff + px;
          const u32x4 b0{p0[0], p0[1], p0[2], p0[3]};
          const u32x4 b1{p0[SPW], p0[SPW + 1], p0[SPW + 2], p0[SPW + 3]};
          acc[px][j] = mma3(a1, b0, acc[px][j]);
          acc[px][j] = mma3(a0, b1, acc[px][j]);
          acc[px][j] = mma3(a0, b0, acc[px][j]);
        }
      }
    }
  }

#undef C3U_SKIP_GLOAD
  // ---- epilogue: acc[px][j][r] <-> cout row (r>>2)*8 + hi*4 + (r&3), low-resolution voxel (wid, 2j + (l31>>4), lx)
  // -> output voxels (2 z + pz, 2 y + py, 2 x + {0, 1}): one 8-byte store per (j, r)
  const int Dfo = 2 * k.D, Hfo = 2 * k.H, Wfo = 2 * k.W;
  const long long Sf = (long long)Dfo * Hfo * Wfo;
  const __amdgpu_buffer_rsrc_t y_dst = __builtin_amdgcn_make_buffer_rsrc(
      y + (long long)n * k.Cout * Sf, 0, (unsigned)((long long)k.Cout * Sf * 4), 0x00020000);
  const unsigned sf4 = (unsigned)Sf * 4u;
  const int gz = z0 + wid;
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = mt * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
    bv[r] = (SKIP2 && k.bias && co < k.Cout) ? k.bias[co] : 0.f;
  }
  float pm = 0.f;
  if (SKIP2 && tid == 0) smax = 0u;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int gy = y0 + 2 * j + (l31 >> 4), gx = x0 + lx;
    const bool vok = gz < k.D && gy < k.H && gx < k.W;
    const unsigned vo = (unsigned)(((2 * gz + pz) * Hfo + 2 * gy + py) * Wfo + 2 * gx) * 4u + (unsigned)(hi * 4) * sf4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rowu = (r >> 2) * 8 + (r & 3);
      const int cou = mt * 32 + rowu;
      const bool ok = vok && (cou + hi * 4) < k.Cout;
      float v0 = acc[0][j][r] * osc_f + bv[r], v1 = acc[1][j][r] * osc_f + bv[r];
      if (SKIP2) {
        if (k.act == 1) { v0 = v0 > 0.f ? v0 : v0 * k.slope; v1 = v1 > 0.f ? v1 : v1 * k.slope; }
        pm = fmaxf(pm, ok ? fmaxf(fabsf(v0), fabsf(v1)) : 0.f);
      }
      typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
      u32x2_ v2;
      v2[0] = __float_as_uint(v0);
      v2[1] = __float_as_uint(v1);
      __builtin_amdgcn_raw_buffer_store_b64(v2, y_dst, (ok && !((C3U_KO & 1) && k.D > 0)) ? vo : OOB, (unsigned)cou * sf4, 0);
    }
  }
  if (SKIP2 && k.y_amax) {
    __syncthreads();
    publish_block_absmax_acc(pm, &smax, k.y_amax);
  }
}

extern "C" long long dfmir_conv3d_up_ws_floats(int Ca, int Cout) {
  if (Ca <= 0 || Cout <= 0) return -1;
  return 4LL * ((Cout + 31) / 32) * ((Ca + 7) / 8) * 1024 * 4 + (long long)((Cout + 31) / 32) * 640 * 4 + 4;
}
extern "C" int dfmir_conv3d_up_ok(int N, int Ca, int Cout, int D, int H, int W) {
  // D, H, W: the low-resolution volume.  W % 4: 16-B patch loads; sizes: 32-bit buffer offsets on both tensors
  static DfOptFlag noup_o{"DFMIR_CONV3D_NO_UPPHASE"};
  if (split3d_off() || noup_o.get()) return 0;
  if (N <= 0 || Ca < 8 || (Ca & 7) || Cout < 8 || D < 2 || H < 2 || W < 4 || (W & 3)) return 0;
  if ((long long)Ca * D * H * W * 4 >= 0x7FFFFFFFLL || (long long)Cout * D * H * W * 8 * 4 >= 0x7FFFFFFFLL) return 0;
  return 1;
}
static int conv3d_up_launch(const float* a, const float* a_amax, int a_amax_n, const float* w_tcc, int Ktot, float* ws,
                            float* y, int N, int Ca, int Cout, int D, int H, int W, const float* b, const float* b_amax,
                            int b_n, int Cb, const float* bias, int act, float slope, float* y_amax, hipStream_t st) {
  const int nchunk = (Ca + 7) / 8, nmt = (Cout + 31) / 32;
  float* trailer = ws + dfmir_conv3d_up_ws_floats(Ca, Cout) - 4;
  if (w_tcc) {
    const long long units = 4LL * nmt * nchunk * 512;
    long long nwg = (units + 1023) / 1024;
    if (nwg > 32) nwg = 32;
    conv3d_up_wsplit_k<<<(unsigned)nwg, 1024, 0, st>>>(w_tcc, reinterpret_cast<u32x4*>(ws), Ca, Ktot, Cout, trailer, Cb);
    if (hipGetLastError() != hipSuccess) return -1;
  }
  C3uP k{N, Ca, Cout, D, H, W, (D + 3) / 4, (H + 7) / 8, (W + 15) / 16, nchunk, a_amax_n, 0,
         b, b_amax, b_n, Cb, bias, act, slope, y_amax};
  k.ntile = (long long)N * k.nz * k.ny * k.nx;
  const dim3 grid((unsigned)(32 * ((k.ntile + 7) / 8)), (unsigned)nmt, 1);
  if (Cb > 0) conv3d_up_phase_k<true><<<grid, 256, 0, st>>>(a, a_amax, reinterpret_cast<const u32x4*>(ws), trailer, y, k);
  else conv3d_up_phase_k<false><<<grid, 256, 0, st>>>(a, a_amax, reinterpret_cast<const u32x4*>(ws), trailer, y, k);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// y [N, Cout, 2D, 2H, 2W] <- the up-sampled channels' share of conv3x3x3(cat(nearest_up2(a), b)); a [N, Ca, D, H, W].
// w_tcc: the layer's forward packing [27][Ktot][Cout] (channels 0 .. Ca - 1 are a's), or NULL when ws already holds
// the phase split of the current weights.
extern "C" int dfmir_conv3d_up_fwd(const float* a, const float* a_amax, int a_amax_n, const float* w_tcc, int Ktot,
                                   float* ws, float* y, int N, int Ca, int Cout, int D, int H, int W, void* stream) {
  DF_ARG_CHECK(a && a_amax && a_amax_n > 0 && ws && y && dfmir_conv3d_up_ok(N, Ca, Cout, D, H, W));
  DF_ARG_CHECK((reinterpret_cast<uintptr_t>(ws) & 15) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(y) & 7) == 0 && (!w_tcc || Ktot >= Ca));
  if (conv3d_up_launch(a, a_amax, a_amax_n, w_tcc, Ktot, ws, y, N, Ca, Cout, D, H, W, nullptr, nullptr, 0, 0, nullptr, 0,
                       0.f, nullptr, (hipStream_t)stream) != 0)
    return df_set_error((int)hipErrorLaunchFailure, __FILE__, __LINE__);
  return 0;
}
// The whole layer in ONE launch when the skip tensor b [N, Cb, 2D, 2H, 2W] has Cb <= 2 channels (the network's input
// images at the top of the U-Net): y = act(conv3x3x3(cat(nearest_up2(a), b)) + bias), y_amax = its range probe.
// Ktot == Ca + Cb.  act: 0 none, 1 LeakyReLU(slope).
extern "C" int dfmir_conv3d_up_skip2_fwd(const float* a, const float* a_amax, int a_amax_n, const float* b,
                                         const float* b_amax, int b_amax_n, int Cb, const float* w_tcc, float* ws,
                                         const float* bias, float* y, float* y_amax, int N, int Ca, int Cout, int D, int H,
                                         int W, int act, float slope, void* stream) {
  DF_ARG_CHECK(a && a_amax && a_amax_n > 0 && b && b_amax && b_amax_n > 0 && ws && y && Cb >= 1 && Cb <= 2 && act >= 0 && act <= 1);
  DF_ARG_CHECK(dfmir_conv3d_up_ok(N, Ca, Cout, D, H, W) && (long long)Cb * D * H * W * 8 * 4 < 0x7FFFFFFFLL);
  DF_ARG_CHECK((reinterpret_cast<uintptr_t>(ws) & 15) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 &&
               (reinterpret_cast<uintptr_t>(b) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 7) == 0);
  if (conv3d_up_launch(a, a_amax, a_amax_n, w_tcc, Ca + Cb, ws, y, N, Ca, Cout, D, H, W, b, b_amax, b_amax_n, Cb, bias, act,
                       slope, y_amax, (hipStream_t)stream) != 0)
    return df_set_error((int)hipErrorLaunchFailure, __FILE__, __LINE__);
  return 0;
}
// ================================================================================================
// d(a) of the same layer, directly at LOW resolution: the adjoint of nearest_up2 (a 2x2x2 sum pool) composed with the
// conv's dgrad is a 4x4x4 stride-2 convolution of dy -- 64 taps per low-resolution voxel instead of 8 x 27.  In parity
// classes: with Yp[co, U] = dy[co, 2U + p],
//     d(a)[ci, V] = sum_p sum_{s in {0,1}^3} sum_co Weff[p][1 - s][ci, co] * Yp[co, V + s - p]
// conv3d_up_dgrad_k: workgroup = 2 x 8 x 16 low-resolution voxels x 32 input channels; for each (pz, py) and chunk of
// 8 output channels it reads the rows 2 (z0 - 1 + hz) + pz, 2 (y0 - 1 + hy) + py of dy ONCE (aligned quads), splits
// them into the two px sub-lattices' patches ([px][split][4 x 10 x 18 positions] x 16 B) and runs 2 x 4 k-steps
// (K = 2 x-slots x 8 channels).  No up-sampled gradient tensor (0.9 GB at 160x192x224), no pooling pass.
// ================================================================================================
// ws[pzy 4][mtile (ci)][chunk (co / 8)][split 2][16 units u = px*8 + sz*4 + sy*2 + sx][32 ci] x (8 co x fp16)
__device__ __forceinline__ void conv3d_up_wsplit_t_body(const float* __restrict__ w, u32x4* __restrict__ ws, int Ka, int Ktot,
                                                        int M, float* __restrict__ trailer, float* sm) {
  // scale from the same BOUND as conv3d_up_wsplit_body: |Weff| <= 8 max|w| (see there)
  float m = wslab_absmax(w, Ka, Ktot, 0, M);
  m = 8.f * block_max(m, sm);
  if (!(m == m)) m = __uint_as_float(0x7f800000u);
  const int ew = scale_exp3(m);
  const float s = pow2f3(ew);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<int*>(trailer)[0] = ew;
  const int nchunk = (M + 7) / 8, nmt = (Ka + 31) / 32;     // reduction = output channels, rows = a's channels
  const int units = 4 * nmt * nchunk * 16 * 32;
  for (int u = blockIdx.x * 1024 + threadIdx.x; u < units; u += gridDim.x * 1024) {
    const int ci_l = u & 31;
    int t = u >> 5;
    const int slot = t & 15; t >>= 4;
    const int ch = t % nchunk; t /= nchunk;
    const int mt = t % nmt, pzy = t / nmt;
    const int ci = mt * 32 + ci_l;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int co = ch * 8 + c;
      v[c] = (ci < Ka && co < M) ? up_weff(w, Ktot, M, ci, co, pzy >> 1, pzy & 1, slot >> 3, 1 - ((slot >> 2) & 1),
                                           1 - ((slot >> 1) & 1), 1 - (slot & 1))
                                 : 0.f;
    }
    u32x4 h, r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned hh, rr;
      split_pair3(v[2 * q], v[2 * q + 1], s, hh, rr);
      h[q] = hh; r[q] = rr;
    }
    const long long base = ((((long long)pzy * nmt + mt) * nchunk + ch) * 2) * 512;
    ws[base + slot * 32 + ci_l] = h;
    ws[base + 512 + slot * 32 + ci_l] = r;
  }
}
__global__ __launch_bounds__(1024) void conv3d_up_wsplit_t_k(const float* __restrict__ w, u32x4* __restrict__ ws, int Ka,
                                                             int Ktot, int M, float* __restrict__ trailer) {
  __shared__ float sm[17];
  conv3d_up_wsplit_t_body(w, ws, Ka, Ktot, M, trailer, sm);
}
// Every split of a network's 3-D conv weights in ONE launch (they all change together, at the optimizer step): 16
// launches of ~20 us each were 7 % of the 128^3 train step.  grid = (32, jobs).
struct DfWsplitJob {
  const float* w;
  u32x4* ws;
  float* trailer;
  int kind;                      // 0: conv3d_wsplit_k, 1: conv3d_up_wsplit_k, 2: conv3d_up_wsplit_t_k
  int K, M, pair, Ktot, koff, Cb, pad_;
};
static_assert(sizeof(DfWsplitJob) == 56, "job table layout (7 x int64 on the host side)");
__global__ __launch_bounds__(1024) void conv3d_wsplit_batch_k(const DfWsplitJob* __restrict__ jobs) {
  __shared__ float sm[17];
  const DfWsplitJob j = jobs[blockIdx.y];
  if (j.kind == 0) conv3d_wsplit_body(j.w, j.ws, j.K, j.M, j.trailer, j.pair, j.Ktot, j.koff, sm);
  else if (j.kind == 1) conv3d_up_wsplit_body(j.w, j.ws, j.K, j.Ktot, j.M, j.trailer, j.Cb, sm);
  else conv3d_up_wsplit_t_body(j.w, j.ws, j.K, j.Ktot, j.M, j.trailer, sm);
}
extern "C" int dfmir_conv3d_wsplit_batch(const void* jobs_dev, int njobs, void* stream) {
  DF_ARG_CHECK(jobs_dev && njobs > 0 && njobs < 65536);
  conv3d_wsplit_batch_k<<<dim3(32, (unsigned)njobs), 1024, 0, (hipStream_t)stream>>>(reinterpret_cast<const DfWsplitJob*>(jobs_dev));
  DF_LAUNCH_CHECK();
  return 0;
}
// 0: 32-row form, 1: plane-pair form, 2: 16-row (M16) form -- the `pair` field of a dfmir_conv3d_wsplit_batch job
extern "C" int dfmir_conv3d_split_is_pair(int cout_used) {
  return (cout_used <= 16 && !pair3d_off()) ? (m16_off() ? 1 : 2) : 0;
}

struct C3dP {
  int N, Ca, Cout, D, H, W;      // D, H, W: the LOW-resolution volume (d(a)); dy is [N, Cout, 2D, 2H, 2W]
  int nz, ny, nx, nchunk, dy_n;
  long long ntile;
  const float* act_src;          // optional: a itself when it is the output of a LeakyReLU feeding only this layer
  float act_slope;
};

__global__ __launch_bounds__(256, 2) void conv3d_up_dgrad_k(const float* __restrict__ dy, const float* __restrict__ dy_amax,
                                                         const u32x4* __restrict__ wsp, const float* __restrict__ w_trailer,
                                                         float* __restrict__ da, float* __restrict__ da_amax, C3dP k) {
  constexpr int TZ = 2, TY = 8, TX = 16, HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
  constexpr int XP = HZ * HY * HX;                        // 720 positions per px sub-lattice and split
  constexpr int WU = 2 * 16 * 32;                         // 1024 units of one (pzy, chunk)'s weights
  constexpr int NJ = 2;
  constexpr int NTASK = HZ * HY * 10, NIT = (NTASK + 255) / 256;   // (row, aligned quad) tasks: 400 -> 2 per thread
  constexpr unsigned OOB = 0x80000000u;
  __shared__ u32x4 Xs[2 * 2 * XP];                        // [px][split][position]
  __shared__ u32x4 Ws[WU];
  __shared__ float red[17];
  __shared__ unsigned smax;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const long long S = (long long)k.D * k.H * k.W;
  const int Df = 2 * k.D, Hf = 2 * k.H, Wf = 2 * k.W;
  const long long Sf = (long long)Df * Hf * Wf;
  const long long per_xcd = (k.ntile + 7) / 8;
  const long long tile = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if ((long long)(blockIdx.x >> 3) >= per_xcd || tile >= k.ntile) return;
  int n, z0, y0, x0;
  {
    long long pid = tile;
    const int bx = (int)(pid % k.nx); pid /= k.nx;
    const int bz = (int)(pid % k.nz); pid /= k.nz;
    const int by = (int)(pid % k.ny);
    n = (int)(pid / k.ny);
    z0 = bz * TZ; y0 = by * TY; x0 = bx * TX;
  }
  const int mt = blockIdx.y;
  const float amax = reduce_absmax(dy_amax, k.dy_n, red);
  const int ex = scale_exp3(amax);
  const int ew = reinterpret_cast<const int*>(w_trailer)[0];
  const float xscale = pow2f3(ex), osc = pow2f3(-ex) * pow2f3(-ew);
  if (tid == 0) smax = 0u;

  const __amdgpu_buffer_rsrc_t y_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dy + (long long)n * k.Cout * Sf), 0, (unsigned)((long long)k.Cout * Sf * 4), 0x00020000);
  const unsigned sf4 = (unsigned)Sf * 4u;
  // staging tasks of this thread: task = row * 10 + q, row = hz * HY + hy; the quad starts at column 2 x0 - 4 + 4 q
  int trow[NIT], tq[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int task = tid + 256 * i;
    trow[i] = task < NTASK ? task / 10 : -1;
    tq[i] = task - 10 * (task / 10);
  }
  // this lane's output voxels: plane wz, rows wy + 2 j + (l31 >> 4), column lx (lane rotation as conv3d_split_k)
  const int wz = wid & 1, wy = 4 * (wid >> 1);
  const int lx = (l31 - 2 * (l31 >> 4)) & 15;
  int pbase[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) pbase[j] = (wz * HY + wy + 2 * j + (l31 >> 4)) * HX + lx + hi;      // + sx = hi

  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  u32x4 rq[NIT][8];
  u32x4 rw[4];
  const int nmt = gridDim.y;
  // loads of sub-phase sp = pzy * nchunk + chunk
#define C3D_GLOAD(sp_)                                                                            \
  {                                                                                               \
    const int pzy_ = (sp_) / k.nchunk, ch_ = (sp_) - pzy_ * k.nchunk;                             \
    const int pz_ = pzy_ >> 1, py_ = pzy_ & 1;                                                    \
    _Pragma("unroll") for (int i = 0; i < NIT; ++i) {                                             \
      const int hz = trow[i] / HY, hy = trow[i] - HY * (trow[i] / HY);                            \
      const int fz = 2 * (z0 - 1 + hz) + pz_, fy = 2 * (y0 - 1 + hy) + py_, fx = 2 * x0 - 4 + 4 * tq[i];   \
      const bool ok = trow[i] >= 0 && (unsigned)fz < (unsigned)Df && (unsigned)fy < (unsigned)Hf && (unsigned)fx < (unsigned)Wf; \
      const unsigned off = ok ? (unsigned)((fz * Hf + fy) * Wf + fx) * 4u : OOB;                  \
      _Pragma("unroll") for (int c = 0; c < 8; ++c)                                               \
        rq[i][c] = __builtin_amdgcn_raw_buffer_load_b128(y_src, ok ? off + (unsigned)(ch_ * 8 + c) * sf4 : OOB, 0, 0); \
    }                                                                                             \
    const u32x4* wp_ = wsp + (((long long)pzy_ * nmt + mt) * k.nchunk + ch_) * WU;                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) rw[j] = wp_[tid + 256 * j];                     \
  }
#define C3D_LSTORE()                                                                              \
  {                                                                                               \
    _Pragma("unroll") for (int i = 0; i < NIT; ++i) {                                             \
      if (trow[i] >= 0) {                                                                         \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                           \
          const int hx = 2 * tq[i] - 1 + (e >> 1);                                                \
          if ((unsigned)hx < (unsigned)HX) {                                                      \
            u32x4 h, r;                                                                           \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                       \
              unsigned hh, rr;                                                                    \
              split_pair3(__uint_as_float(rq[i][2 * q][e]), __uint_as_float(rq[i][2 * q + 1][e]), xscale, hh, rr); \
              h[q] = hh; r[q] = rr;                                                               \
            }                                                                                     \
            const int pos = (e & 1) * 2 * XP + trow[i] * HX + hx;                                 \
            Xs[pos] = h;                                                                          \
            Xs[XP + pos] = r;                                                                     \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
    }                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) Ws[tid + 256 * j] = rw[j];                      \
  }
  const int nsp = 4 * k.nchunk;
  C3D_GLOAD(0);
  C3D_LSTORE();
  __syncthreads();
  for (int sp = 0; sp < nsp; ++sp) {
    const bool more = sp + 1 < nsp;
    const int pzy = sp / k.nchunk, pz = pzy >> 1, py = pzy & 1;
    if (more) C3D_GLOAD(sp + 1);
    // patch offset of slot (sz, sy) for this (pz, py): ((sz + 1 - pz) HY + (sy + 1 - py)) HX + (1 - px)   [+ sx in pbase]
    const int pofs = ((1 - pz) * HY + (1 - py)) * HX;
    // operands one k-step ahead (two register sets), reads pinned between the MFMAs
#define C3D_OPLOAD(b_, tp_)                                                                       \
    {                                                                                             \
      const int px_ = (tp_) >> 2, sz_ = ((tp_) >> 1) & 1, sy_ = (tp_) & 1;                        \
      const int toff_ = px_ * 2 * XP + pofs + (sz_ * HY + sy_) * HX + (1 - px_);                  \
      const int u_ = 2 * (tp_) + hi;                                                              \
      A0[b_] = Ws[u_ * 32 + l31];                                                                 \
      A1[b_] = Ws[512 + u_ * 32 + l31];                                                           \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                            \
        B0[b_][j] = Xs[pbase[j] + toff_];                                                         \
        B1[b_][j] = Xs[XP + pbase[j] + toff_];                                                    \
      }                                                                                           \
    }
    u32x4 A0[2], A1[2], B0[2][NJ], B1[2][NJ];
    C3D_OPLOAD(0, 0)
#pragma unroll
    for (int tp = 0; tp < 8; ++tp) {
      const int cur = tp & 1;
      if (tp + 1 < 8) C3D_OPLOAD(cur ^ 1, tp + 1)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[j] = mma3(A1[cur], B0[cur][j], acc[j]);
        acc[j] = mma3(A0[cur], B1[cur][j], acc[j]);
        acc[j] = mma3(A0[cur], B0[cur][j], acc[j]);
      }
#pragma unroll
      for (int i = 0; i < 3 * NJ; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
      }
    }
#undef C3D_OPLOAD
    if (more) {
      __syncthreads();
      C3D_LSTORE();
      __syncthreads();
    }
  }
#undef C3D_GLOAD
#undef C3D_LSTORE
  // ---- epilogue: acc[j][r] <-> channel row (r>>2)*8 + hi*4 + (r&3) of tile mt, voxel (z0 + wz, y0 + wy + 2j + (l31>>4), x0 + lx)
  const __amdgpu_buffer_rsrc_t a_dst = __builtin_amdgcn_make_buffer_rsrc(
      da + (long long)n * k.Ca * S, 0, (unsigned)((long long)k.Ca * S * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t s_src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>((k.act_src ? k.act_src : da) + (long long)n * k.Ca * S), 0, (unsigned)((long long)k.Ca * S * 4), 0x00020000);
  const unsigned s4 = (unsigned)S * 4u;
  float pm = 0.f;
  const int gz = z0 + wz;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int gy = y0 + wy + 2 * j + (l31 >> 4), gx = x0 + lx;
    const bool vok = gz < k.D && gy < k.H && gx < k.W;
    const unsigned vo = (unsigned)((gz * k.H + gy) * k.W + gx) * 4u + (unsigned)(hi * 4) * s4;
    float av[16];
    if (k.act_src) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = mt * 32 + (r >> 2) * 8 + (r & 3);
        av[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(s_src, (vok && (ci + hi * 4) < k.Ca) ? vo : OOB, (unsigned)ci * s4, 0));
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = mt * 32 + (r >> 2) * 8 + (r & 3);
      const bool ok = vok && (ci + hi * 4) < k.Ca;
      float v = acc[j][r] * osc;
      if (k.act_src) v = av[r] > 0.f ? v : v * k.act_slope;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), a_dst, ok ? vo : OOB, (unsigned)ci * s4, 0);
      pm = fmaxf(pm, ok ? fabsf(v) : 0.f);
    }
  }
  if (da_amax) {
    __syncthreads();
    publish_block_absmax_acc(pm, &smax, da_amax);
  }
}

extern "C" long long dfmir_conv3d_up_dgrad_ws_floats(int Ca, int Cout) {
  if (Ca <= 0 || Cout <= 0) return -1;
  return 4LL * ((Ca + 31) / 32) * ((Cout + 7) / 8) * 1024 * 4 + 4;
}
// da [N, Ca, D, H, W] <- d(a) of y = conv3x3x3(cat(nearest_up2(a), b)) given dy [N, Cout, 2D, 2H, 2W] (the gradient
// w.r.t. the conv's result, i.e. after the activation's backward).  w_tcc: the FORWARD packing [27][Ktot][Cout] or NULL
// (ws holds the current split).  act_src (optional): a, when a is the output of a LeakyReLU(act_slope) that feeds only
// this layer -- da then is the gradient w.r.t. that activation's input.  da_amax (optional): range probe of da.
extern "C" int dfmir_conv3d_up_dgrad(const float* dy, const float* dy_amax, int dy_amax_n, const float* w_tcc, int Ktot,
                                     float* ws, float* da, float* da_amax, const float* act_src, float act_slope, int N,
                                     int Ca, int Cout, int D, int H, int W, void* stream) {
  DF_ARG_CHECK(dy && dy_amax && dy_amax_n > 0 && ws && da && dfmir_conv3d_up_ok(N, Ca, Cout, D, H, W) && (Cout & 7) == 0);
  DF_ARG_CHECK((reinterpret_cast<uintptr_t>(ws) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (!w_tcc || Ktot >= Ca));
  hipStream_t st = (hipStream_t)stream;
  const int nchunk = (Cout + 7) / 8, nmt = (Ca + 31) / 32;
  float* trailer = ws + dfmir_conv3d_up_dgrad_ws_floats(Ca, Cout) - 4;
  if (w_tcc) {
    const long long units = 4LL * nmt * nchunk * 512;
    long long nwg = (units + 1023) / 1024;
    if (nwg > 32) nwg = 32;
    conv3d_up_wsplit_t_k<<<(unsigned)nwg, 1024, 0, st>>>(w_tcc, reinterpret_cast<u32x4*>(ws), Ca, Ktot, Cout, trailer);
    DF_LAUNCH_CHECK();
  }
  C3dP k{N, Ca, Cout, D, H, W, (D + 1) / 2, (H + 7) / 8, (W + 15) / 16, nchunk, dy_amax_n, 0, act_src, act_slope};
  k.ntile = (long long)N * k.nz * k.ny * k.nx;
  const dim3 grid((unsigned)(8 * ((k.ntile + 7) / 8)), (unsigned)nmt);
  conv3d_up_dgrad_k<<<grid, 256, 0, st>>>(dy, dy_amax, reinterpret_cast<const u32x4*>(ws), trailer, da, da_amax, k);
  DF_LAUNCH_CHECK();
  return 0;
}

// The skip channels' share, added to the partial sum already in y: y = act(conv3x3x3(b; rows koff .. koff + Cb - 1 of the
// layer's [27][Ktot][Cout] packing) + bias + y), range probe of y in y_amax.  g describes the conv over b alone
// (Cin = Cb).  w_tcc NULL: ws holds the split of the current weights.
extern "C" int dfmir_conv3d_split_fwd_add(const DfConvGeom* g, const float* b, const float* b_amax, int b_amax_n,
                                          const float* w_tcc, int Ktot, int koff, float* ws, const float* bias, float* y,
                                          float* y_amax, void* stream) {
  DF_ARG_CHECK(g && y && (!w_tcc || (Ktot >= koff + g->Cin && koff >= 0)));
  return conv3d_split_fwd_impl(g, b, b_amax, b_amax_n, w_tcc, ws, bias, y, y_amax, g->Cout, stream, y, 0.f, 2, Ktot, koff);
}

// ================================================================================================
// wgrad:  dWt[tap][ci][co] += sum_v X[ci][v + tap] * dY[co][v]   with the VOXEL as the MFMA K (16 consecutive x).
//
// Workgroup = 4 waves, persistent over 2 x 4 x 16-voxel patches of its share of the volume, ALL input channels
// (<= 48, in chunks of 8) and all output channels (<= 32).  MFMA rows = (tap, ci) of a chunk (27 x 8 = 216 -> 7 row
// tiles of 32), columns = co.  Wave w keeps the accumulators of row tiles {(w+c) mod 4, (w+c) mod 4 + 4} of every
// chunk c (the padding tile 7 rotates over the waves) and runs all 8 k-steps (x-rows) of a patch for them, so no
// cross-wave reduction is needed; one atomicAdd per accumulator at the very end (split-K over workgroups).
// LDS per patch: dY split once, [split][co][row][unit] (unit = 8 voxels x fp16 = 16 B); per chunk: X split into three
// x-aligned copies [split][dx][ci][hz][hy][unit] (copy dx holds x + dx - 1), so every tap reads 16-B aligned units
// (an unaligned ds_read_b128 runs at quarter rate).  A thread converts one 18-voxel row: 9 packed pairs serve the
// copies dx = 0 and dx = 2 (same pairing, one dword apart), v_alignbit makes the dx = 1 pairing.
// ================================================================================================
struct W3sP {
  int N, Cin, Cout, D, H, W;
  int nz, ny, nx;
  long long npatch, per_block;
  long long nslot;               // tr kernel: workgroups per XCD along grid.x (per_block = tiles per XCD)
  int x_n, dy_n;
  int nchunk;                    // chunks of 8 input channels in the layer; blockIdx.y * NCH = this workgroup's first
  long long s_tap, s_row, s_col; // output index = tap' * s_tap + ci * s_row + co * s_col, tap' = flip ? 26 - tap : tap
  int flip;
  float* db;                     // optional bias gradient: db[.] += sum over voxels of the layer's output gradient
  int db_from_x;                 // 0: db indexed by co, summed from the dY operand;  1: by ci, from the X operand (swapped roles)
  // X = cat(nearest_up2(xa), x) never materialised (tr kernel): channels 0 .. Ca - 1 are read from the HALF-resolution
  // tensor xa [N, Ca, D/2, H/2, W/2] at (z >> 1, y >> 1, x >> 1), the remaining Cin - Ca from x [N, Cin - Ca, D, H, W]
  const float* xa;
  int Ca;
  const float* fx;               // deterministic mode (common.h df_acc): dwt holds 64-bit fixed-point sums, db is NULL
};

template <int NCH>
__global__ __launch_bounds__(256) void conv3d_wgrad_split_k(const float* __restrict__ x, const float* __restrict__ x_amax,
                                                            const float* __restrict__ dy, const float* __restrict__ dy_amax,
                                                            float* __restrict__ dwt, W3sP k) {
  constexpr int PZ = 2, PY = 4, HZ = PZ + 2, HY = PY + 2;
  constexpr int CIS = HZ * HY * 2 + 1;                    // 49 units: ci stride == 16 B (mod 256 B)
  constexpr int XSPL = 3 * 8 * CIS;                       // 1176 units per split
  constexpr int COS = 17, YSPL = 32 * COS;                // dY: co stride 17 units, 544 per split
  __shared__ u32x4 Xs[2 * XSPL];
  __shared__ u32x4 Ys[2 * YSPL];
  __shared__ float red[17];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const long long S = (long long)k.D * k.H * k.W;

  const int ex = scale_exp3(reduce_absmax(x_amax, k.x_n, red));
  __syncthreads();
  const int ed = scale_exp3(reduce_absmax(dy_amax, k.dy_n, red));
  const float xscale = pow2f3(ex), dscale = pow2f3(ed), oscale = pow2f3(-ex), oscale2 = pow2f3(-ed);

  f32x16 acc[NCH][2];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][s][r] = 0.f;

  // staging roles
  const bool xrow = tid < 8 * HZ * HY;                    // 192 threads: one (ci, hz, hy) row of 18 voxels each
  const int xci = tid / (HZ * HY), xrem = tid % (HZ * HY), xhz = xrem / HY, xhy = xrem % HY;
  const int yco = tid >> 3, yr = tid & 7;                 // dY: (co, row) -> 16 voxels
  constexpr unsigned OOB = 0x80000000u;

  const int c_base = blockIdx.y * NCH;
  const long long p_begin = (long long)blockIdx.x * k.per_block;
  long long p_end = p_begin + k.per_block;
  if (p_end > k.npatch) p_end = k.npatch;
  if (p_begin >= p_end) return;
  const int niter = (int)(p_end - p_begin) * NCH;

  float rx[18];
  u32x4 ry[4];
  __amdgpu_buffer_rsrc_t x_src, y_src;
  int pz0 = 0, py0 = 0, px0 = 0;
  float bacc = 0.f;                                       // this thread's share of the bias gradient
  const bool db_y = k.db && !k.db_from_x && blockIdx.y == 0;
  const bool db_x = k.db && k.db_from_x && xrow && xhz >= 1 && xhz <= PZ && xhy >= 1 && xhy <= PY;   // the patch's own rows

#define W3S_GLOAD(it_)                                                                            \
  {                                                                                               \
    const long long p_ = p_begin + (it_) / NCH;                                                   \
    const int c_ = (it_) % NCH;                                                                   \
    const int ca_ = c_base + c_;                                                                  \
    long long q_ = p_;                                                                            \
    const int bx_ = (int)(q_ % k.nx); q_ /= k.nx;                                                 \
    const int by_ = (int)(q_ % k.ny); q_ /= k.ny;                                                 \
    const int bz_ = (int)(q_ % k.nz);                                                             \
    const int n_ = (int)(q_ / k.nz);                                                              \
    pz0 = bz_ * PZ; py0 = by_ * PY; px0 = bx_ * 16;                                               \
    x_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)n_ * k.Cin * S), 0,   \
                                              (unsigned)((long long)k.Cin * S * 4), 0x00020000);  \
    {                                                                                             \
      const int gz = pz0 - 1 + xhz, gy = py0 - 1 + xhy, ci = ca_ * 8 + xci;                       \
      const bool rowok = xrow && ci < k.Cin && (unsigned)gz < (unsigned)k.D && (unsigned)gy < (unsigned)k.H;  \
      const unsigned base = (unsigned)(((long long)ci * S + ((long long)gz * k.H + gy) * k.W + px0 - 1) * 4);  \
      _Pragma("unroll") for (int j = 0; j < 18; ++j) {                                            \
        const int gx = px0 - 1 + j;                                                               \
        rx[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(                             \
            x_src, (rowok && (unsigned)gx < (unsigned)k.W) ? base + 4u * j : OOB, 0, 0));         \
      }                                                                                           \
    }                                                                                             \
    if (c_ == 0) {                                                                                \
      y_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + (long long)n_ * k.Cout * S), 0,   \
                                                (unsigned)((long long)k.Cout * S * 4), 0x00020000);   \
      const int gz = pz0 + (yr >> 2), gy = py0 + (yr & 3);                                        \
      const bool rowok = yco < k.Cout && gz < k.D && gy < k.H;                                    \
      const unsigned base = (unsigned)(((long long)yco * S + ((long long)gz * k.H + gy) * k.W + px0) * 4);  \
      _Pragma("unroll") for (int q = 0; q < 4; ++q)                                               \
        ry[q] = __builtin_amdgcn_raw_buffer_load_b128(y_src, (rowok && px0 + 4 * q < k.W) ? base + 16u * q : OOB, 0, 0);  \
    }                                                                                             \
  }

  W3S_GLOAD(0);
  for (int it = 0; it < niter; ++it) {
    const int c = it % NCH;
    __syncthreads();                                       // previous compute is done with the LDS buffers
    if (db_x) {
#pragma unroll
      for (int j = 1; j <= 16; ++j) bacc += rx[j];          // loads outside the frame returned 0
    }
    if (xrow) {
      unsigned ph[9], pr[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) split_pair3(rx[2 * j], rx[2 * j + 1], xscale, ph[j], pr[j]);
      const int ubase = xci * CIS + (xhz * HY + xhy) * 2;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4 h0, h1, h2, r0, r1, r2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          h0[q] = ph[4 * u + q]; r0[q] = pr[4 * u + q];                               // voxels x-1 .. (copy dx = 0)
          h2[q] = ph[4 * u + q + 1]; r2[q] = pr[4 * u + q + 1];                       // voxels x+1 .. (copy dx = 2)
          h1[q] = __builtin_amdgcn_alignbit(ph[4 * u + q + 1], ph[4 * u + q], 16);    // voxels x ..   (copy dx = 1)
          r1[q] = __builtin_amdgcn_alignbit(pr[4 * u + q + 1], pr[4 * u + q], 16);
        }
        Xs[0 * 8 * CIS + ubase + u] = h0; Xs[XSPL + 0 * 8 * CIS + ubase + u] = r0;
        Xs[1 * 8 * CIS + ubase + u] = h1; Xs[XSPL + 1 * 8 * CIS + ubase + u] = r1;
        Xs[2 * 8 * CIS + ubase + u] = h2; Xs[XSPL + 2 * 8 * CIS + ubase + u] = r2;
      }
    }
    if (c == 0) {
      const int ub = yco * COS + yr * 2;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4 h, r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const u32x4 v = ry[2 * u + q];
          if (db_y) bacc += (__uint_as_float(v[0]) + __uint_as_float(v[1])) + (__uint_as_float(v[2]) + __uint_as_float(v[3]));
          unsigned hh, rr;
          split_pair3(__uint_as_float(v[0]), __uint_as_float(v[1]), dscale, hh, rr);
          h[2 * q] = hh; r[2 * q] = rr;
          split_pair3(__uint_as_float(v[2]), __uint_as_float(v[3]), dscale, hh, rr);
          h[2 * q + 1] = hh; r[2 * q + 1] = rr;
        }
        Ys[ub + u] = h;
        Ys[YSPL + ub + u] = r;
      }
    }
    __syncthreads();
    if (it + 1 < niter) W3S_GLOAD(it + 1);

    // this wave's two row tiles of chunk c
    int aoff[2];
    bool tok[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int tile = ((wid + c) & 3) + 4 * s;
      tok[s] = tile < 7;
      int rho = tile * 32 + l31;
      if (rho > 215) rho = 215;                            // padding rows: any valid address (results discarded)
      const int tap = rho >> 3, ci = rho & 7;
      const int dz = tap / 9, dyy = (tap / 3) % 3, dx = tap % 3;
      aoff[s] = (dx * 8 + ci) * CIS + (dz * HY + dyy) * 2 + hi;
    }
    const int boff = l31 * COS + hi;
#pragma unroll
    for (int c2 = 0; c2 < NCH; ++c2) {
      if (c2 != c || c_base + c >= k.nchunk) continue;     // static accumulator index; a padding chunk has no work
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int zy = ((r >> 2) * HY + (r & 3)) * 2;
        const u32x4 b0 = Ys[boff + r * 2], b1 = Ys[YSPL + boff + r * 2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (!tok[s]) continue;
          const u32x4 a0 = Xs[aoff[s] + zy], a1 = Xs[XSPL + aoff[s] + zy];
          acc[c2][s] = mma3(a1, b0, acc[c2][s]);
          acc[c2][s] = mma3(a0, b1, acc[c2][s]);
          acc[c2][s] = mma3(a0, b0, acc[c2][s]);
        }
      }
    }
  }
#undef W3S_GLOAD

  // ---- epilogue: acc[c][s][r] <-> row (r>>2)*8 + hi*4 + (r&3) of tile ((wid+c)&3) + 4s, column co = l31
  const float sc = oscale * oscale2;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int tile = ((wid + c) & 3) + 4 * s;
      if (tile >= 7 || l31 >= k.Cout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = tile * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
        const int tap = rho >> 3, ci = (c_base + c) * 8 + (rho & 7);
        const int to = k.flip ? 26 - tap : tap;
        if (rho < 216 && ci < k.Cin) df_acc(dwt, to * k.s_tap + ci * k.s_row + l31 * k.s_col, acc[c][s][r] * sc, k.fx);
      }
    }
  if (db_y && yco < k.Cout) atomicAdd(&k.db[yco], bacc);
  if (db_x) {
    // the X rows of chunk c were summed while c was staged; bacc mixes the chunks of this workgroup row only when
    // NCH > 1, which the swapped-role launch (one chunk) never has
    const int ci = c_base * 8 + xci;
    if (ci < k.Cin) atomicAdd(&k.db[ci], bacc);
  }
}

// ================================================================================================
// wgrad, transpose-read form (default):  the same GEMM -- rows (tap, ci), columns co, K = voxels -- but the operands
// are read from the FORWARD kernel's LDS images with ds_read_b64_tr_b16 (gfx950): X as [position][8 channels x fp16]
// (16 B per position and split), dY as [voxel][32 output channels x fp16] (64 B).  In a 16-lane group the hardware
// hands lane 4q + c, element j the c-th fp16 at the address supplied by lane 4j + q: with lane (j, q) pointing at
// (voxel j, channel quad q) every lane receives 4 consecutive voxels of its own row -- a K-major MFMA operand out of a
// channel-major image.  A tap is then nothing but a position offset (16-B granular, so dx shifts stay aligned): ONE
// copy of the patch instead of three, the patch is staged exactly like the forward kernel's (8 x buffer_load_dwordx4
// per thread, 4 v_fma_mix per value pair), and a tile can be four times larger for the same LDS.
//
// Workgroup = 4 waves, persistent over 2 x 8 x 16-voxel tiles (16 k-steps = x-rows of 16 voxels); per tile the dY
// image is staged once and the X patch (4 x 10 x 18 positions) once per 8-channel chunk.  Row tile T of a chunk =
// taps 4T .. 4T+3 x 8 channels (7 tiles, tap 27 is padding); wave w owns tiles (w + c) & 3 and that + 4 of chunk c
// (accumulators of <= 3 chunks resident: 96 AGPRs, two workgroups per CU so that one stages while the other computes).
// Bank behaviour of the reads: the 32 lanes served together cover 4 taps x 4 x-positions x 16 B; taps of one
// (dz, dy) are contiguous, the next (dz, dy) must start 112..192 B further (mod 256): row stride 25 positions, plane
// stride 250 -- odd, so that the staging stores of 4 consecutive rows x 2 quads (one ds_write_b128 lane group) fall on
// 8 different 16-B bank slots (measured: all of the kernel's bank conflicts were these stores at stride 24).  dY: 4 voxels x 64 B contiguous; the channel group is XORed with (voxel >> 2) & 3 so that the staging
// stores of a wave (one channel group, 4 voxel quads) spread over the banks.
// ================================================================================================
#ifndef W3T_NOSKIP
#define W3T_NOSKIP 0
#endif
#ifndef W3T_KO
#define W3T_KO 0     // knock-out builds for timing: 1 = no MFMAs, 2 = no prefetch loads, 4 = no convert + LDS store, 8 = no operand reads
#endif
typedef short s16x4_3 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_3* lds_tr_ptr;
__device__ __forceinline__ uint2 tr_read8(unsigned byte_addr) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(uintptr_t)byte_addr));
}
#define TR_PAIR(dst_, addr_)                                                                      \
  {                                                                                               \
    const uint2 u0_ = tr_read8(addr_), u1_ = tr_read8((addr_) + 64u);                             \
    dst_ = u32x4{u0_.x, u0_.y, u1_.x, u1_.y};                                                     \
  }

template <int NCH>
struct W3T {
  static constexpr int TZ = 2, TY = 8, TX = 16, HZ = 4, HY = 10, HXP = 25, SZP = HY * HXP;
  static constexpr int XPOS = HZ * SZP;                  // 1000 units per split
  static constexpr int YU = TZ * TY * TX * 4;            // 1024 units per split
};

// PAIR form (<= 16 output channels: half of the 32 MFMA columns would be padding): the columns are (plane p, co) --
// columns 0-15 take dY of the tile's z-plane 0, columns 16-31 the SAME voxel (y, x) of plane 1 -- and K runs over
// plane 0 only (8 k-steps).  Rows then are (tap', ci) with tap' = (dz' in 0..3, dy, dx) over the patch's four planes:
// row (dz', .) x column (p, .) is a term of dW[dz' - p] (dropped where dz' - p is outside 0..2): 9 row tiles x 8
// k-steps instead of 7 x 16.  A wave owns tiles w and w + 4 and k-steps 2w, 2w + 1 of tile 8 (the partial sums meet
// in the atomics of the epilogue): 18 tile-steps per wave and phase instead of 32.
// UPCAT: X = cat(nearest_up2(xa), x) read in place (see W3sP::xa).  A compile-time variant: a branch around the operand
// prefetch splits the k-loop's basic block and with it the MFMA / LDS / VMEM interleave (measured: +35 % on every shape),
// so both sources are loaded branch-free -- the inactive one with an out-of-range offset (returns 0, no memory access).
template <int NCH, bool PAIR, bool UPCAT>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_tr_k(const float* __restrict__ x, const float* __restrict__ x_amax,
                                                         const float* __restrict__ dy, const float* __restrict__ dy_amax,
                                                         float* __restrict__ dwt, W3sP k) {
  using G = W3T<NCH>;
  constexpr int TZ = G::TZ, TY = G::TY, HY = G::HY, HXP = G::HXP, SZP = G::SZP, XPOS = G::XPOS, YU = G::YU;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ u32x4 Xs[2 * XPOS];
  __shared__ u32x4 Ys[2 * YU];
  __shared__ float red[17];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const long long S = (long long)k.D * k.H * k.W;
  const unsigned s4 = (unsigned)S * 4u;

  const int ex = scale_exp3(reduce_absmax(x_amax, k.x_n, red));
  __syncthreads();
  const int ed = scale_exp3(reduce_absmax(dy_amax, k.dy_n, red));
  const float xscale = pow2f3(ex), dscale = pow2f3(ed), oscale = pow2f3(-ex), oscale2 = pow2f3(-ed);

  constexpr int NA = PAIR ? 3 : 2;                         // accumulators (row tiles) per chunk and wave
  constexpr int NKS = PAIR ? 8 : 16;                       // k-steps per phase
  f32x16 acc[NCH][NA];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int s = 0; s < NA; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][s][r] = 0.f;

  const int c_base = blockIdx.y * NCH;
  int nc = k.nchunk - c_base;                              // chunks of this workgroup row
  if (nc > NCH) nc = NCH;
  // Tiles: workgroup ids go round-robin to the 8 XCDs (one L2 each), so XCD e = id & 7 owns the e-th eighth of the
  // tile list and its J workgroups (slot j = id >> 3) walk it together: in iteration i they hold the J consecutive
  // tiles i J .. i J + J - 1 of the eighth.  The list runs x fastest, then over 2 x 2 blocks of (z, y): a window of J
  // tiles is a few full x-rows of neighbouring (z, y), so the halos (3.75x the tile in the patch) are mostly L2 hits.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const long long t_first = (long long)xcd * k.per_block + slot;       // per_block = tiles per XCD
  long long t_lim = (long long)(xcd + 1) * k.per_block;
  if (t_lim > k.npatch) t_lim = k.npatch;
  const int J = (int)k.nslot;
  if (t_first >= t_lim || nc <= 0) return;
  const int niter = (int)((t_lim - t_first + J - 1) / J);

  // ---- staging roles.  X: thread t < 240 owns the aligned quad qq (x0 - 4 + 4 qq ..) of halo row t / 6 (40 rows of
  // 6 quads; quads 0 and 5 contribute one position each) in the 8 channels of the chunk.  dY: wave w owns channel
  // group w, thread = (row 0..15, quad 0..3) of the tile.
  const bool xt = tid < 240;
  const int xm = tid / 80, xr80 = tid - 80 * xm;           // quad pair m = 0..2; 8 consecutive lanes = 4 rows x 2 quads
  const int xrow = xr80 >> 1, xqq = 2 * xm + (xr80 & 1);
  const int xhz = xrow / HY, xhy = xrow - HY * xhz;
  const int xpos0 = xhz * SZP + xhy * HXP + 4 * xqq - 3;   // unit of element e: xpos0 + e
  const int xe0 = xqq == 0 ? 3 : 0, xe1 = xqq == 5 ? 1 : 4;
  const int yq = tid & 3, yrow = (tid >> 2) & 15;
  const bool yt = wid * 8 < k.Cout;
  // unit of element e: yunit0 + 4 e  (PAIR: plane 1 takes the other 32-B half of the voxel's 64 B, so that the columns
  // of both planes, read together, fall on different banks)
  const int yunit0 = (yrow * 16 + 4 * yq) * 4 + (wid ^ yq ^ (PAIR ? (yrow >> 3) * 2 : 0));
  const bool db_y = k.db && !k.db_from_x && blockIdx.y == 0 && yt;
  const bool db_x = k.db && k.db_from_x && xt && xhz >= 1 && xhz <= TZ && xhy >= 1 && xhy <= TY && xqq >= 1 && xqq <= 4;
  float bacc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) bacc[c] = 0.f;

  // ---- operand addresses (bytes in LDS).  Source role of this lane in its 16-lane group: j = voxel, q = channel quad.
  const int sj = (lane & 15) >> 2, sq = lane & 3, sg = (lane >> 4) & 1;
  const unsigned xs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)Xs;
  const unsigned ys_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)Ys;
  // B (dY): voxel = s * 16 + 8 hi + 4 i + j, channel quad = 4 sg + sq -> unit (2 sg + (sq >> 1)) ^ ((2 hi + i) & 3)
  // PAIR: column group sg = plane, voxel + 128 sg, channel quad sq -> unit (sq >> 1) ^ ((2 hi + i) & 3) ^ 2 sg
  unsigned baddr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    baddr[i] = PAIR ? ys_base + (unsigned)(((128 * sg + 8 * hi + 4 * i + sj) * 4 + ((sq >> 1) ^ ((2 * hi + i) & 3) ^ (2 * sg))) * 16 + (sq & 1) * 8)
                    : ys_base + (unsigned)(((8 * hi + 4 * i + sj) * 4 + ((2 * sg + (sq >> 1)) ^ ((2 * hi + i) & 3))) * 16 + (sq & 1) * 8);

  u32x4 rq[8], ry[8];
  int tn, tz, ty, tx;                                      // tile being LOADED (runs one phase ahead of the compute)
  const int nzb = (k.nz + 1) >> 1;
  const int cells = 4 * nzb * ((k.ny + 1) >> 1);           // (z, y) cells per image incl. the phantom ones of odd counts
#define W3T_DECODE(t_)                                                                            \
  {                                                                                               \
    long long q_ = (t_);                                                                          \
    tx = (int)(q_ % k.nx); q_ /= k.nx;                                                            \
    const int u_ = (int)(q_ % cells);                                                             \
    tn = (int)(q_ / cells);                                                                       \
    const int b_ = u_ >> 2;                                                                       \
    tz = 2 * (b_ % nzb) + (u_ & 1);                                                               \
    ty = 2 * (b_ / nzb) + ((u_ >> 1) & 1);                                                        \
  }
  W3T_DECODE(t_first)
  unsigned gq = OOB, gy_ = OOB, gqa = OOB;
  __amdgpu_buffer_rsrc_t x_src, y_src, xa_src;
  const int Cup = UPCAT ? k.Ca : 0;                         // channels taken from the half-resolution tensor
  const int Dh = k.D >> 1, Hh = k.H >> 1, Wh = k.W >> 1;
  const unsigned sa4 = (unsigned)((long long)Dh * Hh * Wh) * 4u;
#define W3T_TILE_ADDR()                                                                           \
  {                                                                                               \
    const int z0 = tz * TZ, y0 = ty * TY, x0 = tx * 16;                                           \
    x_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (long long)tn * (k.Cin - Cup) * S), 0,   \
                                              (unsigned)((long long)(k.Cin - Cup) * S * 4), 0x00020000);  \
    if (UPCAT) xa_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(k.xa + (long long)tn * Cup * (sa4 >> 2)), 0,   \
                                                        (unsigned)((long long)Cup * sa4), 0x00020000);  \
    y_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy + (long long)tn * k.Cout * S), 0, \
                                              (unsigned)((long long)k.Cout * S * 4), 0x00020000); \
    {                                                                                             \
      const int gz = z0 - 1 + xhz, gyy = y0 - 1 + xhy, gx = x0 - 4 + 4 * xqq;                     \
      gq = (xt && tz < k.nz && ty < k.ny && (unsigned)gz < (unsigned)k.D && (unsigned)gyy < (unsigned)k.H && (unsigned)gx < (unsigned)k.W) \
               ? (unsigned)((gz * k.H + gyy) * k.W + gx) * 4u : OOB;                              \
      gqa = (UPCAT && gq != OOB) ? (unsigned)(((gz >> 1) * Hh + (gyy >> 1)) * Wh + (gx >> 1)) * 4u : OOB;   \
    }                                                                                             \
    {                                                                                             \
      const int gz = z0 + (yrow >> 3), gyy = y0 + (yrow & 7), gx = x0 + 4 * yq;                   \
      gy_ = (yt && tz < k.nz && ty < k.ny && gz < k.D && gyy < k.H && gx < k.W) ? (unsigned)((gz * k.H + gyy) * k.W + gx) * 4u : OOB; \
    }                                                                                             \
  }
#define W3T_GLOAD_X1(ca_, c_)                                                                     \
  if constexpr (UPCAT) {     /* a chunk lies entirely in one of the two tensors (Ca % 8 == 0): wave-uniform selects */ \
    typedef unsigned u32x2w_ __attribute__((ext_vector_type(2)));                                 \
    const bool up_ = (ca_) * 8 < Cup;                                                             \
    const u32x2w_ v2_ = __builtin_amdgcn_raw_buffer_load_b64(                                     \
        xa_src, (up_ && gqa != OOB) ? gqa + (unsigned)((ca_) * 8 + (c_)) * sa4 : OOB, 0, 0);      \
    const u32x4 v4_ = __builtin_amdgcn_raw_buffer_load_b128(                                      \
        x_src, (!up_ && gq != OOB) ? gq + (unsigned)((ca_) * 8 + (c_) - Cup) * s4 : OOB, 0, 0);   \
    rq[c_] = u32x4{v2_[0] | v4_[0], v2_[0] | v4_[1], v2_[1] | v4_[2], v2_[1] | v4_[3]};           \
  } else {                                                                                        \
    rq[c_] = __builtin_amdgcn_raw_buffer_load_b128(x_src, gq == OOB ? OOB : gq + (unsigned)((ca_) * 8 + (c_)) * s4, 0, 0); \
  }
#define W3T_GLOAD_Y1(c_)                                                                          \
  ry[c_] = __builtin_amdgcn_raw_buffer_load_b128(y_src, gy_ == OOB ? OOB : gy_ + (unsigned)(wid * 8 + (c_)) * s4, 0, 0);
#define W3T_STORE_X()                                                                             \
  if (xt) {                                                                                       \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                               \
      if (e < xe0 || e >= xe1) continue;                                                          \
      u32x4 h, r;                                                                                 \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
        unsigned hh, rr;                                                                          \
        split_pair3(__uint_as_float(rq[2 * q][e]), __uint_as_float(rq[2 * q + 1][e]), xscale, hh, rr); \
        h[q] = hh; r[q] = rr;                                                                     \
      }                                                                                           \
      Xs[xpos0 + e] = h;                                                                          \
      Xs[XPOS + xpos0 + e] = r;                                                                   \
    }                                                                                             \
    if (db_x) {                                                                                   \
      _Pragma("unroll") for (int c = 0; c < 8; ++c)                                               \
        bacc[c] += (__uint_as_float(rq[c][0]) + __uint_as_float(rq[c][1])) + (__uint_as_float(rq[c][2]) + __uint_as_float(rq[c][3])); \
    }                                                                                             \
  }
#define W3T_STORE_Y()                                                                             \
  if (yt) {                                                                                       \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                               \
      u32x4 h, r;                                                                                 \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
        unsigned hh, rr;                                                                          \
        split_pair3(__uint_as_float(ry[2 * q][e]), __uint_as_float(ry[2 * q + 1][e]), dscale, hh, rr); \
        h[q] = hh; r[q] = rr;                                                                     \
      }                                                                                           \
      Ys[yunit0 + 4 * e] = h;                                                                     \
      Ys[YU + yunit0 + 4 * e] = r;                                                                \
    }                                                                                             \
    if (db_y) {                                                                                   \
      _Pragma("unroll") for (int c = 0; c < 8; ++c)                                               \
        bacc[c] += (__uint_as_float(ry[c][0]) + __uint_as_float(ry[c][1])) + (__uint_as_float(ry[c][2]) + __uint_as_float(ry[c][3])); \
    }                                                                                             \
  }
  // operands of k-step s_ (x-row (z, y) = (s_ >> 3, s_ & 7)) into register set b_
  // Operand reads of k-step s_ (x-row (z, y) = (s_ >> 3, s_ & 7)).  A (one row tile) is single-buffered and fetched one
  // HALF-step ahead -- tile 1's while tile 0's three MFMAs run and vice versa -- B (shared by both tiles) is
  // double-buffered and fetched one step ahead: 32 operand registers instead of 48 (three chunks of accumulators +
  // both staging sets + 48 did not fit 256 registers).
#define W3T_READ_A(t_, s_)                                                                        \
  {                                                                                               \
    const unsigned ko = (unsigned)((((s_) >> 3) * SZP + ((s_) & 7) * HXP) * 16);                  \
    TR_PAIR(A1[t_], aaddr[t_] + ko + XPOS * 16u)                                                  \
    TR_PAIR(A0[t_], aaddr[t_] + ko)                                                               \
  }
#define W3T_READ_B(B_, b_, s_, off_)                                                              \
  {                                                                                               \
    const uint2 u0_ = tr_read8(baddr[0] + (unsigned)(s_) * 1024u + (off_)), u1_ = tr_read8(baddr[1] + (unsigned)(s_) * 1024u + (off_)); \
    B_[b_] = u32x4{u0_.x, u0_.y, u1_.x, u1_.y};                                                   \
  }
  // the 16 k-steps of chunk c_ (LAST_: the next phase starts a new tile, so dY is prefetched too -- a wave-uniform
  // branch around one load per step; two copies of the loop behind one branch cost 32 registers of accumulator copies)
#define W3T_KLOOP(c_, LAST_)                                                                      \
  {                                                                                               \
    W3T_READ_B(B0, 0, 0, 0u) W3T_READ_B(B1, 0, 0, YU * 16u) W3T_READ_A(0, 0)                      \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    _Pragma("unroll") for (int s = 0; s < NKS; ++s) {                                             \
      const int cur = s & 1;                                                                      \
      if (!(W3T_KO & 8)) {                                                                        \
        if (tok1) { W3T_READ_A(1, s) }                                                            \
        if (s + 1 < NKS) W3T_READ_B(B0, cur ^ 1, s + 1, 0u)                                       \
      }                                                                                           \
      if (!(W3T_KO & 2)) {                                                                        \
        if (s < 8) { W3T_GLOAD_X1(ca_next, s); }                                                  \
        else if (LAST_) { W3T_GLOAD_Y1(s - 8); }                                                  \
      }                                                                                           \
      W3T_MMA(c_, 0)                                                                              \
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                          \
      __builtin_amdgcn_sched_barrier(0);                                                          \
      if (s + 1 < NKS && !(W3T_KO & 8)) {                                                         \
        W3T_READ_A(0, s + 1)                                                                      \
        W3T_READ_B(B1, cur ^ 1, s + 1, YU * 16u)                                                  \
      }                                                                                           \
      if (PAIR && (LAST_) && !(W3T_KO & 2)) { W3T_GLOAD_Y1(s); }   /* 8 k-steps: dY rides in the second half-steps */ \
      if (tok1) { W3T_MMA(c_, 1) }                                                                \
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                          \
      if (PAIR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                          \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    if constexpr (PAIR) {                                                                         \
      /* tile 8, k-steps 2 wid and 2 wid + 1 (wave-uniform offsets: one add per address) */        \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                             \
        const unsigned ko = (unsigned)((2 * wid + u) * HXP * 16), kb = (unsigned)(2 * wid + u) * 1024u; \
        TR_PAIR(A1[u], aaddr[2] + ko + XPOS * 16u)                                                \
        TR_PAIR(A0[u], aaddr[2] + ko)                                                             \
        { const uint2 u0_ = tr_read8(baddr[0] + kb), u1_ = tr_read8(baddr[1] + kb);               \
          B0[u] = u32x4{u0_.x, u0_.y, u1_.x, u1_.y}; }                                            \
        { const uint2 v0_ = tr_read8(baddr[0] + kb + YU * 16u), v1_ = tr_read8(baddr[1] + kb + YU * 16u); \
          B1[u] = u32x4{v0_.x, v0_.y, v1_.x, v1_.y}; }                                            \
      }                                                                                           \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                             \
        acc[c_][NA - 1] = mma3(A1[u], B0[u], acc[c_][NA - 1]);                                    \
        acc[c_][NA - 1] = mma3(A0[u], B1[u], acc[c_][NA - 1]);                                    \
        acc[c_][NA - 1] = mma3(A0[u], B0[u], acc[c_][NA - 1]);                                    \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
  }
#if (W3T_KO & 1)
#define W3T_MMA(c_, t) acc[c_][t][0] += __uint_as_float(A1[t][0] ^ B0[cur][1] ^ A0[t][2] ^ B1[cur][3]);
#else
#define W3T_MMA(c_, t)                                                                            \
  acc[c_][t] = mma3(A1[t], B0[cur], acc[c_][t]);                                                  \
  acc[c_][t] = mma3(A0[t], B1[cur], acc[c_][t]);                                                  \
  acc[c_][t] = mma3(A0[t], B0[cur], acc[c_][t]);
#endif

  u32x4 A0[2], A1[2], B0[2], B1[2];
  unsigned aaddr[NA];

  // prologue: first tile's dY and chunk 0
  W3T_TILE_ADDR();
#pragma unroll
  for (int c = 0; c < 8; ++c) { W3T_GLOAD_X1(c_base, c); W3T_GLOAD_Y1(c); }
  W3T_STORE_X();
  W3T_STORE_Y();
  __syncthreads();

  for (int it = 0; it < niter; ++it) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c >= nc) continue;
      const bool last = (c + 1 == nc);                     // the next phase starts a new tile (or nothing)
      const bool more = !last || (it + 1 < niter);
      if (last) {                                          // advance the load cursor to the next tile
        if (more) { W3T_DECODE(t_first + (long long)(it + 1) * J) W3T_TILE_ADDR(); } else { gq = OOB; gy_ = OOB; }
      }
      const int ca_next = last ? c_base : c_base + c + 1;
      // this wave's two row tiles of chunk c: lane (j, q, g) of tile T supplies tap 4T + 2g + (q >> 1), channel quad q & 1
#pragma unroll
      for (int t = 0; t < NA; ++t) {
        const int tile = PAIR ? (t < 2 ? wid + 4 * t : 8) : ((wid + c) & 3) + 4 * t;
        int tap = tile * 4 + 2 * sg + (sq >> 1);           // PAIR: tap' of 36 (dz' = tap' / 9 in 0..3)
        if (!PAIR && tap > 26) tap = 26;                   // padding rows: any valid address (results discarded)
        const int dz = tap / 9, dyy = (tap / 3) % 3, dx = tap % 3;
        aaddr[t] = xs_base + (unsigned)((dz * SZP + dyy * HXP + dx + 8 * hi + sj) * 16 + (sq & 1) * 8);
      }
      // the wave whose second row tile is the padding tile 7 skips its reads and MFMAs (wave-uniform): no time gained
      // (the phase ends at the barrier) but 1/8 of the matrix and LDS energy of a kernel that sits on the power cap
      const bool tok1 = PAIR || (((wid + c) & 3) != 3) || W3T_NOSKIP;
      W3T_KLOOP(c, last)
      if (more) {
        __syncthreads();
        if (!(W3T_KO & 4)) {
          W3T_STORE_X();
          if (last) W3T_STORE_Y();
        }
        __syncthreads();
      }
    }
  }
#undef W3T_TILE_ADDR
#undef W3T_DECODE
#undef W3T_GLOAD_X1
#undef W3T_GLOAD_Y1
#undef W3T_STORE_X
#undef W3T_STORE_Y
#undef W3T_READ_A
#undef W3T_READ_B
#undef W3T_KLOOP
#undef W3T_MMA

  // ---- epilogue: acc[c][s][r] <-> row (r>>2)*8 + hi*4 + (r&3) of tile ((wid+c)&3) + 4s, column co = l31
  const float sc = oscale * oscale2;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int s = 0; s < NA; ++s) {
      const int tile = PAIR ? (s < 2 ? wid + 4 * s : 8) : ((wid + c) & 3) + 4 * s;
      const int co = PAIR ? (l31 & 15) : l31, pl = PAIR ? (l31 >> 4) : 0;
      if (c >= nc || (!PAIR && tile >= 7) || co >= k.Cout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rho = tile * 32 + (r >> 2) * 8 + hi * 4 + (r & 3);
        int tap = rho >> 3;
        const int ci = (c_base + c) * 8 + (rho & 7);
        bool ok = ci < k.Cin;
        if constexpr (PAIR) {
          const int dz = tap / 9 - pl;                     // row plane dz' against column plane p
          ok = ok && (unsigned)dz < 3u;
          tap = dz * 9 + tap % 9;
        } else {
          ok = ok && rho < 216;
        }
        const int to = k.flip ? 26 - tap : tap;
        if (ok) df_acc(dwt, to * k.s_tap + ci * k.s_row + co * k.s_col, acc[c][s][r] * sc, k.fx);
      }
    }
  if (k.db) {                                              // wave-uniform: every lane of a wave holds the same 8 channels
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float t = wave_sum((db_y || db_x) ? bacc[c] : 0.f);
      const int ch = k.db_from_x ? c_base * 8 + c : wid * 8 + c;
      const int lim = k.db_from_x ? k.Cin : k.Cout;
      if (lane == 0 && ch < lim && t != 0.f) atomicAdd(&k.db[ch], t);
    }
  }
}

static bool split3d_wgrad_common_ok(const DfConvGeom* g) {
  return g->KD == 3 && g->KH == 3 && g->KW == 3 && g->stride == 1 && g->dil == 1 && g->pd == 1 && g->ph == 1 &&
         g->pw == 1 && g->pad_mode == 0 && g->Do == g->Di && g->Ho == g->Hi && g->Wo == g->Wi && g->Di > 1 &&
         (g->Wi & 3) == 0 && (long long)g->Cin * g->Di * g->Hi * g->Wi * 4 < 0x7FFFFFFFLL &&
         (long long)g->Cout * g->Di * g->Hi * g->Wi * 4 < 0x7FFFFFFFLL;
}
// normal roles: rows = (tap, ci), columns = co.  Few output channels (the 16 -> 3 flow conv): SWAPPED roles -- rows =
// (tap, co) gathered from shifted dY, columns = ci: dW[t][ci][co] = sum_u X[ci][u] * dY[co][u - t], i.e. the same kernel
// on (x := dY, dy := X) with the taps flipped and the output transposed (one 8-channel chunk instead of Cin / 8).
static bool split3d_wgrad_geom_ok(const DfConvGeom* g) {
  return split3d_wgrad_common_ok(g) && g->Cin >= 8 && g->Cin <= 128 && g->Cout >= 8 && g->Cout <= 32;
}
static bool split3d_wgrad_swapped_ok(const DfConvGeom* g) {
  return split3d_wgrad_common_ok(g) && g->Cout >= 1 && g->Cout < 8 && g->Cin >= 8 && g->Cin <= 32;
}
extern "C" int dfmir_conv3d_split_wgrad_ok(const DfConvGeom* g) {
  return (g && !split3d_off() && (split3d_wgrad_geom_ok(g) || split3d_wgrad_swapped_ok(g))) ? 1 : 0;
}
static int conv3d_split_wgrad_impl(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                   const float* dy, const float* dy_amax, int dy_amax_n, float* dw_tcc, float* db,
                                   void* stream, const float* xa = nullptr, int Ca = 0, long long s_tap_full = 0);
// The weight gradient of conv3x3x3 over X = cat(nearest_up2(a), b) without building X: a [N, Ca, D/2, H/2, W/2] is read
// at (z >> 1, y >> 1, x >> 1) while the operand patch is staged.  g: the full layer (Cin = Ca + Cb); Ca % 8 == 0; even
// D, H, W; x_amax: a range probe valid for both parts.
extern "C" int dfmir_conv3d_split_wgrad_upcat(const DfConvGeom* g, const float* a, const float* b, int Ca,
                                              const float* x_amax, int x_amax_n, const float* dy, const float* dy_amax,
                                              int dy_amax_n, float* dw_tcc, float* db, void* stream) {
  DF_ARG_CHECK(g && a && b && Ca > 0 && (Ca & 7) == 0 && Ca < g->Cin && !(g->Di & 1) && !(g->Hi & 1) && !(g->Wi & 7));
  static DfOptFlag copies_o{"DFMIR_CONV3D_WGRAD_COPIES"};
  DF_ARG_CHECK(split3d_wgrad_geom_ok(g) && !copies_o.get() && (reinterpret_cast<uintptr_t>(a) & 7) == 0);
  return conv3d_split_wgrad_impl(g, b, x_amax, x_amax_n, dy, dy_amax, dy_amax_n, dw_tcc, db, stream, a, Ca);
}
// The same gradient with the up-sampled channels in PARITY CLASSES (conv3duw.hip: 8 / 27 of their products, one pass over
// dY at 1.0 x) and the skip channels b on the direct kernel, which also carries the bias gradient.  Ca == 32 (every
// decoder level of the VoxelMorph U-Net), Cout a multiple of 8 up to 32.  ws: dfmir_conv3d_upwgrad_ws_floats() floats,
// private to the stream while the call is in flight.
int df_conv3d_upwgrad_launch(const float* a, const float* a_amax, int a_n, const float* b, const float* dy, const float* dy_amax,
                             int dy_n, float* dwt, long long s_tap, float* db, float* ws, int N, int Dl, int Hl, int Wl,
                             int Cout, hipStream_t st);
int df_conv3d_wgrad_march_launch(const float* x, const float* x_amax, int x_n, const float* dy, const float* dy_amax, int dy_n,
                                 float* dwt, float* db, int N, int D, int H, int W, hipStream_t st);
int df_conv3d_flow_wgrad_ok(const DfConvGeom* g, const float* x, const float* dy);
int df_conv3d_flow_wgrad_launch(const float* x, const float* x_amax, int x_n, const float* dy, const float* dy_amax, int dy_n,
                                float* dwt, float* db, int N, int D, int H, int W, int Cout, hipStream_t st);
static bool wgrad_march_geom_ok(const DfConvGeom* g) {
  static DfOptFlag nomarch_o{"DFMIR_CONV3D_NO_WGRAD_MARCH"};
  return g->Cin == 32 && g->Cout == 16 && !split3d_off() && split3d_wgrad_geom_ok(g) && !nomarch_o.get() &&
         (long long)g->Di * g->Hi * g->Wi >= 4096;
}
// what the launcher decides: the geometry AND 16-byte aligned operands (x, dy are read as quads)
static bool wgrad_march_takes(const DfConvGeom* g, const float* x, const float* dy) {
  return wgrad_march_geom_ok(g) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
}
// 1 when dfmir_conv3d_split_wgrad / _db take the marching kernel for this layer (csrc/conv3dwm.hip): _is_march answers
// for 16-byte aligned operands, _is_march_at for the operands given (the SAME predicate the launcher applies)
extern "C" int dfmir_conv3d_wgrad_is_march(const DfConvGeom* g) { return (g && wgrad_march_geom_ok(g)) ? 1 : 0; }
extern "C" int dfmir_conv3d_wgrad_is_march_at(const DfConvGeom* g, const float* x, const float* dy) {
  return (g && wgrad_march_takes(g, x, dy)) ? 1 : 0;
}
static bool upwgrad_geom_ok(const DfConvGeom* g, int Ca) {
  static DfOptFlag off_o{"DFMIR_UPWGRAD_DIRECT"};
  return !off_o.get() && !split3d_off() && split3d_wgrad_common_ok(g) && Ca == 32 && g->Cin > Ca && g->Cin - Ca <= 128 &&
         g->Cout >= 8 && g->Cout <= 32 && (g->Cout & 7) == 0 && !(g->Di & 1) && !(g->Hi & 1) && !(g->Wi & 7);
}
extern "C" int dfmir_conv3d_upwgrad_ok(const DfConvGeom* g, int Ca) { return (g && upwgrad_geom_ok(g, Ca)) ? 1 : 0; }
extern "C" long long dfmir_conv3d_upwgrad_ws_floats(void) { return 2 * 72LL * 1024; }   // (64-bit slots in deterministic mode)
extern "C" int dfmir_conv3d_upwgrad(const DfConvGeom* g, const float* a, const float* b, int Ca, const float* x_amax,
                                    int x_amax_n, const float* dy, const float* dy_amax, int dy_amax_n, float* dw_tcc,
                                    float* db, float* ws, void* stream) {
  DF_ARG_CHECK(g && a && b && ws && x_amax && x_amax_n > 0 && dy && dy_amax && dy_amax_n > 0 && dw_tcc);
  DF_ARG_CHECK(upwgrad_geom_ok(g, Ca) && (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(b) & 7) == 0 &&
               (reinterpret_cast<uintptr_t>(dy) & 15) == 0);   // 8-byte pair loads of a and b, 16-byte quads of dY
  const long long s_tap = (long long)g->Cin * g->Cout;
  // two skip channels (the network's input images at the top level): fused into the same launch, with the bias gradient
  static DfOptFlag nofuse_o{"DFMIR_UPWGRAD_NO_FUSEB"};
  const bool fuse = g->Cin - Ca == 2 && !nofuse_o.get();
  const int rc = df_conv3d_upwgrad_launch(a, x_amax, x_amax_n, fuse ? b : nullptr, dy, dy_amax, dy_amax_n, dw_tcc, s_tap,
                                          fuse ? db : nullptr, ws, g->N, g->Di / 2, g->Hi / 2, g->Wi / 2, g->Cout,
                                          (hipStream_t)stream);
  if (rc || fuse) return rc;
  DfConvGeom gb = *g;
  gb.Cin = g->Cin - Ca;
  return conv3d_split_wgrad_impl(&gb, b, x_amax, x_amax_n, dy, dy_amax, dy_amax_n, dw_tcc + (df_det_fx() ? 2LL : 1LL) * Ca * g->Cout, db,   // (deterministic mode: 8-byte slots)
                                 stream, nullptr, 0, s_tap);
}
extern "C" int dfmir_conv3d_split_wgrad(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                        const float* dy, const float* dy_amax, int dy_amax_n, float* dw_tcc,
                                        void* stream) {
  return conv3d_split_wgrad_impl(g, x, x_amax, x_amax_n, dy, dy_amax, dy_amax_n, dw_tcc, nullptr, stream);
}
extern "C" int dfmir_conv3d_split_wgrad_db(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                           const float* dy, const float* dy_amax, int dy_amax_n, float* dw_tcc,
                                           float* db, void* stream) {
  return conv3d_split_wgrad_impl(g, x, x_amax, x_amax_n, dy, dy_amax, dy_amax_n, dw_tcc, db, stream);
}
static int conv3d_split_wgrad_impl(const DfConvGeom* g, const float* x, const float* x_amax, int x_amax_n,
                                   const float* dy, const float* dy_amax, int dy_amax_n, float* dw_tcc, float* db,
                                   void* stream, const float* xa, int Ca, long long s_tap_full) {
  DF_ARG_CHECK(g && x && x_amax && x_amax_n > 0 && dy && dy_amax && dy_amax_n > 0 && dw_tcc);
  // s_tap_full: the rows of a wider gradient [27][Ctot][Cout] (dw_tcc points at this operand's first row): the skip
  // channels of dfmir_conv3d_upwgrad, any Cin >= 1 (a partial 8-channel chunk reads zeros beyond Cin)
  const bool rows = s_tap_full > 0 && split3d_wgrad_common_ok(g) && g->Cin >= 1 && g->Cin <= 128 && g->Cout >= 8 && g->Cout <= 32;
  DF_ARG_CHECK(!split3d_off() && (rows || split3d_wgrad_geom_ok(g) || split3d_wgrad_swapped_ok(g)));
  hipStream_t st = (hipStream_t)stream;
  // the full-resolution 32 -> 16 layer: all 27 tap matrices resident, z-marching (conv3dwm.hip)
  if (!rows && !xa && wgrad_march_takes(g, x, dy))
    return df_conv3d_wgrad_march_launch(x, x_amax, x_amax_n, dy, dy_amax, dy_amax_n, dw_tcc, db, g->N, g->Di, g->Hi, g->Wi, st);
  const bool swapped = !rows && !split3d_wgrad_geom_ok(g);
  // the flow head 16 -> 3: (co, dx) pairs as MFMA columns, z-marching (conv3dt.hip)
  if (swapped && !xa && df_conv3d_flow_wgrad_ok(g, x, dy))
    return df_conv3d_flow_wgrad_launch(x, x_amax, x_amax_n, dy, dy_amax, dy_amax_n, dw_tcc, db, g->N, g->Di, g->Hi, g->Wi, g->Cout, st);
  W3sP k{};
  k.fx = df_det_fx();
  k.N = g->N; k.D = g->Di; k.H = g->Hi; k.W = g->Wi;
  k.xa = xa; k.Ca = xa ? Ca : 0;
  if (swapped) {
    k.Cin = g->Cout; k.Cout = g->Cin;                        // kernel roles
    k.s_tap = (long long)g->Cin * g->Cout; k.s_row = 1; k.s_col = g->Cout; k.flip = 1;
    k.x_n = dy_amax_n; k.dy_n = x_amax_n;
    k.db = db; k.db_from_x = 1;
  } else {
    k.Cin = g->Cin; k.Cout = g->Cout;
    k.s_tap = rows ? s_tap_full : (long long)g->Cin * g->Cout; k.s_row = g->Cout; k.s_col = 1; k.flip = 0;
    k.x_n = x_amax_n; k.dy_n = dy_amax_n;
    k.db = db; k.db_from_x = 0;
  }
  static DfOptFlag tr_o{"DFMIR_CONV3D_WGRAD_COPIES"};
  const bool tr_off = tr_o.get();   // A/B: the three-copy kernel
  k.nz = (g->Di + 1) / 2; k.ny = tr_off ? (g->Hi + 3) / 4 : (g->Hi + 7) / 8; k.nx = (g->Wi + 15) / 16;
  k.npatch = (long long)g->N * k.nz * k.ny * k.nx;
  long long want = 512;
  if (want > k.npatch) want = k.npatch;
  k.per_block = (k.npatch + want - 1) / want;
  // <= 3 chunks of accumulators per workgroup (96 AGPRs + staging registers: two workgroups per CU, so that one
  // converts while the other computes); more input channels = a second workgroup row, which stages dY again
  k.nchunk = (k.Cin + 7) / 8;
  static DfOptFlag pair_o{"DFMIR_CONV3D_WGRAD_NO_PAIR"};
  const bool pair_off = pair_o.get();
  const bool pairw = !tr_off && !pair_off && !swapped && k.Cout <= 16;   // (swapped flow head: measured slower, 0.44 vs 0.31 ms)         // (kernel roles) plane-pair columns: 3 accumulators per chunk
  const int per_wg = pairw ? (k.nchunk < 2 ? k.nchunk : 2) : (k.nchunk <= 3 ? k.nchunk : (k.nchunk == 4 ? 2 : 3));
  const unsigned gy = (unsigned)((k.nchunk + per_wg - 1) / per_wg);
  if (gy > 1) {                                             // keep the number of workgroups
    want = 512 / gy;
    if (want > k.npatch) want = k.npatch;
    k.per_block = (k.npatch + want - 1) / want;
  }
  unsigned nbx = (unsigned)((k.npatch + k.per_block - 1) / k.per_block);
  if (!tr_off) {
    // tr kernel: the tile list includes the phantom (z, y) cells of odd tile counts (they load nothing); 8 XCD shares,
    // nslot workgroups each (see the kernel)
    k.npatch = (long long)g->N * k.nx * 4 * ((k.nz + 1) / 2) * ((k.ny + 1) / 2);
    k.per_block = (k.npatch + 7) / 8;
    k.nslot = (long long)(512 / gy / 8);
    if (k.nslot > k.per_block) k.nslot = k.per_block;
    if (k.nslot < 1) k.nslot = 1;
    nbx = (unsigned)(8 * k.nslot);
  }
  const float *kx = swapped ? dy : x, *kxa = swapped ? dy_amax : x_amax, *kdy = swapped ? x : dy, *kda = swapped ? x_amax : dy_amax;
#define W3S_LAUNCH(N_)                                                                            \
  {                                                                                               \
    if (tr_off) conv3d_wgrad_split_k<N_><<<dim3(nbx, gy), 256, 0, st>>>(kx, kxa, kdy, kda, dw_tcc, k);   \
    else if (xa) conv3d_wgrad_tr_k<N_, false, true><<<dim3(nbx, gy), 256, 0, st>>>(kx, kxa, kdy, kda, dw_tcc, k);   \
    else conv3d_wgrad_tr_k<N_, false, false><<<dim3(nbx, gy), 256, 0, st>>>(kx, kxa, kdy, kda, dw_tcc, k);   \
  }
  if (pairw && per_wg == 1 && xa) conv3d_wgrad_tr_k<1, true, true><<<dim3(nbx, gy), 256, 0, st>>>(kx, kxa, kdy, kda, dw_tcc, k);
  else if (pairw && xa) conv3d_wgrad_tr_k<2, true, true><<<dim3(nbx, gy), 256, 0, st>>>(kx, kxa, kdy, kda, dw_tcc, k);
  else if (pairw && per_wg == 1) conv3d_wgrad_tr_k<1, true, false><<<dim3(nbx, gy), 256, 0, st>>>(kx, kxa, kdy, kda, dw_tcc, k);
  else if (pairw) conv3d_wgrad_tr_k<2, true, false><<<dim3(nbx, gy), 256, 0, st>>>(kx, kxa, kdy, kda, dw_tcc, k);
  else if (per_wg == 1) W3S_LAUNCH(1)
  else if (per_wg == 2) W3S_LAUNCH(2)
  else W3S_LAUNCH(3)
#undef W3S_LAUNCH
  DF_LAUNCH_CHECK();
  return 0;
}

// PatchNCE head plumbing that keeps the train step off the host: device-side patch-id draws, a multi-source patch
// gather for the key side, and the scalar loss algebra (segment means of the per-row NCE losses, the final affine
// combination of the step's loss terms) as single launches.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// Patch ids.  PatchSampleF draws `torch.randperm(H*W)[:num_patches]` per feature layer and per call
// (models/networks.py:609-610): a uniformly random P-subset of [0, S) in random order; only the SET matters to
// PatchNCELoss (rows are exchangeable: every row is a positive once and a negative of the others).  One workgroup
// per (layer, set) draws it without a sort of S keys:
//   round 0: thread i draws a uniform candidate; the P (candidate, thread) pairs are sorted in LDS (bitonic);
//   a candidate equal to its left neighbour loses (an element accepted in an earlier round sorts first and always
//   wins, then the lower thread id) and redraws in the next round.  Deterministic for a given (seed, counter).
// The generator is counter-based (splitmix64 finaliser over (seed, draw counter, stream, thread, round)); the draw
// counter lives in device memory and is advanced by the last workgroup to finish, so a captured hipGraph replays
// fresh ids every step.  That form needs S >= 2P (every redraw is accepted with probability >= 1/2; the 64-round cap
// is then never reached: 2^-64 per element).  Denser layers (P <= S < 2P, S <= 4096: the 16x16 maps of a 64x64 run)
// take the other branch: S random keys sorted in LDS, the first P positions of that permutation.
// ------------------------------------------------------------------------------------------------
#define DF_IDS_MAXP 1024
#define DF_IDS_MAXPERM 4096
struct DfIdSizes {
  long long S[8];
};

__device__ __forceinline__ unsigned long long df_mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void patch_ids_draw_k(unsigned long long* __restrict__ state, DfIdSizes sizes,
                                                        int n_layers, int n_sets, int P, int Ppad,
                                                        long long* __restrict__ out) {
  // element = (candidate << 12 | fresh << 11 | owner thread) : sorts by candidate, accepted before fresh, then owner
  __shared__ unsigned long long el[DF_IDS_MAXPERM];
  __shared__ unsigned char lost[DF_IDS_MAXP];
  __shared__ int pending;
  const int layer = blockIdx.x / n_sets, set = blockIdx.x - layer * n_sets;
  const unsigned long long S = (unsigned long long)sizes.S[layer];
  const unsigned long long seed = state[0], counter = state[1];
  const unsigned long long stream = df_mix64(seed ^ df_mix64(counter * 0x100000001B3ull + (unsigned long long)blockIdx.x));
  long long* o = out + ((long long)layer * n_sets + set) * P;
  if (S < 2ull * (unsigned long long)P) {
    // dense layer: a full random permutation of [0, S) (distinct keys: 52 random bits | position), first P of it
    int Spad = 1;
    while (Spad < (int)S) Spad <<= 1;
    for (int i = threadIdx.x; i < Spad; i += 256)
      el[i] = i < (int)S ? ((df_mix64(stream ^ ((unsigned long long)i * 0xD6E8FEB86659FD93ull)) >> 12) << 12) | (unsigned long long)i
                         : ~0ull;
    __syncthreads();
    for (int k = 2; k <= Spad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < Spad; i += 256) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long a = el[i], b = el[ixj];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { el[i] = b; el[ixj] = a; }
          }
        }
        __syncthreads();
      }
    for (int i = threadIdx.x; i < P; i += 256) o[i] = (long long)(el[i] & 0xFFFull);
  } else {
  // each thread owns slots tid, tid+256, ... (P <= 1024)
  long long mine[DF_IDS_MAXP / 256];
  bool fresh[DF_IDS_MAXP / 256];
#pragma unroll
  for (int j = 0; j < DF_IDS_MAXP / 256; ++j) { mine[j] = -1; fresh[j] = true; }
  for (int round = 0; round < 64; ++round) {
#pragma unroll
    for (int j = 0; j < DF_IDS_MAXP / 256; ++j) {
      const int slot = threadIdx.x + j * 256;
      if (slot < Ppad) {
        unsigned long long e = ~0ull;                      // padding sorts last
        if (slot < P) {
          if (fresh[j]) {
            const unsigned long long r = df_mix64(stream ^ ((unsigned long long)slot * 0xD6E8FEB86659FD93ull +
                                                            (unsigned long long)round * 0xA0761D6478BD642Full));
            mine[j] = (long long)(((r >> 32) * S) >> 32);  // uniform on [0, S), bias <= S / 2^32
          }
          e = ((unsigned long long)mine[j] << 12) | ((unsigned long long)(fresh[j] ? 1 : 0) << 11) | (unsigned long long)slot;
        }
        el[slot] = e;
      }
    }
    if (threadIdx.x == 0) pending = 0;
    __syncthreads();
    for (int k = 2; k <= Ppad; k <<= 1)                    // bitonic sort of Ppad (power of two) elements
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < Ppad; i += 256) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long a = el[i], b = el[ixj];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { el[i] = b; el[ixj] = a; }
          }
        }
        __syncthreads();
      }
    // duplicates: an element whose left neighbour holds the same candidate redraws (it is fresh by construction:
    // accepted elements are distinct and sort first); verdicts go back to the owners through a flag per slot
    for (int i = threadIdx.x; i < Ppad; i += 256) lost[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x + 1; i < Ppad; i += 256) {
      const unsigned long long e = el[i];
      if (e != ~0ull && (el[i - 1] >> 12) == (e >> 12)) lost[(int)(e & 0x7FFull)] = 1;
    }
    __syncthreads();
    bool any = false;
#pragma unroll
    for (int j = 0; j < DF_IDS_MAXP / 256; ++j) {
      const int slot = threadIdx.x + j * 256;
      if (slot < P) {
        fresh[j] = lost[slot] != 0;
        any = any || fresh[j];
      }
    }
    if (any) atomicOr(&pending, 1);
    __syncthreads();
    const int more = pending;
    __syncthreads();
    if (!more) break;
  }
#pragma unroll
  for (int j = 0; j < DF_IDS_MAXP / 256; ++j) {
    const int slot = threadIdx.x + j * 256;
    if (slot < P) o[slot] = mine[j];
  }
  }
  // the last workgroup to finish advances the draw counter
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned long long done = atomicAdd(&state[2], 1ull);
    if (done == (unsigned long long)gridDim.x - 1) {
      state[2] = 0ull;
      state[1] = counter + 1ull;
      __threadfence();
    }
  }
}

extern "C" int dfmir_patch_ids_draw(unsigned long long* state, const long long* sizes, int n_layers, int n_sets,
                                    int P, long long* out, void* stream) {
  DF_ARG_CHECK(state && sizes && out && n_layers > 0 && n_layers <= 8 && n_sets > 0 && P > 0 && P <= DF_IDS_MAXP);
  DfIdSizes sz{};
  for (int l = 0; l < n_layers; ++l) {
    DF_ARG_CHECK(sizes[l] >= P && sizes[l] < (1LL << 32));
    DF_ARG_CHECK(sizes[l] >= 2LL * P || sizes[l] <= DF_IDS_MAXPERM);   // rejection sampling needs S >= 2P; else sort S keys
    sz.S[l] = sizes[l];
  }
  int Ppad = 1;
  while (Ppad < P) Ppad <<= 1;
  patch_ids_draw_k<<<(unsigned)(n_layers * n_sets), 256, 0, (hipStream_t)stream>>>(state, sz, n_layers, n_sets, P, Ppad,
                                                                               out);
  DF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Key-side gather of several NCE terms at once: group g takes its images from src[g] ([Bper, C, S], e.g. the two
// halves of the feature map forward() tapped) at ids[g][0..P)  ->  channel-major rows out[C][G*Bper*P].
// (PatchSampleF.forward's feat.permute(0,2,3,1).flatten(1,2)[:, patch_id, :], models/networks.py:604-611.)
// ------------------------------------------------------------------------------------------------
struct DfGatherSrcs {
  const float* p[8];
};
__global__ void patch_gather_multi_k(DfGatherSrcs srcs, const long long* __restrict__ ids, float* __restrict__ out,
                                     int G, int Bper, int C, long long S, int P) {
  const long long total = (long long)G * Bper * C * P;
  const long long rows = (long long)G * Bper * P;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    long long t = i / P;
    const int c = (int)(t % C);
    t /= C;
    const int b = (int)(t % Bper), g = (int)(t / Bper);
    out[(long long)c * rows + ((long long)g * Bper + b) * P + p] =
        srcs.p[g][((long long)b * C + c) * S + ids[(long long)g * P + p]];
  }
}
extern "C" int dfmir_patch_gather_fwd_multi(const float* const* srcs, int G, const long long* ids, float* out, int Bper,
                                            int C, long long S, int P, void* stream) {
  DF_ARG_CHECK(srcs && ids && out && G > 0 && G <= 8 && Bper > 0 && C > 0 && S > 0 && P > 0);
  DfGatherSrcs s{};
  for (int g = 0; g < G; ++g) {
    DF_ARG_CHECK(srcs[g] != nullptr);
    s.p[g] = srcs[g];
  }
  patch_gather_multi_k<<<df_grid((long long)G * Bper * C * P, 256, 4096), 256, 0, (hipStream_t)stream>>>(
      s, ids, out, G, Bper, C, S, P);
  DF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// The PatchNCE head of one layer in ONE launch: sample the patches, Linear(C, 256) + ReLU, Linear(256, 256), L2-normalise
// (PatchSampleF.forward, models/networks.py:602-619: feat.permute(0,2,3,1).flatten(1,2)[:, patch_id, :] -> mlp -> l2norm),
// for the rows of G groups (NCE terms) at once.  Replaces gather + two 1x1-conv launches + l2norm (4 launches per layer
// and side, the GEMMs at 44 TF on the generic kernel).
// Workgroup = 32 rows (patches) x all 256 output channels, 4 waves (wave w: channels 64 w .. 64 w + 63 = two 32-row MFMA
// tiles); products on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: the arithmetic of the unfused path).
//   phase 0  gather x[C][32] into LDS (and to `xs` when the backward needs it)
//   phase 1  h = relu(W1^T x + b1): A = W1[k][m] straight from global / L2 (consecutive lanes = consecutive m), B = x from LDS
//   phase 2  y = W2^T h + b2 likewise with h in LDS
//   phase 3  row norms (shuffle across the half-waves, LDS across the waves), out = y / (norm + eps)
// Saved for the backward on request: xs [C][rows], hs [256][rows], ypre [256][rows], nrm [rows].
// ------------------------------------------------------------------------------------------------
typedef float nh_f32x16 __attribute__((ext_vector_type(16)));
struct NceHeadP {
  DfGatherSrcs srcs;
  const long long* ids;
  const float *w1, *b1, *w2, *b2;
  float *out, *nrm, *xs, *hs, *ypre;
  int G, Bper, C, P;
  long long S, rows;
  float eps;
};
constexpr int NH_R = 32, NH_M = 256, NH_CMAX = 256, NH_KB = 8;
__global__ __launch_bounds__(256, 2) void nce_head_fwd_k(NceHeadP k) {
  __shared__ float Xs[NH_CMAX * NH_R];
  __shared__ float Hs[NH_M * NH_R];
  __shared__ float red[4][NH_R];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const long long r0 = (long long)blockIdx.x * NH_R;
  // ---- phase 0: thread (c0 = tid >> 5, r = tid & 31) gathers channels c0, c0 + 8, ... of row r0 + r
  {
    const int r = tid & 31;
    const long long rr = r0 + r;
    const bool rok = rr < k.rows;
    const long long gb = rok ? rr / k.P : 0;
    const int pp = rok ? (int)(rr - gb * k.P) : 0;
    const int g = (int)(gb / k.Bper), b = (int)(gb - (long long)g * k.Bper);
    const long long id = rok ? k.ids[(long long)g * k.P + pp] : 0;
    const float* src = k.srcs.p[g] + (long long)b * k.C * k.S + id;
    const int Cp = (k.C + 2 * NH_KB - 1) / (2 * NH_KB) * (2 * NH_KB);   // the K loop runs in batches of NH_KB row pairs: zero rows up to there
    // 8 scattered loads in flight per thread (each one its own cache line): issued one at a time in front of its LDS
    // store this phase alone cost more than the two GEMMs
    for (int cb = tid >> 5; cb < Cp; cb += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = cb + 8 * u;
        v[u] = (rok && c < k.C) ? src[(long long)c * k.S] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = cb + 8 * u;
        if (c < Cp) {
          Xs[c * NH_R + r] = v[u];
          if (k.xs && rok && c < k.C) k.xs[(long long)c * k.rows + rr] = v[u];
        }
      }
    }
  }
  __syncthreads();
  const int m0 = 64 * wid;
  nh_f32x16 acc[2];
  // ---- phase 1
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  // K loop in batches of NH_KB pairs: the weights of batch j + 1 are fetched (global / L2, one dword per lane and MFMA)
  // while the MFMAs of batch j run -- fetched at the point of use the loop was bound by that latency (107 us per launch)
#define NH_GEMM(W_, KTOT_, SRC_)                                                                  \
  {                                                                                               \
    const float* wp = (W_) + m0 + l31;                                                            \
    const int K2 = ((KTOT_) + 1) >> 1, nb = (K2 + NH_KB - 1) / NH_KB;                             \
    float ca0[NH_KB], ca1[NH_KB], na0[NH_KB], na1[NH_KB];                                         \
    _Pragma("unroll") for (int u = 0; u < NH_KB; ++u) {                                           \
      const int kr = 2 * u + hi;                                                                  \
      const bool ok = kr < (KTOT_);                                                               \
      ca0[u] = ok ? wp[(long long)kr * NH_M] : 0.f;                                               \
      ca1[u] = ok ? wp[(long long)kr * NH_M + 32] : 0.f;                                          \
    }                                                                                             \
    for (int j = 0; j < nb; ++j) {                                                                \
      if (j + 1 < nb) {                                                                           \
        _Pragma("unroll") for (int u = 0; u < NH_KB; ++u) {                                       \
          const int kr = 2 * ((j + 1) * NH_KB + u) + hi;                                          \
          const bool ok = kr < (KTOT_);                                                           \
          na0[u] = ok ? wp[(long long)kr * NH_M] : 0.f;                                           \
          na1[u] = ok ? wp[(long long)kr * NH_M + 32] : 0.f;                                      \
        }                                                                                         \
      }                                                                                           \
      _Pragma("unroll") for (int u = 0; u < NH_KB; ++u) {                                         \
        const int kr = 2 * (j * NH_KB + u) + hi;                                                  \
        const float bv = (SRC_)[kr * NH_R + l31];                                                 \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0[u], bv, acc[0], 0, 0, 0);               \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1[u], bv, acc[1], 0, 0, 0);               \
      }                                                                                           \
      _Pragma("unroll") for (int u = 0; u < NH_KB; ++u) { ca0[u] = na0[u]; ca1[u] = na1[u]; }     \
    }                                                                                             \
  }
  NH_GEMM(k.w1, k.C, Xs)
  // D layout: acc[i][e] <-> channel m0 + 32 i + (e >> 2) * 8 + hi * 4 + (e & 3), row l31
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + 32 * i + (e >> 2) * 8 + hi * 4 + (e & 3);
      float v = acc[i][e] + k.b1[m];
      v = v > 0.f ? v : 0.f;
      Hs[m * NH_R + l31] = v;
      if (k.hs && r0 + l31 < k.rows) k.hs[(long long)m * k.rows + r0 + l31] = v;
      acc[i][e] = 0.f;
    }
  __syncthreads();
  // ---- phase 2
  NH_GEMM(k.w2, NH_M, Hs)
#undef NH_GEMM
  // ---- phase 3: bias, row norms, normalise
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + 32 * i + (e >> 2) * 8 + hi * 4 + (e & 3);
      acc[i][e] += k.b2[m];
      ss += acc[i][e] * acc[i][e];
    }
  ss += __shfl_xor(ss, 32);
  if (hi == 0) red[wid][l31] = ss;
  __syncthreads();
  const float nr = sqrtf((red[0][l31] + red[1][l31]) + (red[2][l31] + red[3][l31]));
  const float inv = 1.f / (nr + k.eps);
  const long long rr = r0 + l31;
  if (rr < k.rows) {
    if (k.nrm && wid == 0 && hi == 0) k.nrm[rr] = nr;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + 32 * i + (e >> 2) * 8 + hi * 4 + (e & 3);
        k.out[(long long)m * k.rows + rr] = acc[i][e] * inv;
        if (k.ypre) k.ypre[(long long)m * k.rows + rr] = acc[i][e];
      }
  }
}
// srcs: HOST array of G <= 8 device pointers (group g's images [Bper, C, S]); ids [G][P]; w1 [C][256], w2 [256][256] in
// the forward packing of dfmir_weight_pack ([Cin][Cout]); out [256][G*Bper*P].  xs / hs / ypre / nrm may be NULL.
extern "C" int dfmir_nce_head_fwd(const float* const* srcs, int G, const long long* ids, const float* w1, const float* b1,
                                  const float* w2, const float* b2, float* out, float* nrm, float* xs, float* hs,
                                  float* ypre, int Bper, int C, long long S, int P, float eps, void* stream) {
  DF_ARG_CHECK(srcs && ids && w1 && b1 && w2 && b2 && out && G > 0 && G <= 8 && Bper > 0 && C > 0 && C <= NH_CMAX && S > 0 && P > 0);
  NceHeadP k{};
  for (int g = 0; g < G; ++g) {
    DF_ARG_CHECK(srcs[g] != nullptr);
    k.srcs.p[g] = srcs[g];
  }
  k.ids = ids; k.w1 = w1; k.b1 = b1; k.w2 = w2; k.b2 = b2;
  k.out = out; k.nrm = nrm; k.xs = xs; k.hs = hs; k.ypre = ypre;
  k.G = G; k.Bper = Bper; k.C = C; k.P = P; k.S = S; k.rows = (long long)G * Bper * P; k.eps = eps;
  nce_head_fwd_k<<<(unsigned)((k.rows + NH_R - 1) / NH_R), 256, 0, (hipStream_t)stream>>>(k);
  DF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// out[t] = scale * sum_l mean(rows[l][t*seg .. (t+1)*seg))  for rows [L][T*seg]: the per-term NCE losses
// `total_nce_loss += loss.mean() * lambda_NCE ... / n_layers` (models/registration_model.py:247-253), all terms and
// layers in one launch (deterministic: one workgroup per term, fixed reduction order).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segment_means_k(const float* __restrict__ rows, float* __restrict__ out, int L,
                                                       int T, long long seg, float scale) {
  __shared__ float sm[17];
  const int t = blockIdx.x;
  float s = 0.f;
  for (int l = 0; l < L; ++l) {
    const float* r = rows + ((long long)l * T + t) * seg;
    for (long long i = threadIdx.x; i < seg; i += 256) s += r[i];
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) out[t] = s * (scale / (float)seg);
}
__global__ void segment_means_bwd_k(const float* __restrict__ g, float* __restrict__ drows, int L, int T, long long seg,
                                    float scale) {
  const long long total = (long long)L * T * seg;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)((i / seg) % T);
    drows[i] = g[t] * (scale / (float)seg);
  }
}
extern "C" int dfmir_segment_means_fwd(const float* rows, float* out, int L, int T, long long seg, float scale,
                                       void* stream) {
  DF_ARG_CHECK(rows && out && L > 0 && T > 0 && seg > 0);
  segment_means_k<<<(unsigned)T, 256, 0, (hipStream_t)stream>>>(rows, out, L, T, seg, scale);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_segment_means_bwd(const float* gout, float* drows, int L, int T, long long seg, float scale,
                                       void* stream) {
  DF_ARG_CHECK(gout && drows && L > 0 && T > 0 && seg > 0);
  segment_means_bwd_k<<<df_grid((long long)L * T * seg, 256, 1024), 256, 0, (hipStream_t)stream>>>(gout, drows, L, T, seg,
                                                                                                scale);
  DF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// out[j] = sum_i M[j][i] * *in[i]   (n_in, n_out <= 8): the scalar algebra that turns the step's loss terms into
// loss_G / loss_R / loss_local / loss_smooth and their sum (models/registration_model.py:163-166,230-234) in one
// launch; backward: din[i] = sum_j M[j][i] * g[j].
// ------------------------------------------------------------------------------------------------
struct DfCombine {
  const float* in[8];
  float M[8][8];
};
__global__ void scalar_combine_k(DfCombine c, int n_in, int n_out, float* __restrict__ out) {
  const int j = threadIdx.x;
  if (j < n_out) {
    float s = 0.f;
    for (int i = 0; i < n_in; ++i) s += c.M[j][i] * c.in[i][0];
    out[j] = s;
  }
}
__global__ void scalar_combine_bwd_k(DfCombine c, const float* __restrict__ g, int n_in, int n_out,
                                     float* __restrict__ din) {
  const int i = threadIdx.x;
  if (i < n_in) {
    float s = 0.f;
    for (int j = 0; j < n_out; ++j) s += c.M[j][i] * g[j];
    din[i] = s;
  }
}
extern "C" int dfmir_scalar_combine_fwd(const float* const* in, int n_in, const float* M, int n_out, float* out,
                                        void* stream) {
  DF_ARG_CHECK(in && M && out && n_in > 0 && n_in <= 8 && n_out > 0 && n_out <= 8);
  DfCombine c{};
  for (int i = 0; i < n_in; ++i) {
    DF_ARG_CHECK(in[i] != nullptr);
    c.in[i] = in[i];
  }
  for (int j = 0; j < n_out; ++j)
    for (int i = 0; i < n_in; ++i) c.M[j][i] = M[j * n_in + i];
  scalar_combine_k<<<1, 64, 0, (hipStream_t)stream>>>(c, n_in, n_out, out);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_scalar_combine_bwd(const float* gout, int n_in, const float* M, int n_out, float* din,
                                        void* stream) {
  DF_ARG_CHECK(gout && M && din && n_in > 0 && n_in <= 8 && n_out > 0 && n_out <= 8);
  DfCombine c{};
  for (int j = 0; j < n_out; ++j)
    for (int i = 0; i < n_in; ++i) c.M[j][i] = M[j * n_in + i];
  scalar_combine_bwd_k<<<1, 64, 0, (hipStream_t)stream>>>(c, gout, n_in, n_out, din);
  DF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// out[i] = max(a[i], b[i]) over DF_PROBE_SLOTS floats: the range probe of cat([up2(a), b]) from its inputs' probes
// (nearest up-sampling and concatenation create no new values).
// ------------------------------------------------------------------------------------------------
__global__ void probe_merge_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
  out[threadIdx.x] = fmaxf(a[threadIdx.x], b[threadIdx.x]);
}
extern "C" int dfmir_probe_merge(const float* a, const float* b, float* out, void* stream) {
  DF_ARG_CHECK(a && b && out);
  probe_merge_k<<<1, DF_PROBE_SLOTS, 0, (hipStream_t)stream>>>(a, b, out);
  DF_LAUNCH_CHECK();
  return 0;
}

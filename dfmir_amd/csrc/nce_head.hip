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

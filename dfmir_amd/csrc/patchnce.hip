// PatchNCE path: patch gather (NCHW -> channel-major [B,C,P] rows), L2 normalisation over
// channels, and the fused InfoNCE loss (logits GEMM + diagonal fill + softmax cross-entropy in one
// kernel, probabilities kept for backward).
#include "common.h"

// ids[G][P]: image b samples the positions of group b / (B / G)  (G = 1: one id set for the whole batch, as
// PatchSampleF draws it; G > 1: several NCE terms' query images stacked along the batch, each with its term's ids)
__global__ void patch_gather_fwd_k(const float* __restrict__ feat, const long long* __restrict__ ids,
                                   float* __restrict__ out, int B, int C, long long S, int P, int bpg) {
  const long long total = (long long)B * C * P;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    const long long bc = i / P;
    const long long b = bc / C, c = bc - b * C;
    out[c * ((long long)B * P) + b * P + p] = feat[bc * S + ids[(b / bpg) * P + p]];
  }
}
// amax (optional): the range probe of dfeat (DF_PROBE_SLOTS floats, as left by the InstanceNorm kernels); kept valid
// by raising a slot to |new value| of every element this scatter touches (ids are distinct within a plane)
// DISTINCT: the ids of a group are a P-SUBSET (as torch.randperm / dfmir_patch_ids_draw yield them), so no two threads of
// the launch touch the same element and the scatter is a plain load + store -- 3.1 M returning L2 atomics per layer
// were the whole cost of this kernel (125 -> ~35 us).  The non-DISTINCT form accumulates arbitrary ids with atomics.
template <bool DISTINCT>
__global__ void patch_gather_bwd_k(const float* __restrict__ dout, const long long* __restrict__ ids,
                                   float* __restrict__ dfeat, int B, int C, long long S, int P, int bpg,
                                   unsigned* __restrict__ amax, unsigned* __restrict__ pmax = nullptr) {
  const long long total = (long long)B * C * P;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int p;
    long long bc, b, c;
    if (total < 0x7FFFFFFFLL) {
      const unsigned u = (unsigned)i;
      p = (int)(u % (unsigned)P);
      const unsigned ubc = u / (unsigned)P;
      bc = ubc; b = ubc / (unsigned)C; c = ubc - (unsigned)b * (unsigned)C;
    } else {
      p = (int)(i % P);
      bc = i / P;
      b = bc / C; c = bc - b * C;
    }
    const float v = dout[c * ((long long)B * P) + b * P + p];
    float* dst = &dfeat[bc * S + ids[(b / bpg) * P + p]];
    float old;
    if (DISTINCT) { old = *dst; *dst = old + v; }
    else old = atomicAdd(dst, v);
    if (amax) {
      float nv = fabsf(old + v);
      if (!(nv == nv)) nv = __uint_as_float(0x7f800000u);
      unsigned* sl = amax + (bc & (DF_PROBE_SLOTS - 1));
      if (__float_as_uint(nv) > *reinterpret_cast<volatile unsigned*>(sl)) atomicMax(sl, __float_as_uint(nv));
      if (pmax && __float_as_uint(nv) > *reinterpret_cast<volatile unsigned*>(pmax + bc)) atomicMax(pmax + bc, __float_as_uint(nv));
    }
  }
}

// y = x / (sqrt(sum_c x^2) + eps)     (models/networks.py:499-502)
// x is channel-major [C][P] with only a few thousand rows P: a workgroup takes 16 rows x 16 channel
// groups (thread = row r, channels cg, cg+16, ...), so 256 rows already fill 16 CUs and the per-thread
// dependent-load chain is C/16 long instead of C (the one-thread-per-row form was pure latency: 94 us).
__global__ __launch_bounds__(256) void l2norm_fwd_k(const float* __restrict__ x, float* __restrict__ y,
                                                    float* __restrict__ nrm, int C, long long P, float eps) {
  __shared__ float red[16][17];
  const int r = threadIdx.x & 15, cg = threadIdx.x >> 4;
  const long long i = (long long)blockIdx.x * 16 + r;
  const bool ok = i < P;
  float s = 0.f;
  if (ok)
    for (int c = cg; c < C; c += 16) {
      const float v = x[(long long)c * P + i];
      s += v * v;
    }
  red[cg][r] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) tot += red[g][r];
  if (!ok) return;
  const float n = sqrtf(tot);
  if (cg == 0) nrm[i] = n;
  const float inv = 1.f / (n + eps);
  for (int c = cg; c < C; c += 16) y[(long long)c * P + i] = x[(long long)c * P + i] * inv;
}
__global__ __launch_bounds__(256) void l2norm_bwd_k(const float* __restrict__ dy, const float* __restrict__ x,
                                                    const float* __restrict__ nrm, float* __restrict__ dx, int C,
                                                    long long P, float eps) {
  __shared__ float red[16][17];
  const int r = threadIdx.x & 15, cg = threadIdx.x >> 4;
  const long long i = (long long)blockIdx.x * 16 + r;
  const bool ok = i < P;
  float s = 0.f;
  if (ok)
    for (int c = cg; c < C; c += 16) s += dy[(long long)c * P + i] * x[(long long)c * P + i];
  red[cg][r] = s;
  __syncthreads();
  float dot = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) dot += red[g][r];
  if (!ok) return;
  const float n = nrm[i];
  const float inv = 1.f / (n + eps);
  const float k2 = n > 0.f ? dot * inv * inv / n : 0.f;
  for (int c = cg; c < C; c += 16)
    dx[(long long)c * P + i] = dy[(long long)c * P + i] * inv - x[(long long)c * P + i] * k2;
}

// ---------------------------------------------------------------------------------------------
// PatchNCE forward.  q, k are channel-major [C][rows] (rows = B*P, row r = b*P + p); consecutive
// groups of R rows share negatives.  One workgroup = TI query rows of one group; thread t owns
// columns j = t (+256 ...), so every k read is coalesced along rows.
// ---------------------------------------------------------------------------------------------
#define NCE_TI 16
__global__ __launch_bounds__(256) void patchnce_fwd_k(const float* __restrict__ q,
                                                      const float* __restrict__ k,
                                                      float* __restrict__ loss, float* __restrict__ probs,
                                                      long long rows, int C, int R, float invT) {
  extern __shared__ float smem[];
  float* qs = smem;                        // [C][TI]
  float* red = smem + (size_t)C * NCE_TI;  // 17 (+pad)
  float* lpos = red + 32;                  // [TI]
  float* rmax = lpos + NCE_TI;             // [TI]
  float* rsum = rmax + NCE_TI;             // [TI]
  const int tid = threadIdx.x;
  const long long r0 = (long long)blockIdx.x * NCE_TI;  // first query row
  const long long g0 = (r0 / R) * R;                     // first row of the group
  const int i0 = (int)(r0 - g0);
  for (int idx = tid; idx < C * NCE_TI; idx += 256) {
    const int ii = idx % NCE_TI, c = idx / NCE_TI;
    qs[c * NCE_TI + ii] = q[(long long)c * rows + r0 + ii];
  }
  __syncthreads();
  const long long ld = (long long)R + 1;
  float tmax[NCE_TI];
#pragma unroll
  for (int ii = 0; ii < NCE_TI; ++ii) tmax[ii] = -3.0e38f;
  for (int jb = 0; jb < R; jb += 256) {
    const int j = jb + tid;
    if (j < R) {
      const float* kp = k + g0 + j;
      float acc[NCE_TI];
#pragma unroll
      for (int ii = 0; ii < NCE_TI; ++ii) acc[ii] = 0.f;
      for (int c = 0; c < C; ++c) {
        const float kv = kp[(long long)c * rows];
#pragma unroll
        for (int ii = 0; ii < NCE_TI; ++ii) acc[ii] = fmaf(qs[c * NCE_TI + ii], kv, acc[ii]);
      }
#pragma unroll
      for (int ii = 0; ii < NCE_TI; ++ii) {
        float v = acc[ii];
        if (j == i0 + ii) {
          lpos[ii] = v * invT;
          v = -10.0f;
        }
        v *= invT;
        probs[(r0 + ii) * ld + 1 + j] = v;
        tmax[ii] = fmaxf(tmax[ii], v);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int ii = 0; ii < NCE_TI; ++ii) {
    const float m = block_max(tmax[ii], red);
    if (tid == 0) rmax[ii] = fmaxf(m, lpos[ii]);
  }
  __syncthreads();
  float tsum[NCE_TI];
#pragma unroll
  for (int ii = 0; ii < NCE_TI; ++ii) tsum[ii] = 0.f;
  for (int jb = 0; jb < R; jb += 256) {
    const int j = jb + tid;
    if (j < R) {
#pragma unroll
      for (int ii = 0; ii < NCE_TI; ++ii) {
        const float e = expf(probs[(r0 + ii) * ld + 1 + j] - rmax[ii]);
        probs[(r0 + ii) * ld + 1 + j] = e;
        tsum[ii] += e;
      }
    }
  }
#pragma unroll
  for (int ii = 0; ii < NCE_TI; ++ii) {
    const float s = block_sum(tsum[ii], red);
    if (tid == 0) rsum[ii] = s + expf(lpos[ii] - rmax[ii]);
  }
  __syncthreads();
  for (int jb = 0; jb < R; jb += 256) {
    const int j = jb + tid;
    if (j < R) {
#pragma unroll
      for (int ii = 0; ii < NCE_TI; ++ii) probs[(r0 + ii) * ld + 1 + j] *= 1.f / rsum[ii];
    }
  }
  if (tid < NCE_TI) {
    const float e0 = expf(lpos[tid] - rmax[tid]);
    probs[(r0 + tid) * ld] = e0 / rsum[tid];
    loss[r0 + tid] = logf(rsum[tid]) + rmax[tid] - lpos[tid];
  }
}

// dq[c][r] = (g_r/T) * ( sum_{j != i} probs[r][1+j] k[c][j] + (probs[r][0]-1) k[c][r] )
#define NCE_JT 32
__global__ __launch_bounds__(256) void patchnce_bwd_k(const float* __restrict__ dloss,
                                                      const float* __restrict__ probs,
                                                      const float* __restrict__ k, float* __restrict__ dq,
                                                      long long rows, int C, int R, float invT) {
  extern __shared__ float smem[];
  const int Cp = C + 1;
  float* ks = smem;                       // [JT][C+1]
  float* dls = ks + (size_t)NCE_JT * Cp;  // [TI][JT]
  float* os = dls + NCE_TI * NCE_JT;      // [C][TI]
  const int tid = threadIdx.x;
  const long long r0 = (long long)blockIdx.x * NCE_TI;
  const long long g0 = (r0 / R) * R;
  const int i0 = (int)(r0 - g0);
  const long long ld = (long long)R + 1;
  const int ncp = (C + 255) / 256;  // channels per thread (<= 4)
  float acc[4][NCE_TI];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int ii = 0; ii < NCE_TI; ++ii) acc[u][ii] = 0.f;

  for (int jb = 0; jb < R; jb += NCE_JT) {
    __syncthreads();
    for (int idx = tid; idx < NCE_JT * C; idx += 256) {
      const int jj = idx % NCE_JT, c = idx / NCE_JT;
      const int j = jb + jj;
      ks[jj * Cp + c] = (j < R) ? k[(long long)c * rows + g0 + j] : 0.f;
    }
    for (int idx = tid; idx < NCE_TI * NCE_JT; idx += 256) {
      const int jj = idx % NCE_JT, ii = idx / NCE_JT;
      const int j = jb + jj;
      float v = 0.f;
      if (j < R && j != i0 + ii) v = probs[(r0 + ii) * ld + 1 + j] * dloss[r0 + ii] * invT;
      dls[ii * NCE_JT + jj] = v;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = tid + 256 * u;
      if (u < ncp && c < C) {
        for (int jj = 0; jj < NCE_JT; ++jj) {
          const float kv = ks[jj * Cp + c];
#pragma unroll
          for (int ii = 0; ii < NCE_TI; ++ii) acc[u][ii] = fmaf(dls[ii * NCE_JT + jj], kv, acc[u][ii]);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = tid + 256 * u;
    if (u < ncp && c < C) {
#pragma unroll
      for (int ii = 0; ii < NCE_TI; ++ii) os[c * NCE_TI + ii] = acc[u][ii];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < C * NCE_TI; idx += 256) {
    const int ii = idx % NCE_TI, c = idx / NCE_TI;
    const long long r = r0 + ii;
    const long long off = (long long)c * rows + r;
    const float dpos = (probs[r * ld] - 1.f) * dloss[r] * invT;
    dq[off] = os[c * NCE_TI + ii] + dpos * k[off];
  }
}

// ---------------------------------------------------------------------------------------------
// Matrix-core forms for the training configuration (R = 256 rows per group, C = 256 channels).  The scalar
// kernels above are latency chains (256 dependent global loads per thread, 96 workgroup barriers in the
// forward): 122 / 126 us per call.  Here the 16 x 256 logits of a workgroup are one small GEMM on
// v_mfma_f32_16x16x4_f32 staged through LDS, and the softmax runs on 16-lane row groups with shuffles only.
// ---------------------------------------------------------------------------------------------
typedef float nce_f4 __attribute__((ext_vector_type(4)));
#define NCE_KP 272   // LDS row stride (== 16 mod 32: the two k-groups of a half-wave hit disjoint banks)

__global__ __launch_bounds__(256) void patchnce_fwd_mfma_k(const float* __restrict__ q, const float* __restrict__ k,
                                                           float* __restrict__ loss, float* __restrict__ probs,
                                                           long long rows, int C, float invT) {
  constexpr int R = 256;
  __shared__ float qs[256 * NCE_TI];        // [c][16 rows]
  __shared__ float Ks[32 * NCE_KP];         // [c in chunk][256 keys]
  __shared__ float sS[NCE_TI * 260];        // logits [16][256] (+pad)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lk = lane >> 4;
  const long long r0 = (long long)blockIdx.x * NCE_TI;
  const long long g0 = (r0 / R) * R;
  const int i0 = (int)(r0 - g0);
  for (int idx = tid; idx < C * NCE_TI; idx += 256) {
    const int ii = idx % NCE_TI, c = idx / NCE_TI;
    qs[c * NCE_TI + ii] = q[(long long)c * rows + r0 + ii];
  }
  nce_f4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = nce_f4{0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < C; c0 += 32) {
    __syncthreads();                        // qs visible (first pass) / previous chunk consumed
#pragma unroll 8
    for (int c = 0; c < 32; ++c) Ks[c * NCE_KP + tid] = k[(long long)(c0 + c) * rows + g0 + tid];
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float a = qs[(c0 + ks * 4 + lk) * NCE_TI + l15];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float b = Ks[(ks * 4 + lk) * NCE_KP + (wid * 4 + t) * 16 + l15];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) sS[(lk * 4 + r) * 260 + (wid * 4 + t) * 16 + l15] = acc[t][r];
  __syncthreads();
  // row ii = tid >> 4 is handled by 16 consecutive lanes; lane `sub` takes columns sub, sub+16, ...
  const int ii = tid >> 4, sub = tid & 15;
  const long long ld = (long long)R + 1;
  float v[16];
  float lp = 0.f, m = -3.0e38f;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int j = sub + 16 * u;
    float x = sS[ii * 260 + j];
    if (j == i0 + ii) { lp = x * invT; x = -10.0f; }
    x *= invT;
    v[u] = x;
    m = fmaxf(m, x);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); lp += __shfl_xor(lp, o); }
  m = fmaxf(m, lp);                         // lp: exactly one lane of the row held the diagonal, the others 0
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u) { v[u] = expf(v[u] - m); sum += v[u]; }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float e0 = expf(lp - m);
  sum += e0;
  const float inv = 1.f / sum;
  float* pr = probs + (r0 + ii) * ld;
#pragma unroll
  for (int u = 0; u < 16; ++u) pr[1 + sub + 16 * u] = v[u] * inv;
  if (sub == 0) {
    pr[0] = e0 / sum;
    loss[r0 + ii] = logf(sum) + m - lp;
  }
}

// dq[c][r] = (g_r/T) * ( sum_{j != i} probs[r][1+j] k[c][j] + (probs[r][0]-1) k[c][r] ): M = c, N = 16 rows, K = j
__global__ __launch_bounds__(256) void patchnce_bwd_mfma_k(const float* __restrict__ dloss,
                                                           const float* __restrict__ probs,
                                                           const float* __restrict__ k, float* __restrict__ dq,
                                                           long long rows, int C, float invT) {
  constexpr int R = 256, JT = 32;
  __shared__ float ks[JT * NCE_KP];         // [j in chunk][c]
  __shared__ float dls[JT * NCE_TI];        // [j in chunk][16 rows]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, lk = lane >> 4;
  const long long r0 = (long long)blockIdx.x * NCE_TI;
  const long long g0 = (r0 / R) * R;
  const int i0 = (int)(r0 - g0);
  const long long ld = (long long)R + 1;
  const int ntile = C / 16;                 // 16-channel tiles; wave w owns tiles 4w .. 4w+3
  nce_f4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = nce_f4{0.f, 0.f, 0.f, 0.f};
  for (int jb = 0; jb < R; jb += JT) {
    __syncthreads();
    for (int idx = tid; idx < JT * C; idx += 256) {
      const int jj = idx % JT, c = idx / JT;
      ks[jj * NCE_KP + c] = k[(long long)c * rows + g0 + jb + jj];
    }
    for (int idx = tid; idx < NCE_TI * JT; idx += 256) {
      const int jj = idx % JT, ii = idx / JT;
      const int j = jb + jj;
      dls[jj * NCE_TI + ii] = (j != i0 + ii) ? probs[(r0 + ii) * ld + 1 + j] * dloss[r0 + ii] * invT : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < JT / 4; ++s) {
      const float b = dls[(s * 4 + lk) * NCE_TI + l15];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ct = wid * 4 + t;
        const float a = ct < ntile ? ks[(s * 4 + lk) * NCE_KP + ct * 16 + l15] : 0.f;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  const long long r = r0 + l15;             // D layout: column = row of the workgroup, rows = channels
  const float dpos = (probs[r * ld] - 1.f) * dloss[r] * invT;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ct = wid * 4 + t;
    if (ct >= ntile) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long off = (long long)(ct * 16 + lk * 4 + e) * rows + r;
      dq[off] = acc[t][e] + dpos * k[off];
    }
  }
}

// ---------------------------------------------------------------------------------------------
extern "C" int dfmir_patch_gather_fwd(const float* feat, const long long* ids, float* out, int B, int C,
                                      long long S, int P, void* stream) {
  DF_ARG_CHECK(feat && ids && out && B > 0 && C > 0 && S > 0 && P > 0);
  patch_gather_fwd_k<<<df_grid((long long)B * C * P, 256, 4096), 256, 0, (hipStream_t)stream>>>(
      feat, ids, out, B, C, S, P, B);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_patch_gather_fwd_g(const float* feat, const long long* ids, float* out, int B, int C,
                                        long long S, int P, int G, void* stream) {
  DF_ARG_CHECK(feat && ids && out && B > 0 && C > 0 && S > 0 && P > 0 && G > 0 && B % G == 0);
  patch_gather_fwd_k<<<df_grid((long long)B * C * P, 256, 4096), 256, 0, (hipStream_t)stream>>>(
      feat, ids, out, B, C, S, P, B / G);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_patch_gather_bwd_g(const float* dout, const long long* ids, float* dfeat, int B, int C,
                                        long long S, int P, int G, float* dfeat_amax, void* stream) {
  DF_ARG_CHECK(dout && ids && dfeat && B > 0 && C > 0 && S > 0 && P > 0 && G > 0 && B % G == 0);
  patch_gather_bwd_k<true><<<df_grid((long long)B * C * P, 256, 4096), 256, 0, (hipStream_t)stream>>>(
      dout, ids, dfeat, B, C, S, P, B / G, reinterpret_cast<unsigned*>(dfeat_amax));
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_patch_gather_bwd_gp(const float* dout, const long long* ids, float* dfeat, int B, int C,
                                         long long S, int P, int G, float* dfeat_amax, float* dfeat_pmax, void* stream) {
  DF_ARG_CHECK(dout && ids && dfeat && dfeat_amax && dfeat_pmax && B > 0 && C > 0 && S > 0 && P > 0 && G > 0 && B % G == 0);
  patch_gather_bwd_k<true><<<df_grid((long long)B * C * P, 256, 4096), 256, 0, (hipStream_t)stream>>>(
      dout, ids, dfeat, B, C, S, P, B / G, reinterpret_cast<unsigned*>(dfeat_amax), reinterpret_cast<unsigned*>(dfeat_pmax));
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_patch_gather_bwd_any(const float* dout, const long long* ids, float* dfeat, int B, int C,
                                          long long S, int P, int G, float* dfeat_amax, float* dfeat_pmax, void* stream) {
  DF_ARG_CHECK(dout && ids && dfeat && B > 0 && C > 0 && S > 0 && P > 0 && G > 0 && B % G == 0);
  DF_ARG_CHECK(dfeat_amax || !dfeat_pmax);
  patch_gather_bwd_k<false><<<df_grid((long long)B * C * P, 256, 4096), 256, 0, (hipStream_t)stream>>>(
      dout, ids, dfeat, B, C, S, P, B / G, reinterpret_cast<unsigned*>(dfeat_amax), reinterpret_cast<unsigned*>(dfeat_pmax));
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_patch_gather_bwd(const float* dout, const long long* ids, float* dfeat, int B, int C,
                                      long long S, int P, void* stream) {
  DF_ARG_CHECK(dout && ids && dfeat && B > 0 && C > 0 && S > 0 && P > 0);
  patch_gather_bwd_k<false><<<df_grid((long long)B * C * P, 256, 4096), 256, 0, (hipStream_t)stream>>>(
      dout, ids, dfeat, B, C, S, P, B, nullptr);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_patch_gather_bwd_amax(const float* dout, const long long* ids, float* dfeat, int B, int C,
                                           long long S, int P, float* dfeat_amax, void* stream) {
  DF_ARG_CHECK(dout && ids && dfeat && dfeat_amax && B > 0 && C > 0 && S > 0 && P > 0);
  patch_gather_bwd_k<true><<<df_grid((long long)B * C * P, 256, 4096), 256, 0, (hipStream_t)stream>>>(
      dout, ids, dfeat, B, C, S, P, B, reinterpret_cast<unsigned*>(dfeat_amax));
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_l2norm_fwd(const float* x, float* y, float* norm, int C, long long rows, float eps,
                                void* stream) {
  DF_ARG_CHECK(x && y && norm && C > 0 && rows > 0);
  l2norm_fwd_k<<<(unsigned)((rows + 15) / 16), 256, 0, (hipStream_t)stream>>>(x, y, norm, C, rows, eps);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_l2norm_bwd(const float* dy, const float* x, const float* norm, float* dx, int C,
                                long long rows, float eps, void* stream) {
  DF_ARG_CHECK(dy && x && norm && dx && C > 0 && rows > 0);
  l2norm_bwd_k<<<(unsigned)((rows + 15) / 16), 256, 0, (hipStream_t)stream>>>(dy, x, norm, dx, C, rows, eps);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_patchnce_fwd(const float* q, const float* k, float* loss, float* probs, long long rows,
                                  int C, int G, float T, void* stream) {
  DF_ARG_CHECK(q && k && loss && probs && rows > 0 && C > 0 && G > 0 && T > 0.f);
  DF_ARG_CHECK(rows % G == 0);
  const int R = (int)(rows / G);
  DF_ARG_CHECK(R % NCE_TI == 0);
  if (R == 256 && C <= 256 && (C & 31) == 0) {
    patchnce_fwd_mfma_k<<<(unsigned)(rows / NCE_TI), 256, 0, (hipStream_t)stream>>>(q, k, loss, probs, rows, C, 1.f / T);
    DF_LAUNCH_CHECK();
    return 0;
  }
  const size_t sh = ((size_t)C * NCE_TI + 32 + 3 * NCE_TI) * sizeof(float);
  DF_ARG_CHECK(sh <= 64 * 1024);
  patchnce_fwd_k<<<(unsigned)(rows / NCE_TI), 256, sh, (hipStream_t)stream>>>(q, k, loss, probs, rows, C, R,
                                                                         1.f / T);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_patchnce_bwd(const float* dloss, const float* probs, const float* k, float* dq,
                                  long long rows, int C, int G, float T, void* stream) {
  DF_ARG_CHECK(dloss && probs && k && dq && rows > 0 && C > 0 && G > 0 && T > 0.f);
  DF_ARG_CHECK(rows % G == 0);
  const int R = (int)(rows / G);
  DF_ARG_CHECK(R % NCE_TI == 0 && C <= 1024);
  if (R == 256 && C <= 256 && (C & 15) == 0) {
    patchnce_bwd_mfma_k<<<(unsigned)(rows / NCE_TI), 256, 0, (hipStream_t)stream>>>(dloss, probs, k, dq, rows, C, 1.f / T);
    DF_LAUNCH_CHECK();
    return 0;
  }
  const size_t sh = ((size_t)NCE_JT * (C + 1) + NCE_TI * NCE_JT + (size_t)C * NCE_TI) * sizeof(float);
  DF_ARG_CHECK(sh <= 64 * 1024);
  patchnce_bwd_k<<<(unsigned)(rows / NCE_TI), 256, sh, (hipStream_t)stream>>>(dloss, probs, k, dq, rows, C, R,
                                                                         1.f / T);
  DF_LAUNCH_CHECK();
  return 0;
}

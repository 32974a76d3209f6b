// PatchNCE path: patch gather (NCHW -> channel-major [B,C,P] rows), L2 normalisation over
// channels, and the fused InfoNCE loss (logits GEMM + diagonal fill + softmax cross-entropy in one
// kernel, probabilities kept for backward).
#include "common.h"

__global__ void patch_gather_fwd_k(const float* __restrict__ feat, const long long* __restrict__ ids,
                                   float* __restrict__ out, int B, int C, long long S, int P) {
  const long long total = (long long)B * C * P;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    const long long bc = i / P;
    const long long b = bc / C, c = bc - b * C;
    out[c * ((long long)B * P) + b * P + p] = feat[bc * S + ids[p]];
  }
}
__global__ void patch_gather_bwd_k(const float* __restrict__ dout, const long long* __restrict__ ids,
                                   float* __restrict__ dfeat, int B, int C, long long S, int P) {
  const long long total = (long long)B * C * P;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    const long long bc = i / P;
    const long long b = bc / C, c = bc - b * C;
    atomicAdd(&dfeat[bc * S + ids[p]], dout[c * ((long long)B * P) + b * P + p]);
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
extern "C" int dfmir_patch_gather_fwd(const float* feat, const long long* ids, float* out, int B, int C,
                                      long long S, int P, void* stream) {
  DF_ARG_CHECK(feat && ids && out && B > 0 && C > 0 && S > 0 && P > 0);
  patch_gather_fwd_k<<<df_grid((long long)B * C * P, 256, 4096), 256, 0, (hipStream_t)stream>>>(
      feat, ids, out, B, C, S, P);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_patch_gather_bwd(const float* dout, const long long* ids, float* dfeat, int B, int C,
                                      long long S, int P, void* stream) {
  DF_ARG_CHECK(dout && ids && dfeat && B > 0 && C > 0 && S > 0 && P > 0);
  patch_gather_bwd_k<<<df_grid((long long)B * C * P, 256, 4096), 256, 0, (hipStream_t)stream>>>(
      dout, ids, dfeat, B, C, S, P);
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
  const size_t sh = ((size_t)NCE_JT * (C + 1) + NCE_TI * NCE_JT + (size_t)C * NCE_TI) * sizeof(float);
  DF_ARG_CHECK(sh <= 64 * 1024);
  patchnce_bwd_k<<<(unsigned)(rows / NCE_TI), 256, sh, (hipStream_t)stream>>>(dloss, probs, k, dq, rows, C, R,
                                                                         1.f / T);
  DF_LAUNCH_CHECK();
  return 0;
}

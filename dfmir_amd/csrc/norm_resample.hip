// HBM-bound per-plane kernels of the translation generator: InstanceNorm(+ReLU,+residual),
// activation backward, anti-aliased 2x down/up-sampling, reflection pad, nearest-up + concat.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// InstanceNorm: one (n,c) plane per workgroup; two-pass (mean, then centred variance) so the
// result tracks torch's numerics instead of the cancellation-prone E[x^2]-E[x]^2 form.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void instnorm_fwd_k(const float* __restrict__ x,
                                                      const float* __restrict__ res,
                                                      float* __restrict__ y, float* __restrict__ mean_o,
                                                      float* __restrict__ rstd_o, long long S,
                                                      float eps, int relu) {
  __shared__ float sm[17];
  const long long base = (long long)blockIdx.x * S;
  const float* xp = x + base;
  const bool vec = (S & 3) == 0;
  float s = 0.f;
  if (vec) {
    const float4* x4 = reinterpret_cast<const float4*>(xp);
    for (long long i = threadIdx.x; i < (S >> 2); i += 256) {
      const float4 v = x4[i];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (long long i = threadIdx.x; i < S; i += 256) s += xp[i];
  }
  const float mean = block_sum(s, sm) / (float)S;
  float q = 0.f;
  if (vec) {
    const float4* x4 = reinterpret_cast<const float4*>(xp);
    for (long long i = threadIdx.x; i < (S >> 2); i += 256) {
      const float4 v = x4[i];
      const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  } else {
    for (long long i = threadIdx.x; i < S; i += 256) {
      const float a = xp[i] - mean;
      q += a * a;
    }
  }
  const float var = block_sum(q, sm) / (float)S;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_o[blockIdx.x] = mean;
    rstd_o[blockIdx.x] = rstd;
  }
  float* yp = y + base;
  const float* rp = res ? res + base : nullptr;
  if (vec) {
    const float4* x4 = reinterpret_cast<const float4*>(xp);
    const float4* r4 = reinterpret_cast<const float4*>(rp);
    float4* y4 = reinterpret_cast<float4*>(yp);
    for (long long i = threadIdx.x; i < (S >> 2); i += 256) {
      float4 v = x4[i];
      v.x = (v.x - mean) * rstd; v.y = (v.y - mean) * rstd;
      v.z = (v.z - mean) * rstd; v.w = (v.w - mean) * rstd;
      if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (rp) {
        const float4 r = r4[i];
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      y4[i] = v;
    }
  } else {
    for (long long i = threadIdx.x; i < S; i += 256) {
      float v = (xp[i] - mean) * rstd;
      if (relu) v = fmaxf(v, 0.f);
      if (rp) v += rp[i];
      yp[i] = v;
    }
  }
}

// Register-resident variants for the generator's plane sizes (64^2, 128^2, 256^2): the plane is read
// from HBM exactly once (float4 per lane, E float4 per thread), both reductions run on registers.
template <int NT, int E>
__global__ __launch_bounds__(NT) void instnorm_fwd_reg_k(const float* __restrict__ x,
                                                         const float* __restrict__ res,
                                                         float* __restrict__ y, float* __restrict__ mean_o,
                                                         float* __restrict__ rstd_o, float eps, int relu,
                                                         float* __restrict__ amax) {
  __shared__ float sm[17];
  __shared__ unsigned smax;
  if (threadIdx.x == 0) smax = 0u;
  constexpr long long S = (long long)NT * 4 * E;
  const long long base = (long long)blockIdx.x * S;
  const float4* x4 = reinterpret_cast<const float4*>(x + base);
  float4 v[E];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    v[i] = x4[threadIdx.x + NT * i];
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = block_sum(s, sm) / (float)S;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  const float var = block_sum(q, sm) / (float)S;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_o[blockIdx.x] = mean;
    rstd_o[blockIdx.x] = rstd;
  }
  const float4* r4 = res ? reinterpret_cast<const float4*>(res + base) : nullptr;
  float4* y4 = reinterpret_cast<float4*>(y + base);
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    float4 o;
    o.x = (v[i].x - mean) * rstd; o.y = (v[i].y - mean) * rstd;
    o.z = (v[i].z - mean) * rstd; o.w = (v[i].w - mean) * rstd;
    if (relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    if (r4) {
      const float4 r = r4[threadIdx.x + NT * i];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    if (y) y4[threadIdx.x + NT * i] = o;                    // y == NULL: statistics + range probe only (dfmir_instnorm_stats)
    am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  if (amax) publish_block_absmax_acc(am, &smax, amax);   // range probe for the next conv's fp16x2 split
}
template <int NT, int E, bool COLS>
__global__ __launch_bounds__(NT) void instnorm_bwd_reg_k(const float* __restrict__ dy,
                                                         const float* __restrict__ x,
                                                         const float* __restrict__ mean_i,
                                                         const float* __restrict__ rstd_i,
                                                         float* __restrict__ dx, int relu,
                                                         float* __restrict__ amax, float* __restrict__ cols, int W,
                                                         float* __restrict__ pmax) {
  // pmax (optional, needs amax): pmax[plane] <- max |dx| of this plane (per-channel scales of the split wgrad)
  // cols (optional, W % 4 == 0): cols[plane][2][H] <- the first and last column of dx, for the ring kernel of a
  // reflect-padded conv's dgrad (a column of an NCHW tensor is one cache line per element to read back)
  __shared__ float sm[17];
  __shared__ unsigned smax;
  if (threadIdx.x == 0) smax = 0u;
  constexpr long long S = (long long)NT * 4 * E;
  // (<1024,16>: 2 x 64 values per thread at the 128-register limit of a 1024-thread workgroup: 29 registers spill to
  // scratch; parking part of g in LDS did not change that -- the pressure is in the load phase -- and the border-column
  // code, 13 more, is compiled out for the plane sizes that never need it: COLS)
  const long long base = (long long)blockIdx.x * S;
  const float mean = mean_i[blockIdx.x], rstd = rstd_i[blockIdx.x];
  const float4* x4 = reinterpret_cast<const float4*>(x + base);
  const float4* g4 = reinterpret_cast<const float4*>(dy + base);
  float4 xh[E], g[E];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const float4 xv = x4[threadIdx.x + NT * i];
    float4 gi = g4[threadIdx.x + NT * i];
    xh[i].x = (xv.x - mean) * rstd; xh[i].y = (xv.y - mean) * rstd;
    xh[i].z = (xv.z - mean) * rstd; xh[i].w = (xv.w - mean) * rstd;
    if (relu) {
      if (!(xh[i].x > 0.f)) gi.x = 0.f;
      if (!(xh[i].y > 0.f)) gi.y = 0.f;
      if (!(xh[i].z > 0.f)) gi.z = 0.f;
      if (!(xh[i].w > 0.f)) gi.w = 0.f;
    }
    s1 += (gi.x + gi.y) + (gi.z + gi.w);
    s2 += (gi.x * xh[i].x + gi.y * xh[i].y) + (gi.z * xh[i].z + gi.w * xh[i].w);
    g[i] = gi;
  }
  const float m1 = block_sum(s1, sm) / (float)S;
  const float m2 = block_sum(s2, sm) / (float)S;
  float4* d4 = reinterpret_cast<float4*>(dx + base);
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const float4 gi = g[i];
    float4 o;
    o.x = rstd * (gi.x - m1 - xh[i].x * m2); o.y = rstd * (gi.y - m1 - xh[i].y * m2);
    o.z = rstd * (gi.z - m1 - xh[i].z * m2); o.w = rstd * (gi.w - m1 - xh[i].w * m2);
    d4[threadIdx.x + NT * i] = o;
    am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    if (COLS && cols) {
      const int e0 = 4 * (threadIdx.x + NT * i), H = (int)(S / W);
      float* cp = cols + (long long)blockIdx.x * 2 * H;
      if (e0 % W == 0) cp[e0 / W] = o.x;
      if ((e0 + 3) % W == W - 1) cp[H + (e0 + 3) / W] = o.w;
    }
  }
  if (amax) publish_block_absmax_acc(am, &smax, amax);
  if (amax && pmax && threadIdx.x == 0) pmax[blockIdx.x] = __uint_as_float(smax);
}

// dx = rstd * (g - mean(g) - xhat*mean(g*xhat)),  g = dy * [xhat>0 if relu]
__global__ __launch_bounds__(256) void instnorm_bwd_k(const float* __restrict__ dy,
                                                      const float* __restrict__ x,
                                                      const float* __restrict__ mean_i,
                                                      const float* __restrict__ rstd_i,
                                                      float* __restrict__ dx, long long S, int relu) {
  __shared__ float sm[17];
  const long long base = (long long)blockIdx.x * S;
  const float mean = mean_i[blockIdx.x], rstd = rstd_i[blockIdx.x];
  const float* xp = x + base;
  const float* gp = dy + base;
  float* dp = dx + base;
  float s1 = 0.f, s2 = 0.f;
  if ((S & 3) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(xp);
    const float4* g4 = reinterpret_cast<const float4*>(gp);
    float4* d4 = reinterpret_cast<float4*>(dp);
    const long long S4 = S >> 2;
    for (long long i = threadIdx.x; i < S4; i += 256) {
      const float4 xv = x4[i];
      float4 g = g4[i];
      const float a = (xv.x - mean) * rstd, b = (xv.y - mean) * rstd, c = (xv.z - mean) * rstd, d = (xv.w - mean) * rstd;
      if (relu) {
        if (!(a > 0.f)) g.x = 0.f;
        if (!(b > 0.f)) g.y = 0.f;
        if (!(c > 0.f)) g.z = 0.f;
        if (!(d > 0.f)) g.w = 0.f;
      }
      s1 += (g.x + g.y) + (g.z + g.w);
      s2 += (g.x * a + g.y * b) + (g.z * c + g.w * d);
    }
    const float m1 = block_sum(s1, sm) / (float)S;
    const float m2 = block_sum(s2, sm) / (float)S;
    for (long long i = threadIdx.x; i < S4; i += 256) {
      const float4 xv = x4[i];
      float4 g = g4[i];
      const float a = (xv.x - mean) * rstd, b = (xv.y - mean) * rstd, c = (xv.z - mean) * rstd, d = (xv.w - mean) * rstd;
      if (relu) {
        if (!(a > 0.f)) g.x = 0.f;
        if (!(b > 0.f)) g.y = 0.f;
        if (!(c > 0.f)) g.z = 0.f;
        if (!(d > 0.f)) g.w = 0.f;
      }
      float4 o;
      o.x = rstd * (g.x - m1 - a * m2); o.y = rstd * (g.y - m1 - b * m2);
      o.z = rstd * (g.z - m1 - c * m2); o.w = rstd * (g.w - m1 - d * m2);
      d4[i] = o;
    }
    return;
  }
  for (long long i = threadIdx.x; i < S; i += 256) {
    const float xh = (xp[i] - mean) * rstd;
    float g = gp[i];
    if (relu && !(xh > 0.f)) g = 0.f;
    s1 += g;
    s2 += g * xh;
  }
  const float m1 = block_sum(s1, sm) / (float)S;
  const float m2 = block_sum(s2, sm) / (float)S;
  for (long long i = threadIdx.x; i < S; i += 256) {
    const float xh = (xp[i] - mean) * rstd;
    float g = gp[i];
    if (relu && !(xh > 0.f)) g = 0.f;
    dp[i] = rstd * (g - m1 - xh * m2);
  }
}

__global__ void act_bwd_k(const float* __restrict__ dy, const float* __restrict__ y,
                          float* __restrict__ dx, long long n, int act, float slope) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float yv = y[i], g = dy[i];
    dx[i] = (act == 2) ? g * (1.f - yv * yv) : (yv > 0.f ? g : g * slope);
  }
}
// 16-byte accesses (n % 4 == 0, aligned pointers)
__global__ __launch_bounds__(256) void act_bwd_v4_k(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                    float4* __restrict__ dx, long long n4, int act, float slope) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 yv = y[i], g = dy[i];
    float4 o;
    o.x = (act == 2) ? g.x * (1.f - yv.x * yv.x) : (yv.x > 0.f ? g.x : g.x * slope);
    o.y = (act == 2) ? g.y * (1.f - yv.y * yv.y) : (yv.y > 0.f ? g.y : g.y * slope);
    o.z = (act == 2) ? g.z * (1.f - yv.z * yv.z) : (yv.z > 0.f ? g.z : g.z * slope);
    o.w = (act == 2) ? g.w * (1.f - yv.w * yv.w) : (yv.w > 0.f ? g.w : g.w * slope);
    dx[i] = o;
  }
}

// the same, 4 elements per thread, leaving the range probe of dx (DF_PROBE_SLOTS accumulating slots) for the split
// convolutions that consume it (dgrad / wgrad of the layer in front of the activation)
__global__ __launch_bounds__(256) void act_bwd_amax_k(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dx, long long n, int act, float slope,
                                                      float* __restrict__ amax) {
  __shared__ unsigned smax;
  if (threadIdx.x == 0) smax = 0u;
  __syncthreads();
  const long long n4 = n >> 2;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 yv = reinterpret_cast<const float4*>(y)[i], g = reinterpret_cast<const float4*>(dy)[i];
    float4 o;
    o.x = (act == 2) ? g.x * (1.f - yv.x * yv.x) : (yv.x > 0.f ? g.x : g.x * slope);
    o.y = (act == 2) ? g.y * (1.f - yv.y * yv.y) : (yv.y > 0.f ? g.y : g.y * slope);
    o.z = (act == 2) ? g.z * (1.f - yv.z * yv.z) : (yv.z > 0.f ? g.z : g.z * slope);
    o.w = (act == 2) ? g.w * (1.f - yv.w * yv.w) : (yv.w > 0.f ? g.w : g.w * slope);
    reinterpret_cast<float4*>(dx)[i] = o;
    m = fmaxf(fmaxf(m, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float yv = y[i], g = dy[i];
    const float o = (act == 2) ? g * (1.f - yv * yv) : (yv > 0.f ? g : g * slope);
    dx[i] = o;
    m = fmaxf(m, fabsf(o));
  }
  publish_block_absmax_acc(m, &smax, amax);
}

// ---------------------------------------------------------------------------------------------
// Downsample: y[oy][ox] = sum_{a,b} f[a]f[b] x[refl(2oy+a-1)][refl(2ox+b-1)], f = [1,2,1]/4
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int refl1(int c, int n) {
  if (c < 0) c = -c;
  if (c >= n) c = 2 * (n - 1) - c;
  return c < 0 ? 0 : c;
}
__global__ void blur_down_fwd_k(const float* __restrict__ x, float* __restrict__ y, int planes, int H,
                                int W, int Ho, int Wo) {
  const long long total = (long long)planes * Ho * Wo;
  const float f[3] = {0.25f, 0.5f, 0.25f};
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    long long r = i / Wo;
    const int oy = (int)(r % Ho);
    const long long pl = r / Ho;
    const float* xp = x + pl * H * W;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int iy = refl1(2 * oy + a - 1, H);
      float rs = 0.f;
#pragma unroll
      for (int b = 0; b < 3; ++b) rs += f[b] * xp[(long long)iy * W + refl1(2 * ox + b - 1, W)];
      s += f[a] * rs;
    }
    y[i] = s;
  }
}
// Adjoint of the 1-D operator above for input index i: at most 3 (output index, weight) pairs, held
// in registers (weight 0 = unused slot, index clamped in range).
//   2o + a - 1 = i  =>  i even: (i/2, 1/2);  i odd: ((i+1)/2, 1/4) and ((i-1)/2, 1/4)
//   reflected halo: padded -1 -> i == 1 adds (0, 1/4);  padded n -> i == n-2 adds ((n-1)/2, 1/4) for odd n
__device__ __forceinline__ void blur_down_adj(int i, int n, int no, int (&o)[3], float (&w)[3]) {
  if ((i & 1) == 0) {
    o[0] = i >> 1; w[0] = 0.5f;
    o[1] = 0; w[1] = 0.f;
  } else {
    o[0] = (i - 1) >> 1; w[0] = 0.25f;
    const int o1 = (i + 1) >> 1;
    o[1] = o1 < no ? o1 : 0; w[1] = o1 < no ? 0.25f : 0.f;
  }
  o[2] = 0; w[2] = 0.f;
  if (i == 1) { o[2] = 0; w[2] = 0.25f; }
  if (i == n - 2 && (n & 1)) {
    if (i == 1) {  // n == 3: both halo extras land on the middle sample -> (0, 1/2), (1, 1/2)
      o[0] = 0; w[0] = 0.5f; o[1] = 1; w[1] = 0.5f; o[2] = 0; w[2] = 0.f;
    } else {
      o[2] = (n - 1) >> 1; w[2] = 0.25f;
    }
  }
}
__global__ void blur_down_bwd_k(const float* __restrict__ dy, float* __restrict__ dx, int planes, int H,
                                int W, int Ho, int Wo) {
  const long long total = (long long)planes * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W);
    long long r = i / W;
    const int iy = (int)(r % H);
    const long long pl = r / H;
    const float* gp = dy + pl * Ho * Wo;
    int oy[3], oxx[3];
    float wy[3], wx[3];
    blur_down_adj(iy, H, Ho, oy, wy);
    blur_down_adj(ix, W, Wo, oxx, wx);
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float* row = gp + (long long)oy[a] * Wo;
      s += wy[a] * (wx[0] * row[oxx[0]] + wx[1] * row[oxx[1]] + wx[2] * row[oxx[2]]);
    }
    dx[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// InstanceNorm + ReLU + Downsample in one pass over the plane (models/networks.py:984-996: the full-resolution
// IN/ReLU output feeds only the blur-pool and is needed by nobody's backward -- InstanceNorm's adjoint wants x, mean,
// rstd; the blur's wants only its output gradient).  Forward: the plane is read once into registers (as
// instnorm_fwd_reg_k), normalised, and pushed band by band (RB = NT*4/W rows) through a 2*RB-row LDS ring from which
// every thread takes one output of the 3x3 [1 2 1]^2/16 stride-2 stencil (reflect at row / column -1): 1.25 plane
// transfers instead of 3.25.  Backward: d(IN output) is the blur's adjoint of dz, gathered from LDS-staged dz rows
// (<= 2x2 taps per pixel, blur_down_adj), once for the two InstanceNorm sums and once more for the result, so that
// only x stays in registers (the 1024-thread form has 128): 2.25-2.5 transfers instead of 4.25.
// ---------------------------------------------------------------------------------------------
template <int NT, int E>
__global__ __launch_bounds__(NT) void in_relu_blurdown_fwd_k(const float* __restrict__ x, float* __restrict__ z,
                                                             float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                             int W, float eps, float* __restrict__ amax) {
  constexpr long long S = (long long)NT * 4 * E;
  extern __shared__ float ring[];                        // 2*RB rows x W
  __shared__ float sm[17];
  __shared__ unsigned smax;
  if (threadIdx.x == 0) smax = 0u;
  const int H = (int)(S / W), Wo = W >> 1, Ho = H >> 1, RB = NT * 4 / W;
  const long long base = (long long)blockIdx.x * S;
  const float4* x4 = reinterpret_cast<const float4*>(x + base);
  float4 v[E];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    v[i] = x4[threadIdx.x + NT * i];
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = block_sum(s, sm) / (float)S;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  const float var = block_sum(q, sm) / (float)S;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_o[blockIdx.x] = mean;
    rstd_o[blockIdx.x] = rstd;
  }
  float* zp = z + (long long)blockIdx.x * Ho * Wo;
  const int lrow = (threadIdx.x * 4) / W, lcol = (threadIdx.x * 4) % W;   // this thread's place inside a band
  const int orow = threadIdx.x / Wo, ocol = threadIdx.x % Wo;            // ... and its output inside the band's outputs
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < E; ++i) {
    float4 o;
    o.x = fmaxf((v[i].x - mean) * rstd, 0.f); o.y = fmaxf((v[i].y - mean) * rstd, 0.f);
    o.z = fmaxf((v[i].z - mean) * rstd, 0.f); o.w = fmaxf((v[i].w - mean) * rstd, 0.f);
    am = fmaxf(fmaxf(am, fmaxf(o.x, o.y)), fmaxf(o.z, o.w));
    __syncthreads();                                     // readers of the half about to be overwritten are done
    *reinterpret_cast<float4*>(&ring[(((i & 1) * RB) + lrow) * W + lcol]) = o;
    __syncthreads();
    // output row oy = i*RB/2 + orow reads input rows 2oy-1 .. 2oy+1 (row -1 reflects to row 1)
    const int oy = i * (RB >> 1) + orow;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int gr = 2 * oy + a - 1;
      gr = gr < 0 ? -gr : gr;
      const float* row = ring + (gr % (2 * RB)) * W;
      const int c1 = 2 * ocol, c0 = c1 == 0 ? 1 : c1 - 1, c2 = c1 + 1;
      const float r = 0.25f * row[c0] + 0.5f * row[c1] + 0.25f * row[c2];
      acc += (a == 1 ? 0.5f : 0.25f) * r;
    }
    zp[oy * Wo + ocol] = acc;
  }
  if (amax) publish_block_absmax_acc(am, &smax, amax);     // bounds the blur's output too (convex combination)
}

template <int NT, int E>
__global__ __launch_bounds__(NT) void in_relu_blurdown_bwd_k(const float* __restrict__ dz, const float* __restrict__ x,
                                                             const float* __restrict__ mean_i,
                                                             const float* __restrict__ rstd_i, float* __restrict__ dx,
                                                             int W, float* __restrict__ amax, float* __restrict__ pmax) {
  // x is NOT kept in registers here (2 x E unrolled bands with the blur's tap geometry on top of 4*E registers of x
  // spilled in every form tried): both passes stream it band by band, one band ahead; the second read of the 64-256 KB
  // plane comes back from L2 / the Infinity Cache.
  constexpr long long S = (long long)NT * 4 * E;
  extern __shared__ float zr[];                          // (RB/2 + 1) rows of dz x Wo
  __shared__ float sm[17];
  __shared__ unsigned smax;
  if (threadIdx.x == 0) smax = 0u;
  const int H = (int)(S / W), Wo = W >> 1, Ho = H >> 1, RB = NT * 4 / W, ZR = (RB >> 1) + 1;
  const long long base = (long long)blockIdx.x * S;
  const float mean = mean_i[blockIdx.x], rstd = rstd_i[blockIdx.x];
  const float4* x4 = reinterpret_cast<const float4*>(x + base);
  const float* gz = dz + (long long)blockIdx.x * Ho * Wo;
  const int lrow = (threadIdx.x * 4) / W, ix = (threadIdx.x * 4) % W;
  const int c = ix >> 1;                                 // dz columns c, c+1, c+2 serve inputs ix..ix+3
  const bool c2 = c + 2 < Wo;
  float m1 = 0.f, m2 = 0.f, s1 = 0.f, s2 = 0.f, am = 0.f;
  float4* d4 = reinterpret_cast<float4*>(dx + base);
  // dz rows of a band go global -> registers one band AHEAD and registers -> LDS between the barriers, x is read two
  // bands ahead: with the loads issued between the barriers every band paid a full memory latency (A/B in one trace:
  // 1.80 -> 1.68 ms at 256^2, 0.83 -> 0.69 ms at 128^2; reading dz straight from global memory without LDS and
  // barriers measured no better: 1.66 / 0.84 ms)
  constexpr int NZ = 2;                                  // dz values per thread and band: ZR * Wo <= 2 NT
  for (int pass = 0; pass < 2; ++pass) {
    float4 xa = x4[threadIdx.x], xb = x4[threadIdx.x + (E > 1 ? NT : 0)];
    float zv[NZ];
#define IRB_LOADZ(i_)                                                                             \
    _Pragma("unroll") for (int q = 0; q < NZ; ++q) {                                              \
      const int u = threadIdx.x + q * NT;                                                         \
      const int zrow = (i_) * (RB >> 1) + u / Wo;                                                 \
      zv[q] = (u < ZR * Wo && zrow < Ho) ? gz[zrow * Wo + (u % Wo)] : 0.f;                        \
    }
    IRB_LOADZ(0)
#pragma unroll 1
    for (int i = 0; i < E; ++i) {
      const float4 xv = xa;
      xa = xb;
      if (i + 2 < E) xb = x4[threadIdx.x + NT * (i + 2)];
      __syncthreads();
      const int z0 = i * (RB >> 1);
#pragma unroll
      for (int q = 0; q < NZ; ++q) {
        const int u = threadIdx.x + q * NT;
        if (u < ZR * Wo) zr[u] = zv[q];
      }
      __syncthreads();
      if (i + 1 < E) { IRB_LOADZ(i + 1) }
      // g = blur adjoint of dz at this thread's float4 of band i, masked by ReLU (xhat > 0)
      int oy[3]; float wy[3];
      blur_down_adj(i * RB + lrow, H, Ho, oy, wy);
      float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (wy[a] == 0.f) continue;
        const float* row = zr + (oy[a] - z0) * Wo + c;
        const float g0 = row[0], g1 = row[1], g2 = c2 ? row[2] : 0.f;
        g[0] += wy[a] * (0.5f * g0);
        g[1] += wy[a] * (0.25f * g0 + 0.25f * g1 + (ix == 0 ? 0.25f * g0 : 0.f));   // column 1 also gets the reflected -1
        g[2] += wy[a] * (0.5f * g1);
        g[3] += wy[a] * (0.25f * g1 + 0.25f * g2);
      }
      const float h[4] = {(xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(h[e] > 0.f)) g[e] = 0.f;
      if (pass == 0) {
        s1 += (g[0] + g[1]) + (g[2] + g[3]);
        s2 += (g[0] * h[0] + g[1] * h[1]) + (g[2] * h[2] + g[3] * h[3]);
      } else {
        float4 o;
        o.x = rstd * (g[0] - m1 - h[0] * m2); o.y = rstd * (g[1] - m1 - h[1] * m2);
        o.z = rstd * (g[2] - m1 - h[2] * m2); o.w = rstd * (g[3] - m1 - h[3] * m2);
        d4[threadIdx.x + NT * i] = o;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
      }
    }
    if (pass == 0) {
      m1 = block_sum(s1, sm) / (float)S;
      m2 = block_sum(s2, sm) / (float)S;
    }
  }
#undef IRB_LOADZ
  if (amax) publish_block_absmax_acc(am, &smax, amax);
  if (amax && pmax && threadIdx.x == 0) pmax[blockIdx.x] = __uint_as_float(smax);
}

// The 256 x 256 planes take the ring-free kernels of in_blurdown.hip (built without packed-fp32 VALU instructions).
int df_in_relu_blurdown_fwd256_launch(const float* x, float* z, float* mean, float* rstd, int planes, float eps, float* z_amax,
                                      hipStream_t st, int W);
int df_in_relu_blurdown_bwd256_launch(const float* dz, const float* x, const float* mean, const float* rstd, float* dx,
                                      int planes, float* dx_amax, float* dx_pmax, hipStream_t st, int W);

// ---------------------------------------------------------------------------------------------
// Upsample (replicate pad 1 + conv_transpose 4x4 stride 2 pad 2, cropped) == per axis
//   out[2m]   = 0.75 x[m] + 0.25 x[max(m-1,0)]
//   out[2m+1] = 0.75 x[m] + 0.25 x[min(m+1,L-1)]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void blur_up_src(int o, int L, int& i0, int& i1) {
  const int m = o >> 1;
  i0 = m;
  i1 = (o & 1) ? (m + 1 < L ? m + 1 : L - 1) : (m > 0 ? m - 1 : 0);
}
__global__ void blur_up_fwd_k(const float* __restrict__ x, float* __restrict__ y, int planes, int H,
                              int W) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long long total = (long long)planes * Ho * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    long long r = i / Wo;
    const int oy = (int)(r % Ho);
    const long long pl = r / Ho;
    const float* xp = x + pl * H * W;
    int y0, y1, x0, x1;
    blur_up_src(oy, H, y0, y1);
    blur_up_src(ox, W, x0, x1);
    const float r0 = 0.75f * xp[(long long)y0 * W + x0] + 0.25f * xp[(long long)y0 * W + x1];
    const float r1 = 0.75f * xp[(long long)y1 * W + x0] + 0.25f * xp[(long long)y1 * W + x1];
    y[i] = 0.75f * r0 + 0.25f * r1;
  }
}
__device__ __forceinline__ int blur_up_adj(int m, int L, int* oo, float* ww) {
  int cnt = 0;
  oo[cnt] = 2 * m; ww[cnt++] = 0.75f;
  oo[cnt] = 2 * m + 1; ww[cnt++] = 0.75f;
  if (m + 1 <= L - 1) { oo[cnt] = 2 * m + 2; ww[cnt++] = 0.25f; }   // out[2(m+1)] uses x[m]
  if (m == 0) { oo[cnt] = 0; ww[cnt++] = 0.25f; }                    // clamp at the low edge
  if (m >= 1) { oo[cnt] = 2 * m - 1; ww[cnt++] = 0.25f; }            // out[2(m-1)+1] uses x[m]
  if (m == L - 1) { oo[cnt] = 2 * L - 1; ww[cnt++] = 0.25f; }        // clamp at the high edge
  return cnt;
}
__global__ void blur_up_bwd_k(const float* __restrict__ dy, float* __restrict__ dx, int planes, int H,
                              int W) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long long total = (long long)planes * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W);
    long long r = i / W;
    const int iy = (int)(r % H);
    const long long pl = r / H;
    const float* gp = dy + pl * Ho * Wo;
    int oy[6], oxx[6];
    float wy[6], wx[6];
    const int ny = blur_up_adj(iy, H, oy, wy);
    const int nx = blur_up_adj(ix, W, oxx, wx);
    float s = 0.f;
    for (int a = 0; a < ny; ++a) {
      float rs = 0.f;
      for (int b = 0; b < nx; ++b) rs += wx[b] * gp[(long long)oy[a] * Wo + oxx[b]];
      s += wy[a] * rs;
    }
    dx[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// 4-outputs-per-thread forms of the four resampling kernels (W % 8 == 0 / W % 4 == 0, even H): 16-B loads
// and stores, each loaded value reused by up to 4 outputs, 32-bit index math, one grid row per plane.
// Same arithmetic (and summation order) as the scalar kernels above, which remain the general fallback.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blur_down_fwd_v4_k(const float* __restrict__ x, float* __restrict__ y,
                                                          int H, int W) {
  const int Ho = H >> 1, Wo = W >> 1, G = Wo >> 2;          // G groups of 4 outputs per row
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Ho * G) return;
  const int oy = i / G, ox = (i - oy * G) << 2;
  const float* xp = x + (long long)blockIdx.y * H * W;
  float h[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int iy = refl1(2 * oy + a - 1, H);
    const float* row = xp + iy * W + 2 * ox;
    const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 4);
    const float l = ox ? row[-1] : row[1];                  // reflect: column -1 -> column 1
    const float v[9] = {l, v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float rs = 0.f;
      rs += 0.25f * v[2 * j]; rs += 0.5f * v[2 * j + 1]; rs += 0.25f * v[2 * j + 2];
      h[a][j] = rs;
    }
  }
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = 0.f;
    t += 0.25f * h[0][j]; t += 0.5f * h[1][j]; t += 0.25f * h[2][j];
    o[j] = t;
  }
  *reinterpret_cast<float4*>(y + ((long long)blockIdx.y * Ho + oy) * Wo + ox) = make_float4(o[0], o[1], o[2], o[3]);
}
// adjoint of the above (even H, W): per axis, input i gets g[i/2]/2 (i even) or (g[(i-1)/2] + g[(i+1)/2])/4
// (i odd; the second term only while (i+1)/2 < n/2), plus g[0]/4 at i == 1 (the reflected column -1)
__global__ __launch_bounds__(256) void blur_down_bwd_v4_k(const float* __restrict__ dy, float* __restrict__ dx,
                                                          int H, int W) {
  const int Ho = H >> 1, Wo = W >> 1, G = W >> 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * G) return;
  const int iy = i / G, ix = (i - iy * G) << 2;
  const float* gp = dy + (long long)blockIdx.y * Ho * Wo;
  int oy[3]; float wy[3];
  blur_down_adj(iy, H, Ho, oy, wy);
  const int c = ix >> 1;                                     // dy columns c, c+1, c+2 serve inputs ix..ix+3
  const bool c2 = c + 2 < Wo;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (wy[a] == 0.f) continue;
    const float* row = gp + oy[a] * Wo + c;
    const float g0 = row[0], g1 = row[1], g2 = c2 ? row[2] : 0.f;
    const float r0 = 0.5f * g0;
    const float r1 = 0.25f * g0 + 0.25f * g1 + (ix == 0 ? 0.25f * g0 : 0.f);   // column 1 also gets the reflected -1
    const float r2 = 0.5f * g1;
    const float r3 = 0.25f * g1 + 0.25f * g2;
    s[0] += wy[a] * r0; s[1] += wy[a] * r1; s[2] += wy[a] * r2; s[3] += wy[a] * r3;
  }
  *reinterpret_cast<float4*>(dx + ((long long)blockIdx.y * H + iy) * W + ix) = make_float4(s[0], s[1], s[2], s[3]);
}
// thread = 2 input columns of one row -> one float4 of each of the two output rows: every store instruction of a wave
// writes 1 KB of consecutive bytes (with 4 input columns per thread the two float4 stores of a row interleaved at 32-byte
// stride).  The neighbouring columns are the neighbouring lanes' values (consecutive lanes = consecutive pairs of a row; at
// a row's ends the clamp takes the thread's own): lane-strided 4-byte loads made this kernel TA-bound.
__global__ __launch_bounds__(256) void blur_up_fwd_v4_k(const float* __restrict__ x, float* __restrict__ y,
                                                        int H, int W) {
  const int G = W >> 1, Wo = 2 * W, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * G) return;
  const int iy = i / G, ix = (i - iy * G) << 1;
  const float* xp = x + (long long)blockIdx.y * H * W;
  float v[3][4];                                             // rows iy-1, iy, iy+1 (clamped); columns ix-1 .. ix+2 (clamped)
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int r = iy + a - 1;
    r = r < 0 ? 0 : (r > H - 1 ? H - 1 : r);
    const float* row = xp + r * W + ix;
    const float2 q = *reinterpret_cast<const float2*>(row);
    const float lft = __shfl_up(q.y, 1, 64), rgt = __shfl_down(q.x, 1, 64);
    v[a][0] = ix ? (lane ? lft : row[-1]) : q.x;
    v[a][1] = q.x; v[a][2] = q.y;
    v[a][3] = (ix + 2 < W) ? (lane < 63 ? rgt : row[2]) : q.y;
  }
  float* yp = y + ((long long)blockIdx.y * 2 * H + 2 * iy) * Wo + 2 * ix;
#pragma unroll
  for (int p = 0; p < 2; ++p) {                              // output rows 2iy (neighbour iy-1) and 2iy+1 (neighbour iy+1)
    const int nb = p ? 2 : 0;
    float o[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {                          // output columns 2(ix+j) (neighbour -1) and +1 (neighbour +1)
        const int xn = q ? j + 2 : j;
        const float r0 = 0.75f * v[1][j + 1] + 0.25f * v[1][xn];
        const float r1 = 0.75f * v[nb][j + 1] + 0.25f * v[nb][xn];
        o[2 * j + q] = 0.75f * r0 + 0.25f * r1;
      }
    }
    *reinterpret_cast<float4*>(yp + p * Wo) = make_float4(o[0], o[1], o[2], o[3]);
  }
}
// adjoint: per axis the 4-tap filter (1/4, 3/4, 3/4, 1/4) over dy[2m-1 .. 2m+2] with clamped indices
__global__ __launch_bounds__(256) void blur_up_bwd_v4_k(const float* __restrict__ dy, float* __restrict__ dx,
                                                        int H, int W) {
  const int G = W >> 2, Wo = 2 * W, Ho = 2 * H, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * G) return;
  const int iy = i / G, ix = (i - iy * G) << 2;
  const float* gp = dy + (long long)blockIdx.y * Ho * Wo;
  const float w4[4] = {0.25f, 0.75f, 0.75f, 0.25f};
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    int r = 2 * iy - 1 + a;
    r = r < 0 ? 0 : (r > Ho - 1 ? Ho - 1 : r);
    const float* row = gp + r * Wo + 2 * ix;
    const float4 q0 = *reinterpret_cast<const float4*>(row), q1 = *reinterpret_cast<const float4*>(row + 4);
    const float lft = __shfl_up(q1.w, 1, 64), rgt = __shfl_down(q0.x, 1, 64);      // see blur_up_fwd_v4_k
    const float l = ix ? (lane ? lft : row[-1]) : q0.x, rr = (2 * ix + 8 < Wo) ? (lane < 63 ? rgt : row[8]) : q1.w;
    const float v[10] = {l, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, rr};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float hs = 0.25f * v[2 * j] + 0.75f * v[2 * j + 1] + 0.75f * v[2 * j + 2] + 0.25f * v[2 * j + 3];
      s[j] += w4[a] * hs;
    }
  }
  *reinterpret_cast<float4*>(dx + ((long long)blockIdx.y * H + iy) * W + ix) = make_float4(s[0], s[1], s[2], s[3]);
}

// ---------------------------------------------------------------------------------------------
__global__ void reflect_pad2d_fwd_k(const float* __restrict__ x, float* __restrict__ y, int planes,
                                    int H, int W, int p) {
  const int Ho = H + 2 * p, Wo = W + 2 * p;
  const long long total = (long long)planes * Ho * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    long long r = i / Wo;
    const int oy = (int)(r % Ho);
    const long long pl = r / Ho;
    y[i] = x[pl * H * W + (long long)refl1(oy - p, H) * W + refl1(ox - p, W)];
  }
}
// padded positions q (in padded coords) that read input index i
__device__ __forceinline__ int reflect_adj(int i, int n, int p, int* qq) {
  int cnt = 0;
  qq[cnt++] = i + p;
  if (i >= 1 && i <= p) qq[cnt++] = p - i;
  if (i <= n - 2 && i >= n - 1 - p) qq[cnt++] = p + 2 * (n - 1) - i;
  return cnt;
}
__global__ void reflect_pad2d_bwd_k(const float* __restrict__ dy, const float* __restrict__ add,
                                    float* __restrict__ dx, int planes, int H, int W, int p) {
  const int Ho = H + 2 * p, Wo = W + 2 * p;
  const long long total = (long long)planes * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W);
    long long r = i / W;
    const int iy = (int)(r % H);
    const long long pl = r / H;
    const float* gp = dy + pl * Ho * Wo;
    int qy[3], qx[3];
    const int ny = reflect_adj(iy, H, p, qy);
    const int nx = reflect_adj(ix, W, p, qx);
    float s = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) s += gp[(long long)qy[a] * Wo + qx[b]];
    dx[i] = add ? s + add[i] : s;
  }
}

// pad 1, W % 4 == 0, H, W >= 8: 4 outputs per thread (one 16-B load of the padded row + the two halo columns
// at the row ends), rows 1 and H-2 also take the halo rows 0 and H+1.  `add` (optional): a second gradient of the
// same tensor -- the residual branch of a ResnetBlock -- summed in the same pass instead of by a separate kernel
__global__ __launch_bounds__(256) void reflect_pad1_bwd_v4_k(const float* __restrict__ dy, const float* __restrict__ add,
                                                             float* __restrict__ dx, int H, int W) {
  typedef unsigned rp_u32x4 __attribute__((ext_vector_type(4)));
  const int G = W >> 2, Wo = W + 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * G) return;
  const int y = i / G, x0 = (i - y * G) << 2;
  const float* gp = dy + (long long)blockIdx.y * (H + 2) * Wo;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gp), 0,
                                                                      (unsigned)((H + 2) * Wo) * 4u, 0x00020000);
  int rows[2] = {y + 1, y + 1};
  int nr = 1;
  if (y == 1) { rows[1] = 0; nr = 2; }
  else if (y == H - 2) { rows[1] = H + 1; nr = 2; }
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int a = 0; a < nr; ++a) {
    const int rb = rows[a] * Wo;
    const rp_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(rb + x0 + 1) * 4u, 0, 0);
    s[0] += __uint_as_float(v[0]); s[1] += __uint_as_float(v[1]);
    s[2] += __uint_as_float(v[2]); s[3] += __uint_as_float(v[3]);
    if (x0 == 0) s[1] += gp[rb];                  // output column 1 <- padded column 0
    if (x0 == W - 4) s[2] += gp[rb + W + 1];      // output column W-2 <- padded column W+1
  }
  const long long o = ((long long)blockIdx.y * H + y) * W + x0;
  if (add) {
    const float4 r = *reinterpret_cast<const float4*>(add + o);
    s[0] += r.x; s[1] += r.y; s[2] += r.z; s[3] += r.w;
  }
  *reinterpret_cast<float4*>(dx + o) = make_float4(s[0], s[1], s[2], s[3]);
}

// ---------------------------------------------------------------------------------------------
// y[n][c] = c < Ca ? a[n][c][z/sd][y/2][x/2] : b[n][c-Ca][z][y][x]
// ---------------------------------------------------------------------------------------------
__global__ void upcat_fwd_k(const float* __restrict__ a, const float* __restrict__ b,
                            float* __restrict__ y, int N, int Ca, int Cb, int Da, int Ha, int Wa, int sd) {
  const int Do = Da * sd, Ho = Ha * 2, Wo = Wa * 2, C = Ca + Cb;
  const long long So = (long long)Do * Ho * Wo, Sa = (long long)Da * Ha * Wa;
  const long long total = (long long)N * C * So;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wo);
    long long r = i / Wo;
    const int yy = (int)(r % Ho); r /= Ho;
    const int z = (int)(r % Do); r /= Do;
    const int c = (int)(r % C);
    const long long n = r / C;
    float v;
    if (c < Ca) v = a[(n * Ca + c) * Sa + ((long long)(z / sd) * Ha + (yy >> 1)) * Wa + (x >> 1)];
    else v = b[(n * Cb + (c - Ca)) * So + ((long long)z * Ho + yy) * Wo + x];
    y[i] = v;
  }
}
// the same, 4 x-consecutive outputs per thread (Wa even): one float2 of `a` (or one float4 of `b`) -> one float4 store.
// IDX = unsigned when the element count is below 2^31 (these kernels and the three below): the 64-bit division chain of
// the coordinate decode was most of a thread's instructions
template <typename IDX>
__global__ __launch_bounds__(256) void upcat_fwd_v4_k(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ y, int N, int Ca, int Cb, int Da, int Ha,
                                                      int Wa, int sd) {
  const int Do = Da * sd, Ho = Ha * 2, Wo = Wa * 2, C = Ca + Cb, Wq = Wo >> 2;
  const long long So = (long long)Do * Ho * Wo, Sa = (long long)Da * Ha * Wa;
  const IDX total = (IDX)((long long)N * C * Do * Ho * Wq);
  for (IDX i = (IDX)blockIdx.x * 256 + threadIdx.x; i < total; i += (IDX)gridDim.x * 256) {
    const int xq = (int)(i % (IDX)Wq);
    IDX r = i / (IDX)Wq;
    const int yy = (int)(r % (IDX)Ho); r /= (IDX)Ho;
    const int z = (int)(r % (IDX)Do); r /= (IDX)Do;
    const int c = (int)(r % (IDX)C);
    const long long n = (long long)(r / (IDX)C);
    float4 v;
    if (c < Ca) {
      const float2 t = *reinterpret_cast<const float2*>(a + (n * Ca + c) * Sa + ((long long)(z / sd) * Ha + (yy >> 1)) * Wa + 2 * xq);
      v = make_float4(t.x, t.x, t.y, t.y);
    } else {
      v = *reinterpret_cast<const float4*>(b + (n * Cb + (c - Ca)) * So + ((long long)z * Ho + yy) * Wo + 4 * xq);
    }
    *reinterpret_cast<float4*>(y + ((n * C + c) * Do + z) * (long long)Ho * Wo + (long long)yy * Wo + 4 * xq) = v;
  }
}
// d(a)[n][c][z][y][x] = the sum of dy over the sd x 2 x 2 outputs that copied it; V2: a thread owns two x-consecutive
// inputs (one float4 of dy per output row)
template <typename IDX, bool V2>
__global__ __launch_bounds__(256) void upcat_bwd_a_k(const float* __restrict__ dy, float* __restrict__ da, int N, int Ca, int Cb,
                                                     int Da, int Ha, int Wa, int sd) {
  const int Do = Da * sd, Ho = Ha * 2, Wo = Wa * 2, C = Ca + Cb, Wh = V2 ? (Wa >> 1) : Wa;
  const long long So = (long long)Do * Ho * Wo;
  const IDX total = (IDX)((long long)N * Ca * Da * Ha * Wh);
  for (IDX i = (IDX)blockIdx.x * 256 + threadIdx.x; i < total; i += (IDX)gridDim.x * 256) {
    const int x = (int)(i % (IDX)Wh);
    IDX r = i / (IDX)Wh;
    const int yy = (int)(r % (IDX)Ha); r /= (IDX)Ha;
    const int z = (int)(r % (IDX)Da); r /= (IDX)Da;
    const int c = (int)(r % (IDX)Ca);
    const long long n = (long long)(r / (IDX)Ca);
    const float* gp = dy + (n * C + c) * So;
    float s0 = 0.f, s1 = 0.f;
    for (int dz = 0; dz < sd; ++dz)
#pragma unroll
      for (int dyy = 0; dyy < 2; ++dyy) {
        const float* row = gp + ((long long)(z * sd + dz) * Ho + (2 * yy + dyy)) * Wo;
        if (V2) {
          const float4 t = *reinterpret_cast<const float4*>(row + 4 * x);
          s0 += t.x + t.y; s1 += t.z + t.w;
        } else {
          s0 += row[2 * x] + row[2 * x + 1];
        }
      }
    float* o = da + (((n * Ca + c) * Da + z) * (long long)Ha + yy) * Wa;
    if (V2) *reinterpret_cast<float2*>(o + 2 * x) = make_float2(s0, s1);
    else o[x] = s0;
  }
}
// y[n] = cat(a[n], b[n]) along channels (dir 0) or its two slice copies (dir 1; a NULL destination is skipped); SA = Ca*S,
// SB = Cb*S per-sample element counts IN UNITS OF V floats.  upcat's d(b) is the b slice of this copy.
template <typename IDX, int V>
__global__ __launch_bounds__(256) void cat_channels_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                      long long N, long long SA, long long SB, int dir) {
  typedef float vec_t __attribute__((ext_vector_type(V)));
  const IDX ST = (IDX)(SA + SB), total = (IDX)(N * (SA + SB));
  const vec_t* av = reinterpret_cast<const vec_t*>(a);
  const vec_t* bv = reinterpret_cast<const vec_t*>(b);
  vec_t* yv = reinterpret_cast<vec_t*>(y);
  vec_t* aw = const_cast<vec_t*>(av);
  vec_t* bw = const_cast<vec_t*>(bv);
  for (IDX i = (IDX)blockIdx.x * 256 + threadIdx.x; i < total; i += (IDX)gridDim.x * 256) {
    const IDX n = i / ST, r = i - n * ST;
    if (dir == 0) yv[i] = r < (IDX)SA ? av[(long long)n * SA + r] : bv[(long long)n * SB + (r - (IDX)SA)];
    else if (r < (IDX)SA) { if (aw) aw[(long long)n * SA + r] = yv[i]; }
    else { if (bw) bw[(long long)n * SB + (r - (IDX)SA)] = yv[i]; }
  }
}
static void cat_channels_launch(const float* a, const float* b, float* y, long long N, long long SA, long long SB, int dir,
                                hipStream_t st) {
  const bool v4 = ((SA | SB) & 3) == 0 &&
                  ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  const long long V = v4 ? 4 : 1, total = N * (SA + SB) / V;
  const unsigned grid = df_grid(total, 256, 16384);
  const bool small = total < 0x7FFFFFFFLL;
  if (v4) {
    if (small) cat_channels_k<unsigned, 4><<<grid, 256, 0, st>>>(a, b, y, N, SA / 4, SB / 4, dir);
    else cat_channels_k<long long, 4><<<grid, 256, 0, st>>>(a, b, y, N, SA / 4, SB / 4, dir);
  } else {
    if (small) cat_channels_k<unsigned, 1><<<grid, 256, 0, st>>>(a, b, y, N, SA, SB, dir);
    else cat_channels_k<long long, 1><<<grid, 256, 0, st>>>(a, b, y, N, SA, SB, dir);
  }
}
__global__ void scale_k(const float* __restrict__ x, float* __restrict__ y, long long n, float mult) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    y[i] = mult * x[i];
}

// ---------------------------------------------------------------------------------------------
int df_absmax_launch(const float* x, long long n, float* out, hipStream_t st, bool zero_first);
// y_amax / dx_amax (may be NULL): DF_PROBE_SLOTS (64) floats, zero-initialised by the caller; their maximum is raised
// to max |output| -- the range
// probe the fp16x2 conv kernels need for their next input, produced while the output is still in registers (one
// conditional atomic per workgroup, common.h publish_block_absmax_acc).
// Statistics only: mean / rstd of every plane and the range probe of relu?(IN(x)), nothing written back -- the consumer
// normalises while it stages its operand (dfmir_conv3x3_fwd_norm).  Plane sizes of the register-resident kernels only.
extern "C" int dfmir_instnorm_stats_ok(long long S) { return (S == 4096 || S == 16384 || S == 65536) ? 1 : 0; }
extern "C" int dfmir_instnorm_stats(const float* x, float* mean, float* rstd, int planes, long long S, float eps, int relu,
                                    float* y_amax, void* stream) {
  DF_ARG_CHECK(x && mean && rstd && planes > 0 && dfmir_instnorm_stats_ok(S));
  hipStream_t st = (hipStream_t)stream;
  if (S == 4096) instnorm_fwd_reg_k<256, 4><<<planes, 256, 0, st>>>(x, nullptr, nullptr, mean, rstd, eps, relu, y_amax);
  else if (S == 16384) instnorm_fwd_reg_k<256, 16><<<planes, 256, 0, st>>>(x, nullptr, nullptr, mean, rstd, eps, relu, y_amax);
  else instnorm_fwd_reg_k<1024, 16><<<planes, 1024, 0, st>>>(x, nullptr, nullptr, mean, rstd, eps, relu, y_amax);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_instnorm_fwd(const float* x, const float* res, float* y, float* mean, float* rstd,
                                  int planes, long long S, float eps, int relu, float* y_amax, void* stream) {
  DF_ARG_CHECK(x && y && mean && rstd && planes > 0 && S > 0);
  hipStream_t st = (hipStream_t)stream;
  if (S == 4096) instnorm_fwd_reg_k<256, 4><<<planes, 256, 0, st>>>(x, res, y, mean, rstd, eps, relu, y_amax);
  else if (S == 16384) instnorm_fwd_reg_k<256, 16><<<planes, 256, 0, st>>>(x, res, y, mean, rstd, eps, relu, y_amax);
  else if (S == 65536) instnorm_fwd_reg_k<1024, 16><<<planes, 1024, 0, st>>>(x, res, y, mean, rstd, eps, relu, y_amax);
  else {
    instnorm_fwd_k<<<planes, 256, 0, st>>>(x, res, y, mean, rstd, S, eps, relu);
    if (y_amax) {   // generic plane size: a pass over the output
      DF_LAUNCH_CHECK();
      const int rc = df_absmax_launch(y, (long long)planes * S, y_amax, st, false);
      if (rc) return df_set_error(rc, __FILE__, __LINE__);
    }
  }
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_instnorm_bwd_cols_ok(long long S, int W) {
  return ((S == 4096 || S == 16384) && W >= 4 && (W & 3) == 0 && S % W == 0) ? 1 : 0;
}
static int instnorm_bwd_impl(const float* dy, const float* x, const float* mean, const float* rstd, float* dx,
                             int planes, long long S, int relu, float* dx_amax, float* dx_cols, int W, void* stream,
                             float* dx_pmax = nullptr);
extern "C" int dfmir_instnorm_bwd_pmax_ok(long long S) { return (S == 4096 || S == 16384 || S == 65536) ? 1 : 0; }
extern "C" int dfmir_instnorm_bwd_pmax(const float* dy, const float* x, const float* mean, const float* rstd, float* dx,
                                       int planes, long long S, int relu, float* dx_amax, float* dx_cols, int W,
                                       float* dx_pmax, void* stream) {
  DF_ARG_CHECK(dx_amax && dx_pmax && dfmir_instnorm_bwd_pmax_ok(S) && (!dx_cols || dfmir_instnorm_bwd_cols_ok(S, W)));
  return instnorm_bwd_impl(dy, x, mean, rstd, dx, planes, S, relu, dx_amax, dx_cols, W, stream, dx_pmax);
}
extern "C" int dfmir_instnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                  float* dx, int planes, long long S, int relu, float* dx_amax, void* stream) {
  return instnorm_bwd_impl(dy, x, mean, rstd, dx, planes, S, relu, dx_amax, nullptr, 0, stream);
}
extern "C" int dfmir_instnorm_bwd_cols(const float* dy, const float* x, const float* mean, const float* rstd,
                                       float* dx, int planes, long long S, int relu, float* dx_amax, float* dx_cols,
                                       int W, void* stream) {
  DF_ARG_CHECK(dx_cols && dfmir_instnorm_bwd_cols_ok(S, W));
  return instnorm_bwd_impl(dy, x, mean, rstd, dx, planes, S, relu, dx_amax, dx_cols, W, stream);
}
static int instnorm_bwd_impl(const float* dy, const float* x, const float* mean, const float* rstd, float* dx,
                             int planes, long long S, int relu, float* dx_amax, float* dx_cols, int W, void* stream,
                             float* dx_pmax) {
  DF_ARG_CHECK(dy && x && mean && rstd && dx && planes > 0 && S > 0);
  hipStream_t st = (hipStream_t)stream;
  if (S == 4096 && dx_cols) instnorm_bwd_reg_k<256, 4, true><<<planes, 256, 0, st>>>(dy, x, mean, rstd, dx, relu, dx_amax, dx_cols, W, dx_pmax);
  else if (S == 4096) instnorm_bwd_reg_k<256, 4, false><<<planes, 256, 0, st>>>(dy, x, mean, rstd, dx, relu, dx_amax, nullptr, 0, dx_pmax);
  else if (S == 16384 && dx_cols) instnorm_bwd_reg_k<256, 16, true><<<planes, 256, 0, st>>>(dy, x, mean, rstd, dx, relu, dx_amax, dx_cols, W, dx_pmax);
  else if (S == 16384) instnorm_bwd_reg_k<256, 16, false><<<planes, 256, 0, st>>>(dy, x, mean, rstd, dx, relu, dx_amax, nullptr, 0, dx_pmax);
  else if (S == 65536 && dx_cols) return df_set_error(-1, __FILE__, __LINE__);     // dfmir_instnorm_bwd_cols_ok excludes it
  else if (S == 65536) instnorm_bwd_reg_k<1024, 16, false><<<planes, 1024, 0, st>>>(dy, x, mean, rstd, dx, relu, dx_amax, nullptr, 0, dx_pmax);
  else {
    instnorm_bwd_k<<<planes, 256, 0, st>>>(dy, x, mean, rstd, dx, S, relu);
    if (dx_amax) {
      DF_LAUNCH_CHECK();
      const int rc = df_absmax_launch(dx, (long long)planes * S, dx_amax, st, false);
      if (rc) return df_set_error(rc, __FILE__, __LINE__);
    }
  }
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_in_relu_blurdown_ok(int H, int W) {
  return ((H == 256 && W == 256) || (H == 128 && W == 128)) ? 1 : 0;
}
extern "C" int dfmir_in_relu_blurdown_fwd(const float* x, float* z, float* mean, float* rstd, int planes, int H, int W,
                                          float eps, float* z_amax, void* stream) {
  DF_ARG_CHECK(x && z && mean && rstd && planes > 0 && dfmir_in_relu_blurdown_ok(H, W));
  hipStream_t st = (hipStream_t)stream;
  static DfOptFlag banded_o{"DFMIR_IN_BLUR_BANDED"};
  const bool banded = banded_o.get();      // A/B: the LDS-ring form on the 256^2 planes
  if (!banded) return df_in_relu_blurdown_fwd256_launch(x, z, mean, rstd, planes, eps, z_amax, st, W);
  if (W == 256) in_relu_blurdown_fwd_k<1024, 16><<<planes, 1024, 2 * 16 * 256 * 4, st>>>(x, z, mean, rstd, W, eps, z_amax);
  else in_relu_blurdown_fwd_k<256, 16><<<planes, 256, 2 * 8 * 128 * 4, st>>>(x, z, mean, rstd, W, eps, z_amax);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_in_relu_blurdown_bwd(const float* dz, const float* x, const float* mean, const float* rstd, float* dx,
                                          int planes, int H, int W, float* dx_amax, float* dx_pmax, void* stream) {
  DF_ARG_CHECK(dz && x && mean && rstd && dx && planes > 0 && dfmir_in_relu_blurdown_ok(H, W));
  hipStream_t st = (hipStream_t)stream;
  static DfOptFlag banded_o{"DFMIR_IN_BLUR_BANDED"};
  const bool banded = banded_o.get();
  if (!banded) return df_in_relu_blurdown_bwd256_launch(dz, x, mean, rstd, dx, planes, dx_amax, dx_pmax, st, W);
  if (W == 256) in_relu_blurdown_bwd_k<1024, 16><<<planes, 1024, (8 + 1) * 128 * 4, st>>>(dz, x, mean, rstd, dx, W, dx_amax, dx_pmax);
  else in_relu_blurdown_bwd_k<256, 16><<<planes, 256, (4 + 1) * 64 * 4, st>>>(dz, x, mean, rstd, dx, W, dx_amax, dx_pmax);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_act_bwd(const float* dy, const float* y, float* dx, long long n, int act,
                             float slope, void* stream) {
  DF_ARG_CHECK(dy && y && dx && n > 0 && (act == 1 || act == 2));
  if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0)
    act_bwd_v4_k<<<df_grid(n / 4, 256, 8192), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(y), reinterpret_cast<float4*>(dx), n / 4, act, slope);
  else
    act_bwd_k<<<df_grid(n, 256, 4096), 256, 0, (hipStream_t)stream>>>(dy, y, dx, n, act, slope);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_act_bwd_amax(const float* dy, const float* y, float* dx, long long n, int act, float slope,
                                  float* dx_amax, void* stream) {
  DF_ARG_CHECK(dy && y && dx && dx_amax && n > 0 && (act == 1 || act == 2));
  DF_ARG_CHECK(((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0);
  act_bwd_amax_k<<<df_grid((n + 3) / 4, 256, 4096), 256, 0, (hipStream_t)stream>>>(dy, y, dx, n, act, slope, dx_amax);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_blur_down_fwd(const float* x, float* y, int planes, int H, int W, void* stream) {
  DF_ARG_CHECK(x && y && planes > 0 && H > 1 && W > 1);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if ((H & 1) == 0 && (W & 7) == 0 && planes <= 65535 && (long long)H * W < (1LL << 30)) {
    blur_down_fwd_v4_k<<<dim3((unsigned)(((H / 2) * (W / 8) + 255) / 256), (unsigned)planes), 256, 0,
                         (hipStream_t)stream>>>(x, y, H, W);
    DF_LAUNCH_CHECK();
    return 0;
  }
  blur_down_fwd_k<<<df_grid((long long)planes * Ho * Wo, 256, 8192), 256, 0, (hipStream_t)stream>>>(
      x, y, planes, H, W, Ho, Wo);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_blur_down_bwd(const float* dy, float* dx, int planes, int H, int W, void* stream) {
  DF_ARG_CHECK(dy && dx && planes > 0 && H > 1 && W > 1);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if ((H & 1) == 0 && (W & 3) == 0 && H >= 4 && W >= 4 && planes <= 65535 && (long long)H * W < (1LL << 30)) {
    blur_down_bwd_v4_k<<<dim3((unsigned)((H * (W / 4) + 255) / 256), (unsigned)planes), 256, 0,
                         (hipStream_t)stream>>>(dy, dx, H, W);
    DF_LAUNCH_CHECK();
    return 0;
  }
  blur_down_bwd_k<<<df_grid((long long)planes * H * W, 256, 8192), 256, 0, (hipStream_t)stream>>>(
      dy, dx, planes, H, W, Ho, Wo);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_blur_up_fwd(const float* x, float* y, int planes, int H, int W, void* stream) {
  DF_ARG_CHECK(x && y && planes > 0 && H > 0 && W > 0);
  if ((W & 1) == 0 && planes <= 65535 && (long long)H * W < (1LL << 28)) {
    blur_up_fwd_v4_k<<<dim3((unsigned)((H * (W / 2) + 255) / 256), (unsigned)planes), 256, 0,
                       (hipStream_t)stream>>>(x, y, H, W);
    DF_LAUNCH_CHECK();
    return 0;
  }
  blur_up_fwd_k<<<df_grid((long long)planes * H * W * 4, 256, 8192), 256, 0, (hipStream_t)stream>>>(
      x, y, planes, H, W);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_blur_up_bwd(const float* dy, float* dx, int planes, int H, int W, void* stream) {
  DF_ARG_CHECK(dy && dx && planes > 0 && H > 0 && W > 0);
  if ((W & 3) == 0 && planes <= 65535 && (long long)H * W < (1LL << 28)) {
    blur_up_bwd_v4_k<<<dim3((unsigned)((H * (W / 4) + 255) / 256), (unsigned)planes), 256, 0,
                       (hipStream_t)stream>>>(dy, dx, H, W);
    DF_LAUNCH_CHECK();
    return 0;
  }
  blur_up_bwd_k<<<df_grid((long long)planes * H * W, 256, 8192), 256, 0, (hipStream_t)stream>>>(
      dy, dx, planes, H, W);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_reflect_pad2d_fwd(const float* x, float* y, int planes, int H, int W, int p,
                                       void* stream) {
  DF_ARG_CHECK(x && y && planes > 0 && p >= 0 && p < H && p < W);
  reflect_pad2d_fwd_k<<<df_grid((long long)planes * (H + 2 * p) * (W + 2 * p), 256, 8192), 256, 0,
                        (hipStream_t)stream>>>(x, y, planes, H, W, p);
  DF_LAUNCH_CHECK();
  return 0;
}
static int reflect_pad2d_bwd_impl(const float* dy, const float* add, float* dx, int planes, int H, int W, int p,
                                  void* stream) {
  DF_ARG_CHECK(dy && dx && planes > 0 && p >= 0 && p < H && p < W);
  if (p == 1 && (W & 3) == 0 && H >= 8 && W >= 8 && planes <= 65535 && (long long)(H + 2) * (W + 2) < (1LL << 29)) {
    reflect_pad1_bwd_v4_k<<<dim3((unsigned)((H * (W / 4) + 255) / 256), (unsigned)planes), 256, 0,
                            (hipStream_t)stream>>>(dy, add, dx, H, W);
    DF_LAUNCH_CHECK();
    return 0;
  }
  reflect_pad2d_bwd_k<<<df_grid((long long)planes * H * W, 256, 8192), 256, 0, (hipStream_t)stream>>>(
      dy, add, dx, planes, H, W, p);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_reflect_pad2d_bwd(const float* dy, float* dx, int planes, int H, int W, int p,
                                       void* stream) {
  return reflect_pad2d_bwd_impl(dy, nullptr, dx, planes, H, W, p, stream);
}
extern "C" int dfmir_reflect_pad2d_bwd_add(const float* dy, const float* add, float* dx, int planes, int H, int W,
                                           int p, void* stream) {
  DF_ARG_CHECK(add != nullptr);
  return reflect_pad2d_bwd_impl(dy, add, dx, planes, H, W, p, stream);
}
extern "C" int dfmir_upcat_fwd(const float* a, const float* b, float* y, int N, int Ca, int Cb, int Da,
                               int Ha, int Wa, int sd, void* stream) {
  DF_ARG_CHECK(a && b && y && N > 0 && Ca > 0 && Cb > 0 && (sd == 1 || sd == 2));
  const long long total = (long long)N * (Ca + Cb) * Da * sd * Ha * 2 * Wa * 2;
  if ((Wa & 1) == 0 && ((reinterpret_cast<uintptr_t>(a) & 7) | (reinterpret_cast<uintptr_t>(b) & 15) | (reinterpret_cast<uintptr_t>(y) & 15)) == 0)
  {
    if (total / 4 < 0x7FFFFFFFLL)
      upcat_fwd_v4_k<unsigned><<<df_grid(total / 4, 256, 16384), 256, 0, (hipStream_t)stream>>>(a, b, y, N, Ca, Cb, Da, Ha, Wa, sd);
    else
      upcat_fwd_v4_k<long long><<<df_grid(total / 4, 256, 16384), 256, 0, (hipStream_t)stream>>>(a, b, y, N, Ca, Cb, Da, Ha, Wa, sd);
  } else
    upcat_fwd_k<<<df_grid(total, 256, 16384), 256, 0, (hipStream_t)stream>>>(a, b, y, N, Ca, Cb, Da, Ha, Wa, sd);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_upcat_bwd(const float* dy, float* da, float* db, int N, int Ca, int Cb, int Da,
                               int Ha, int Wa, int sd, void* stream) {
  DF_ARG_CHECK(dy && (da || db) && N > 0 && Ca > 0 && Cb > 0 && (sd == 1 || sd == 2));   // NULL: that gradient is not wanted
  const long long Sa = (long long)Da * Ha * Wa, So = Sa * sd * 4;
  hipStream_t st = (hipStream_t)stream;
  if (da) {
    const bool v2 = (Wa & 1) == 0 && ((reinterpret_cast<uintptr_t>(dy) & 15) | (reinterpret_cast<uintptr_t>(da) & 7)) == 0;
    const long long total = (long long)N * Ca * Sa / (v2 ? 2 : 1);
    const unsigned grid = df_grid(total, 256, 16384);
    const bool small = total < 0x7FFFFFFFLL;
    if (v2) {
      if (small) upcat_bwd_a_k<unsigned, true><<<grid, 256, 0, st>>>(dy, da, N, Ca, Cb, Da, Ha, Wa, sd);
      else upcat_bwd_a_k<long long, true><<<grid, 256, 0, st>>>(dy, da, N, Ca, Cb, Da, Ha, Wa, sd);
    } else {
      if (small) upcat_bwd_a_k<unsigned, false><<<grid, 256, 0, st>>>(dy, da, N, Ca, Cb, Da, Ha, Wa, sd);
      else upcat_bwd_a_k<long long, false><<<grid, 256, 0, st>>>(dy, da, N, Ca, Cb, Da, Ha, Wa, sd);
    }
    DF_LAUNCH_CHECK();
  }
  if (db) {
    cat_channels_launch(nullptr, db, const_cast<float*>(dy), N, (long long)Ca * So, (long long)Cb * So, 1, st);
    DF_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int dfmir_cat_channels_fwd(const float* a, const float* b, float* y, long long N, long long SA,
                                      long long SB, void* stream) {
  DF_ARG_CHECK(a && b && y && N > 0 && SA > 0 && SB > 0);
  cat_channels_launch(a, b, y, N, SA, SB, 0, (hipStream_t)stream);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_cat_channels_bwd(const float* dy, float* da, float* db, long long N, long long SA,
                                      long long SB, void* stream) {
  DF_ARG_CHECK(dy && N > 0 && SA > 0 && SB > 0);
  cat_channels_launch(da, db, const_cast<float*>(dy), N, SA, SB, 1, (hipStream_t)stream);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_scale(const float* x, float* y, long long n, float mult, void* stream) {
  DF_ARG_CHECK(x && y && n > 0);
  scale_k<<<df_grid(n, 256, 8192), 256, 0, (hipStream_t)stream>>>(x, y, n, mult);
  DF_LAUNCH_CHECK();
  return 0;
}

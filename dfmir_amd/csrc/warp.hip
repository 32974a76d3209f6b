// Displacement-field warps (SpatialTransformer), the fused VecInt step and ResizeTransform.
// All HBM-bound gathers: one thread per output voxel, lanes along x (coalesced flow / output
// streams, L1/L2-served neighbourhood reads of src), channel loop inside the thread so the
// interpolation weights are computed once per voxel.  The kernels take the voxel displacement
// directly -- the reference's normalise-to-[-1,1] / permute / grid_sample round trip
// (torchvoxelmorph/layers.py:36-48) is never materialised.
#include "common.h"

// ------------------------------------------------------------------------------------------ 2-D
__global__ __launch_bounds__(256) void warp2d_fwd_k(const float* __restrict__ src,
                                                    const float* __restrict__ flow,
                                                    float* __restrict__ out, int B, int C, int H, int W,
                                                    int mode, int add_identity) {
  const long long HW = (long long)H * W;
  const long long total = (long long)B * HW;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const long long r = i / W;
  const int y = (int)(r % H);
  const long long b = r / H;
  const long long sp = (long long)y * W + x;
  const float fy = (float)y + flow[(b * 2 + 0) * HW + sp];
  const float fx = (float)x + flow[(b * 2 + 1) * HW + sp];
  const float* sb = src + b * C * HW;
  float* ob = out + b * C * HW + sp;
  if (mode == 1) {
    const int yy = (int)nearbyintf(fy), xx = (int)nearbyintf(fx);
    const bool v = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    for (int c = 0; c < C; ++c) {
      float val = v ? sb[c * HW + (long long)yy * W + xx] : 0.f;
      if (add_identity) val += sb[c * HW + sp];
      ob[c * HW] = val;
    }
    return;
  }
  const float y0f = floorf(fy), x0f = floorf(fx);
  const int y0 = (int)y0f, x0 = (int)x0f;
  const float wy1 = fy - y0f, wx1 = fx - x0f, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
  const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
  const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
  const long long o00 = (long long)y0 * W + x0;
  for (int c = 0; c < C; ++c) {
    const float* sc = sb + c * HW;
    const float v00 = (vy0 && vx0) ? sc[o00] : 0.f;
    const float v01 = (vy0 && vx1) ? sc[o00 + 1] : 0.f;
    const float v10 = (vy1 && vx0) ? sc[o00 + W] : 0.f;
    const float v11 = (vy1 && vx1) ? sc[o00 + W + 1] : 0.f;
    float val = wy0 * (wx0 * v00 + wx1 * v01) + wy1 * (wx0 * v10 + wx1 * v11);
    if (add_identity) val += sc[sp];
    ob[c * HW] = val;
  }
}

__global__ __launch_bounds__(256) void warp2d_bwd_k(const float* __restrict__ dout,
                                                    const float* __restrict__ src,
                                                    const float* __restrict__ flow,
                                                    float* __restrict__ dsrc, float* __restrict__ dflow,
                                                    int B, int C, int H, int W, int add_identity,
                                                    int flow_into_src) {
  const long long HW = (long long)H * W;
  const long long total = (long long)B * HW;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const long long r = i / W;
  const int y = (int)(r % H);
  const long long b = r / H;
  const long long sp = (long long)y * W + x;
  const float fy = (float)y + flow[(b * 2 + 0) * HW + sp];
  const float fx = (float)x + flow[(b * 2 + 1) * HW + sp];
  const float y0f = floorf(fy), x0f = floorf(fx);
  const int y0 = (int)y0f, x0 = (int)x0f;
  const float wy1 = fy - y0f, wx1 = fx - x0f, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
  const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
  const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
  const long long o00 = (long long)y0 * W + x0;
  const float* sb = src + b * C * HW;
  float* db = dsrc ? dsrc + b * C * HW : nullptr;
  const float* gb = dout + b * C * HW + sp;
  float gy = 0.f, gx = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = gb[c * HW];
    const float* sc = sb + c * HW;
    const float v00 = (vy0 && vx0) ? sc[o00] : 0.f;
    const float v01 = (vy0 && vx1) ? sc[o00 + 1] : 0.f;
    const float v10 = (vy1 && vx0) ? sc[o00 + W] : 0.f;
    const float v11 = (vy1 && vx1) ? sc[o00 + W + 1] : 0.f;
    gy += g * (wx0 * (v10 - v00) + wx1 * (v11 - v01));
    gx += g * (wy0 * (v01 - v00) + wy1 * (v11 - v10));
    if (db) {
      float* dc = db + c * HW;
      if (vy0 && vx0) atomicAdd(dc + o00, g * wy0 * wx0);
      if (vy0 && vx1) atomicAdd(dc + o00 + 1, g * wy0 * wx1);
      if (vy1 && vx0) atomicAdd(dc + o00 + W, g * wy1 * wx0);
      if (vy1 && vx1) atomicAdd(dc + o00 + W + 1, g * wy1 * wx1);
      if (add_identity) atomicAdd(dc + sp, g);
    }
  }
  if (flow_into_src) {
    atomicAdd(db + 0 * HW + sp, gy);
    atomicAdd(db + 1 * HW + sp, gx);
  } else if (dflow) {
    dflow[(b * 2 + 0) * HW + sp] = gy;
    dflow[(b * 2 + 1) * HW + sp] = gx;
  }
}

// ------------------------------------------------------------------------------------------ 3-D
__global__ __launch_bounds__(256) void warp3d_fwd_k(const float* __restrict__ src,
                                                    const float* __restrict__ flow,
                                                    float* __restrict__ out, int B, int C, int D, int H,
                                                    int W, int mode, int add_identity) {
  const long long HW = (long long)H * W, S = HW * D;
  const long long total = (long long)B * S;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  long long r = i / W;
  const int y = (int)(r % H); r /= H;
  const int z = (int)(r % D);
  const long long b = r / D;
  const long long sp = ((long long)z * H + y) * W + x;
  const float fz = (float)z + flow[(b * 3 + 0) * S + sp];
  const float fy = (float)y + flow[(b * 3 + 1) * S + sp];
  const float fx = (float)x + flow[(b * 3 + 2) * S + sp];
  const float* sb = src + b * C * S;
  float* ob = out + b * C * S + sp;
  if (mode == 1) {
    const int zz = (int)nearbyintf(fz), yy = (int)nearbyintf(fy), xx = (int)nearbyintf(fx);
    const bool v = (unsigned)zz < (unsigned)D && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    for (int c = 0; c < C; ++c) {
      float val = v ? sb[c * S + ((long long)zz * H + yy) * W + xx] : 0.f;
      if (add_identity) val += sb[c * S + sp];
      ob[c * S] = val;
    }
    return;
  }
  const float z0f = floorf(fz), y0f = floorf(fy), x0f = floorf(fx);
  const int z0 = (int)z0f, y0 = (int)y0f, x0 = (int)x0f;
  const float wz1 = fz - z0f, wy1 = fy - y0f, wx1 = fx - x0f;
  const float wz0 = 1.f - wz1, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
  const bool vz0 = (unsigned)z0 < (unsigned)D, vz1 = (unsigned)(z0 + 1) < (unsigned)D;
  const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
  const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
  const long long o000 = ((long long)z0 * H + y0) * W + x0;
  for (int c = 0; c < C; ++c) {
    const float* sc = sb + c * S;
    const float v000 = (vz0 && vy0 && vx0) ? sc[o000] : 0.f;
    const float v001 = (vz0 && vy0 && vx1) ? sc[o000 + 1] : 0.f;
    const float v010 = (vz0 && vy1 && vx0) ? sc[o000 + W] : 0.f;
    const float v011 = (vz0 && vy1 && vx1) ? sc[o000 + W + 1] : 0.f;
    const float v100 = (vz1 && vy0 && vx0) ? sc[o000 + HW] : 0.f;
    const float v101 = (vz1 && vy0 && vx1) ? sc[o000 + HW + 1] : 0.f;
    const float v110 = (vz1 && vy1 && vx0) ? sc[o000 + HW + W] : 0.f;
    const float v111 = (vz1 && vy1 && vx1) ? sc[o000 + HW + W + 1] : 0.f;
    float val = wz0 * (wy0 * (wx0 * v000 + wx1 * v001) + wy1 * (wx0 * v010 + wx1 * v011)) +
                wz1 * (wy0 * (wx0 * v100 + wx1 * v101) + wy1 * (wx0 * v110 + wx1 * v111));
    if (add_identity) val += sc[sp];
    ob[c * S] = val;
  }
}

__global__ __launch_bounds__(256) void warp3d_bwd_k(const float* __restrict__ dout,
                                                    const float* __restrict__ src,
                                                    const float* __restrict__ flow,
                                                    float* __restrict__ dsrc, float* __restrict__ dflow,
                                                    int B, int C, int D, int H, int W, int add_identity,
                                                    int flow_into_src) {
  const long long HW = (long long)H * W, S = HW * D;
  const long long total = (long long)B * S;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  long long r = i / W;
  const int y = (int)(r % H); r /= H;
  const int z = (int)(r % D);
  const long long b = r / D;
  const long long sp = ((long long)z * H + y) * W + x;
  const float fz = (float)z + flow[(b * 3 + 0) * S + sp];
  const float fy = (float)y + flow[(b * 3 + 1) * S + sp];
  const float fx = (float)x + flow[(b * 3 + 2) * S + sp];
  const float z0f = floorf(fz), y0f = floorf(fy), x0f = floorf(fx);
  const int z0 = (int)z0f, y0 = (int)y0f, x0 = (int)x0f;
  const float wz1 = fz - z0f, wy1 = fy - y0f, wx1 = fx - x0f;
  const float wz0 = 1.f - wz1, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
  const bool vz0 = (unsigned)z0 < (unsigned)D, vz1 = (unsigned)(z0 + 1) < (unsigned)D;
  const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
  const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
  const long long o000 = ((long long)z0 * H + y0) * W + x0;
  const float* sb = src + b * C * S;
  float* db = dsrc ? dsrc + b * C * S : nullptr;
  const float* gb = dout + b * C * S + sp;
  float gz = 0.f, gy = 0.f, gx = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = gb[c * S];
    const float* sc = sb + c * S;
    const float v000 = (vz0 && vy0 && vx0) ? sc[o000] : 0.f;
    const float v001 = (vz0 && vy0 && vx1) ? sc[o000 + 1] : 0.f;
    const float v010 = (vz0 && vy1 && vx0) ? sc[o000 + W] : 0.f;
    const float v011 = (vz0 && vy1 && vx1) ? sc[o000 + W + 1] : 0.f;
    const float v100 = (vz1 && vy0 && vx0) ? sc[o000 + HW] : 0.f;
    const float v101 = (vz1 && vy0 && vx1) ? sc[o000 + HW + 1] : 0.f;
    const float v110 = (vz1 && vy1 && vx0) ? sc[o000 + HW + W] : 0.f;
    const float v111 = (vz1 && vy1 && vx1) ? sc[o000 + HW + W + 1] : 0.f;
    const float p0 = wy0 * (wx0 * v000 + wx1 * v001) + wy1 * (wx0 * v010 + wx1 * v011);
    const float p1 = wy0 * (wx0 * v100 + wx1 * v101) + wy1 * (wx0 * v110 + wx1 * v111);
    gz += g * (p1 - p0);
    gy += g * (wz0 * (wx0 * (v010 - v000) + wx1 * (v011 - v001)) +
               wz1 * (wx0 * (v110 - v100) + wx1 * (v111 - v101)));
    gx += g * (wz0 * (wy0 * (v001 - v000) + wy1 * (v011 - v010)) +
               wz1 * (wy0 * (v101 - v100) + wy1 * (v111 - v110)));
    if (db) {
      float* dc = db + c * S;
      if (vz0 && vy0 && vx0) atomicAdd(dc + o000, g * wz0 * wy0 * wx0);
      if (vz0 && vy0 && vx1) atomicAdd(dc + o000 + 1, g * wz0 * wy0 * wx1);
      if (vz0 && vy1 && vx0) atomicAdd(dc + o000 + W, g * wz0 * wy1 * wx0);
      if (vz0 && vy1 && vx1) atomicAdd(dc + o000 + W + 1, g * wz0 * wy1 * wx1);
      if (vz1 && vy0 && vx0) atomicAdd(dc + o000 + HW, g * wz1 * wy0 * wx0);
      if (vz1 && vy0 && vx1) atomicAdd(dc + o000 + HW + 1, g * wz1 * wy0 * wx1);
      if (vz1 && vy1 && vx0) atomicAdd(dc + o000 + HW + W, g * wz1 * wy1 * wx0);
      if (vz1 && vy1 && vx1) atomicAdd(dc + o000 + HW + W + 1, g * wz1 * wy1 * wx1);
      if (add_identity) atomicAdd(dc + sp, g);
    }
  }
  if (flow_into_src) {
    atomicAdd(db + 0 * S + sp, gz);
    atomicAdd(db + 1 * S + sp, gy);
    atomicAdd(db + 2 * S + sp, gx);
  } else if (dflow) {
    dflow[(b * 3 + 0) * S + sp] = gz;
    dflow[(b * 3 + 1) * S + sp] = gy;
    dflow[(b * 3 + 2) * S + sp] = gx;
  }
}

// --------------------------------------------------------------------------------- resize
// F.interpolate(mode=(bi|tri)linear, align_corners=True): src = dst*(in-1)/(out-1).
__device__ __forceinline__ void lin_src(int o, float scale, int in, int& i0, int& i1, float& l1) {
  const float f = scale * (float)o;
  i0 = (int)f;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = f - (float)i0;
}
__global__ __launch_bounds__(256) void resize_fwd_k(const float* __restrict__ x, float* __restrict__ y,
                                                    int planes, int Di, int Hi, int Wi, int Do, int Ho,
                                                    int Wo, float sd, float sh, float sw, float mult) {
  const long long So = (long long)Do * Ho * Wo, Si = (long long)Di * Hi * Wi;
  const long long total = (long long)planes * So;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int ox, oy, oz;
  long long pl;
  if (total < 0x7FFFFFFFLL) {        // 32-bit coordinate decode (the 64-bit divisions cost more than the 8 taps)
    unsigned r = (unsigned)i;
    ox = (int)(r % (unsigned)Wo); r /= (unsigned)Wo;
    oy = (int)(r % (unsigned)Ho); r /= (unsigned)Ho;
    oz = (int)(r % (unsigned)Do);
    pl = r / (unsigned)Do;
  } else {
    long long r = i;
    ox = (int)(r % Wo); r /= Wo;
    oy = (int)(r % Ho); r /= Ho;
    oz = (int)(r % Do);
    pl = r / Do;
  }
  int z0, z1, y0, y1, x0, x1;
  float lz, ly, lx;
  lin_src(oz, sd, Di, z0, z1, lz);
  lin_src(oy, sh, Hi, y0, y1, ly);
  lin_src(ox, sw, Wi, x0, x1, lx);
  const float* xp = x + pl * Si;
#define XR(z_, y_, x_) xp[((long long)(z_) * Hi + (y_)) * Wi + (x_)]
  const float a0 = (1.f - lx) * XR(z0, y0, x0) + lx * XR(z0, y0, x1);
  const float a1 = (1.f - lx) * XR(z0, y1, x0) + lx * XR(z0, y1, x1);
  const float b0 = (1.f - lx) * XR(z1, y0, x0) + lx * XR(z1, y0, x1);
  const float b1 = (1.f - lx) * XR(z1, y1, x0) + lx * XR(z1, y1, x1);
#undef XR
  const float v = (1.f - lz) * ((1.f - ly) * a0 + ly * a1) + lz * ((1.f - ly) * b0 + ly * b1);
  y[i] = mult * v;
}
// Wo % 4 == 0: a thread writes 4 x-consecutive outputs (one 16-B store; the z / y terms are shared, the <= 5 distinct
// input columns of the quad come from L1) -- the x2 flow up-sampling writes 83 MB per call at 160x192x224.
// IDX = unsigned when the quad count fits 32 bits: the three 64-bit divisions of the coordinate decode were ~600 of the
// thread's ~800 instructions (66 us per call for 92 MB)
template <typename IDX>
__global__ __launch_bounds__(256) void resize_fwd_v4_k(const float* __restrict__ x, float* __restrict__ y,
                                                       int planes, int Di, int Hi, int Wi, int Do, int Ho,
                                                       int Wo, float sd, float sh, float sw, float mult) {
  const long long Si = (long long)Di * Hi * Wi;
  const int Wq = Wo >> 2;
  const long long total = (long long)planes * Do * Ho * Wq;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  IDX r = (IDX)i;
  const int oq = (int)(r % (IDX)Wq); r /= (IDX)Wq;
  const int oy = (int)(r % (IDX)Ho); r /= (IDX)Ho;
  const int oz = (int)(r % (IDX)Do);
  const long long pl = (long long)(r / (IDX)Do);
  int z0, z1, y0, y1;
  float lz, ly;
  lin_src(oz, sd, Di, z0, z1, lz);
  lin_src(oy, sh, Hi, y0, y1, ly);
  const float* xp = x + pl * Si;
  const float* r00 = xp + ((long long)z0 * Hi + y0) * Wi;
  const float* r01 = xp + ((long long)z0 * Hi + y1) * Wi;
  const float* r10 = xp + ((long long)z1 * Hi + y0) * Wi;
  const float* r11 = xp + ((long long)z1 * Hi + y1) * Wi;
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int x0, x1;
    float lx;
    lin_src(4 * oq + e, sw, Wi, x0, x1, lx);
    const float a0 = (1.f - lx) * r00[x0] + lx * r00[x1];
    const float a1 = (1.f - lx) * r01[x0] + lx * r01[x1];
    const float b0 = (1.f - lx) * r10[x0] + lx * r10[x1];
    const float b1 = (1.f - lx) * r11[x0] + lx * r11[x1];
    o[e] = mult * ((1.f - lz) * ((1.f - ly) * a0 + ly * a1) + lz * ((1.f - ly) * b0 + ly * b1));
  }
  *reinterpret_cast<float4*>(y + ((pl * Do + oz) * Ho + oy) * (long long)Wo + 4 * oq) = make_float4(o[0], o[1], o[2], o[3]);
}
// Row form of the same (Wi % 4 == 0, Wo % 4 == 0, Wi <= RF_MAXW): one WAVE per output row (plane, oz, oy).  Its four source
// rows (z0 | z1) x (y0 | y1) are copied to LDS with one 16-byte load per lane and row, then every lane interpolates four
// outputs along x from LDS -- the same expression in the same order as the kernels above (bit-identical results), but the
// 32 four-byte gathers per thread are gone: the x2 up-sampling of the flow at 160x192x224 took 82 us for 93 MB.
constexpr int RF_MAXW = 512;
__global__ __launch_bounds__(256) void resize_rows_fwd_k(const float* __restrict__ x, float* __restrict__ y,
                                                         int planes, int Di, int Hi, int Wi, int Do, int Ho,
                                                         int Wo, float sd, float sh, float sw, float mult) {
  __shared__ __attribute__((aligned(16))) float buf[4][4][RF_MAXW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long Si = (long long)Di * Hi * Wi;
  const unsigned nrow = (unsigned)planes * (unsigned)Do * (unsigned)Ho;
  const int Wiq = Wi >> 2, Woq = Wo >> 2;
  float (*b)[RF_MAXW] = buf[wv];
  for (unsigned row = blockIdx.x * 4u + (unsigned)wv; row < nrow; row += gridDim.x * 4u) {
    const unsigned rz = row / (unsigned)Ho;
    const int oy = (int)(row - rz * (unsigned)Ho), oz = (int)(rz % (unsigned)Do);
    const long long pl = (long long)(rz / (unsigned)Do);
    int z0, z1, y0, y1;
    float lz, ly;
    lin_src(oz, sd, Di, z0, z1, lz);
    lin_src(oy, sh, Hi, y0, y1, ly);
    const float* xp = x + pl * Si;
    const float* r00 = xp + ((long long)z0 * Hi + y0) * Wi;
    const float* r01 = xp + ((long long)z0 * Hi + y1) * Wi;
    const float* r10 = xp + ((long long)z1 * Hi + y0) * Wi;
    const float* r11 = xp + ((long long)z1 * Hi + y1) * Wi;
    for (int q = lane; q < Wiq; q += 64) {
      const float4 v0 = *reinterpret_cast<const float4*>(r00 + 4 * q), v1 = *reinterpret_cast<const float4*>(r01 + 4 * q);
      const float4 v2 = *reinterpret_cast<const float4*>(r10 + 4 * q), v3 = *reinterpret_cast<const float4*>(r11 + 4 * q);
      *reinterpret_cast<float4*>(&b[0][4 * q]) = v0; *reinterpret_cast<float4*>(&b[1][4 * q]) = v1;
      *reinterpret_cast<float4*>(&b[2][4 * q]) = v2; *reinterpret_cast<float4*>(&b[3][4 * q]) = v3;
    }
    // (one wave: its LDS operations complete in order; the fence keeps the compiler from moving the reads up)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* yrow = y + ((pl * Do + oz) * Ho + oy) * (long long)Wo;
    for (int oq = lane; oq < Woq; oq += 64) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int x0, x1;
        float lx;
        lin_src(4 * oq + e, sw, Wi, x0, x1, lx);
        const float a0 = (1.f - lx) * b[0][x0] + lx * b[0][x1];
        const float a1 = (1.f - lx) * b[1][x0] + lx * b[1][x1];
        const float b0 = (1.f - lx) * b[2][x0] + lx * b[2][x1];
        const float b1 = (1.f - lx) * b[3][x0] + lx * b[3][x1];
        o[e] = mult * ((1.f - lz) * ((1.f - ly) * a0 + ly * a1) + lz * ((1.f - ly) * b0 + ly * b1));
      }
      *reinterpret_cast<float4*>(yrow + 4 * oq) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}
// Adjoint in GATHER form: input sample i collects from the outputs o whose interpolation window
// touches it (o in [(i-1)/scale, (i+1)/scale]); the weight is recomputed with the forward's own
// lin_src(), so forward and backward use bit-identical coefficients.  No atomics, no pre-zeroing.
__device__ __forceinline__ void lin_range(int i, float scale, int out, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  lo = (int)floorf((float)(i - 1) / scale) - 1;
  hi = (int)ceilf((float)(i + 1) / scale) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out - 1 ? out - 1 : hi;
}
__device__ __forceinline__ float lin_weight(int o, float scale, int in, int i) {
  int i0, i1;
  float l1;
  lin_src(o, scale, in, i0, i1, l1);
  return (i0 == i ? 1.f - l1 : 0.f) + (i1 == i ? l1 : 0.f);
}
__global__ __launch_bounds__(256) void resize_bwd_k(const float* __restrict__ dy, float* __restrict__ dx,
                                                    int planes, int Di, int Hi, int Wi, int Do, int Ho,
                                                    int Wo, float sd, float sh, float sw, float mult) {
  const long long So = (long long)Do * Ho * Wo, Si = (long long)Di * Hi * Wi;
  const long long total = (long long)planes * Si;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int ix, iy, iz;
  long long pl;
  if (total < 0x7FFFFFFFLL) {
    unsigned r = (unsigned)i;
    ix = (int)(r % (unsigned)Wi); r /= (unsigned)Wi;
    iy = (int)(r % (unsigned)Hi); r /= (unsigned)Hi;
    iz = (int)(r % (unsigned)Di);
    pl = r / (unsigned)Di;
  } else {
    long long r = i;
    ix = (int)(r % Wi); r /= Wi;
    iy = (int)(r % Hi); r /= Hi;
    iz = (int)(r % Di);
    pl = r / Di;
  }
  const float* gp = dy + pl * So;
  int zl, zh, yl, yh, xl, xh;
  lin_range(iz, sd, Do, zl, zh);
  lin_range(iy, sh, Ho, yl, yh);
  lin_range(ix, sw, Wo, xl, xh);
  float acc = 0.f;
  constexpr int MW = 8;                       // windows of the x2 / x0.5 resizes of the path are 7 / 4 candidates wide
  if (zh - zl < MW && yh - yl < MW && xh - xl < MW) {
    // per-axis weights once per thread (the triple loop used to recompute lin_weight for every candidate)
    float wzv[MW], wyv[MW], wxv[MW];
#pragma unroll
    for (int t = 0; t < MW; ++t) {
      wzv[t] = (zl + t <= zh) ? lin_weight(zl + t, sd, Di, iz) : 0.f;
      wyv[t] = (yl + t <= yh) ? lin_weight(yl + t, sh, Hi, iy) : 0.f;
      wxv[t] = (xl + t <= xh) ? lin_weight(xl + t, sw, Wi, ix) : 0.f;
    }
#pragma unroll
    for (int a = 0; a < MW; ++a) {
      if (wzv[a] == 0.f) continue;
#pragma unroll
      for (int b = 0; b < MW; ++b) {
        if (wyv[b] == 0.f) continue;
        const float* row = gp + ((long long)(zl + a) * Ho + (yl + b)) * Wo + xl;
        float racc = 0.f;
#pragma unroll
        for (int c = 0; c < MW; ++c)
          if (wxv[c] != 0.f) racc += wxv[c] * row[c];
        acc += wzv[a] * wyv[b] * racc;
      }
    }
  } else {
    for (int oz = zl; oz <= zh; ++oz) {
      const float wz = lin_weight(oz, sd, Di, iz);
      if (wz == 0.f) continue;
      for (int oy = yl; oy <= yh; ++oy) {
        const float wy = lin_weight(oy, sh, Hi, iy);
        if (wy == 0.f) continue;
        const float* row = gp + ((long long)oz * Ho + oy) * Wo;
        float racc = 0.f;
        for (int ox = xl; ox <= xh; ++ox) racc += lin_weight(ox, sw, Wi, ix) * row[ox];
        acc += wz * wy * racc;
      }
    }
  }
  dx[i] = mult * acc;
}

// Separable adjoint: the trilinear weights are a product of per-axis weights, so the adjoint is three 1-D adjoints
// (W, then H, then D), each a gather of <= 4 non-zero candidates along ONE axis with the forward's own lin_src()
// coefficients.  The one-pass kernel above visits up to 7^3 candidates per input voxel (0.19 ms per call on the x2
// flow up-sampling at 160x192x224); the three passes move 82.6 + 2 (41.3 + 20.6) + 10.3 MB.
// in [outer][olen][inner] -> out [outer][ilen][inner];  V = floats per thread along `inner`
template <int V, typename IDX>
__global__ __launch_bounds__(256) void resize_axis_bwd_k(const float* __restrict__ in, float* __restrict__ out,
                                                         long long outer, int ilen, int olen, long long inner,
                                                         float scale, float mult) {
  const long long nin = inner / V;
  const long long total = outer * ilen * nin;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  // (IDX = unsigned when the thread count fits 32 bits: see resize_fwd_v4_k)
  const IDX q = (IDX)t % (IDX)nin;
  IDX r = (IDX)t / (IDX)nin;
  const int i = (int)(r % (IDX)ilen);
  const long long ou = (long long)(r / (IDX)ilen);
  int lo, hi;
  lin_range(i, scale, olen, lo, hi);
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
  const float* base = in + ou * olen * inner + (long long)q * V;
  for (int o = lo; o <= hi; ++o) {
    const float w = lin_weight(o, scale, ilen, i);
    if (w == 0.f) continue;
    const float* p = base + (long long)o * inner;
    if (V == 4) {
      const float4 v = *reinterpret_cast<const float4*>(p);
      acc[0] += w * v.x; acc[1] += w * v.y; acc[V > 2 ? 2 : 0] += w * v.z; acc[V > 3 ? 3 : 0] += w * v.w;
    } else {
      acc[0] += w * p[0];
    }
  }
  float* op = out + (ou * ilen + i) * inner + (long long)q * V;
  if (V == 4) *reinterpret_cast<float4*>(op) = make_float4(mult * acc[0], mult * acc[1], mult * acc[V > 2 ? 2 : 0], mult * acc[V > 3 ? 3 : 0]);
  else op[0] = mult * acc[0];
}

// The W pass with FOUR consecutive inputs per thread (ilen % 4 == 0): the candidate outputs of the four windows are walked
// once, in ascending order, and each adds to the (<= 2) inputs its interpolation touched -- the same weights and the same
// order of additions per input as resize_axis_bwd_k<1>, i.e. bit-identical results, for a third of the instructions (one
// lin_src() per candidate instead of one per candidate and input: that pass was bound by the vector ALUs, 50 us for 31 MB).
template <typename IDX>
__global__ __launch_bounds__(256) void resize_w_bwd4_k(const float* __restrict__ in, float* __restrict__ out,
                                                       long long rows, int ilen, int olen, float scale, float mult) {
  const IDX nin = (IDX)(ilen >> 2);
  const IDX total = (IDX)rows * nin;
  const long long tt = (long long)blockIdx.x * 256 + threadIdx.x;
  if (tt >= (long long)total) return;
  const IDX t = (IDX)tt;
  const int ib = 4 * (int)(t % nin);
  const long long row = (long long)(t / nin);
  int lo, hi, lo3, hi0;
  lin_range(ib, scale, olen, lo, hi0);
  lin_range(ib + 3, scale, olen, lo3, hi);
  lo = lo < lo3 ? lo : lo3;
  hi = hi > hi0 ? hi : hi0;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* p = in + row * olen;
  for (int o = lo; o <= hi; ++o) {
    int i0, i1;
    float l1;
    lin_src(o, scale, ilen, i0, i1, l1);
    const float g = p[o];
    const int d0 = i0 - ib, d1 = i1 - ib;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float w = (d0 == k ? 1.f - l1 : 0.f) + (d1 == k ? l1 : 0.f);
      if (w != 0.f) acc[k] += w * g;
    }
  }
  *reinterpret_cast<float4*>(out + row * ilen + ib) = make_float4(mult * acc[0], mult * acc[1], mult * acc[2], mult * acc[3]);
}

// The W pass from a TABLE: the candidate window of input i and its weights depend on i only, not on the row.  A workgroup
// builds (first candidate with a non-zero weight, RT_NC weights) for all ilen inputs in LDS once -- with lin_range() /
// lin_weight(), i.e. the same numbers as the kernels above -- and then walks rows: per input RT_NC independent loads and
// multiply-adds, zero weights skipped, ascending candidates: bit-identical to resize_axis_bwd_k<1>.  (resize_w_bwd4_k's
// candidate loop waits for one 4-byte load per trip: 75 us for the x2 adjoint at 160x192x224.)  The host takes this form
// when every window fits: floor(2 / scale) + 2 <= RT_NC.
constexpr int RT_NC = 6, RT_MAXW = 512;
__global__ __launch_bounds__(256) void resize_w_bwd_tab_k(const float* __restrict__ in, float* __restrict__ out,
                                                          unsigned rows, int ilen, int olen, float scale, float mult) {
  __shared__ float tw[RT_MAXW][RT_NC];
  __shared__ int tlo[RT_MAXW];
  for (int i = threadIdx.x; i < ilen; i += 256) {
    int lo, hi;
    lin_range(i, scale, olen, lo, hi);
    int first = hi;                                       // first candidate with a non-zero weight (hi if there is none)
    for (int o = hi; o >= lo; --o)
      if (lin_weight(o, scale, ilen, i) != 0.f) first = o;
    tlo[i] = first;
#pragma unroll
    for (int c = 0; c < RT_NC; ++c) tw[i][c] = (first + c <= hi) ? lin_weight(first + c, scale, ilen, i) : 0.f;
  }
  __syncthreads();
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (unsigned row = blockIdx.x * 4u + (unsigned)wv; row < rows; row += gridDim.x * 4u) {
    const float* p = in + (long long)row * olen;
    float* q = out + (long long)row * ilen;
    for (int i = lane; i < ilen; i += 64) {
      const int first = tlo[i];
      float g[RT_NC], w[RT_NC];
#pragma unroll
      for (int c = 0; c < RT_NC; ++c) {
        w[c] = tw[i][c];
        const int o = first + c < olen ? first + c : olen - 1;
        g[c] = p[o];
      }
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < RT_NC; ++c)
        if (w[c] != 0.f) acc += w[c] * g[c];
      q[i] = mult * acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-windowed kernels (warp_win.hip): taken whenever W % 4 == 0 in linear mode.
int df_warp_win_fwd_try(int nd, const float* src, const float* flow, float* out, int B, int C, int D, int H, int W,
                        int add_identity, hipStream_t st);
int df_warp_win_bwd_try(int nd, const float* dout, const float* src, const float* flow, float* dsrc, float* dflow,
                        int B, int C, int D, int H, int W, int add_identity, int flow_into_src, hipStream_t st);

extern "C" int dfmir_warp2d_fwd(const float* src, const float* flow, float* out, int B, int C, int H,
                                int W, int mode, int add_identity, void* stream) {
  DF_ARG_CHECK(src && flow && out && B > 0 && C > 0 && H > 0 && W > 0);
  DF_ARG_CHECK(!add_identity || C == 2);
  const long long total = (long long)B * H * W;
  if (mode == 0 && df_warp_win_fwd_try(2, src, flow, out, B, C, 1, H, W, add_identity, (hipStream_t)stream)) {
    DF_LAUNCH_CHECK();
    return 0;
  }
  warp2d_fwd_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, flow, out, B, C, H,
                                                                                W, mode, add_identity);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_warp2d_bwd(const float* dout, const float* src, const float* flow, float* dsrc,
                                float* dflow, int B, int C, int H, int W, int add_identity,
                                int flow_into_src, void* stream) {
  DF_ARG_CHECK(dout && src && flow && B > 0 && C > 0 && H > 0 && W > 0);
  DF_ARG_CHECK(!flow_into_src || (dsrc && C == 2));
  const long long total = (long long)B * H * W;
  if (df_warp_win_bwd_try(2, dout, src, flow, dsrc, dflow, B, C, 1, H, W, add_identity, flow_into_src, (hipStream_t)stream)) {
    DF_LAUNCH_CHECK();
    return 0;
  }
  warp2d_bwd_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      dout, src, flow, dsrc, dflow, B, C, H, W, add_identity, flow_into_src);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_warp3d_fwd(const float* src, const float* flow, float* out, int B, int C, int D,
                                int H, int W, int mode, int add_identity, void* stream) {
  DF_ARG_CHECK(src && flow && out && B > 0 && C > 0 && D > 0 && H > 0 && W > 0);
  DF_ARG_CHECK(!add_identity || C == 3);
  const long long total = (long long)B * D * H * W;
  if (mode == 0 && df_warp_win_fwd_try(3, src, flow, out, B, C, D, H, W, add_identity, (hipStream_t)stream)) {
    DF_LAUNCH_CHECK();
    return 0;
  }
  warp3d_fwd_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      src, flow, out, B, C, D, H, W, mode, add_identity);
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_warp3d_bwd(const float* dout, const float* src, const float* flow, float* dsrc,
                                float* dflow, int B, int C, int D, int H, int W, int add_identity,
                                int flow_into_src, void* stream) {
  DF_ARG_CHECK(dout && src && flow && B > 0 && C > 0 && D > 0 && H > 0 && W > 0);
  DF_ARG_CHECK(!flow_into_src || (dsrc && C == 3));
  const long long total = (long long)B * D * H * W;
  if (df_warp_win_bwd_try(3, dout, src, flow, dsrc, dflow, B, C, D, H, W, add_identity, flow_into_src, (hipStream_t)stream)) {
    DF_LAUNCH_CHECK();
    return 0;
  }
  warp3d_bwd_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      dout, src, flow, dsrc, dflow, B, C, D, H, W, add_identity, flow_into_src);
  DF_LAUNCH_CHECK();
  return 0;
}
long long df_warp_win_bwd_own_ws(int nd, int B, int C, int D, int H, int W);
int df_warp_win_bwd_own_try(int nd, const float* dout, const float* src, const float* flow, float* dsrc, float* dflow,
                            int B, int C, int D, int H, int W, int add_identity, int flow_into_src, float* ws,
                            hipStream_t st);
extern "C" long long dfmir_warp_bwd_own_ws_floats(int nd, int B, int C, int D, int H, int W) {
  if ((nd != 2 && nd != 3) || B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return -1;
  return df_warp_win_bwd_own_ws(nd, B, C, nd == 3 ? D : 1, H, W);
}
extern "C" int dfmir_warp_bwd_own(int nd, const float* dout, const float* src, const float* flow, float* dsrc,
                                  float* dflow, int B, int C, int D, int H, int W, int add_identity, int flow_into_src,
                                  float* ws, void* stream) {
  DF_ARG_CHECK((nd == 2 || nd == 3) && dout && src && flow && dsrc && ws && B > 0 && C > 0 && D > 0 && H > 0 && W > 0);
  DF_ARG_CHECK(!flow_into_src || C == nd);
  DF_ARG_CHECK((reinterpret_cast<uintptr_t>(ws) & 15) == 0 && df_warp_win_bwd_own_ws(nd, B, C, nd == 3 ? D : 1, H, W) > 0);
  const int rc = df_warp_win_bwd_own_try(nd, dout, src, flow, dsrc, dflow, B, C, nd == 3 ? D : 1, H, W, add_identity,
                                         flow_into_src, ws, (hipStream_t)stream);
  if (rc != 1) return df_set_error((int)hipErrorInvalidValue, __FILE__, __LINE__);
  return 0;
}
static inline float lin_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
extern "C" int dfmir_resize_fwd(const float* x, float* y, int planes, int Di, int Hi, int Wi, int Do,
                                int Ho, int Wo, float mult, void* stream) {
  DF_ARG_CHECK(x && y && planes > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0);
  const long long total = (long long)planes * Do * Ho * Wo;
  static DfOptFlag no_rows{"DFMIR_RESIZE_NO_ROWS"};          // A/B: the per-thread gather kernels
  if (!no_rows.get() && (Wo & 3) == 0 && (Wi & 3) == 0 && Wi <= RF_MAXW &&
      ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 && (long long)planes * Do * Ho < 0x7FFFFFFFLL) {
    const long long nrow = (long long)planes * Do * Ho;
    resize_rows_fwd_k<<<df_grid(nrow, 4, 256 * 20), 256, 0, (hipStream_t)stream>>>(
        x, y, planes, Di, Hi, Wi, Do, Ho, Wo, lin_scale(Di, Do), lin_scale(Hi, Ho), lin_scale(Wi, Wo), mult);
  } else if ((Wo & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && total / 4 < (1LL << 31) * 256) {
    if (total / 4 < 0xFFFFFFFFLL)
      resize_fwd_v4_k<unsigned><<<(unsigned)((total / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
          x, y, planes, Di, Hi, Wi, Do, Ho, Wo, lin_scale(Di, Do), lin_scale(Hi, Ho), lin_scale(Wi, Wo), mult);
    else
      resize_fwd_v4_k<long long><<<(unsigned)((total / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
          x, y, planes, Di, Hi, Wi, Do, Ho, Wo, lin_scale(Di, Do), lin_scale(Hi, Ho), lin_scale(Wi, Wo), mult);
  } else {
    resize_fwd_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        x, y, planes, Di, Hi, Wi, Do, Ho, Wo, lin_scale(Di, Do), lin_scale(Hi, Ho), lin_scale(Wi, Wo), mult);
  }
  DF_LAUNCH_CHECK();
  return 0;
}
// Separable form of dfmir_resize_bwd; ws holds the two intermediates (dfmir_resize_bwd_ws_floats).
extern "C" long long dfmir_resize_bwd_ws_floats(int planes, int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
  if (planes <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0) return 0;
  return (long long)planes * Do * Ho * Wi + (long long)planes * Do * Hi * Wi;
}
template <int V>
static void resize_axis_launch(const float* in, float* out, long long outer, int ilen, int olen, long long inner,
                               float scale, float mult, hipStream_t st) {
  const long long total = outer * ilen * (inner / V);
  if (total < 0xFFFFFFFFLL && inner / V < 0xFFFFFFFFLL)
    resize_axis_bwd_k<V, unsigned><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(in, out, outer, ilen, olen, inner, scale, mult);
  else
    resize_axis_bwd_k<V, long long><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(in, out, outer, ilen, olen, inner, scale, mult);
}
extern "C" int dfmir_resize_bwd_sep(const float* dy, float* dx, int planes, int Di, int Hi, int Wi, int Do,
                                    int Ho, int Wo, float mult, float* ws, void* stream) {
  DF_ARG_CHECK(dy && dx && ws && planes > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0);
  DF_ARG_CHECK((reinterpret_cast<uintptr_t>(ws) & 15) == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  float* t1 = ws;                                               // [planes][Do][Ho][Wi]
  float* t2 = ws + (((long long)planes * Do * Ho * Wi + 3) & ~3LL);   // [planes][Do][Hi][Wi]
  static DfOptFlag no_rows{"DFMIR_RESIZE_NO_ROWS"};
  const long long wrows = (long long)planes * Do * Ho;
  const float wscale = lin_scale(Wi, Wo);
  if (!no_rows.get() && Wi <= RT_MAXW && wscale > 0.f && (int)(2.f / wscale) + 2 <= RT_NC && wrows < 0x7FFFFFFFLL) {
    resize_w_bwd_tab_k<<<df_grid(wrows, 4 * 16, 256 * 8), 256, 0, st>>>(dy, t1, (unsigned)wrows, Wi, Wo, wscale, 1.f);
  } else if ((Wi & 3) == 0 && !no_rows.get()) {
    const long long total = wrows * (Wi / 4);
    if (total < 0xFFFFFFFFLL) resize_w_bwd4_k<unsigned><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(dy, t1, wrows, Wi, Wo, lin_scale(Wi, Wo), 1.f);
    else resize_w_bwd4_k<long long><<<(unsigned)((total + 255) / 256), 256, 0, st>>>(dy, t1, wrows, Wi, Wo, lin_scale(Wi, Wo), 1.f);
  } else {
    resize_axis_launch<1>(dy, t1, wrows, Wi, Wo, 1, lin_scale(Wi, Wo), 1.f, st);
  }
  if ((Wi & 3) == 0) {
    resize_axis_launch<4>(t1, t2, (long long)planes * Do, Hi, Ho, Wi, lin_scale(Hi, Ho), 1.f, st);
    resize_axis_launch<4>(t2, dx, (long long)planes, Di, Do, (long long)Hi * Wi, lin_scale(Di, Do), mult, st);
  } else {
    resize_axis_launch<1>(t1, t2, (long long)planes * Do, Hi, Ho, Wi, lin_scale(Hi, Ho), 1.f, st);
    resize_axis_launch<1>(t2, dx, (long long)planes, Di, Do, (long long)Hi * Wi, lin_scale(Di, Do), mult, st);
  }
  DF_LAUNCH_CHECK();
  return 0;
}
extern "C" int dfmir_resize_bwd(const float* dy, float* dx, int planes, int Di, int Hi, int Wi, int Do,
                                int Ho, int Wo, float mult, void* stream) {
  DF_ARG_CHECK(dy && dx && planes > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0);
  const long long total = (long long)planes * Di * Hi * Wi;
  resize_bwd_k<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      dy, dx, planes, Di, Hi, Wi, Do, Ho, Wo, lin_scale(Di, Do), lin_scale(Hi, Ho), lin_scale(Wi, Wo), mult);
  DF_LAUNCH_CHECK();
  return 0;
}

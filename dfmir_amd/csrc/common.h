// Shared device helpers for the dfmir_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "../../include/dfmir_hip.h"

#define DF_LAUNCH_CHECK()                                  \
  do {                                                     \
    hipError_t e_ = hipGetLastError();                     \
    if (e_ != hipSuccess) return df_set_error((int)e_, __FILE__, __LINE__); \
  } while (0)
#define DF_ARG_CHECK(cond)                                 \
  do {                                                     \
    if (!(cond)) return df_set_error(-1, __FILE__, __LINE__); \
  } while (0)

int df_set_error(int code, const char* file, int line);

// Library options (include/dfmir_hip.h, "Options"): PROCESS-GLOBAL A/B switches.  A value set through
// dfmir_set_option() wins; otherwise the environment variable of the same name is read.  Sites cache the parsed
// value together with the option generation, which every dfmir_set_option() call bumps.
// df_opt_get copies the current value out UNDER the table's lock (a concurrent dfmir_set_option on the same name cannot
// free it under the reader); the cached flags keep (generation, value) in ONE atomic word, so launching threads may
// race on them freely.
bool df_opt_get(const char* name, char* buf, int buf_len);   // false when unset; value truncated to buf_len - 1
int df_opt_gen();
static inline bool df_opt_on(const char* name) {             // set to anything but "" / "0"
  char b[8];
  return df_opt_get(name, b, sizeof(b)) && b[0] != 0 && !(b[0] == '0' && b[1] == 0);
}
struct DfOptFlag {                      // "is the option ON?"  (unset, "" and "0" are off)
  const char* name;
  std::atomic<long long> st{-1};        // (generation << 1) | value
  bool get() {
    const int g = df_opt_gen();
    const long long s = st.load(std::memory_order_relaxed);
    if (s >= 0 && (int)(s >> 1) == g) return (s & 1) != 0;
    const bool v = df_opt_on(name);
    st.store(((long long)g << 1) | (v ? 1 : 0), std::memory_order_relaxed);
    return v;
  }
};
struct DfOptInt {                       // integer value, `def` when unset
  const char* name;
  int def;
  std::atomic<long long> st{-1};        // (generation << 32) | (unsigned) value
  int get() {
    const int g = df_opt_gen();
    const long long s = st.load(std::memory_order_relaxed);
    if (s >= 0 && (int)(s >> 32) == g) return (int)(unsigned)(s & 0xffffffffLL);
    char b[32];
    const int v = df_opt_get(name, b, sizeof(b)) ? atoi(b) : def;
    st.store(((long long)g << 32) | (unsigned)v, std::memory_order_relaxed);
    return v;
  }
};

static inline unsigned df_grid(long long n, int bs, long long cap = 1LL << 20) {
  long long b = (n + bs - 1) / bs;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

// Compute units of the CURRENT device (persistent kernels launch one workgroup per CU); cached per device ordinal, so a
// process that drives several (possibly different) devices gets each one's own count.
static inline int df_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  cache[dev].store(v, std::memory_order_relaxed);
  return v;
}

// ---- deterministic accumulation (dfmir_det_begin / dfmir_det_end: include/dfmir_hip.h, "Deterministic weight gradients").
// Between the two calls every weight- / bias-gradient kernel launched by THIS host thread adds its partial sums as 64-bit
// FIXED-POINT integers into the scratch the caller handed over: integer addition is associative, so the result does not
// depend on the order in which workgroups finish (fp32 atomics make every weight gradient run-to-run different in its
// last bits).  fx = device pointer to {scale, 1 / scale} (a power of two such that the bound count * max|x| * max|dy| maps
// below 2^61), nullptr outside a begin / end pair -- the kernels then add floats as before.
const float* df_det_fx();
__device__ __forceinline__ void df_acc(float* base, long long idx, float v, const float* fx) {
  if (fx) atomicAdd(reinterpret_cast<unsigned long long*>(base) + idx, (unsigned long long)__float2ll_rn(v * fx[0]));
  else atomicAdd(base + idx, v);
}

// Zero a few floats on the stream with a KERNEL.  Not hipMemsetAsync: a captured step turns that into a hipGraph
// memset node, and on ROCm 7.2 such nodes are not reliably ordered with the kernel nodes around them -- the third of
// four identical masked-L1 calls in one captured graph read a stale workspace at every replay
// (scripts/graph_probe.py memset_order).
static __global__ void df_zero_small_k(float* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
static inline hipError_t df_zero_async(float* p, int n, hipStream_t st) {
  df_zero_small_k<<<(unsigned)((n + 63) / 64), 64, 0, st>>>(p, n);
  return hipGetLastError();
}

// 64-wide wavefront reductions (CDNA wave = 64 lanes).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). Result valid in every thread.
__device__ __forceinline__ float block_sum(float v, float* sm /* >= 17 floats */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  if (wid == 0) {
    float t = (lane < nw) ? sm[lane] : 0.f;
    t = wave_sum(t);
    if (lane == 0) sm[16] = t;
  }
  __syncthreads();
  return sm[16];
}
__device__ __forceinline__ float block_max(float v, float* sm) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  if (wid == 0) {
    float t = (lane < nw) ? sm[lane] : -3.0e38f;
    t = wave_max(t);
    if (lane == 0) sm[16] = t;
  }
  __syncthreads();
  return sm[16];
}
// Range probes (max |.|) travel as arrays of partial maxima: a producer writes one value per workgroup
// (plain stores, no global atomics), the consumer reduces the array in its prologue.
//   producer: smax zeroed before a barrier; at the end every wave folds its maximum into smax (LDS atomic on
//   the bit pattern: non-negative floats order like unsigned integers), one barrier, thread 0 stores.
__device__ __forceinline__ void publish_block_absmax(float m, unsigned* smax, float* out_block) {
  const float t = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(smax, __float_as_uint(t == t ? t : __uint_as_float(0x7f800000u)));
  __syncthreads();
  if (threadIdx.x == 0) *out_block = __uint_as_float(*smax);
}
//   accumulating form: DF_PROBE_SLOTS floats per tensor, zero-initialised by the host; thread 0 of every producer
//   workgroup folds its block's maximum into slot (blockIdx.x mod DF_PROBE_SLOTS) with a fire-and-forget atomic max
//   (spread over 64 addresses: ~128 per address for 8192 planes; a single address serialised them, a read-before-
//   atomic filter put an L2 round trip at the end of every short workgroup).  The consumers reduce 64 floats instead
//   of one per plane (8192 per workgroup of the conv kernels used to cost them 2-4 %).
#define DF_PROBE_SLOTS 64
__device__ __forceinline__ void publish_block_absmax_acc(float m, unsigned* smax, float* out) {
  const float t = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(smax, __float_as_uint(t == t ? t : __uint_as_float(0x7f800000u)));
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(out) + (blockIdx.x & (DF_PROBE_SLOTS - 1)), *smax);
}
//   consumer: sm = >= 17 floats of LDS scratch
__device__ __forceinline__ float reduce_absmax(const float* __restrict__ a, int n, float* sm) {
  if (n <= 1) return a[0];
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, a[i]);
  return block_max(m, sm);
}

// Input coordinate of output index o / tap t along one axis.
//   c = o*stride - pad + t ; dil>1 means the input is (virtually) zero-dilated
//   (transposed convolution == dgrad of a strided conv).
// pad_mode 0 = zero padding (returns false when outside), 1 = reflect (always inside).
__device__ __forceinline__ bool df_in_coord(int o, int t, int stride, int pad, int dil, int size,
                                            int pad_mode, int& idx) {
  int c = o * stride - pad + t;
  if (dil == 1) {
    if (pad_mode == 1) {
      if (c < 0) c = -c;
      if (c >= size) c = 2 * (size - 1) - c;
      c = c < 0 ? 0 : (c >= size ? size - 1 : c);
      idx = c;
      return true;
    }
    idx = c;
    return (unsigned)c < (unsigned)size;
  }
  if (c < 0) { idx = 0; return false; }
  int q = c / dil;
  idx = q;
  return (q * dil == c) && (q < size);
}

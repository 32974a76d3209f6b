// Micro-benchmark (GPU box): does ds_read_b128 accept 2-byte-misaligned addresses, and at what cost?
// build: hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(unsigned* out, int shift_bytes, int stride_bytes, int iters, int check) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32768];   // 64 KB of bf16-sized cells
  const int t = threadIdx.x;
  for (int i = t; i < 32768; i += 256) lds[i] = (unsigned short)i;
  __syncthreads();
  const char* base = reinterpret_cast<const char*>(lds);
  u32x4 acc = {0u, 0u, 0u, 0u};
  int off = (t & 63) * stride_bytes + shift_bytes + (t >> 6) * 8192;
  for (int it = 0; it < iters; ++it) {
    u32x4 v;
    // force a single 128-bit LDS read at a possibly misaligned address
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)(base + off) ) : "memory");
    acc += v;
    off = (off + 16) & 8191 | ((t >> 6) * 8192) | 0;
    off = ((off & ~1) | 0) + 0;
    if (shift_bytes & 2) off |= 2;
  }
  if (check) {
    u32x4 v;
    const int o2 = (t & 63) * stride_bytes + shift_bytes;
    asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)(base + o2)) : "memory");
    out[t * 4 + 0] = v[0]; out[t * 4 + 1] = v[1]; out[t * 4 + 2] = v[2]; out[t * 4 + 3] = v[3];
  } else {
    out[blockIdx.x * 256 + t] = acc[0] + acc[1] + acc[2] + acc[3];
  }
}

int main() {
  unsigned* out; hipMalloc(&out, 1 << 22);
  unsigned h[1024];
  for (int shift : {0, 2, 4, 6, 8}) {
    k<<<1, 256>>>(out, shift, 80, 1, 1);
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    // lane 1: offset 80 + shift bytes -> first cell index (80 + shift) / 2
    const unsigned want0 = (80 + shift) / 2;
    const unsigned got0 = h[4] & 0xffff, got1 = h[4] >> 16, got7 = h[7] >> 16;
    printf("shift %d B: lane1 cells %u,%u,...,%u (want %u,%u,...,%u) %s\n", shift, got0, got1, got7, want0, want0 + 1, want0 + 7,
           (got0 == want0 && got1 == want0 + 1 && got7 == want0 + 7) ? "OK" : "MISMATCH");
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int stride : {16, 80}) for (int shift : {0, 2, 4, 8}) {
    k<<<1024, 256>>>(out, shift, stride, 100, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<1024, 256>>>(out, shift, stride, 4000, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("stride %2d B shift %d B: %.3f ms  (%.2f ns per wave-read per CU-quarter)\n", stride, shift, ms, ms * 1e6 / 4000 / 4);
  }
  return 0;
}

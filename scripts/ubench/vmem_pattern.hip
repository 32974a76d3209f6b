// Micro-benchmark (GPU box): what does a wave's 16-byte-per-lane buffer load cost the CU's vector memory path as a
// function of HOW its 64 lanes are laid out over cache lines?  One 512-thread workgroup per CU (8 waves, like the 2-D
// weight-gradient kernel), every wave issues `iters` x 6 loads back to back from an L2-resident working set and waits.
//   pattern 0: 64 lanes contiguous (1 KB = 8 lines)
//   pattern 1: quads of 4 contiguous lanes (64 B), quads 16 KB apart            (16 x 64-B pieces per instruction)
//   pattern 2: every lane its own line, 16 KB apart                              (64 lines per instruction: channel per lane)
//   pattern 3: lane pairs in one 64-B segment, non-contiguous (bytes 0-15, 32-47) (32 lines, 64 pieces)
//   pattern 4: as 2, 4-byte loads (dword)
// build: hipcc --offload-arch=gfx950 -O3 -o vmem_pattern vmem_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(const float* __restrict__ src, unsigned* out, int pattern, int iters) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned HW4 = 4096u * 4u;                       // one 64 x 64 plane
  unsigned off;
  if (pattern == 0) off = (unsigned)lane * 16u;
  else if (pattern == 1) off = (unsigned)(lane >> 2) * HW4 + (unsigned)(lane & 3) * 16u;
  else if (pattern == 2 || pattern == 4) off = (unsigned)lane * HW4;
  else off = (unsigned)(lane >> 1) * HW4 + (unsigned)(lane & 1) * 32u;
  off += (unsigned)wid * 256u + (unsigned)(blockIdx.x & 7) * 1024u;      // waves / workgroups on different rows
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 64u * HW4 + 65536u, 0x00020000);
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    u32x4 v[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const unsigned o = off + (unsigned)(((it * 6 + j) & 15) * 4096);     // walk over 16 row groups: stays L2-resident (4 MB)
      if (pattern == 4) { v[j] = u32x4{__builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0), 0u, 0u, 0u}; }
      else v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) acc += v[j];
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  float* src; unsigned* out;
  hipMalloc(&src, 64u * 4096u * 4u + 65536u + 4096);
  hipMemset(src, 0, 64u * 4096u * 4u + 65536u);
  hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[5] = {"contiguous 1 KB", "quads of 64 B", "lane per line (16 B)", "lane pairs per segment", "lane per line (4 B)"};
  for (int p = 0; p < 5; ++p) {
    k<<<256, 512>>>(src, out, p, 50);
    hipDeviceSynchronize();
    const int iters = 2000;
    hipEventRecord(e0);
    k<<<256, 512>>>(src, out, p, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / (iters * 6.0 * 8.0);        // ns per wave-load per CU
    printf("pattern %d %-24s: %.3f ms, %.1f ns per wave-instruction per CU (~%.0f cycles at 2.0 GHz), %.0f GB/s useful per CU\n",
           p, names[p], ms, per, per * 2.0, (p == 4 ? 256.0 : 1024.0) / per);
  }
  return 0;
}

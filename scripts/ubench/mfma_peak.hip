// Micro-benchmark (GPU box): sustained matrix-pipe rate of v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x2_f32
// under back-to-back issue, alone and with LDS operand reads interleaved (the conv kernels' inner pattern).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: bf16 MFMA only, 1: + ds_read_b128 (12 per 24), 2: fp32 MFMA only, 3/4: bf16 + 2/4 VALU per MFMA
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2048];
  const int t = threadIdx.x;
  for (int i = t; i < 2048; i += 256) lds[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u32x4 av[6], bv[6];
  for (int i = 0; i < 6; ++i) { av[i] = lds[(t + 64 * i) & 2047]; bv[i] = lds[(t + 64 * i + 7) & 2047]; }
  float vx[4] = {1.f + t, 2.f, 3.f, 4.f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3 || MODE == 4) {
      constexpr int NV = (MODE == 3) ? 12 : 24;   // x4 accumulators -> 48 / 96 VALU per 24 MFMAs
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int a = 0; a < 4; ++a) vx[a] = vx[a] * 1.0001f + 0.5f;
    }
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i) { av[i] = lds[(t + 64 * i + it) & 2047]; bv[i] = lds[(t + 64 * i + it + 7) & 2047]; }
    }
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (MODE == 5) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[p]), __builtin_bit_cast(bf16x8, bv[(p + a) % 6]), acc[0], 0, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
          for (int q = 0; q < 1; ++q)
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av[p][a]), __uint_as_float(bv[p][a]), acc[a], 0, 0, 0);
        } else {
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[p]), __builtin_bit_cast(bf16x8, bv[(p + a) % 6]), acc[a], 0, 0, 0);
        }
      }
    if (MODE == 1) {
      __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
      for (int i = 0; i < 0; ++i) {}
    }
  }
  float s = vx[0] + vx[1] + vx[2] + vx[3];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + t] = s;
}


// Software-pipelined operand reads (the conv kernels' real pattern): the 12 ds_read_b128 of iteration i+1 are issued
// between the 24 MFMAs of iteration i.  SPACING 1: one read after each of the first 12 MFMAs (bunched);
// SPACING 2: one read after every second MFMA (even).  NRD reads per 24 MFMAs.
template <int SPACING, int NRD>
__global__ __launch_bounds__(256) void kp(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[2048];
  const int t = threadIdx.x;
  for (int i = t; i < 2048; i += 256) lds[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  u32x4 v[2][16];
  for (int i = 0; i < 16; ++i) v[0][i] = lds[(t + 64 * i) & 2047];
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < NRD; ++i) v[half ^ 1][i] = lds[(t + 64 * i + it + half) & 2047];
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int a = 0; a < 4; ++a)
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v[half][(p * 2) % NRD]),
                                                           __builtin_bit_cast(bf16x8, v[half][(p * 2 + a + 1) % NRD]), acc[a], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (SPACING == 1) { if (i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        else if (SPACING == 2) { if ((i & 1) == 1 && (i >> 1) < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        else { if (i < 24 && (i * NRD) / 24 != ((i + 1) * NRD) / 24) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + t] = s;
}
template <int SPACING, int NRD>
void runp(const char* name, int waves_per_simd) {
  float* out; hipMalloc(&out, 1 << 24);
  const int blocks = 256 * waves_per_simd, iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kp<SPACING, NRD><<<blocks, 256>>>(out, 200);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kp<SPACING, NRD><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)blocks * 4 * iters * 24;
  printf("%-44s %d wave/SIMD  %8.3f ms  %8.1f TFLOP/s  (%.2f ns per MFMA per SIMD)\n", name, waves_per_simd, ms,
         mf * 32768.0 / ms / 1e9, ms * 1e6 / ((double)iters * 24 * waves_per_simd));
  hipFree(out);
}

template <int MODE>
void run(const char* name, double flop_per_mfma, int waves_per_simd) {
  float* out; hipMalloc(&out, 1 << 24);
  const int blocks = 256 * waves_per_simd, iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 200);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)blocks * 4 * iters * 24;
  printf("%-44s %d wave/SIMD  %8.3f ms  %8.1f TFLOP/s  (%.2f ns per MFMA per SIMD)\n", name, waves_per_simd, ms,
         mf * flop_per_mfma / ms / 1e9, ms * 1e6 / ((double)iters * 24 * waves_per_simd));
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("bf16 32x32x16, registers only", 32768.0, w);
    run<1>("bf16 32x32x16 + 12 ds_read_b128 per 24", 32768.0, w);
    run<2>("fp32 32x32x2, registers only", 4096.0, w);
    run<3>("bf16 32x32x16 + 2 VALU per MFMA", 32768.0, w);
    run<4>("bf16 32x32x16 + 4 VALU per MFMA", 32768.0, w);
    run<5>("bf16 32x32x16, ONE accumulator chain", 32768.0, w);
    runp<1, 12>("pipelined 12 reads / 24, bunched 1:1", w);
    runp<2, 12>("pipelined 12 reads / 24, every 2nd MFMA", w);
    runp<1, 16>("pipelined 16 reads / 24, bunched 1:1", w);
    runp<3, 16>("pipelined 16 reads / 24, spread", w);
    runp<3, 8>("pipelined 8 reads / 24, spread", w);
  }
  return 0;
}

// Micro-benchmark (GPU box): how long a NON-matrix wave takes for a fixed instruction sequence while the other wave
// of its SIMD issues v_mfma_f32_32x32x16_f16 back to back -- the situation of the "store" wave group in the
// ping-pong conv kernels.  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run the MFMA loop (or idle),
// waves 4-7 time SEQ repetitions of a test sequence with s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_under_mfma valu_under_mfma.hip ; run: ./valu_under_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int REP = 16;   // repetitions of the 32-instruction test sequence

// MODE: 0 dependent v_add_f32 chain | 1 eight independent v_add_f32 chains | 2 independent v_pk_mul_f32
//       3 independent v_cvt_pk_f16_f32 (cvt_pkrtz) | 4 v_cvt_f32_f16 | 5 ds_write_b128 | 6 v_mov_b32 | 7 s_nop-only (s_sleep-free spin)
template <int MODE, bool MATE>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned* ticks, int mfma_iters) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[4096];
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  for (int i = t; i < 4096; i += 512) lds[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  __syncthreads();
  float s = 0.f;
  if (w < 4) {
    if (MATE) {
      f32x16 acc[4];
      for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
      u32x4 av = lds[t], bv = lds[t + 512];
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
          for (int a = 0; a < 4; ++a)
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), acc[a], 0, 0, 0);
      }
      for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    }
  } else {
    // let the MFMA waves get going
    for (int i = 0; i < 50; ++i) __builtin_amdgcn_s_sleep(10);
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.f + lane + i;
    f32x2 p2[8];
    for (int i = 0; i < 8; ++i) p2[i] = f32x2{1.f + i, 2.f + lane};
    unsigned h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32x4 wv = u32x4{(unsigned)t, 1u, 2u, 3u};
    const float one = 1.0001f;
    double d2[4] = {1.0 + lane, 2.0, 3.0, 4.0};
    const double done = 1.0001;
    const f32x2 one2 = f32x2{1.0001f, 0.9999f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < REP; ++r) {
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(one));
        else if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(one));
        else if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p2[q & 7]) : "v"(one2));
        else if (MODE == 3) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[q & 7]) : "v"(v[q & 7]), "v"(v[(q + 1) & 7]));
        else if (MODE == 4) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[q & 7]) : "v"(h[(q + 3) & 7]));
        else if (MODE == 5) { if ((q & 3) == 0) lds[2048 + ((t - 256 + 256 * (q >> 2)) & 2047)] = wv; }
        else if (MODE == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(h[q & 7]) : "v"(v[(q + 3) & 7]));
        else if (MODE == 8) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[q & 7]) : "v"(v[q & 7]), "v"(v[(q + 1) & 7]));
        else if (MODE == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2[q & 7]) : "v"(one2));
        else if (MODE == 10) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[q & 7]) : "v"(v[q & 7]), "v"(v[(q + 1) & 7]));
        else if (MODE == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2[q & 7]) : "v"(one2));
        else if (MODE == 12) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(h[q & 7]) : "v"(h[(q + 3) & 7]));
        else if (MODE == 13) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v[q & 7]) : "v"(h[(q + 3) & 7]));
        else if (MODE == 14) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d2[q & 3]) : "v"(done));
        else if (MODE == 15) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(one));
        else if (MODE == 16) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(h[q & 7]) : "v"(v[q & 7]), "v"(one), "v"(h[(q + 3) & 7]));
        else if (MODE == 17) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(h[q & 7]) : "v"(h[(q + 3) & 7]), "v"(h[(q + 5) & 7]));
        else asm volatile("s_nop 0");
      }
    }
    if (MODE == 5) asm volatile("s_waitcnt lgkmcnt(0)");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 17) ticks[w - 4] = (unsigned)(t1 - t0);
    for (int i = 0; i < 8; ++i) s += v[i] + p2[i][0] + p2[i][1] + (float)h[i] + (float)d2[i & 3];
  }
  out[blockIdx.x * 512 + t] = s;
}

template <int MODE>
void run(const char* name, int ninstr) {
  float* out; unsigned* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 64);
  unsigned h[2][4];
  for (int rep = 0; rep < 2; ++rep) {   // second launch is the measured one
    k<MODE, false><<<256, 512>>>(out, ticks, 0);
    hipMemcpy(h[0], ticks, 16, hipMemcpyDeviceToHost);
    k<MODE, true><<<256, 512>>>(out, ticks, 3000);
    hipMemcpy(h[1], ticks, 16, hipMemcpyDeviceToHost);
  }
  printf("%-34s alone %6u ticks (%5.1f / instr)   beside an MFMA wave %6u ticks (%5.1f / instr)\n", name, h[0][0],
         (double)h[0][0] / ninstr, h[1][0], (double)h[1][0] / ninstr);
  hipFree(out); hipFree(ticks);
}

int main() {
  printf("512 instructions per measurement (ds_write_b128: 128), s_memtime ticks; MFMA = v_mfma_f32_32x32x16_f16 x 4 accumulators\n");
  run<0>("v_add_f32, one dependent chain", 512);
  run<1>("v_add_f32, 8 independent chains", 512);
  run<2>("v_pk_mul_f32, 8 independent", 512);
  run<3>("v_cvt_pkrtz_f16_f32, independent", 512);
  run<4>("v_cvt_f32_f16, independent", 512);
  run<6>("v_mov_b32, independent", 512);
  run<7>("s_nop 0", 512);
  run<15>("v_mul_f32, independent", 512);
  run<9>("v_pk_add_f32, 8 independent", 512);
  run<11>("v_pk_fma_f32, 8 independent", 512);
  run<8>("v_cvt_pk_f16_f32 (RNE)", 512);
  run<10>("v_cvt_pk_bf16_f32", 512);
  run<12>("v_pk_add_f16", 512);
  run<13>("v_cvt_f32_f16 sdwa WORD_1", 512);
  run<14>("v_fma_f64, 4 independent", 512);
  run<16>("v_fma_mixlo_f16 (f32,f32,-f16)", 512);
  run<17>("v_alignbit_b32", 512);
  run<5>("ds_write_b128 (+ final wait)", 128);
  return 0;
}

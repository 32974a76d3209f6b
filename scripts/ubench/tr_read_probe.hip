#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  int h[64]; short o[256];
  // lane l supplies address of 4 contiguous shorts: element index = l * 100 (8-B aligned: 100*2 = 200 B, ok)
  for (int l = 0; l < 64; ++l) h[l] = l * 100;
  int* d; short* dout; hipMalloc(&d, 256); hipMalloc(&dout, 512);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, dout);
  hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]);
}

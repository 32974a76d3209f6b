// Shared by the fp32-MFMA (conv3x3.hip) and split-bf16 (conv3x3s.hip) 3x3 stride-1 kernels.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct Conv3P {
  int N, Cin, Cout, Hi, Wi, Ho, Wo, pad, pad_mode, act;
  float slope;
  int tiles_per_img;
};

__device__ __forceinline__ int halo_offset(int iy, int ix, int Hi, int Wi, int pad_mode) {
  if (pad_mode == 1) {
    if (iy < 0) iy = -iy;
    if (iy >= Hi) iy = 2 * (Hi - 1) - iy;
    if (ix < 0) ix = -ix;
    if (ix >= Wi) ix = 2 * (Wi - 1) - ix;
    iy = iy < 0 ? 0 : (iy >= Hi ? Hi - 1 : iy);
    ix = ix < 0 ? 0 : (ix >= Wi ? Wi - 1 : ix);
    return iy * Wi + ix;
  }
  return ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) ? iy * Wi + ix : -1;
}

// largest halo patch (positions) any BN-pixel run of a Wo-wide, HWo-pixel image can need
static inline int worst_npos(int Wo, int HWo, int BN) {
  if (HWo <= BN) {  // one tile per image
    const int rows = (HWo + Wo - 1) / Wo;
    return rows == 1 ? 3 * (HWo + 2) : (rows + 2) * (Wo + 2);
  }
  if (Wo >= BN) return (Wo % BN == 0) ? 3 * (BN + 2) : 4 * (Wo + 2);
  const int rows = (BN % Wo == 0) ? BN / Wo : (BN + Wo - 1) / Wo + 1;
  return (rows + 2) * (Wo + 2);
}

// Packed-weight buffer: [T][K][M] fp32, rounded up to 4 floats, then -- for 9-tap kernels -- the
// split-bf16 section [ceil(K/8)][3 splits][9 taps][M] x (8 bf16 = 16 B)   (K = reduction channels,
// M = produced channels: Cin/Cout forward, Cout/Cin for the dgrad packing).
static inline long long df_pack_tcc_floats(long long K, long long M, int T) { return ((long long)T * K * M + 3) & ~3LL; }
static inline long long df_pack_split_floats(long long K, long long M, int T) {
  return T == 9 ? ((K + 7) / 8) * 27 * M * 4 : 0;
}

// One job of the batched weight packing (dfmir_weight_pack_batch): device-side view with the derived pointers.
struct DfPackJobDev {
  const float* w;      // [Cout][Cin][T]
  float* o;            // packed buffer: [T][K][M] fp32 first
  float* part;         // probe slots (partial maxima of |w|) or NULL
  float* sec;          // split section (16-B units) or NULL
  float* trailer;      // split trailer or NULL
  int Cout, Cin, T, mode;
  int nblk;            // workgroups of the pack pass = number of partial maxima
  int nsplit;          // workgroups of the split pass (0: no split section)
};
int df_weight_split_batch_launch(const DfPackJobDev* jobs_dev, int njobs, int max_nsplit, hipStream_t st);
void df_weight_split_fill(DfPackJobDev* j);   // fills part / sec / trailer / nsplit for a T == 9 job


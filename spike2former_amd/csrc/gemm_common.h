// Types and small device helpers shared by the GEMM translation units (gemm.hip, gemm_bf16.hip, gemm_dx.hip).
#pragma once
#include "s2f_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ unsigned short s2f_f2bf(float f) {      // round-to-nearest-even fp32 -> bf16 (finite inputs)
  unsigned int u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float s2f_bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }

// (v0, v1) -> three packed bf16 pairs hi, mid, lo with v = hi + mid + lo to 24 bits; v_cvt_pk_bf16_f32 rounds to nearest
// even, the residuals are exact fp32 subtractions.
__device__ __forceinline__ unsigned int s2f_pack2(f32x2 v) {
  bf16x2 b = __builtin_convertvector(v, bf16x2);
  return *reinterpret_cast<unsigned int*>(&b);
}
__device__ __forceinline__ void s2f_split3x2(float v0, float v1, unsigned int& h, unsigned int& m, unsigned int& l) {
  h = s2f_pack2(f32x2{v0, v1});
  const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
  m = s2f_pack2(f32x2{r0, r1});
  l = s2f_pack2(f32x2{r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u)});
}

// Geometry of the implicit 3x3 convolution mode (stride 1, padding 1): the B operand is the activation [C][H][W] itself and
// "row k, column n" of the virtual im2col matrix is  x[c][y + ky - 1][x + kx - 1],  n = y W + x, with the TAP-MAJOR row order
// k = (3 ky + kx) C + c  (C % 32 == 0: every 32-wide contraction step lies inside one tap).
struct Conv3 {
  int H, W, C;
};
// pixel index -> (row, column) of a W-wide map; log_w >= 0: W = 2^log_w (shift / mask), log_w < 0: any width (one division)
__device__ __forceinline__ int conv3_row(int l, int W, int log_w) { return log_w >= 0 ? l >> log_w : l / W; }
__device__ __forceinline__ int conv3_col(int l, int W, int log_w) { return log_w >= 0 ? l & (W - 1) : l - (l / W) * W; }

// LDS transpose read (gfx950 ds_read_b64_tr_b16): within a 16-lane group lane i supplies the 8-byte-aligned address of
// 4 consecutive 16-bit elements -- row (i >> 2), columns 4 (i & 3) .. +3 of a 4 x 16 block when addressed as below -- and
// receives column i of that block: element j = block[j][i]  (verified on hardware by tools/micro/tr_probe.hip).
__device__ __forceinline__ s16x4 s2f_lds_tr16(const unsigned short* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}

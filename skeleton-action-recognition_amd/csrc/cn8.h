// cn8.h -- the bf16 activation layout of the bf16 configuration and its register helpers.
//
// "CN8": an activation with C channels over n columns (column = (b*T + t)*V + v as in the fp32 CN layout) is stored as
// G = ceil(C/8) planes of n 16-byte UNITS; unit (g, col) holds the 8 bfloat16 channels 8g .. 8g+7 of that column
// (channels >= C are zero).  Why this shape on MI355X:
//   * it IS the k-innermost LDS operand image of v_mfma_f32_32x32x16_bf16 (8 consecutive k = channels per lane), so
//     the conv stagers copy HBM -> LDS in 16-byte pieces with no transposition, and a temporal tap is a shift by
//     whole units whatever the parity of V = 25;
//   * an MFMA accumulator holds 4 consecutive channels of one column in registers 4g .. 4g+3 of a lane: the epilogue
//     stores 8 bytes per lane and half-unit, 512 contiguous bytes per wave instruction;
//   * the weight-gradient kernels (contraction over columns) read the same image through ds_read_b64_tr_b16;
//   * every element-wise pass moves 16 bytes per lane.
#pragma once
#include "sar_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// bfloat16 -> float is a 16-bit shift; element 2i is the low half of dword i
__device__ __forceinline__ void cn8_unpack(const uint4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16);
  f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16);
  f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16);
  f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16);
  f[7] = __uint_as_float(u.w & 0xffff0000u);
}

__device__ __forceinline__ void cn8_unpack4(const uint2& u, float (&f)[4]) {
  f[0] = __uint_as_float(u.x << 16);
  f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16);
  f[3] = __uint_as_float(u.y & 0xffff0000u);
}

// float -> bfloat16, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned cn8_pack2(float lo, float hi) {
  bf16x2 p;
  p[0] = (__bf16)lo;
  p[1] = (__bf16)hi;
  return *reinterpret_cast<unsigned*>(&p);
}

__device__ __forceinline__ uint4 cn8_pack(const float (&f)[8]) {
  return make_uint4(cn8_pack2(f[0], f[1]), cn8_pack2(f[2], f[3]), cn8_pack2(f[4], f[5]), cn8_pack2(f[6], f[7]));
}

__device__ __forceinline__ uint2 cn8_pack4(const float (&f)[4]) {
  return make_uint2(cn8_pack2(f[0], f[1]), cn8_pack2(f[2], f[3]));
}

// per-channel parameter vector of a unit's 8 channels (0 beyond C)
__device__ __forceinline__ void cn8_params(const float* __restrict__ p, int g, int C, float fill, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (p && 8 * g + j < C) ? p[8 * g + j] : fill;
}

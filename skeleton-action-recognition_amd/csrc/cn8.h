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

// max(x, 0) of two packed bfloat16: the sign bit is the int16 sign, so one v_pk_max_i16 clears the negative halves
// (-0 included).  ReLU commutes with the rounding (round-to-nearest keeps the sign), so relu(bf16(x)) == bf16(relu(x)).
typedef short cn8_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cn8_relu2(unsigned p) {
  cn8_s16x2 v = *reinterpret_cast<cn8_s16x2*>(&p);
  const cn8_s16x2 z = {0, 0};
  v = __builtin_elementwise_max(v, z);
  return *reinterpret_cast<unsigned*>(&v);
}

// The folded BatchNorm (+ ReLU) of a consumer's operand staging on one unit: 8 shifts / masks, 8 fma, 4 v_cvt_pk_bf16_f32,
// 4 v_pk_max_i16, 4 v_and (keep = 0 forces an exact zero: temporal padding, columns outside the sequence).
__device__ __forceinline__ uint4 cn8_bn_relu_unit(const uint4& u, const float (&sc)[8], const float (&sh)[8], bool relu,
                                                  unsigned keep) {
  float f[8];
  cn8_unpack(u, f);
#pragma unroll
  for (int q = 0; q < 8; ++q) f[q] = fmaf(f[q], sc[q], sh[q]);
  unsigned p[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) p[q] = cn8_pack2(f[2 * q], f[2 * q + 1]);
  if (relu) {
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = cn8_relu2(p[q]);
  }
  return make_uint4(p[0] & keep, p[1] & keep, p[2] & keep, p[3] & keep);
}

// scale / shift of the 8 channels cb .. cb+7 (0 beyond C).  Whole groups are fetched as two 16-byte scalar loads per
// vector (the address is wave-uniform); a ragged last group falls back to guarded element loads.
typedef float cn8_f32x4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void cn8_params8(const float* __restrict__ p, int cb, int C, float (&o)[8]) {
  if (cb + 8 <= C) {
    const cn8_f32x4u a = *reinterpret_cast<const cn8_f32x4u*>(p + cb), b = *reinterpret_cast<const cn8_f32x4u*>(p + cb + 4);
    o[0] = a[0], o[1] = a[1], o[2] = a[2], o[3] = a[3], o[4] = b[0], o[5] = b[1], o[6] = b[2], o[7] = b[3];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (cb + j < C) ? p[cb + j] : 0.f;
  }
}

// per-channel parameter vector of a unit's 8 channels (0 beyond C)
__device__ __forceinline__ void cn8_params(const float* __restrict__ p, int g, int C, float fill, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (p && 8 * g + j < C) ? p[8 * g + j] : fill;
}

// split_terms.h -- the term arithmetic of the "split" kernels (conv_gemm_split.hip, conv2d_split.hip): which terms an fp32 operand is
// written as, which products are issued and in which order, the power-of-two scale of an fp16 operand from its bound, and the
// split of 8 consecutive-channel values into 16-byte MFMA units.  Include inside the unit's anonymous namespace (after sar_common.h).
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int AR_B1 = SAR_SPLIT_BF16X1, AR_B3 = SAR_SPLIT_BF16X3, AR_B6 = SAR_SPLIT_BF16X6, AR_B9 = SAR_SPLIT_BF16X9,
              AR_H3 = SAR_SPLIT_F16X3, AR_H3S = SAR_SPLIT_F16X3S, AR_H3A = SAR_SPLIT_F16X3A;
constexpr float H3_LO = 2048.f;   // f16x3 (two accumulators): the second term carries 2^11

constexpr bool ar_f16(int ar) { return ar == AR_H3 || ar == AR_H3S || ar == AR_H3A; }
constexpr bool ar_two_acc(int ar) { return ar == AR_H3; }
// terms of the W-side operand / of the source-side operand.  f16x3a (the product arithmetic) is ASYMMETRIC: the well-conditioned
// operand (the weights: max / typical magnitude ~ 4) carries THREE images -- w0 = fp16(s w), w1 = fp16(s w - w0), w0 2^-11 -- and
// the wide-range operand (activations, gradients) two -- x0 = fp16(s x), x1' = fp16((s x - x0) 2^11) -- so that the products
// w0 x0 + w1 x0 + (w0 2^-11)(x1' 2^11 ...) = w0 x0 + w1 x0 + w0 x1 meet in ONE accumulator and the low term of the wide operand is
// never an fp16 subnormal: 22 significant bits for every source element within 2^-29 of the tensor's bound (f16x3s: 2^-18 --
// a gradient tensor with a few outliers lost its second term on most elements: 1e-4 errors in the whole-model parity test).
constexpr int ar_nta(int ar) { return ar == AR_B1 ? 1 : ((ar == AR_B3 || ar == AR_H3 || ar == AR_H3S) ? 2 : 3); }
constexpr int ar_ntb(int ar) { return ar == AR_B1 ? 1 : ((ar == AR_B3 || ar_f16(ar)) ? 2 : 3); }
constexpr int ar_nprod(int ar) { return ar == AR_B1 ? 1 : (ar == AR_B3 || ar_f16(ar)) ? 3 : (ar == AR_B6 ? 6 : 9); }
// product p of an arithmetic: (W term, src term), smallest magnitude first
constexpr int ar_pi(int ar, int p) {
  constexpr int i9[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
  return ar == AR_H3A ? 2 - p : ar == AR_B1 ? 0 : (ar == AR_B3 || ar_f16(ar)) ? (p == 1 ? 1 : 0) : i9[p + 9 - ar_nprod(ar)];
}
constexpr int ar_pj(int ar, int p) {
  constexpr int j9[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
  return ar == AR_H3A ? (p == 0 ? 1 : 0) : ar == AR_B1 ? 0 : (ar == AR_B3 || ar_f16(ar)) ? (p == 0 ? 1 : 0) : j9[p + 9 - ar_nprod(ar)];
}

#include "split_scale.h"   // scale_exp, bound_nonfinite, split_unscale

__device__ __forceinline__ unsigned pk_bf16(float x, float y) {
  bf16x2 p;
  p[0] = (__bf16)x;
  p[1] = (__bf16)y;
  return *reinterpret_cast<unsigned*>(&p);
}
__device__ __forceinline__ unsigned pk_f16(float x, float y) {
  f16x2 p;
  p[0] = (_Float16)x;
  p[1] = (_Float16)y;
  return *reinterpret_cast<unsigned*>(&p);
}

// f16x3a: the conditioned operand's third image is (first image) x 2^-11.  The kernels neither move it to LDS nor read it: a fragment of
// it is four v_pk_mul_f16 on the fragment of the first image -- an exact power-of-two scaling with the fp16 rounding the pack kernel
// applies -- which removes a third of the weight DMA pieces and a fifth of the k-steps' LDS reads (round 6; bit-identical results).
constexpr int nta_lds(int ar) { return ar == AR_H3A ? ar_nta(ar) - 1 : ar_nta(ar); }
__device__ __forceinline__ uint4 third_image(const uint4& a0) {
  const f16x8 w0 = *reinterpret_cast<const f16x8*>(&a0);
  const f16x8 w2 = w0 * (_Float16)(1.f / H3_LO);
  return *reinterpret_cast<const uint4*>(&w2);
}

// 8 consecutive-channel values of one row / column -> the term units (k-innermost: element j of a unit = channel j).  WSIDE: the
// W-side (well-conditioned) operand's images, else the source side's.
template <int AR, bool WSIDE>
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&u)[WSIDE ? ar_nta(AR) : ar_ntb(AR)], float scale) {
  constexpr int NT = WSIDE ? ar_nta(AR) : ar_ntb(AR);
  unsigned w[NT][4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float x = v[2 * p], y = v[2 * p + 1];
    if constexpr (ar_f16(AR)) {
      x = __builtin_amdgcn_fmed3f(x * scale, -65504.f, 65504.f);   // scale == 1 where the caller folded it into the prologue
      y = __builtin_amdgcn_fmed3f(y * scale, -65504.f, 65504.f);
      f16x2 h;
      h[0] = (_Float16)x;
      h[1] = (_Float16)y;
      w[0][p] = *reinterpret_cast<unsigned*>(&h);
      // the remainder: up-scaled by 2^11 on the source side of f16x3a and in f16x3 (two accumulators)
      const float lo = (ar_two_acc(AR) || (AR == AR_H3A && !WSIDE)) ? H3_LO : 1.f;
      w[1][p] = pk_f16((x - (float)h[0]) * lo, (y - (float)h[1]) * lo);
      if constexpr (AR == AR_H3A && WSIDE) w[2][p] = pk_f16((float)h[0] * (1.f / H3_LO), (float)h[1] * (1.f / H3_LO));
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const unsigned b = pk_bf16(x, y);
        w[t][p] = b;
        if (t + 1 < NT) {   // the remainder is exact in fp32
          x -= __uint_as_float(b << 16);
          y -= __uint_as_float(b & 0xffff0000u);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) u[t] = make_uint4(w[t][0], w[t][1], w[t][2], w[t][3]);
}

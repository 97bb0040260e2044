// split_scale.h -- operand scales of the fp16 split arithmetic from the bound cells (include/sar_hip.h: cells hold the BITS of a
// non-negative float), shared by conv_gemm_split.hip / conv2d_split.hip (through split_terms.h) and conv_wgrad_split.hip.
#pragma once

// power-of-two scale exponent of an operand from (the bits of) an upper bound of its magnitudes: bound * 2^e in [2^14, 2^15)
__host__ __device__ __forceinline__ int scale_exp(unsigned bound_bits) {
  const int fl = (int)((bound_bits >> 23) & 0xffu) - 127;
  const int e = 14 - fl;
  return e > 100 ? 100 : (e < -100 ? -100 : e);
}

// A cell that holds the bits of Inf or NaN: the operand tensor -- or the statistics its bound was formed from -- is not finite (the
// producers raise the cells by UNSIGNED maxima of the value bits, in which NaN > Inf > every finite magnitude).  The operand clamp
// (+-65504 behind the scale) would turn such an operand into finite fp16 terms and a diverged step into finite garbage; instead the
// factor that undoes the operand scales in the epilogue becomes NaN and every output of the launch is NaN -- what the fp32
// kernels and the reference's framework ops propagate from a non-finite operand through a contraction.
__host__ __device__ __forceinline__ bool bound_nonfinite(unsigned bound_bits) { return bound_bits >= 0x7f800000u; }

__device__ __forceinline__ float split_unscale(int ea, int eb, unsigned bound_a, unsigned bound_b) {
  return (bound_nonfinite(bound_a) || bound_nonfinite(bound_b)) ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + eb));
}

// box_probe.hip -- what THIS box sustains (bench.py "box" object): dense matrix-instruction loops on the
// three pipes the conv kernels use (v_mfma_f32_32x32x2_f32, v_mfma_f32_32x32x16_f16, v_mfma_f32_32x32x16_bf16) with the shader clock
// the chip held under each (s_memtime cycles over s_memrealtime 100 MHz ticks, per workgroup), and a float4 copy.  The peaks of
// MI355X_MICROARCH.md stay the denominators of every `frac`; `frac_of_box` divides by what these loops reached on the same box,
// so a slow box (boxes differ by ~10 %) reads as a slow box and not as a regression.  Nothing in the product path calls these.
#include "sar_common.h"

typedef _Float16 bp_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bp_bf16x8 __attribute__((ext_vector_type(8)));

// Four independent accumulator chains per wave (the dependent-issue latency of a 32x32 MFMA is covered by the other three),
// four waves per workgroup, two workgroups per CU: the densest issue the matrix pipe accepts.  Operands are loop-invariant
// registers: no LDS, no memory -- the number is the pipe's, at the clock the power management holds under it.
template <int KIND>
__global__ __launch_bounds__(256, 2) void box_mfma_kernel(float* __restrict__ sink, uint32_t* __restrict__ clocks, int iters) {
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_amdgcn_s_memtime();
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float fa = (float)(threadIdx.x & 7) * 0.125f, fb = (float)(blockIdx.x & 3) * 0.25f;
  if constexpr (KIND == 0) {
    const float a0 = fa, a1 = fa + 1.f, b0 = fb, b1 = fb - 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
      }
    }
  } else if constexpr (KIND == 1) {
    bp_f16x8 a0, a1, b0, b1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a0[q] = (_Float16)(fa + q * 0.01f), a1[q] = (_Float16)(fa - q * 0.01f);
      b0[q] = (_Float16)(fb + q * 0.02f), b1[q] = (_Float16)(fb - q * 0.02f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
      }
    }
  } else {
    bp_bf16x8 a0, a1, b0, b1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a0[q] = (__bf16)(fa + q * 0.01f), a1[q] = (__bf16)(fa - q * 0.01f);
      b0[q] = (__bf16)(fb + q * 0.02f), b1[q] = (__bf16)(fb - q * 0.02f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) r += acc[j][q];
  sink[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
  if (clocks != nullptr && threadIdx.x == 0) {
    clocks[2 * blockIdx.x] = (uint32_t)(__builtin_amdgcn_s_memtime() - ct0);          // shader-clock cycles of this workgroup
    clocks[2 * blockIdx.x + 1] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt0);  // the same span in 100 MHz ticks
  }
}

// four float4 units per thread, all loads issued before the first store (the loop shape the streaming kernels of
// elementwise.hip use): 2 x n x 4 bytes of HBM traffic per launch
__global__ __launch_bounds__(256) void box_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  const int64_t i0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  float4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (i0 + u * 256 < n4) v[u] = src[i0 + u * 256];
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (i0 + u * 256 < n4) dst[i0 + u * 256] = v[u];
}

extern "C" int sar_box_mfma(int kind, int blocks, int iters, float* sink, uint32_t* clocks, sar_stream_t s) {
  SAR_REQUIRE(kind >= 0 && kind <= 2 && blocks > 0 && iters > 0 && sink != nullptr, "sar_box_mfma: bad arguments");
  hipStream_t st = (hipStream_t)s;
  if (kind == 0) box_mfma_kernel<0><<<blocks, 256, 0, st>>>(sink, clocks, iters);
  else if (kind == 1) box_mfma_kernel<1><<<blocks, 256, 0, st>>>(sink, clocks, iters);
  else box_mfma_kernel<2><<<blocks, 256, 0, st>>>(sink, clocks, iters);
  SAR_LAUNCH_CHECK("sar_box_mfma");
  return 0;
}

extern "C" int64_t sar_box_mfma_flops(int kind, int blocks, int iters) {
  // 4 waves x iters x 8 x 4 instructions, each 32 x 32 x K multiply-adds
  return (int64_t)blocks * 4 * iters * 32 * (2LL * 32 * 32 * (kind == 0 ? 2 : 16));
}

extern "C" int sar_box_copy_f32(const float* src, float* dst, int64_t n, sar_stream_t s) {
  SAR_REQUIRE(src != nullptr && dst != nullptr && n > 0 && n % 4 == 0, "sar_box_copy_f32: n must be a positive multiple of 4");
  SAR_REQUIRE(((uintptr_t)src | (uintptr_t)dst) % 16 == 0, "sar_box_copy_f32: pointers must be 16-byte aligned");
  const int64_t n4 = n / 4;
  box_copy_kernel<<<(unsigned)((n4 + 1023) / 1024), 256, 0, (hipStream_t)s>>>(reinterpret_cast<const float4*>(src),
                                                                                reinterpret_cast<float4*>(dst), n4);
  SAR_LAUNCH_CHECK("sar_box_copy_f32");
  return 0;
}

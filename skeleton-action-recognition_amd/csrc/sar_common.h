// Shared host/device helpers for libsar_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sar_hip.h"

void sar_set_error(const char* fmt, ...);

// the caller-owned launch context of include/sar_hip.h: side streams + fork / join events on ONE device
struct sar_context {
  int device;
  int nstreams;            // 3, or 0 when SAR_C2D_PARITY_STREAMS=0 disabled the fan-out at creation
  hipStream_t s[3];
  hipEvent_t fork, join[3];
};

#define SAR_REQUIRE(cond, ...)                  \
  do {                                          \
    if (!(cond)) {                              \
      sar_set_error(__VA_ARGS__);               \
      return SAR_E_ARG;                         \
    }                                           \
  } while (0)

#define SAR_LAUNCH_CHECK(name)                                        \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) {                                          \
      sar_set_error("%s: %s", name, hipGetErrorString(e__));          \
      return (int)e__;                                                \
    }                                                                 \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// MFMA 32x32x2 f32 fragment maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&31][k = l>>5];  B: lane l holds B[k = l>>5][j = l&31]
//   C/D: col j = l&31, row i = (reg&3) + 8*(reg>>2) + 4*(l>>5)
__device__ __forceinline__ int mfma_row(int reg, int hi) { return (reg & 3) + 8 * (reg >> 2) + 4 * hi; }

__device__ __forceinline__ int floordiv(int a, int b) {  // b > 0
  int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}

// sum over the 32 lanes of each half-wave (xor masks < 32 never cross the halves)
__device__ __forceinline__ float half_wave_sum(float x) {
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 8);
  x += __shfl_xor(x, 4);
  x += __shfl_xor(x, 2);
  x += __shfl_xor(x, 1);
  return x;
}

__device__ __forceinline__ float wave_sum(float x) {
  x = half_wave_sum(x);
  x += __shfl_xor(x, 32);
  return x;
}

__device__ __forceinline__ double wave_sum_d(double x) {
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
  return x;
}

static inline hipStream_t as_stream(sar_stream_t s) { return (hipStream_t)s; }

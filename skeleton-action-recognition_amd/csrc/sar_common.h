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

inline bool aux_even_frames(const sar_conv_desc& d) { return d.mode == SAR_CONV_GRAPH && (d.g_flags & SAR_GRAPH_AUX_EVEN_FRAMES) != 0; }

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

// The same total in EVERY lane without the LDS crossbar: __shfl_xor is ds_bpermute_b32 (a dependent LDS round trip per step, six
// per sum: the 48 bias sums of a graph weight-gradient workgroup took ~23 000 cycles, tools/stamps8.sh); the butterfly inside a
// row of 16 lanes is four DPP adds, the four row totals are read with v_readlane.  Summation order differs from wave_sum.
template <int CTRL>
__device__ __forceinline__ float dpp_lanes(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float x) {
  x += dpp_lanes<0xB1>(x);    // quad_perm [1, 0, 3, 2]
  x += dpp_lanes<0x4E>(x);    // quad_perm [2, 3, 0, 1]
  x += dpp_lanes<0x141>(x);   // row_half_mirror: lane i <-> 7 - i of each 8
  x += dpp_lanes<0x140>(x);   // row_mirror: lane i <-> 15 - i of each 16
  const int xi = __float_as_int(x);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(xi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(xi, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(xi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(xi, 48));
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ double wave_sum_d(double x) {
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
  return x;
}

// ---- BatchNorm-backward "tail" of the reduce kernels (include/sar_hip.h: sar_bn_tail)
// The hand-off follows MI355X_MICROARCH.md "inter-workgroup visibility", first row of the sc1 table: the partials are stored
// with agent-scope (sc1) stores by lanes of ONE wave, that wave waits for them (vmcnt(0)), then ONE lane adds to the channel's
// counter; the workgroup whose add came last reads every partial with agent-scope (sc1) loads.  No L2 write-back, no
// invalidate (an agent-scope release fence per workgroup made the fp32 step 10 % slower: each one flushes the XCD's L2).
__device__ __forceinline__ void bn_tail_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float bn_tail_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Called by ONE lane of the wave that stored the partials, after them: true for the last workgroup of `total`.
__device__ __forceinline__ bool bn_tail_last_arriver(int32_t* ticket, int total) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's sc1 stores have completed
  const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old != total - 1) return false;
  __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
  return true;
}
// One wave: the fp64 sums of one channel's partials (float4 rows: sum dz, sum dz xhat-numerator of u, the same of r) in a fixed
// order -- lane l adds rows l, l + 64, .., then the xor tree -- and, on lane 0, what sar_bn_bwd_finalize_f32(centered) writes.
__device__ __forceinline__ void bn_tail_channel(const float* __restrict__ partials, int nparts, int c, const sar_bn_tail& t,
                                                const float* __restrict__ mu_p, const float* __restrict__ mr_p, bool has_r) {
  const int lane = threadIdx.x & 63;
  const float* p = partials + (int64_t)c * nparts * 4;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int i = lane; i < nparts; i += 64) {
    s1 += (double)bn_tail_load(p + 4 * i);
    s2 += (double)bn_tail_load(p + 4 * i + 1);
    if (has_r) s3 += (double)bn_tail_load(p + 4 * i + 2);
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  s3 = wave_sum_d(s3);
  if (lane == 0) {
    {
      const double m = mu_p ? (double)mu_p[c] : 0.0, rs = t.rstd[c], g = t.gamma ? (double)t.gamma[c] : 1.0;
      const double dg = rs * s2, a = s1 / t.count, b = dg / t.count;
      if (t.dgamma) t.dgamma[c] = (float)dg;
      if (t.dbeta) t.dbeta[c] = (float)s1;
      t.k1[c] = (float)(g * rs);
      t.k2[c] = (float)(-g * rs * rs * b);
      t.k3[c] = (float)(g * rs * (m * rs * b - a));
    }
    if (has_r) {
      const double m = mr_p ? (double)mr_p[c] : 0.0, rs = t.rrstd[c], g = t.rgamma ? (double)t.rgamma[c] : 1.0;
      const double dg = rs * s3, a = s1 / t.count, b = dg / t.count;
      if (t.rdgamma) t.rdgamma[c] = (float)dg;
      if (t.rdbeta) t.rdbeta[c] = (float)s1;
      t.rk1[c] = (float)(g * rs);
      t.rk2[c] = (float)(-g * rs * rs * b);
      t.rk3[c] = (float)(g * rs * (m * rs * b - a));
    }
  }
}

static inline hipStream_t as_stream(sar_stream_t s) { return (hipStream_t)s; }

// ---- LDS overlay / reuse checks (make ldsdebug -> sar_amd/libsar_hip_ldsdebug.so, tests/test_gpu_lds_overlays.py).
// Every LDS region of the conv kernels is written, read and REWRITTEN (next stage, per-row parameters, the epilogue's
// transpose area); each rewrite is ordered behind the last read by a barrier that the comments at the rewrite name.  A missing
// barrier is a race that ordinary runs almost never lose (the round-3 bias-row race of conv_gemm.hip: 1 launch in 1 000).  In
// a -DSAR_DEBUG_LDS build ONE wave of every workgroup (a different one from workgroup to workgroup) sleeps ~32 000 cycles in
// front of each of its last-read sites: if no barrier holds the rewriting waves back they are several stages ahead by then and
// the sleeper reads the wrong stage's operands -- a deterministic parity failure instead of a rare one.  The build with
// -DSAR_DEBUG_LDS_DROP_BIAS_BARRIER (the round-3 bug put back) is kept to show that the instrument fires.
#ifdef SAR_DEBUG_LDS
#define SAR_LDS_SKEW()                                                        \
  do {                                                                        \
    if (((threadIdx.x >> 6) & 3) == ((blockIdx.x + (blockIdx.x >> 3)) & 3)) { \
      asm volatile("s_sleep 127\n\ts_sleep 127\n\ts_sleep 127\n\ts_sleep 127" ::: "memory"); /* no LDS read may move across */ \
    }                                                                         \
  } while (0)
#else
#define SAR_LDS_SKEW() \
  do {                 \
  } while (0)
#endif

// elementwise_cn8.hip -- the HBM-bound passes of the ST-GCN step on bf16 CN8 activations (cn8.h): block tail
// (BatchNorm + residual + ReLU, models/stgcn.py:37,62-63) forward / backward, BatchNorm-backward apply, pooling.
// One thread = one 16-byte unit (8 channels of one column): every access is a coalesced 1 KiB per wave instruction, the
// per-channel parameters of the unit's 8 channels live in registers, arithmetic is fp32, results are rounded to
// bfloat16 once (round to nearest even).  Reductions: 8 channels x k sums per thread -> wave shuffles -> LDS -> one
// partial per workgroup and channel in the SAME partial layouts as the fp32 kernels (elementwise.hip), so the finalize
// kernels are shared.  No atomics: deterministic.
#include "cn8.h"

namespace {

constexpr int TPB = 256;

// Streaming kernels process EW_U units per thread and iteration, all loads first: with one unit per thread a 960 000-column plane
// was 3 750 workgroups of ONE 16-byte unit per thread, each paying its 24-48 parameter loads for 12 KB of traffic.
constexpr int EW_U = 4;
inline int unit_blocks(int64_t n, int per_thread = 1) {
  int64_t b = (n + (int64_t)TPB * per_thread - 1) / ((int64_t)TPB * per_thread);
  if (b < 1) b = 1;
  if (b > 65535) b = 65535;
  return (int)b;
}

// 8 bits of a unit: bit j = stored channel 8 g + j is > 0 (taken from the ROUNDED bfloat16 that is stored: a positive fp32
// that rounds to +0 counts as 0, exactly as a test of the stored value would)
__device__ __forceinline__ unsigned cn8_positive_bits(const uint4& p) {
  const unsigned w[4] = {p.x, p.y, p.z, p.w};
  unsigned m = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    m |= ((short)(w[q] & 0xffffu) > 0 ? 1u : 0u) << (2 * q);
    m |= ((short)(w[q] >> 16) > 0 ? 1u : 0u) << (2 * q + 1);
  }
  return m;
}

__global__ __launch_bounds__(TPB) void bn_add_relu_fwd_cn8_kernel(const uint4* __restrict__ u, const float* __restrict__ sc,
                                                                  const float* __restrict__ sh, int res_kind,
                                                                  const uint4* __restrict__ r, const float* __restrict__ rsc,
                                                                  const float* __restrict__ rsh, uint4* __restrict__ y, int C,
                                                                  int64_t n, int64_t ld, unsigned char* __restrict__ mask = nullptr) {
  const int g = blockIdx.y;
  float a[8], b[8], ra[8], rb[8];
  cn8_params(sc, g, C, 0.f, a);
  cn8_params(sh, g, C, 0.f, b);
  cn8_params(res_kind == 2 ? rsc : nullptr, g, C, 1.f, ra);
  cn8_params(res_kind == 2 ? rsh : nullptr, g, C, 0.f, rb);
  const int64_t base = (int64_t)g * ld;
  for (int64_t i0 = (int64_t)blockIdx.x * (TPB * EW_U) + threadIdx.x; i0 < n; i0 += (int64_t)gridDim.x * (TPB * EW_U)) {
    uint4 xu[EW_U], xr[EW_U];
#pragma unroll
    for (int q = 0; q < EW_U; ++q) {
      const int64_t i = i0 + q * TPB, ic = i < n ? i : n - 1;
      xu[q] = u[base + ic];
      if (res_kind) xr[q] = r[base + ic];
    }
#pragma unroll
    for (int q = 0; q < EW_U; ++q) {
      const int64_t i = i0 + q * TPB;
      if (i >= n) break;
      float uv[8], rv[8], o[8];
      cn8_unpack(xu[q], uv);
      if (res_kind) cn8_unpack(xr[q], rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float z = fmaf(uv[j], a[j], b[j]);
        if (res_kind) z += fmaf(rv[j], ra[j], rb[j]);
        o[j] = fmaxf(z, 0.f);
      }
      const uint4 p = cn8_pack(o);
      y[base + i] = p;
      if (mask) mask[base + i] = (unsigned char)cn8_positive_bits(p);   // uniform
    }
  }
}

// partials[C][nparts][4] = (sum dz, sum dz (u - mu), sum dz (r - mr), 0), dz = dy where y > 0
// (y == nullptr: the ReLU mask comes from `mask`, one byte per unit written by the forward tail, instead of from the stored y)
template <bool TAIL>
__global__ __launch_bounds__(TPB) void bn_add_relu_bwd_reduce_cn8_kernel(const uint4* __restrict__ dy, const uint4* __restrict__ y,
                                                                         const uint4* __restrict__ u, const uint4* __restrict__ r,
                                                                         const float* __restrict__ mu_p,
                                                                         const float* __restrict__ mr_p,
                                                                         float* __restrict__ partials, int C, int64_t n,
                                                                         int64_t ld, const sar_bn_tail tail,
                                                                         const unsigned char* __restrict__ mask = nullptr) {
  const int g = blockIdx.y;
  float mu[8], mr[8];
  cn8_params(mu_p, g, C, 0.f, mu);
  cn8_params(r ? mr_p : nullptr, g, C, 0.f, mr);
  const int64_t base = (int64_t)g * ld;
  float acc[3][8];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    float gv[8], uv[8], rv[8];
    cn8_unpack(dy[base + i], gv);
    unsigned mb;
    if (mask) mb = mask[base + i];   // uniform
    else mb = cn8_positive_bits(y[base + i]);
    cn8_unpack(u[base + i], uv);
    if (r) cn8_unpack(r[base + i], rv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dz = ((mb >> j) & 1u) ? gv[j] : 0.f;
      acc[0][j] += dz;
      acc[1][j] = fmaf(dz, uv[j] - mu[j], acc[1][j]);
      if (r) acc[2][j] = fmaf(dz, rv[j] - mr[j], acc[2][j]);
    }
  }
  __shared__ float red[4][24];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = wave_sum(acc[q][j]);
      if (lane == 0) red[wave][q * 8 + j] = t;
    }
  __syncthreads();
  if (threadIdx.x < 24) {
    const int q = threadIdx.x >> 3, j = threadIdx.x & 7;
    const int c = 8 * g + j;
    if (c < C) {
      const int k = threadIdx.x;
      float* pp = partials + ((int64_t)c * gridDim.x + blockIdx.x) * 4;
      const float v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
      if constexpr (TAIL) {
        bn_tail_store(pp + q, v);
      } else {
        pp[q] = v;
        if (q == 0) pp[3] = 0.f;
      }
    }
  }
  if constexpr (TAIL) {   // the last workgroup of the plane finalises its 8 channels, two per wave (sar_bn_tail)
    __shared__ int last;  // (the 24 storing lanes are lanes of wave 0, as is the lane that takes the ticket)
    if (threadIdx.x == 0) last = bn_tail_last_arriver(tail.ticket + g, (int)gridDim.x) ? 1 : 0;
    __syncthreads();
    if (last) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = 8 * g + 2 * wave + j;
        if (c < C) bn_tail_channel(partials, (int)gridDim.x, c, tail, mu_p, mr_p, r != nullptr);
      }
    }
  }
}

// du = k1 dz + k2 u + k3 ; dr = rk1 dz + rk2 r + rk3 ; dz_out = dz   (dz = dy where y > 0)
__global__ __launch_bounds__(TPB) void bn_add_relu_bwd_apply_cn8_kernel(
    const uint4* __restrict__ dy, const uint4* __restrict__ y, const uint4* __restrict__ u, const uint4* __restrict__ r,
    const float* __restrict__ k1, const float* __restrict__ k2, const float* __restrict__ k3, const float* __restrict__ rk1,
    const float* __restrict__ rk2, const float* __restrict__ rk3, uint4* du, uint4* dr, uint4* dz_out, int C, int64_t n,
    int64_t ld, const unsigned char* __restrict__ mask = nullptr) {
  const int g = blockIdx.y;
  float a1[8], a2[8], a3[8], b1[8], b2[8], b3[8];
  cn8_params(k1, g, C, 0.f, a1);
  cn8_params(k2, g, C, 0.f, a2);
  cn8_params(k3, g, C, 0.f, a3);
  cn8_params(dr ? rk1 : nullptr, g, C, 0.f, b1);
  cn8_params(dr ? rk2 : nullptr, g, C, 0.f, b2);
  cn8_params(dr ? rk3 : nullptr, g, C, 0.f, b3);
  const int64_t base = (int64_t)g * ld;
  for (int64_t i0 = (int64_t)blockIdx.x * (TPB * EW_U) + threadIdx.x; i0 < n; i0 += (int64_t)gridDim.x * (TPB * EW_U)) {
    uint4 xg[EW_U], xu[EW_U], xr[EW_U];
    unsigned xm[EW_U];
#pragma unroll
    for (int q = 0; q < EW_U; ++q) {   // (dz_out may be dy: a thread stores only the units it has already read)
      const int64_t i = i0 + q * TPB, ic = i < n ? i : n - 1;
      xg[q] = dy[base + ic];
      if (mask) xm[q] = mask[base + ic];   // uniform
      else xm[q] = cn8_positive_bits(y[base + ic]);
      xu[q] = u[base + ic];
      if (dr) xr[q] = r[base + ic];
    }
#pragma unroll
    for (int q = 0; q < EW_U; ++q) {
      const int64_t i = i0 + q * TPB;
      if (i >= n) break;
      float gv[8], uv[8], rv[8], o1[8], o2[8], o3[8];
      cn8_unpack(xg[q], gv);
      const unsigned mb = xm[q];
      cn8_unpack(xu[q], uv);
      if (dr) cn8_unpack(xr[q], rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dz = ((mb >> j) & 1u) ? gv[j] : 0.f;
        o3[j] = dz;
        o1[j] = fmaf(a1[j], dz, fmaf(a2[j], uv[j], a3[j]));
        if (dr) o2[j] = fmaf(b1[j], dz, fmaf(b2[j], rv[j], b3[j]));
      }
      du[base + i] = cn8_pack(o1);
      if (dr) dr[base + i] = cn8_pack(o2);
      if (dz_out) dz_out[base + i] = cn8_pack(o3);
    }
  }
}

__global__ __launch_bounds__(TPB) void affine2_cn8_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                          const float* __restrict__ k1, const float* __restrict__ k2,
                                                          const float* __restrict__ k3, uint4* out, int C, int64_t n, int64_t ld) {
  const int g = blockIdx.y;
  float a1[8], a2[8], a3[8];
  cn8_params(k1, g, C, 0.f, a1);
  cn8_params(k2, g, C, 0.f, a2);
  cn8_params(k3, g, C, 0.f, a3);
  const int64_t base = (int64_t)g * ld;
  for (int64_t i0 = (int64_t)blockIdx.x * (TPB * EW_U) + threadIdx.x; i0 < n; i0 += (int64_t)gridDim.x * (TPB * EW_U)) {
    uint4 xa[EW_U], xb[EW_U];
#pragma unroll
    for (int q = 0; q < EW_U; ++q) {   // (out may be a: a thread stores only the units it has already read)
      const int64_t i = i0 + q * TPB, ic = i < n ? i : n - 1;
      xa[q] = a[base + ic];
      xb[q] = b[base + ic];
    }
#pragma unroll
    for (int q = 0; q < EW_U; ++q) {
      const int64_t i = i0 + q * TPB;
      if (i >= n) break;
      float av[8], bv[8], o[8];
      cn8_unpack(xa[q], av);
      cn8_unpack(xb[q], bv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf(a1[j], av[j], fmaf(a2[j], bv[j], a3[j]));
      out[base + i] = cn8_pack(o);
    }
  }
}

// one wave per (sample, channel group): 8 fp32 sums per lane, two independent chains
__global__ __launch_bounds__(TPB) void pool_fwd_cn8_kernel(const uint4* __restrict__ y, int64_t ld, int span, float inv, int C,
                                                           int N, float* __restrict__ feat) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6), g = blockIdx.y;
  if (n >= N) return;
  const uint4* p = y + (int64_t)g * ld + (int64_t)n * span;
  float a0[8], a1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a0[j] = a1[j] = 0.f;
  int i = lane;
  for (; i + 64 < span; i += 128) {
    float v0[8], v1[8];
    cn8_unpack(p[i], v0);
    cn8_unpack(p[i + 64], v1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0[j] += v0[j];
      a1[j] += v1[j];
    }
  }
  if (i < span) {
    float v0[8];
    cn8_unpack(p[i], v0);
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] += v0[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float t = wave_sum(a0[j] + a1[j]);
    if (lane == 0 && 8 * g + j < C) feat[(int64_t)n * C + 8 * g + j] = t * inv;
  }
}

__global__ __launch_bounds__(TPB) void pool_bwd_cn8_kernel(const float* __restrict__ dfeat, int64_t ld, int span, float inv, int C,
                                                           uint4* __restrict__ dy) {
  const int n = blockIdx.x, g = blockIdx.y;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (8 * g + j < C) ? dfeat[(int64_t)n * C + 8 * g + j] * inv : 0.f;
  const uint4 q = cn8_pack(v);
  uint4* p = dy + (int64_t)g * ld + (int64_t)n * span;
  for (int i = threadIdx.x; i < span; i += TPB) p[i] = q;
}

// fp32 CN matrix [C][ld] <-> CN8 (tests, layout conversion at the boundary of the bf16 engine)
__global__ __launch_bounds__(TPB) void cn_to_cn8_kernel(const float* __restrict__ x, int64_t ld_x, uint4* __restrict__ out,
                                                        int64_t ld_o, int C, int64_t n) {
  const int g = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (8 * g + j < C) ? x[(int64_t)(8 * g + j) * ld_x + i] : 0.f;
    out[(int64_t)g * ld_o + i] = cn8_pack(v);
  }
}

__global__ __launch_bounds__(TPB) void cn8_to_cn_kernel(const uint4* __restrict__ x, int64_t ld_x, float* __restrict__ out,
                                                        int64_t ld_o, int C, int64_t n) {
  const int g = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    float v[8];
    cn8_unpack(x[(int64_t)g * ld_x + i], v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (8 * g + j < C) out[(int64_t)(8 * g + j) * ld_o + i] = v[j];
  }
}

#ifdef SAR_DEBUG
// diagnostic: leave a known pattern in the LDS of every CU (a kernel that reads LDS it never wrote then shows it)
__global__ __launch_bounds__(TPB) void poison_lds_kernel(unsigned pattern, unsigned* sink) {
  extern __shared__ unsigned dyn[];
  for (int i = threadIdx.x; i < 16384; i += TPB) dyn[i] = pattern;
  __syncthreads();
  if (dyn[(threadIdx.x * 61) & 16383] != pattern) sink[0] = 1;   // keeps the stores alive
}
#endif

inline bool al16(std::initializer_list<const void*> ptrs) {
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

}  // namespace

#define CN8_G(C) (((C) + 7) / 8)

extern "C" int sar_bn_add_relu_fwd_cn8(const void* u, const float* sc, const float* sh, int res_kind, const void* r,
                                       const float* rsc, const float* rsh, void* y, int C, int64_t n, int64_t ld,
                                       sar_stream_t s) {
  SAR_REQUIRE(u && sc && sh && y && C > 0 && n > 0 && ld >= n, "sar_bn_add_relu_fwd_cn8: bad arguments");
  SAR_REQUIRE(res_kind >= 0 && res_kind <= 2 && (res_kind == 0 || r) && (res_kind != 2 || (rsc && rsh)),
              "sar_bn_add_relu_fwd_cn8: residual arguments");
  SAR_REQUIRE(al16({u, r, y}), "sar_bn_add_relu_fwd_cn8: CN8 tensors must be 16-byte aligned");
  hipLaunchKernelGGL(bn_add_relu_fwd_cn8_kernel, dim3(unit_blocks(n, EW_U), CN8_G(C)), dim3(TPB), 0, as_stream(s), (const uint4*)u, sc,
                     sh, res_kind, (const uint4*)r, rsc, rsh, (uint4*)y, C, n, ld);
  SAR_LAUNCH_CHECK("sar_bn_add_relu_fwd_cn8");
  return 0;
}

extern "C" int sar_bn_add_relu_fwd_mask_cn8(const void* u, const float* sc, const float* sh, int res_kind, const void* r,
                                            const float* rsc, const float* rsh, void* y, void* mask, int C, int64_t n, int64_t ld,
                                            sar_stream_t s) {
  SAR_REQUIRE(u && sc && sh && y && mask && C > 0 && n > 0 && ld >= n, "sar_bn_add_relu_fwd_mask_cn8: bad arguments");
  SAR_REQUIRE(res_kind >= 0 && res_kind <= 2 && (res_kind == 0 || r) && (res_kind != 2 || (rsc && rsh)),
              "sar_bn_add_relu_fwd_mask_cn8: residual arguments");
  SAR_REQUIRE(al16({u, r, y}), "sar_bn_add_relu_fwd_mask_cn8: CN8 tensors must be 16-byte aligned");
  hipLaunchKernelGGL(bn_add_relu_fwd_cn8_kernel, dim3(unit_blocks(n, EW_U), CN8_G(C)), dim3(TPB), 0, as_stream(s), (const uint4*)u, sc,
                     sh, res_kind, (const uint4*)r, rsc, rsh, (uint4*)y, C, n, ld, (unsigned char*)mask);
  SAR_LAUNCH_CHECK("sar_bn_add_relu_fwd_mask_cn8");
  return 0;
}

extern "C" int sar_bn_add_relu_bwd_reduce_cn8(const void* dy, const void* y, const void* u, const void* r, const float* mean_u,
                                              const float* mean_r, float* partials, int nparts, int C, int64_t n, int64_t ld,
                                              sar_stream_t s) {
  SAR_REQUIRE(dy && y && u && partials && nparts > 0 && nparts <= 65535 && C > 0 && n > 0 && ld >= n,
              "sar_bn_add_relu_bwd_reduce_cn8: bad arguments");
  SAR_REQUIRE(al16({dy, y, u, r}), "sar_bn_add_relu_bwd_reduce_cn8: CN8 tensors must be 16-byte aligned");
  hipLaunchKernelGGL(bn_add_relu_bwd_reduce_cn8_kernel<false>, dim3(nparts, CN8_G(C)), dim3(TPB), 0, as_stream(s), (const uint4*)dy,
                     (const uint4*)y, (const uint4*)u, (const uint4*)r, mean_u, mean_r, partials, C, n, ld, sar_bn_tail());
  SAR_LAUNCH_CHECK("sar_bn_add_relu_bwd_reduce_cn8");
  return 0;
}

extern "C" int sar_bn_add_relu_bwd_reduce_mask_cn8(const void* dy, const void* mask, const void* u, const void* r,
                                                   const float* mean_u, const float* mean_r, float* partials, int nparts, int C,
                                                   int64_t n, int64_t ld, sar_stream_t s) {
  SAR_REQUIRE(dy && mask && u && partials && nparts > 0 && nparts <= 65535 && C > 0 && n > 0 && ld >= n,
              "sar_bn_add_relu_bwd_reduce_mask_cn8: bad arguments");
  SAR_REQUIRE(al16({dy, u, r}), "sar_bn_add_relu_bwd_reduce_mask_cn8: CN8 tensors must be 16-byte aligned");
  hipLaunchKernelGGL(bn_add_relu_bwd_reduce_cn8_kernel<false>, dim3(nparts, CN8_G(C)), dim3(TPB), 0, as_stream(s), (const uint4*)dy,
                     (const uint4*)nullptr, (const uint4*)u, (const uint4*)r, mean_u, mean_r, partials, C, n, ld, sar_bn_tail(),
                     (const unsigned char*)mask);
  SAR_LAUNCH_CHECK("sar_bn_add_relu_bwd_reduce_mask_cn8");
  return 0;
}

extern "C" int sar_bn_add_relu_bwd_reduce_tail_cn8(const void* dy, const void* y, const void* u, const void* r, const float* mean_u,
                                                   const float* mean_r, float* partials, int nparts, int C, int64_t n, int64_t ld,
                                                   const sar_bn_tail* tail, sar_stream_t s) {
  SAR_REQUIRE(dy && y && u && partials && nparts > 0 && nparts <= 65535 && C > 0 && n > 0 && ld >= n,
              "sar_bn_add_relu_bwd_reduce_tail_cn8: bad arguments");
  SAR_REQUIRE(al16({dy, y, u, r, partials}), "sar_bn_add_relu_bwd_reduce_tail_cn8: CN8 tensors and partials must be 16-byte aligned");
  SAR_REQUIRE(tail && tail->ticket && tail->count > 0 && tail->rstd && tail->k1 && tail->k2 && tail->k3,
              "sar_bn_add_relu_bwd_reduce_tail_cn8: ticket, count, rstd and k1..k3 are required");
  SAR_REQUIRE(!r || (tail->rrstd && tail->rk1 && tail->rk2 && tail->rk3), "sar_bn_add_relu_bwd_reduce_tail_cn8: residual-branch outputs");
  hipLaunchKernelGGL(bn_add_relu_bwd_reduce_cn8_kernel<true>, dim3(nparts, CN8_G(C)), dim3(TPB), 0, as_stream(s), (const uint4*)dy,
                     (const uint4*)y, (const uint4*)u, (const uint4*)r, mean_u, mean_r, partials, C, n, ld, *tail);
  SAR_LAUNCH_CHECK("sar_bn_add_relu_bwd_reduce_tail_cn8");
  return 0;
}

extern "C" int sar_bn_add_relu_bwd_apply_cn8(const void* dy, const void* y, const void* u, const void* r, const float* k1,
                                             const float* k2, const float* k3, const float* rk1, const float* rk2,
                                             const float* rk3, void* du, void* dr, void* dz_out, int C, int64_t n, int64_t ld,
                                             sar_stream_t s) {
  SAR_REQUIRE(dy && y && u && k1 && k2 && k3 && du && C > 0 && n > 0 && ld >= n, "sar_bn_add_relu_bwd_apply_cn8: bad arguments");
  SAR_REQUIRE(!dr || (r && rk1 && rk2 && rk3), "sar_bn_add_relu_bwd_apply_cn8: residual-branch arguments");
  SAR_REQUIRE(al16({dy, y, u, r, du, dr, dz_out}), "sar_bn_add_relu_bwd_apply_cn8: CN8 tensors must be 16-byte aligned");
  hipLaunchKernelGGL(bn_add_relu_bwd_apply_cn8_kernel, dim3(unit_blocks(n, EW_U), CN8_G(C)), dim3(TPB), 0, as_stream(s),
                     (const uint4*)dy, (const uint4*)y, (const uint4*)u, (const uint4*)r, k1, k2, k3, rk1, rk2, rk3, (uint4*)du,
                     (uint4*)dr, (uint4*)dz_out, C, n, ld);
  SAR_LAUNCH_CHECK("sar_bn_add_relu_bwd_apply_cn8");
  return 0;
}

extern "C" int sar_bn_add_relu_bwd_apply_mask_cn8(const void* dy, const void* mask, const void* u, const void* r, const float* k1,
                                                  const float* k2, const float* k3, const float* rk1, const float* rk2,
                                                  const float* rk3, void* du, void* dr, void* dz_out, int C, int64_t n, int64_t ld,
                                                  sar_stream_t s) {
  SAR_REQUIRE(dy && mask && u && k1 && k2 && k3 && du && C > 0 && n > 0 && ld >= n, "sar_bn_add_relu_bwd_apply_mask_cn8: bad arguments");
  SAR_REQUIRE(!dr || (r && rk1 && rk2 && rk3), "sar_bn_add_relu_bwd_apply_mask_cn8: residual-branch arguments");
  SAR_REQUIRE(al16({dy, u, r, du, dr, dz_out}), "sar_bn_add_relu_bwd_apply_mask_cn8: CN8 tensors must be 16-byte aligned");
  hipLaunchKernelGGL(bn_add_relu_bwd_apply_cn8_kernel, dim3(unit_blocks(n, EW_U), CN8_G(C)), dim3(TPB), 0, as_stream(s),
                     (const uint4*)dy, (const uint4*)nullptr, (const uint4*)u, (const uint4*)r, k1, k2, k3, rk1, rk2, rk3, (uint4*)du,
                     (uint4*)dr, (uint4*)dz_out, C, n, ld, (const unsigned char*)mask);
  SAR_LAUNCH_CHECK("sar_bn_add_relu_bwd_apply_mask_cn8");
  return 0;
}

extern "C" int sar_affine2_cn8(const void* a, const void* b, const float* k1, const float* k2, const float* k3, void* out, int C,
                               int64_t n, int64_t ld, sar_stream_t s) {
  SAR_REQUIRE(a && b && k1 && k2 && k3 && out && C > 0 && n > 0 && ld >= n, "sar_affine2_cn8: bad arguments");
  SAR_REQUIRE(al16({a, b, out}), "sar_affine2_cn8: CN8 tensors must be 16-byte aligned");
  hipLaunchKernelGGL(affine2_cn8_kernel, dim3(unit_blocks(n, EW_U), CN8_G(C)), dim3(TPB), 0, as_stream(s), (const uint4*)a,
                     (const uint4*)b, k1, k2, k3, (uint4*)out, C, n, ld);
  SAR_LAUNCH_CHECK("sar_affine2_cn8");
  return 0;
}

extern "C" int sar_pool_fwd_cn8(const void* y, int64_t ld, int C, int B, int TV, int Mp, float* feat, sar_stream_t s) {
  SAR_REQUIRE(y && feat && C > 0 && B > 0 && TV > 0 && Mp > 0 && B % Mp == 0 && ld >= (int64_t)B * TV && al16({y}),
              "sar_pool_fwd_cn8: bad arguments");
  const int N = B / Mp, span = Mp * TV;
  hipLaunchKernelGGL(pool_fwd_cn8_kernel, dim3((N + TPB / 64 - 1) / (TPB / 64), CN8_G(C)), dim3(TPB), 0, as_stream(s),
                     (const uint4*)y, ld, span, 1.0f / (float)span, C, N, feat);
  SAR_LAUNCH_CHECK("sar_pool_fwd_cn8");
  return 0;
}

extern "C" int sar_pool_bwd_cn8(const float* dfeat, int64_t ld, int C, int B, int TV, int Mp, void* dy, sar_stream_t s) {
  SAR_REQUIRE(dfeat && dy && C > 0 && B > 0 && TV > 0 && Mp > 0 && B % Mp == 0 && ld >= (int64_t)B * TV && al16({dy}),
              "sar_pool_bwd_cn8: bad arguments");
  const int span = Mp * TV;
  hipLaunchKernelGGL(pool_bwd_cn8_kernel, dim3(B / Mp, CN8_G(C)), dim3(TPB), 0, as_stream(s), dfeat, ld, span,
                     1.0f / (float)span, C, (uint4*)dy);
  SAR_LAUNCH_CHECK("sar_pool_bwd_cn8");
  return 0;
}

#ifdef SAR_DEBUG
#include "../../include/sar_hip_debug.h"
extern "C" int sar_debug_poison_lds(unsigned pattern, void* sink, sar_stream_t s) {
  SAR_REQUIRE(sink, "sar_debug_poison_lds: sink required");
  hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(TPB), 65536, as_stream(s), pattern, (unsigned*)sink);
  SAR_LAUNCH_CHECK("sar_debug_poison_lds");
  return 0;
}
#endif

extern "C" int sar_cn_to_cn8(const float* x, int64_t ld_x, void* out, int64_t ld_out, int C, int64_t n, sar_stream_t s) {
  SAR_REQUIRE(x && out && C > 0 && n > 0 && ld_x >= n && ld_out >= n && al16({out}), "sar_cn_to_cn8: bad arguments");
  hipLaunchKernelGGL(cn_to_cn8_kernel, dim3(unit_blocks(n), CN8_G(C)), dim3(TPB), 0, as_stream(s), x, ld_x, (uint4*)out, ld_out,
                     C, n);
  SAR_LAUNCH_CHECK("sar_cn_to_cn8");
  return 0;
}

extern "C" int sar_cn8_to_cn(const void* x, int64_t ld_x, float* out, int64_t ld_out, int C, int64_t n, sar_stream_t s) {
  SAR_REQUIRE(x && out && C > 0 && n > 0 && ld_x >= n && ld_out >= n && al16({x}), "sar_cn8_to_cn: bad arguments");
  hipLaunchKernelGGL(cn8_to_cn_kernel, dim3(unit_blocks(n), CN8_G(C)), dim3(TPB), 0, as_stream(s), (const uint4*)x, ld_x, out,
                     ld_out, C, n);
  SAR_LAUNCH_CHECK("sar_cn8_to_cn");
  return 0;
}

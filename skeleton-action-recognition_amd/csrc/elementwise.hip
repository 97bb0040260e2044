// elementwise.hip -- HBM-bound kernels of the ST-GCN step on the CN layout (gfx950):
// batch-norm finalisation, the block tail (BN + residual + ReLU) forward/backward, data_bn with the
// (N,C,T,V,M) -> CN re-layout, the classifier head, softmax cross-entropy, Nesterov SGD.
// All row-wise kernels move 16 B per lane when the row stride allows it (float4), are grid-strided
// inside a row and reduce through wave shuffles -> LDS -> one partial per workgroup (no atomics, so
// every reduction is deterministic).  Reference call sites are cited in include/sar_hip.h.
#include <stdarg.h>
#include "sar_common.h"
#include <stdlib.h>
#include "cn8.h"

// ------------------------------------------------------------------------------------ error state
static thread_local char g_err[512] = "no error";
void sar_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* sar_last_error_string(void) { return g_err; }
extern "C" int sar_version(void) { return 100; }
extern "C" int sar_context_create(sar_context** out) {
  SAR_REQUIRE(out != nullptr, "sar_context_create: null output pointer");
  *out = nullptr;
  sar_context* c = new sar_context();
  c->nstreams = 0;
  hipError_t e = hipGetDevice(&c->device);
  const char* env = getenv("SAR_C2D_PARITY_STREAMS");      // experiment switch: 0 = contexts without side streams
  const bool want = !(env && env[0] == '0');
  if (e == hipSuccess && want) {
    e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
    int made = 0;
    for (; e == hipSuccess && made < 3; ++made) {
      e = hipStreamCreateWithFlags(&c->s[made], hipStreamNonBlocking);
      if (e == hipSuccess) {
        e = hipEventCreateWithFlags(&c->join[made], hipEventDisableTiming);
        if (e != hipSuccess) hipStreamDestroy(c->s[made]);
      }
      if (e != hipSuccess) break;
    }
    if (e == hipSuccess) c->nstreams = 3;
    else {
      for (int i = 0; i < made; ++i) {
        hipStreamDestroy(c->s[i]);
        hipEventDestroy(c->join[i]);
      }
      hipEventDestroy(c->fork);
    }
  }
  if (e != hipSuccess) {
    sar_set_error("sar_context_create: %s", hipGetErrorString(e));
    delete c;
    return (int)e;
  }
  *out = c;
  return 0;
}

extern "C" int sar_context_destroy(sar_context* c) {
  if (!c) return 0;
  if (c->nstreams == 3) {
    for (int i = 0; i < 3; ++i) {
      hipStreamDestroy(c->s[i]);
      hipEventDestroy(c->join[i]);
    }
    hipEventDestroy(c->fork);
  }
  delete c;
  return 0;
}

// A stream confined to a subset of the compute units (the caller owns it: sar_stream_destroy).  The weight-gradient stream of the
// engines runs long-lived workgroups (340-1 050 us each, two per CU, every vector register of the SIMDs): a short kernel of the main
// chain (the 23 BatchNorm finalisations of a step: 8 us alone, 119 us in the step) cannot be placed until one of them retires.  With
// the side stream masked off a few CUs the main chain always finds free slots there.
extern "C" int sar_stream_create_cu_mask(const uint32_t* mask, int nwords, sar_stream_t* out) {
  SAR_REQUIRE(mask != nullptr && nwords > 0 && nwords <= 64 && out != nullptr, "sar_stream_create_cu_mask: bad arguments");
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask);
  if (e != hipSuccess) {
    sar_set_error("sar_stream_create_cu_mask: %s", hipGetErrorString(e));
    return (int)e;
  }
  *out = (sar_stream_t)st;
  return 0;
}
extern "C" int sar_stream_destroy(sar_stream_t s) {
  if (!s) return 0;
  hipError_t e = hipStreamDestroy((hipStream_t)s);
  if (e != hipSuccess) {
    sar_set_error("sar_stream_destroy: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int sar_struct_size(int which) { return which == 0 ? (int)sizeof(sar_conv_desc) : which == 1 ? (int)sizeof(sar_wgrad_desc) : which == 2 ? (int)sizeof(sar_conv2d_desc) : -1; }

namespace {

constexpr int TPB = 256;

// block-wide sum of NV values; result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red /* [4][NV] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
  __syncthreads();
  if (threadIdx.x == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = red[i] + red[NV + i] + red[2 * NV + i] + red[3 * NV + i];
}

// ------------------------------------------------------------------------------------ BN finalize
__global__ __launch_bounds__(TPB) void bn_finalize_kernel(const float* __restrict__ partials, int nparts, double count,
                                                          float eps, float momentum, int unbiased_running,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float* mean_o,
                                                          float* rstd_o, float* scale_o, float* shift_o) {
  const int c = blockIdx.x;
  // (sum, sum of squares) pairs: 8-byte loads, four independent fp64 chains per thread so that the loads overlap
  const float2* p = reinterpret_cast<const float2*>(partials) + (int64_t)c * nparts;
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
  int i = threadIdx.x;
  for (; i + 3 * TPB < nparts; i += 4 * TPB) {
    float2 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = p[i + q * TPB];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a1[q] += (double)v[q].x;
      a2[q] += (double)v[q].y;
    }
  }
  for (; i < nparts; i += TPB) {
    const float2 v = p[i];
    a1[0] += (double)v.x;
    a2[0] += (double)v.y;
  }
  double s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  __shared__ double red[2][4];
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    if (mean_o) mean_o[c] = (float)mean;
    if (rstd_o) rstd_o[c] = (float)rstd;
    scale_o[c] = (float)(g * rstd);
    shift_o[c] = (float)(b - mean * g * rstd);
    if (running_mean) {
      const double vr = (unbiased_running && count > 1.0) ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((double)momentum * running_mean[c] + (1.0 - (double)momentum) * mean);
      running_var[c] = (float)((double)momentum * running_var[c] + (1.0 - (double)momentum) * vr);
    }
  }
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.0f / sqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * rstd;
  shift[c] = b - rm[c] * g * rstd;
}

__global__ __launch_bounds__(TPB) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nparts,
                                                              int64_t chan_stride, int64_t part_stride, int off1, int off2,
                                                              int centered, double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              float* dgamma, float* dbeta, float* k1, float* k2, float* k3) {
  const int c = blockIdx.x;
  const float* p = partials + (int64_t)c * chan_stride;
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};   // four independent chains: the loads overlap
  int i = threadIdx.x;
  for (; i + 3 * TPB < nparts; i += 4 * TPB) {
    float u[4], w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u[q] = p[(int64_t)(i + q * TPB) * part_stride + off1];
      w[q] = p[(int64_t)(i + q * TPB) * part_stride + off2];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a1[q] += (double)u[q];
      a2[q] += (double)w[q];
    }
  }
  for (; i < nparts; i += TPB) {
    a1[0] += (double)p[(int64_t)i * part_stride + off1];
    a2[0] += (double)p[(int64_t)i * part_stride + off2];
  }
  double s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  __shared__ double red[2][4];
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s1;
    red[1][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const double m = mean[c], rs = rstd[c], g = gamma ? (double)gamma[c] : 1.0;
    const double dg = centered ? rs * s2 : rs * (s2 - m * s1);  // sum dz * xhat
    if (dgamma) dgamma[c] = (float)dg;
    if (dbeta) dbeta[c] = (float)s1;
    const double a = s1 / count, b = dg / count;
    if (k1) {
      k1[c] = (float)(g * rs);
      k2[c] = (float)(-g * rs * rs * b);
      k3[c] = (float)(g * rs * (m * rs * b - a));
    }
  }
}

// ------------------------------------------------------------------------------------ data_bn
// One workgroup per (n, c) slab x[n][c][T][V][M] (contiguous, M fastest).
__device__ __forceinline__ float load_frame(const float* slab, int t, int v, int m, int V, int M, const int* bone_parent) {
  float val = slab[(t * V + v) * M + m];
  if (bone_parent) {
    const int v2 = bone_parent[v];
    if (v2 >= 0) val -= slab[(t * V + v2) * M + m];
  }
  return val;
}
// optional motion stream (data_gen/gen_motion_data.py:24-27): frame t+1 minus frame t of the joint OR bone data
// (each a float32 value, so the two roundings of the offline passes are reproduced), last frame 0
__device__ __forceinline__ float load_joint(const float* slab, int t, int v, int m, int V, int M, const int* bone_parent,
                                            int motion, int T) {
  if (!motion) return load_frame(slab, t, v, m, V, M, bone_parent);
  if (t >= T - 1) return 0.f;
  return load_frame(slab, t + 1, v, m, V, M, bone_parent) - load_frame(slab, t, v, m, V, M, bone_parent);
}

template <bool BWD, bool DY_CN8 = false>
__global__ __launch_bounds__(TPB) void data_bn_reduce_kernel(const float* __restrict__ x, int N, int C, int T, int V, int M,
                                                             const int* __restrict__ bone_parent, int motion,
                                                             const float* __restrict__ dy, int64_t ld_dy,
                                                             const float* __restrict__ mean,
                                                             float* __restrict__ partials) {
  const int n = blockIdx.x / C, c = blockIdx.x - n * C;
  const float* slab = x + (int64_t)blockIdx.x * T * V * M;
  const int VM = V * M;
  const int groups = TPB / VM;  // row groups working in parallel
  const int col = threadIdx.x % VM, rg = threadIdx.x / VM;
  const int v = col / M, m = col - v * M;
  float s1 = 0.f, s2 = 0.f;
  const float mu = (BWD && mean && rg < groups) ? mean[v * C + c] : 0.f;
  if (rg < groups) {
    for (int t = rg; t < T; t += groups) {
      const float val = load_joint(slab, t, v, m, V, M, bone_parent, motion, T) - mu;
      if (BWD) {
        float g;
        if (DY_CN8) {   // unit 0 of the column holds channels 0..7 of the (C <= 8)-channel input gradient
          const unsigned short h = reinterpret_cast<const unsigned short*>(dy)[(((int64_t)(n * M + m) * T + t) * V + v) * 8 + c];
          g = __uint_as_float((unsigned)h << 16);
        } else {
          g = dy[(int64_t)c * ld_dy + ((int64_t)(n * M + m) * T + t) * V + v];
        }
        s1 += g;
        s2 = fmaf(g, val, s2);
      } else {
        s1 += val;
        s2 = fmaf(val, val, s2);
      }
    }
  }
  __shared__ float r1[TPB], r2[TPB];
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.x < V) {
    float a = 0.f, b = 0.f;
    for (int g = 0; g < groups; ++g)
      for (int mm = 0; mm < M; ++mm) {
        a += r1[g * VM + threadIdx.x * M + mm];
        b += r2[g * VM + threadIdx.x * M + mm];
      }
    float* pp = partials + ((int64_t)(threadIdx.x * C + c) * N + n) * 2;
    pp[0] = a;
    pp[1] = b;
  }
}

__global__ __launch_bounds__(TPB) void data_bn_apply_kernel(const float* __restrict__ x, int N, int C, int T, int V, int M,
                                                            const int* __restrict__ bone_parent, int motion,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* __restrict__ out, int64_t ld_out) {
  const int n = blockIdx.x / C, c = blockIdx.x - n * C;
  const float* slab = x + (int64_t)blockIdx.x * T * V * M;
  const int VM = V * M, total = T * VM;
  for (int e = threadIdx.x; e < total; e += TPB) {
    const int t = e / VM, r = e - t * VM;
    const int v = r / M, m = r - v * M;
    const float val = load_joint(slab, t, v, m, V, M, bone_parent, motion, T);
    const int ch = v * C + c;
    out[(int64_t)c * ld_out + ((int64_t)(n * M + m) * T + t) * V + v] = fmaf(val, scale[ch], shift[ch]);
  }
}

// CN8 output (cn8.h): one unit per column holds the C <= 8 input channels.  Workgroup = (sample n, frame chunk); a thread
// owns one (t, v, m) position and reads its C channel values (each read is coalesced over positions).
__global__ __launch_bounds__(TPB) void data_bn_apply_cn8_kernel(const float* __restrict__ x, int N, int C, int T, int V, int M,
                                                                const int* __restrict__ bone_parent, int motion,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                uint4* __restrict__ out) {
  const int n = blockIdx.x;
  const int VM = V * M, total = T * VM;
  const int per = (total + gridDim.y - 1) / gridDim.y;
  const int e_lo = blockIdx.y * per, e_hi = (e_lo + per < total) ? e_lo + per : total;
  for (int e = e_lo + threadIdx.x; e < e_hi; e += TPB) {
    const int t = e / VM, r = e - t * VM;
    const int v = r / M, m = r - v * M;
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      o[c] = 0.f;
      if (c < C) {
        const float* slab = x + (int64_t)(n * C + c) * T * VM;
        const int ch = v * C + c;
        o[c] = fmaf(load_joint(slab, t, v, m, V, M, bone_parent, motion, T), scale[ch], shift[ch]);
      }
    }
    out[((int64_t)(n * M + m) * T + t) * V + v] = cn8_pack(o);
  }
}

// ------------------------------------------------------------------------------------ row-wise kernels
template <int VEC> struct VecT;
template <> struct VecT<4> { typedef float4 T; };
template <> struct VecT<1> { typedef float T; };
template <int VEC> __device__ __forceinline__ void ld(const float* p, float (&r)[VEC]) {
  if (VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(p); r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w; }
  else r[0] = *p;
}
template <int VEC> __device__ __forceinline__ void st(float* p, const float (&r)[VEC]) {
  if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  else *p = r[0];
}

// amax by-product of a row-wise pass (the operand bounds of the split arithmetic, include/sar_hip.h: cells): the block's largest
// |value| as float bits; one atomic max per workgroup, and only when it would raise the cell (maxima are order-independent)
// running maximum on the BITS of |v| (unsigned order: NaN > Inf > every finite magnitude -- fmaxf would drop a NaN and the split
// kernels would rescale a non-finite tensor into finite fp16 terms; with the bits in the cell they emit NaN, split_scale.h)
__device__ __forceinline__ unsigned amax_bits(unsigned m, float v) {
  const unsigned b = __float_as_uint(v) & 0x7fffffffu;
  return b > m ? b : m;
}
__device__ __forceinline__ void block_amax(unsigned m, unsigned* __restrict__ cell) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)m, o);
    m = t > m ? t : m;
  }
  __shared__ unsigned wm[TPB / 64];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = wm[0];
#pragma unroll
    for (int i = 1; i < TPB / 64; ++i) a = wm[i] > a ? wm[i] : a;
    if (a > __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(cell, a);
  }
}

// mask (VEC == 4 only): one byte per float4 of y, bit j = element j > 0, row stride ldm / 4 bytes -- read by the two backward
// passes instead of y (sar_bn_add_relu_*_mask_f32)
template <int VEC>
__global__ __launch_bounds__(TPB) void bn_add_relu_fwd_kernel(const float* __restrict__ u, const float* __restrict__ sc,
                                                              const float* __restrict__ sh, int res_kind,
                                                              const float* __restrict__ r, const float* __restrict__ rsc,
                                                              const float* __restrict__ rsh, float* __restrict__ y,
                                                              int64_t n, int64_t ldm, unsigned char* __restrict__ mask = nullptr,
                                                              unsigned* __restrict__ amax = nullptr) {
  const int c = blockIdx.y;
  const float a = sc[c], b = sh[c];
  const float ra = (res_kind == 2) ? rsc[c] : 1.f, rb = (res_kind == 2) ? rsh[c] : 0.f;
  const int64_t base = (int64_t)c * ldm;
  unsigned am = 0u;   // bits of the largest |y|
  for (int64_t i = ((int64_t)blockIdx.x * TPB + threadIdx.x) * VEC; i < n; i += (int64_t)gridDim.x * TPB * VEC) {
    float uv[VEC], rv[VEC], o[VEC];
    ld<VEC>(u + base + i, uv);
    if (res_kind) ld<VEC>(r + base + i, rv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float z = fmaf(uv[j], a, b);
      if (res_kind) z += fmaf(rv[j], ra, rb);
      o[j] = z < 0.f ? 0.f : z;   // ReLU that keeps NaN (fmaxf would return 0: a diverged step must not look finite downstream)
      if (amax) am = amax_bits(am, o[j]);   // uniform
    }
    st<VEC>(y + base + i, o);
    if (VEC == 4 && mask) {   // uniform
      unsigned mb = 0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) mb |= (o[j] > 0.f ? 1u : 0u) << j;
      mask[(base + i) >> 2] = (unsigned char)mb;
    }
  }
  if (amax) block_amax(am, amax);
}

template <int VEC, bool TAIL = false>
__global__ __launch_bounds__(TPB) void bn_add_relu_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                     const float* __restrict__ u, const float* __restrict__ r,
                                                                     const float* __restrict__ mu_p, const float* __restrict__ mr_p,
                                                                     float* __restrict__ partials, int64_t n, int64_t ldm,
                                                                     const sar_bn_tail tail = sar_bn_tail(),
                                                                     const unsigned char* __restrict__ mask = nullptr) {
  const int c = blockIdx.y;
  const float mu = mu_p ? mu_p[c] : 0.f, mr = (r && mr_p) ? mr_p[c] : 0.f;
  const int64_t base = (int64_t)c * ldm;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int64_t i = ((int64_t)blockIdx.x * TPB + threadIdx.x) * VEC; i < n; i += (int64_t)gridDim.x * TPB * VEC) {
    float g[VEC], yv[VEC], uv[VEC], rv[VEC];
    ld<VEC>(dy + base + i, g);
    unsigned mb = 0;
    if (VEC == 4 && mask) {   // uniform
      mb = mask[(base + i) >> 2];
    } else {
      ld<VEC>(y + base + i, yv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) mb |= (yv[j] > 0.f ? 1u : 0u) << j;
    }
    ld<VEC>(u + base + i, uv);
    if (r) ld<VEC>(r + base + i, rv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float dz = ((mb >> j) & 1u) ? g[j] : 0.f;
      acc[0] += dz;
      acc[1] = fmaf(dz, uv[j] - mu, acc[1]);
      if (r) acc[2] = fmaf(dz, rv[j] - mr, acc[2]);
    }
  }
  __shared__ float red[4 * 3];
  block_sum<3>(acc, red);
  if (threadIdx.x == 0) {
    float* pp = partials + ((int64_t)c * gridDim.x + blockIdx.x) * 4;
    if constexpr (TAIL) {
      bn_tail_store(pp, acc[0]);
      bn_tail_store(pp + 1, acc[1]);
      bn_tail_store(pp + 2, acc[2]);
    } else {
      pp[0] = acc[0];
      pp[1] = acc[1];
      pp[2] = acc[2];
      pp[3] = 0.f;
    }
  }
  if constexpr (TAIL) {   // the last workgroup of the channel finalises it (sar_bn_tail)
    __shared__ int last;
    if (threadIdx.x == 0) last = bn_tail_last_arriver(tail.ticket + c, (int)gridDim.x) ? 1 : 0;
    __syncthreads();
    if (last && threadIdx.x < 64) bn_tail_channel(partials, (int)gridDim.x, c, tail, mu_p, mr_p, r != nullptr);
  }
}

template <int VEC>
__global__ __launch_bounds__(TPB) void bn_add_relu_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ u, const float* __restrict__ r,
    const float* __restrict__ k1, const float* __restrict__ k2, const float* __restrict__ k3, const float* __restrict__ rk1,
    const float* __restrict__ rk2, const float* __restrict__ rk3, float* du, float* dr, float* dz_out, int64_t n,
    int64_t ldm, const unsigned char* __restrict__ mask = nullptr, unsigned* __restrict__ amax = nullptr,
    unsigned* __restrict__ amax_r = nullptr) {
  const int c = blockIdx.y;
  const int64_t base = (int64_t)c * ldm;
  const float a1 = k1[c], a2 = k2[c], a3 = k3[c];
  const float b1 = dr ? rk1[c] : 0.f, b2 = dr ? rk2[c] : 0.f, b3 = dr ? rk3[c] : 0.f;
  unsigned am = 0u, amr = 0u;
  for (int64_t i = ((int64_t)blockIdx.x * TPB + threadIdx.x) * VEC; i < n; i += (int64_t)gridDim.x * TPB * VEC) {
    float g[VEC], yv[VEC], uv[VEC], rv[VEC], o1[VEC], o2[VEC], o3[VEC];
    ld<VEC>(dy + base + i, g);
    unsigned mb = 0;
    if (VEC == 4 && mask) {   // uniform
      mb = mask[(base + i) >> 2];
    } else {
      ld<VEC>(y + base + i, yv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) mb |= (yv[j] > 0.f ? 1u : 0u) << j;
    }
    ld<VEC>(u + base + i, uv);
    if (dr) ld<VEC>(r + base + i, rv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float dz = ((mb >> j) & 1u) ? g[j] : 0.f;
      o3[j] = dz;
      o1[j] = fmaf(a1, dz, fmaf(a2, uv[j], a3));
      if (amax) am = amax_bits(am, o1[j]);   // uniform
      if (dr) {
        o2[j] = fmaf(b1, dz, fmaf(b2, rv[j], b3));
        if (amax_r) amr = amax_bits(amr, o2[j]);   // uniform
      }
    }
    st<VEC>(du + base + i, o1);
    if (dr) st<VEC>(dr + base + i, o2);
    if (dz_out) st<VEC>(dz_out + base + i, o3);
  }
  if (amax) block_amax(am, amax);
  if (amax_r && dr) {
    __syncthreads();   // block_amax's LDS words are re-used
    block_amax(amr, amax_r);
  }
}

template <int VEC>
__global__ __launch_bounds__(TPB) void affine2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ k1, const float* __restrict__ k2,
                                                      const float* __restrict__ k3, float* out, int64_t n, int64_t ldm,
                                                      unsigned* __restrict__ amax = nullptr) {
  const int c = blockIdx.y;
  const int64_t base = (int64_t)c * ldm;
  const float a1 = k1[c], a2 = k2[c], a3 = k3[c];
  unsigned am = 0u;
  for (int64_t i = ((int64_t)blockIdx.x * TPB + threadIdx.x) * VEC; i < n; i += (int64_t)gridDim.x * TPB * VEC) {
    float av[VEC], bv[VEC], o[VEC];
    ld<VEC>(a + base + i, av);
    ld<VEC>(b + base + i, bv);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      o[j] = fmaf(a1, av[j], fmaf(a2, bv[j], a3));
      if (amax) am = amax_bits(am, o[j]);   // uniform
    }
    st<VEC>(out + base + i, o);
  }
  if (amax) block_amax(am, amax);
}

inline bool vec4_ok(int64_t n, int64_t ldm, std::initializer_list<const void*> ptrs) {
  if ((n & 3) || (ldm & 3)) return false;
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

// one vector chunk per thread: the plain "no loop" shape streams fastest on this part (tools/stream_bench.hip:
// 5.85 TB/s against 5.36 for ~8 grid-stride iterations per thread); the kernels keep their loops for rows longer
// than 65535 * TPB * vec elements
inline int row_blocks(int64_t n, int vec) {
  int64_t per = (int64_t)TPB * vec;
  int64_t b = (n + per - 1) / per;
  if (b < 1) b = 1;
  if (b > 65535) b = 65535;
  return (int)b;
}

// ------------------------------------------------------------------------------------ head
// one wave per (sample n, channel c): 4 independent partial sums per lane, no LDS, no barrier
__global__ __launch_bounds__(TPB) void pool_fwd_kernel(const float* __restrict__ y, int64_t ldm, int span, float inv,
                                                       int C, int N, float* __restrict__ feat) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6), c = blockIdx.y;
  if (n >= N) return;
  const float* p = y + (int64_t)c * ldm + (int64_t)n * span;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = lane;
  for (; i + 192 < span; i += 256) {
    a0 += p[i];
    a1 += p[i + 64];
    a2 += p[i + 128];
    a3 += p[i + 192];
  }
  for (; i < span; i += 64) a0 += p[i];
  const float t = wave_sum((a0 + a1) + (a2 + a3));
  if (lane == 0) feat[(int64_t)n * C + c] = t * inv;
}

// one workgroup per sample: the channel range is split over the 4 waves (two chains per lane), partials combined in
// a fixed order
__global__ __launch_bounds__(TPB) void fc_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ W,
                                                     const float* __restrict__ bias, int C, int K, float* __restrict__ logits) {
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* f = feat + (int64_t)n * C;
  const int cq = (C + 3) / 4;
  const int cb = wave * cq, ce = (cb + cq < C) ? cb + cq : C;
  __shared__ float part[4][64];
  for (int k0 = 0; k0 < K; k0 += 64) {
    const int k = k0 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (k < K) {
      int c = cb;
      for (; c + 1 < ce; c += 2) {
        s0 = fmaf(f[c], W[(int64_t)c * K + k], s0);
        s1 = fmaf(f[c + 1], W[(int64_t)(c + 1) * K + k], s1);
      }
      if (c < ce) s0 = fmaf(f[c], W[(int64_t)c * K + k], s0);
    }
    part[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && k < K)
      logits[(int64_t)n * K + k] = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) + (bias ? bias[k] : 0.f);
    __syncthreads();
  }
}

__global__ __launch_bounds__(TPB) void softmax_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         int N, int K, float inv_gbs, float* loss_sum,
                                                         float* __restrict__ dlogits, float* __restrict__ probs) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float lsum = 0.f;
  for (int n = wave; n < N; n += 4) {
    const float* row = logits + (int64_t)n * K;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, row[k]);
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    float se = 0.f;
    for (int k = lane; k < K; k += 64) se += expf(row[k] - mx);
    se = wave_sum(se);
    // a label outside [0, K) (e.g. 120-class data with --num-classes 60) must not read out of bounds and must not
    // pass silently: the loss (and this row's gradient) become NaN, which the host loops surface
    const int64_t lab64 = labels[n];
    const bool lab_ok = lab64 >= 0 && lab64 < K;
    const int lab = lab_ok ? (int)lab64 : 0;
    const float lse = logf(se) + mx;
    if (lane == 0) lsum += lab_ok ? (lse - row[lab]) : NAN;
    const float inv = 1.f / se;
    for (int k = lane; k < K; k += 64) {
      const float p = expf(row[k] - mx) * inv;
      if (probs) probs[(int64_t)n * K + k] = p;
      if (dlogits) dlogits[(int64_t)n * K + k] = lab_ok ? (p - (k == lab ? 1.f : 0.f)) * inv_gbs : NAN;
    }
  }
  __shared__ float red[4];
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0 && loss_sum) loss_sum[0] = (red[0] + red[1] + red[2] + red[3]) * inv_gbs;
}

__global__ void fc_bwd_w_kernel(const float* __restrict__ feat, const float* __restrict__ dl, int N, int C, int K,
                                float* __restrict__ dW, float* __restrict__ dbias) {
  const int c = blockIdx.x;  // blockIdx.x == C handles the bias
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float s = 0.f;
    if (c < C) {
      for (int n = 0; n < N; ++n) s = fmaf(feat[(int64_t)n * C + c], dl[(int64_t)n * K + k], s);
      dW[(int64_t)c * K + k] = s;
    } else {
      for (int n = 0; n < N; ++n) s += dl[(int64_t)n * K + k];
      dbias[k] = s;
    }
  }
}

__global__ void fc_bwd_x_kernel(const float* __restrict__ W, const float* __restrict__ dl, int C, int K,
                                float* __restrict__ dfeat) {
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(dl[(int64_t)n * K + k], W[(int64_t)c * K + k], s);
    dfeat[(int64_t)n * C + c] = s;
  }
}

__global__ __launch_bounds__(TPB) void pool_bwd_kernel(const float* __restrict__ dfeat, int64_t ldm, int span, float inv,
                                                       int C, float* __restrict__ dy) {
  const int n = blockIdx.x, c = blockIdx.y;
  const float val = dfeat[(int64_t)n * C + c] * inv;
  float* p = dy + (int64_t)c * ldm + (int64_t)n * span;
  for (int i = threadIdx.x; i < span; i += TPB) p[i] = val;
}

// ------------------------------------------------------------------------------------ optimizer / misc
__global__ __launch_bounds__(TPB) void sgd_nesterov_kernel(float* __restrict__ w, float* __restrict__ v,
                                                           const float* __restrict__ g, int64_t n,
                                                           const float* __restrict__ lr_dev, float momentum) {
  const float lr = lr_dev[0];
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const float gi = g[i];
    const float vn = momentum * v[i] - lr * gi;
    v[i] = vn;
    w[i] = w[i] + momentum * vn - lr * gi;
  }
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
  __shared__ float tile[32][33];
  const int64_t boff = (int64_t)blockIdx.z * R * Cc;
  int r = blockIdx.y * 32 + threadIdx.y, c = blockIdx.x * 32 + threadIdx.x;
  for (int j = 0; j < 32; j += 8)
    if (r + j < R && c < Cc) tile[threadIdx.y + j][threadIdx.x] = in[boff + (int64_t)(r + j) * Cc + c];
  __syncthreads();
  c = blockIdx.x * 32 + threadIdx.y;
  r = blockIdx.y * 32 + threadIdx.x;
  for (int j = 0; j < 32; j += 8)
    if (c + j < Cc && r < R) out[boff + (int64_t)(c + j) * R + r] = tile[threadIdx.x][threadIdx.y + j];
}

// many 3-d re-layouts in one launch (blockIdx.y = item): out[dst_off + (i*d1 + j)*d2 + k] = in[src_off + i*s0 + j*s1 + k*s2]
__global__ __launch_bounds__(TPB) void permute3_batch_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             const sar_permute_item* __restrict__ items) {
  const sar_permute_item it = items[blockIdx.y];
  const int64_t n = (int64_t)it.d0 * it.d1 * it.d2;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int kk = (int)(i % it.d2);
    const int64_t r = i / it.d2;
    const int j = (int)(r % it.d1), ii = (int)(r / it.d1);
    out[it.dst_off + i] = in[it.src_off + ii * it.s0 + j * it.s1 + kk * it.s2];
  }
}

}  // namespace

// ====================================================================================== C ABI
extern "C" int sar_permute3_batch_f32(const float* in, float* out, const sar_permute_item* items, int nitems, int64_t max_elems,
                                      sar_stream_t s) {
  SAR_REQUIRE(in && out && items && nitems > 0 && nitems <= 65535 && max_elems > 0, "sar_permute3_batch: bad arguments");
  int64_t blocks = (max_elems + TPB - 1) / TPB;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(permute3_batch_kernel, dim3((unsigned)blocks, nitems), dim3(TPB), 0, as_stream(s), in, out, items);
  SAR_LAUNCH_CHECK("sar_permute3_batch_f32");
  return 0;
}

extern "C" int sar_bn_finalize_f32(const float* partials, int nparts, int C, double count, float eps, float momentum,
                                   int unbiased_running, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float* mean, float* rstd, float* scale, float* shift,
                                   sar_stream_t s) {
  SAR_REQUIRE(partials && nparts > 0 && C > 0 && count > 0 && scale && shift, "sar_bn_finalize: bad arguments");
  SAR_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "sar_bn_finalize: running stats mismatch");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(TPB), 0, as_stream(s), partials, nparts, count, eps, momentum,
                     unbiased_running, gamma, beta, running_mean, running_var, mean, rstd, scale, shift);
  SAR_LAUNCH_CHECK("sar_bn_finalize_f32");
  return 0;
}

extern "C" int sar_bn_eval_affine_f32(const float* gamma, const float* beta, const float* running_mean,
                                      const float* running_var, float eps, int C, float* scale, float* shift,
                                      sar_stream_t s) {
  SAR_REQUIRE(running_mean && running_var && scale && shift && C > 0, "sar_bn_eval_affine: bad arguments");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(s), gamma, beta, running_mean,
                     running_var, eps, C, scale, shift);
  SAR_LAUNCH_CHECK("sar_bn_eval_affine_f32");
  return 0;
}

extern "C" int sar_bn_bwd_finalize_f32(const float* partials, int nparts, int64_t chan_stride, int64_t part_stride,
                                       int off1, int off2, int centered, int C, double count, const float* gamma, const float* mean,
                                       const float* rstd, float* dgamma, float* dbeta, float* k1, float* k2, float* k3,
                                       sar_stream_t s) {
  SAR_REQUIRE(partials && nparts > 0 && C > 0 && count > 0 && mean && rstd, "sar_bn_bwd_finalize: bad arguments");
  SAR_REQUIRE((k1 == nullptr) == (k2 == nullptr) && (k1 == nullptr) == (k3 == nullptr), "sar_bn_bwd_finalize: k1/k2/k3");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(TPB), 0, as_stream(s), partials, nparts, chan_stride,
                     part_stride, off1, off2, centered, count, gamma, mean, rstd, dgamma, dbeta, k1, k2, k3);
  SAR_LAUNCH_CHECK("sar_bn_bwd_finalize_f32");
  return 0;
}

static int data_bn_check(const float* x, int N, int C, int T, int V, int M) {
  SAR_REQUIRE(x && N > 0 && C > 0 && T > 0 && V > 0 && M > 0, "sar_data_bn: bad sizes");
  SAR_REQUIRE(V * M <= TPB, "sar_data_bn: V*M = %d exceeds %d", V * M, TPB);
  return 0;
}

extern "C" int sar_data_bn_stats_f32(const float* x, int N, int C, int T, int V, int M, const int32_t* bone_parent,
                                     int motion, float* partials, sar_stream_t s) {
  int rc = data_bn_check(x, N, C, T, V, M);
  if (rc) return rc;
  SAR_REQUIRE(partials, "sar_data_bn_stats: null partials");
  hipLaunchKernelGGL(data_bn_reduce_kernel<false>, dim3(N * C), dim3(TPB), 0, as_stream(s), x, N, C, T, V, M,
                     bone_parent, motion, (const float*)nullptr, (int64_t)0, (const float*)nullptr, partials);
  SAR_LAUNCH_CHECK("sar_data_bn_stats_f32");
  return 0;
}

extern "C" int sar_data_bn_apply_f32(const float* x, int N, int C, int T, int V, int M, const int32_t* bone_parent,
                                     int motion, const float* scale, const float* shift, float* out, int64_t ld_out, sar_stream_t s) {
  int rc = data_bn_check(x, N, C, T, V, M);
  if (rc) return rc;
  SAR_REQUIRE(scale && shift && out && ld_out >= (int64_t)N * M * T * V, "sar_data_bn_apply: bad arguments");
  hipLaunchKernelGGL(data_bn_apply_kernel, dim3(N * C), dim3(TPB), 0, as_stream(s), x, N, C, T, V, M, bone_parent, motion,
                     scale, shift, out, ld_out);
  SAR_LAUNCH_CHECK("sar_data_bn_apply_f32");
  return 0;
}

extern "C" int sar_data_bn_bwd_reduce_f32(const float* x, int N, int C, int T, int V, int M, const int32_t* bone_parent,
                                          int motion, const float* dy, int64_t ld_dy, const float* mean, float* partials, sar_stream_t s) {
  int rc = data_bn_check(x, N, C, T, V, M);
  if (rc) return rc;
  SAR_REQUIRE(dy && partials && ld_dy >= (int64_t)N * M * T * V, "sar_data_bn_bwd_reduce: bad arguments");
  hipLaunchKernelGGL(data_bn_reduce_kernel<true>, dim3(N * C), dim3(TPB), 0, as_stream(s), x, N, C, T, V, M, bone_parent,
                     motion, dy, ld_dy, mean, partials);
  SAR_LAUNCH_CHECK("sar_data_bn_bwd_reduce_f32");
  return 0;
}

extern "C" int sar_data_bn_apply_cn8(const float* x, int N, int C, int T, int V, int M, const int32_t* bone_parent, int motion,
                                     const float* scale, const float* shift, void* out, int64_t ld_out, sar_stream_t s) {
  int rc = data_bn_check(x, N, C, T, V, M);
  if (rc) return rc;
  SAR_REQUIRE(scale && shift && out && C <= 8 && ld_out >= (int64_t)N * M * T * V && ((uintptr_t)out & 15) == 0,
              "sar_data_bn_apply_cn8: bad arguments (C <= 8, 16-byte aligned output)");
  hipLaunchKernelGGL(data_bn_apply_cn8_kernel, dim3(N, 16), dim3(TPB), 0, as_stream(s), x, N, C, T, V, M, bone_parent, motion,
                     scale, shift, (uint4*)out);
  SAR_LAUNCH_CHECK("sar_data_bn_apply_cn8");
  return 0;
}

extern "C" int sar_data_bn_bwd_reduce_cn8(const float* x, int N, int C, int T, int V, int M, const int32_t* bone_parent,
                                          int motion, const void* dy, int64_t ld_dy, const float* mean, float* partials,
                                          sar_stream_t s) {
  int rc = data_bn_check(x, N, C, T, V, M);
  if (rc) return rc;
  SAR_REQUIRE(dy && partials && C <= 8 && ld_dy >= (int64_t)N * M * T * V, "sar_data_bn_bwd_reduce_cn8: bad arguments");
  hipLaunchKernelGGL((data_bn_reduce_kernel<true, true>), dim3(N * C), dim3(TPB), 0, as_stream(s), x, N, C, T, V, M, bone_parent,
                     motion, (const float*)dy, ld_dy, mean, partials);
  SAR_LAUNCH_CHECK("sar_data_bn_bwd_reduce_cn8");
  return 0;
}

extern "C" int sar_bn_add_relu_fwd_f32(const float* u, const float* sc, const float* sh, int res_kind, const float* r,
                                       const float* rsc, const float* rsh, float* y, int C, int64_t n, int64_t ldm,
                                       sar_stream_t s) {
  SAR_REQUIRE(u && sc && sh && y && C > 0 && n > 0 && ldm >= n, "sar_bn_add_relu_fwd: bad arguments");
  SAR_REQUIRE(res_kind >= 0 && res_kind <= 2 && (res_kind == 0 || r) && (res_kind != 2 || (rsc && rsh)),
              "sar_bn_add_relu_fwd: residual arguments");
  if (vec4_ok(n, ldm, {u, r, y})) {
    hipLaunchKernelGGL(bn_add_relu_fwd_kernel<4>, dim3(row_blocks(n, 4), C), dim3(TPB), 0, as_stream(s), u, sc, sh,
                       res_kind, r, rsc, rsh, y, n, ldm);
  } else {
    hipLaunchKernelGGL(bn_add_relu_fwd_kernel<1>, dim3(row_blocks(n, 1), C), dim3(TPB), 0, as_stream(s), u, sc, sh,
                       res_kind, r, rsc, rsh, y, n, ldm);
  }
  SAR_LAUNCH_CHECK("sar_bn_add_relu_fwd_f32");
  return 0;
}

// the block tail with a 1-bit ReLU mask (include/sar_hip.h): rows of 4-element groups only (n, ld multiples of 4, 16-byte aligned)
static int bn_add_relu_fwd_mask_impl(const float* u, const float* sc, const float* sh, int res_kind, const float* r,
                                     const float* rsc, const float* rsh, float* y, void* mask, uint32_t* amax, int C, int64_t n, int64_t ldm,
                                     sar_stream_t s) {
  SAR_REQUIRE(u && sc && sh && y && mask && C > 0 && n > 0 && ldm >= n, "sar_bn_add_relu_fwd_mask: bad arguments");
  SAR_REQUIRE(res_kind >= 0 && res_kind <= 2 && (res_kind == 0 || r) && (res_kind != 2 || (rsc && rsh)),
              "sar_bn_add_relu_fwd_mask: residual arguments");
  if (!vec4_ok(n, ldm, {u, r, y})) {
    sar_set_error("sar_bn_add_relu_fwd_mask: rows must be 16-byte aligned multiples of 4 elements (n=%lld, ld=%lld)", (long long)n, (long long)ldm);
    return SAR_E_UNSUP;
  }
  hipLaunchKernelGGL(bn_add_relu_fwd_kernel<4>, dim3(row_blocks(n, 4), C), dim3(TPB), 0, as_stream(s), u, sc, sh, res_kind, r, rsc,
                     rsh, y, n, ldm, (unsigned char*)mask, amax);
  SAR_LAUNCH_CHECK("sar_bn_add_relu_fwd_mask_f32");
  return 0;
}
extern "C" int sar_bn_add_relu_fwd_mask_f32(const float* u, const float* sc, const float* sh, int res_kind, const float* r,
                                            const float* rsc, const float* rsh, float* y, void* mask, int C, int64_t n, int64_t ldm,
                                            sar_stream_t s) {
  return bn_add_relu_fwd_mask_impl(u, sc, sh, res_kind, r, rsc, rsh, y, mask, nullptr, C, n, ldm, s);
}
extern "C" int sar_bn_add_relu_fwd_mask_amax_f32(const float* u, const float* sc, const float* sh, int res_kind, const float* r,
                                                 const float* rsc, const float* rsh, float* y, void* mask, uint32_t* amax_y, int C,
                                                 int64_t n, int64_t ldm, sar_stream_t s) {
  SAR_REQUIRE(amax_y != nullptr, "sar_bn_add_relu_fwd_mask_amax: null cell");
  return bn_add_relu_fwd_mask_impl(u, sc, sh, res_kind, r, rsc, rsh, y, mask, amax_y, C, n, ldm, s);
}

extern "C" int sar_bn_add_relu_bwd_reduce_mask_f32(const float* dy, const void* mask, const float* u, const float* r,
                                                   const float* mu, const float* mr, float* partials, int nparts, int C, int64_t n,
                                                   int64_t ldm, sar_stream_t s) {
  SAR_REQUIRE(dy && mask && u && partials && nparts > 0 && nparts <= 65535 && C > 0 && n > 0 && ldm >= n,
              "sar_bn_add_relu_bwd_reduce_mask: bad arguments");
  if (!vec4_ok(n, ldm, {dy, u, r})) {
    sar_set_error("sar_bn_add_relu_bwd_reduce_mask: rows must be 16-byte aligned multiples of 4 elements");
    return SAR_E_UNSUP;
  }
  hipLaunchKernelGGL((bn_add_relu_bwd_reduce_kernel<4, false>), dim3(nparts, C), dim3(TPB), 0, as_stream(s), dy, (const float*)nullptr,
                     u, r, mu, mr, partials, n, ldm, sar_bn_tail(), (const unsigned char*)mask);
  SAR_LAUNCH_CHECK("sar_bn_add_relu_bwd_reduce_mask_f32");
  return 0;
}

static int bn_add_relu_bwd_apply_mask_impl(const float* dy, const void* mask, const float* u, const float* r,
                                           const float* k1, const float* k2, const float* k3, const float* rk1,
                                           const float* rk2, const float* rk3, float* du, float* dr, float* dz_out, uint32_t* amax,
                                           int C, int64_t n, int64_t ldm, sar_stream_t s, uint32_t* amax_r = nullptr) {
  SAR_REQUIRE(dy && mask && u && k1 && k2 && k3 && du && C > 0 && n > 0 && ldm >= n, "sar_bn_add_relu_bwd_apply_mask: bad arguments");
  SAR_REQUIRE(!dr || (r && rk1 && rk2 && rk3), "sar_bn_add_relu_bwd_apply_mask: residual arguments");
  if (!vec4_ok(n, ldm, {dy, u, r, du, dr, dz_out})) {
    sar_set_error("sar_bn_add_relu_bwd_apply_mask: rows must be 16-byte aligned multiples of 4 elements");
    return SAR_E_UNSUP;
  }
  hipLaunchKernelGGL(bn_add_relu_bwd_apply_kernel<4>, dim3(row_blocks(n, 4), C), dim3(TPB), 0, as_stream(s), dy, (const float*)nullptr,
                     u, r, k1, k2, k3, rk1, rk2, rk3, du, dr, dz_out, n, ldm, (const unsigned char*)mask, amax, amax_r);
  SAR_LAUNCH_CHECK("sar_bn_add_relu_bwd_apply_mask_f32");
  return 0;
}
extern "C" int sar_bn_add_relu_bwd_apply_mask_f32(const float* dy, const void* mask, const float* u, const float* r,
                                                  const float* k1, const float* k2, const float* k3, const float* rk1,
                                                  const float* rk2, const float* rk3, float* du, float* dr, float* dz_out,
                                                  int C, int64_t n, int64_t ldm, sar_stream_t s) {
  return bn_add_relu_bwd_apply_mask_impl(dy, mask, u, r, k1, k2, k3, rk1, rk2, rk3, du, dr, dz_out, nullptr, C, n, ldm, s);
}
extern "C" int sar_bn_add_relu_bwd_apply_mask_amax_f32(const float* dy, const void* mask, const float* u, const float* r,
                                                       const float* k1, const float* k2, const float* k3, const float* rk1,
                                                       const float* rk2, const float* rk3, float* du, float* dr, float* dz_out,
                                                       uint32_t* amax_du, uint32_t* amax_dr, int C, int64_t n, int64_t ldm, sar_stream_t s) {
  SAR_REQUIRE(amax_du != nullptr, "sar_bn_add_relu_bwd_apply_mask_amax: null cell");
  SAR_REQUIRE(!amax_dr || dr, "sar_bn_add_relu_bwd_apply_mask_amax: amax_dr without dr");
  return bn_add_relu_bwd_apply_mask_impl(dy, mask, u, r, k1, k2, k3, rk1, rk2, rk3, du, dr, dz_out, amax_du, C, n, ldm, s, amax_dr);
}

extern "C" int sar_bn_add_relu_bwd_reduce_f32(const float* dy, const float* y, const float* u, const float* r,
                                              const float* mu, const float* mr, float* partials, int nparts, int C, int64_t n, int64_t ldm, sar_stream_t s) {
  SAR_REQUIRE(dy && y && u && partials && nparts > 0 && nparts <= 65535 && C > 0 && n > 0 && ldm >= n,
              "sar_bn_add_relu_bwd_reduce: bad arguments");
  if (vec4_ok(n, ldm, {dy, y, u, r})) {
    hipLaunchKernelGGL(bn_add_relu_bwd_reduce_kernel<4>, dim3(nparts, C), dim3(TPB), 0, as_stream(s), dy, y, u, r,
                       mu, mr, partials, n, ldm);
  } else {
    hipLaunchKernelGGL(bn_add_relu_bwd_reduce_kernel<1>, dim3(nparts, C), dim3(TPB), 0, as_stream(s), dy, y, u, r,
                       mu, mr, partials, n, ldm);
  }
  SAR_LAUNCH_CHECK("sar_bn_add_relu_bwd_reduce_f32");
  return 0;
}

extern "C" int sar_bn_add_relu_bwd_reduce_tail_f32(const float* dy, const float* y, const float* u, const float* r,
                                                   const float* mu, const float* mr, float* partials, int nparts, int C, int64_t n,
                                                   int64_t ldm, const sar_bn_tail* tail, sar_stream_t s) {
  SAR_REQUIRE(dy && y && u && partials && nparts > 0 && nparts <= 65535 && C > 0 && n > 0 && ldm >= n,
              "sar_bn_add_relu_bwd_reduce_tail: bad arguments");
  SAR_REQUIRE(tail && tail->ticket && tail->count > 0 && tail->rstd && tail->k1 && tail->k2 && tail->k3,
              "sar_bn_add_relu_bwd_reduce_tail: ticket, count, rstd and k1..k3 are required");
  SAR_REQUIRE(!r || (tail->rrstd && tail->rk1 && tail->rk2 && tail->rk3), "sar_bn_add_relu_bwd_reduce_tail: residual-branch outputs");
  SAR_REQUIRE(((uintptr_t)partials & 15) == 0, "sar_bn_add_relu_bwd_reduce_tail: partials must be 16-byte aligned");
  if (vec4_ok(n, ldm, {dy, y, u, r})) {
    hipLaunchKernelGGL((bn_add_relu_bwd_reduce_kernel<4, true>), dim3(nparts, C), dim3(TPB), 0, as_stream(s), dy, y, u, r, mu, mr,
                       partials, n, ldm, *tail);
  } else {
    hipLaunchKernelGGL((bn_add_relu_bwd_reduce_kernel<1, true>), dim3(nparts, C), dim3(TPB), 0, as_stream(s), dy, y, u, r, mu, mr,
                       partials, n, ldm, *tail);
  }
  SAR_LAUNCH_CHECK("sar_bn_add_relu_bwd_reduce_tail_f32");
  return 0;
}

extern "C" int sar_bn_add_relu_bwd_apply_f32(const float* dy, const float* y, const float* u, const float* r,
                                             const float* k1, const float* k2, const float* k3, const float* rk1,
                                             const float* rk2, const float* rk3, float* du, float* dr, float* dz_out,
                                             int C, int64_t n, int64_t ldm, sar_stream_t s) {
  SAR_REQUIRE(dy && y && u && k1 && k2 && k3 && du && C > 0 && n > 0 && ldm >= n, "sar_bn_add_relu_bwd_apply: bad arguments");
  SAR_REQUIRE(!dr || (r && rk1 && rk2 && rk3), "sar_bn_add_relu_bwd_apply: residual arguments");
  if (vec4_ok(n, ldm, {dy, y, u, r, du, dr, dz_out})) {
    hipLaunchKernelGGL(bn_add_relu_bwd_apply_kernel<4>, dim3(row_blocks(n, 4), C), dim3(TPB), 0, as_stream(s), dy, y, u,
                       r, k1, k2, k3, rk1, rk2, rk3, du, dr, dz_out, n, ldm);
  } else {
    hipLaunchKernelGGL(bn_add_relu_bwd_apply_kernel<1>, dim3(row_blocks(n, 1), C), dim3(TPB), 0, as_stream(s), dy, y, u,
                       r, k1, k2, k3, rk1, rk2, rk3, du, dr, dz_out, n, ldm);
  }
  SAR_LAUNCH_CHECK("sar_bn_add_relu_bwd_apply_f32");
  return 0;
}

static int affine2_impl(const float* a, const float* b, const float* k1, const float* k2, const float* k3,
                        float* out, uint32_t* amax, int C, int64_t n, int64_t ldm, sar_stream_t s) {
  SAR_REQUIRE(a && b && k1 && k2 && k3 && out && C > 0 && n > 0 && ldm >= n, "sar_affine2: bad arguments");
  if (vec4_ok(n, ldm, {a, b, out})) {
    hipLaunchKernelGGL(affine2_kernel<4>, dim3(row_blocks(n, 4), C), dim3(TPB), 0, as_stream(s), a, b, k1, k2, k3, out, n,
                       ldm, amax);
  } else {
    hipLaunchKernelGGL(affine2_kernel<1>, dim3(row_blocks(n, 1), C), dim3(TPB), 0, as_stream(s), a, b, k1, k2, k3, out, n,
                       ldm, amax);
  }
  SAR_LAUNCH_CHECK("sar_affine2_f32");
  return 0;
}
extern "C" int sar_affine2_f32(const float* a, const float* b, const float* k1, const float* k2, const float* k3,
                               float* out, int C, int64_t n, int64_t ldm, sar_stream_t s) {
  return affine2_impl(a, b, k1, k2, k3, out, nullptr, C, n, ldm, s);
}
extern "C" int sar_affine2_amax_f32(const float* a, const float* b, const float* k1, const float* k2, const float* k3,
                                    float* out, uint32_t* amax_out, int C, int64_t n, int64_t ldm, sar_stream_t s) {
  SAR_REQUIRE(amax_out != nullptr, "sar_affine2_amax: null cell");
  return affine2_impl(a, b, k1, k2, k3, out, amax_out, C, n, ldm, s);
}

extern "C" int sar_pool_fwd_f32(const float* y, int64_t ldm, int C, int B, int TV, int Mp, float* feat, sar_stream_t s) {
  SAR_REQUIRE(y && feat && C > 0 && B > 0 && TV > 0 && Mp > 0 && B % Mp == 0 && ldm >= (int64_t)B * TV,
              "sar_pool_fwd: bad arguments");
  const int span = Mp * TV;
  const int N = B / Mp;
  hipLaunchKernelGGL(pool_fwd_kernel, dim3((N + TPB / 64 - 1) / (TPB / 64), C), dim3(TPB), 0, as_stream(s), y, ldm, span,
                     1.0f / (float)span, C, N, feat);
  SAR_LAUNCH_CHECK("sar_pool_fwd_f32");
  return 0;
}

extern "C" int sar_fc_fwd_f32(const float* feat, const float* W, const float* bias, int N, int C, int K, float* logits,
                              sar_stream_t s) {
  SAR_REQUIRE(feat && W && logits && N > 0 && C > 0 && K > 0, "sar_fc_fwd: bad arguments");
  hipLaunchKernelGGL(fc_fwd_kernel, dim3(N), dim3(TPB), 0, as_stream(s), feat, W, bias, C, K, logits);
  SAR_LAUNCH_CHECK("sar_fc_fwd_f32");
  return 0;
}

extern "C" int sar_softmax_ce_f32(const float* logits, const int64_t* labels, int N, int K, float inv_global_batch,
                                  float* loss_sum, float* dlogits, float* probs, sar_stream_t s) {
  SAR_REQUIRE(logits && labels && N > 0 && K > 0, "sar_softmax_ce: bad arguments");
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(TPB), 0, as_stream(s), logits, labels, N, K, inv_global_batch,
                     loss_sum, dlogits, probs);
  SAR_LAUNCH_CHECK("sar_softmax_ce_f32");
  return 0;
}

extern "C" int sar_fc_bwd_f32(const float* feat, const float* W, const float* dlogits, int N, int C, int K, float* dW,
                              float* dbias, float* dfeat, sar_stream_t s) {
  // (round 6) dW == dbias == NULL: only the feature gradient (the launch on the backward chain); dfeat == NULL: only dW / dbias (they
  // feed nothing but the optimizer: an engine may issue them on another stream)
  SAR_REQUIRE(feat && W && dlogits && N > 0 && C > 0 && K > 0 && ((dW != nullptr) == (dbias != nullptr)) && (dW || dfeat),
              "sar_fc_bwd: bad arguments");
  if (dW) hipLaunchKernelGGL(fc_bwd_w_kernel, dim3(C + 1), dim3(128), 0, as_stream(s), feat, dlogits, N, C, K, dW, dbias);
  if (dfeat) hipLaunchKernelGGL(fc_bwd_x_kernel, dim3(N), dim3(256), 0, as_stream(s), W, dlogits, C, K, dfeat);
  SAR_LAUNCH_CHECK("sar_fc_bwd_f32");
  return 0;
}

extern "C" int sar_pool_bwd_f32(const float* dfeat, int64_t ldm, int C, int B, int TV, int Mp, float* dy, sar_stream_t s) {
  SAR_REQUIRE(dfeat && dy && C > 0 && B > 0 && TV > 0 && Mp > 0 && B % Mp == 0 && ldm >= (int64_t)B * TV,
              "sar_pool_bwd: bad arguments");
  const int span = Mp * TV;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(B / Mp, C), dim3(TPB), 0, as_stream(s), dfeat, ldm, span, 1.0f / (float)span,
                     C, dy);
  SAR_LAUNCH_CHECK("sar_pool_bwd_f32");
  return 0;
}

extern "C" int sar_sgd_nesterov_f32(float* w, float* v, const float* g, int64_t n, const float* lr_dev, float momentum,
                                    sar_stream_t s) {
  SAR_REQUIRE(w && v && g && lr_dev && n > 0, "sar_sgd_nesterov: bad arguments");
  int blocks = (int)((n + TPB - 1) / TPB);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(blocks), dim3(TPB), 0, as_stream(s), w, v, g, n, lr_dev, momentum);
  SAR_LAUNCH_CHECK("sar_sgd_nesterov_f32");
  return 0;
}

extern "C" int sar_transpose_f32(const float* in, float* out, int batch, int R, int Cc, sar_stream_t s) {
  SAR_REQUIRE(in && out && batch > 0 && R > 0 && Cc > 0, "sar_transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3((Cc + 31) / 32, (R + 31) / 32, batch), dim3(32, 8), 0, as_stream(s), in, out,
                     R, Cc);
  SAR_LAUNCH_CHECK("sar_transpose_f32");
  return 0;
}

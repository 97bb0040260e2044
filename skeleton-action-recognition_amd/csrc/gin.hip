// gin.hip -- the pieces of the graph isomorphism convolution (models/gcn.py:112-163 GraphIsoConvTD, used by
// models/stgin.py:24-25) that the ST-GCN kernels do not already cover.  fp32 CN layout.
//
//   x' = einsum('nctv,kvw->nkctw', x, concat(A, diag(1 + epsilon)))      graph_dense.hip with the table built here
//   per slice k:  Conv2D(h,1x1) -> BN -> ReLU -> Conv2D(h,1x1) -> BN -> ReLU         sar_conv_gemm_f32 (taps = 1; the first
//                                                                                     BN+ReLU folded into the second conv)
//   s = sum_k relu(bn2_k(a2_k))                                                       gin_sum_fwd (here) + the BatchNorm
//                                                                                     partial sums of s for the tgcn's BN
// The K branches of a layer are stacked along the channel axis ([K*C][n] tensors, K*C-channel BatchNorm state), so that
// one launch covers all of them.  HBM-bound element-wise kernels: 16 B per lane, one partial per workgroup and row, no
// atomics (deterministic).
#include "sar_common.h"

namespace {

constexpr int TPB = 256;

template <int VEC> __device__ __forceinline__ void ld(const float* p, float (&r)[VEC]) {
  if (VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(p); r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w; }
  else r[0] = *p;
}
template <int VEC> __device__ __forceinline__ void st(float* p, const float (&r)[VEC]) {
  if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  else *p = r[0];
}

template <int NV> __device__ __forceinline__ void block_sum(float (&v)[NV], float* red /* [4][NV] */) {
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

inline bool vec4_ok(int64_t n, std::initializer_list<int64_t> lds, std::initializer_list<const void*> ptrs) {
  if (n & 3) return false;
  for (int64_t l : lds)
    if (l & 3) return false;
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

inline int row_blocks(int64_t n, int vec) {
  int64_t per = (int64_t)TPB * vec;
  int64_t b = (n + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

// s[c] = sum_k relu(a[k C + c] * scale[k C + c] + shift[k C + c]);  partials[c][block][2] = (sum s, sum s^2)
template <int VEC>
__global__ __launch_bounds__(TPB) void gin_sum_fwd_kernel(const float* __restrict__ a, int64_t ld_a, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int K, int C, int64_t n,
                                                          float* __restrict__ s, int64_t ld_s, float* __restrict__ partials) {
  const int c = blockIdx.y;
  float acc[2] = {0.f, 0.f};
  for (int64_t i = ((int64_t)blockIdx.x * TPB + threadIdx.x) * VEC; i < n; i += (int64_t)gridDim.x * TPB * VEC) {
    float o[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o[j] = 0.f;
    for (int k = 0; k < K; ++k) {
      const int r = k * C + c;
      const float sc = scale[r], sh = shift[r];
      float v[VEC];
      ld<VEC>(a + (int64_t)r * ld_a + i, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] += fmaxf(fmaf(v[j], sc, sh), 0.f);
    }
    st<VEC>(s + (int64_t)c * ld_s + i, o);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      acc[0] += o[j];
      acc[1] = fmaf(o[j], o[j], acc[1]);
    }
  }
  if (partials) {
    __shared__ float red[4 * 2];
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
      float* pp = partials + ((int64_t)c * gridDim.x + blockIdx.x) * 2;
      pp[0] = acc[0];
      pp[1] = acc[1];
    }
  }
}

// row r = k C + c:  dz = ds[c] where relu'(bn(a[r])) else 0.
// APPLY 0: partials[r][block][2] = (sum dz, sum dz (a[r] - mean[r]));  APPLY 1: da[r] = k1[r] dz + k2[r] a[r] + k3[r]
template <int VEC, int APPLY>
__global__ __launch_bounds__(TPB) void gin_bwd_kernel(const float* __restrict__ ds, int64_t ld_ds, const float* __restrict__ a,
                                                      int64_t ld_a, const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ p1, const float* __restrict__ p2,
                                                      const float* __restrict__ p3, int C, int64_t n, float* out, int64_t ld_out) {
  const int r = blockIdx.y, c = r % C;
  const float sc = scale[r], sh = shift[r];
  const float q1 = p1[r], q2 = APPLY ? p2[r] : 0.f, q3 = APPLY ? p3[r] : 0.f;      // APPLY 0: q1 = mean
  float acc[2] = {0.f, 0.f};
  for (int64_t i = ((int64_t)blockIdx.x * TPB + threadIdx.x) * VEC; i < n; i += (int64_t)gridDim.x * TPB * VEC) {
    float g[VEC], v[VEC], o[VEC];
    ld<VEC>(ds + (int64_t)c * ld_ds + i, g);
    ld<VEC>(a + (int64_t)r * ld_a + i, v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float dz = fmaf(v[j], sc, sh) > 0.f ? g[j] : 0.f;
      if (APPLY) {
        o[j] = fmaf(q1, dz, fmaf(q2, v[j], q3));
      } else {
        acc[0] += dz;
        acc[1] = fmaf(dz, v[j] - q1, acc[1]);
      }
    }
    if (APPLY) st<VEC>(out + (int64_t)r * ld_out + i, o);
  }
  if (!APPLY) {
    __shared__ float red[4 * 2];
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
      float* pp = out + ((int64_t)r * gridDim.x + blockIdx.x) * 2;
      pp[0] = acc[0];
      pp[1] = acc[1];
    }
  }
}

// table[k][a][b] = A[k][b][a] (k < K-1), table[K-1] = (1 + eps) I;  scale[c] = 1 + eps
__global__ void gin_adjacency_kernel(const float* __restrict__ A, int Km1, int V, const float* __restrict__ eps, float* table,
                                     float* scale, int C, float* slice_scale) {
  const float e = 1.f + eps[0];
  if (slice_scale && blockIdx.x == 0 && (int)threadIdx.x <= Km1) slice_scale[threadIdx.x] = (int)threadIdx.x < Km1 ? 1.f : e;
  const int VV = V * V, total = (Km1 + 1) * VV;
  if (table)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
      const int k = i / VV, rem = i - k * VV, r = rem / V, col = rem - r * V;
      table[i] = (k < Km1) ? A[k * VV + col * V + r] : (r == col ? e : 0.f);
    }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < C; i += gridDim.x * blockDim.x) scale[i] = e;
}

// deps = <G, W> (fp64 accumulation, fixed order);  G *= (1 + eps)
__global__ __launch_bounds__(1024) void gin_eps_grad_kernel(float* G, const float* __restrict__ W, int64_t n,
                                                            const float* __restrict__ eps, float* deps) {
  const float e = 1.f + eps[0];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float g = G[i];
    acc += (double)g * (double)W[i];
    G[i] = g * e;
  }
  __shared__ double red[16];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    deps[0] = (float)t;
  }
}

// Frame-local weighted gathers for a FIXED sparse adjacency (<= 4 entries per joint: sar_amd/graph_tables.py), the form the
// ST-GCN kernels fold into their operand loads, as stand-alone HBM-bound passes for the graph isomorphism convolution:
//   MODE 0 (sum):    out[m, (t,w)] = sum_k scale[k] sum_j wt[k][w][j] in[k F + m, (t, idx[k][w][j])]  (+ add[m, (t,w)])
//   MODE 1 (expand): out[k F + m, (t,v)] = sum_j wt[k][v][j] in[m, (t, idx[k][v][j])]
// One thread per (row m, column); the K V 4 table entries sit in LDS; neighbours of a joint lie in the same frame (<= 128 B
// away), so the gathers are cache hits and the pass moves each tensor once.
constexpr int GG_KMAX = 8, GG_VMAX = 32, GG_NZ = 4;
struct GGnz { int nz[GG_KMAX]; };      // entries per joint actually used by slice k (1 for an identity slice): uniform loop bounds
template <int MODE>
__global__ __launch_bounds__(TPB) void graph_gather_kernel(const float* __restrict__ in, int64_t ld_in, const int32_t* __restrict__ idx,
                                                           const float* __restrict__ wt, const float* __restrict__ scale, int K,
                                                           int F, int V, int64_t n, float* __restrict__ out, int64_t ld_out,
                                                           const float* __restrict__ add, int64_t ld_add, const GGnz nzs) {
  __shared__ int sidx[GG_KMAX * GG_VMAX * GG_NZ];
  __shared__ float swt[GG_KMAX * GG_VMAX * GG_NZ];
  for (int i = threadIdx.x; i < K * V * GG_NZ; i += TPB) {
    sidx[i] = idx[i];
    swt[i] = wt[i] * (scale ? scale[i / (V * GG_NZ)] : 1.f);
  }
  __syncthreads();
  const int m = blockIdx.y;
  for (int64_t col = (int64_t)blockIdx.x * TPB + threadIdx.x; col < n; col += (int64_t)gridDim.x * TPB) {
    const int w = (int)(col % V);
    const int64_t fb = col - w;
    if (MODE == 0) {
      float acc = add ? add[(int64_t)m * ld_add + col] : 0.f;
      for (int k = 0; k < K; ++k) {
        const float* row = in + (int64_t)(k * F + m) * ld_in + fb;
        const int e = (k * V + w) * GG_NZ;
        const int nz = nzs.nz[k];
        for (int j = 0; j < nz; ++j) acc = fmaf(swt[e + j], row[sidx[e + j]], acc);
      }
      out[(int64_t)m * ld_out + col] = acc;
    } else {
      const float* row = in + (int64_t)m * ld_in + fb;
      for (int k = 0; k < K; ++k) {
        const int e = (k * V + w) * GG_NZ;
        const int nz = nzs.nz[k];
        float acc = 0.f;
        for (int j = 0; j < nz; ++j) acc = fmaf(swt[e + j], row[sidx[e + j]], acc);
        out[(int64_t)(k * F + m) * ld_out + col] = acc;
      }
    }
  }
}

}  // namespace

extern "C" int sar_gin_nparts(int64_t n) { return n > 0 ? row_blocks(n, (n & 3) ? 1 : 4) : SAR_E_ARG; }

extern "C" int sar_gin_sum_fwd_f32(const float* a, int64_t ld_a, const float* scale, const float* shift, int K, int C, int64_t n,
                                   float* s, int64_t ld_s, float* partials, sar_stream_t st_) {
  SAR_REQUIRE(a && scale && shift && s && K > 0 && C > 0 && n > 0 && ld_a >= n && ld_s >= n, "sar_gin_sum_fwd: bad arguments");
  const bool v4 = vec4_ok(n, {ld_a, ld_s}, {a, s});
  SAR_REQUIRE(v4 || (n & 3) || !partials, "sar_gin_sum_fwd: with partials and n %% 4 == 0 the rows must be 16-byte aligned");
  const int nb = sar_gin_nparts(n);
  if (v4)
    hipLaunchKernelGGL(gin_sum_fwd_kernel<4>, dim3(nb, C), dim3(TPB), 0, as_stream(st_), a, ld_a, scale, shift, K, C, n, s, ld_s,
                       partials);
  else
    hipLaunchKernelGGL(gin_sum_fwd_kernel<1>, dim3(row_blocks(n, 1), C), dim3(TPB), 0, as_stream(st_), a, ld_a, scale, shift, K,
                       C, n, s, ld_s, partials);
  SAR_LAUNCH_CHECK("sar_gin_sum_fwd_f32");
  return 0;
}

extern "C" int sar_gin_bwd_reduce_f32(const float* ds, int64_t ld_ds, const float* a, int64_t ld_a, const float* scale,
                                      const float* shift, const float* mean, int K, int C, int64_t n, float* partials,
                                      sar_stream_t st_) {
  SAR_REQUIRE(ds && a && scale && shift && mean && partials && K > 0 && C > 0 && n > 0 && ld_a >= n && ld_ds >= n,
              "sar_gin_bwd_reduce: bad arguments");
  const bool v4 = vec4_ok(n, {ld_a, ld_ds}, {a, ds});
  SAR_REQUIRE(v4 || (n & 3), "sar_gin_bwd_reduce: with n %% 4 == 0 the rows must be 16-byte aligned");
  const int nb = sar_gin_nparts(n);
  if (v4)
    hipLaunchKernelGGL((gin_bwd_kernel<4, 0>), dim3(nb, K * C), dim3(TPB), 0, as_stream(st_), ds, ld_ds, a, ld_a, scale, shift,
                       mean, (const float*)nullptr, (const float*)nullptr, C, n, partials, (int64_t)0);
  else
    hipLaunchKernelGGL((gin_bwd_kernel<1, 0>), dim3(nb, K * C), dim3(TPB), 0, as_stream(st_), ds, ld_ds, a, ld_a, scale, shift,
                       mean, (const float*)nullptr, (const float*)nullptr, C, n, partials, (int64_t)0);
  SAR_LAUNCH_CHECK("sar_gin_bwd_reduce_f32");
  return 0;
}

extern "C" int sar_gin_bwd_apply_f32(const float* ds, int64_t ld_ds, const float* a, int64_t ld_a, const float* scale,
                                     const float* shift, const float* k1, const float* k2, const float* k3, int K, int C,
                                     int64_t n, float* da, int64_t ld_da, sar_stream_t st_) {
  SAR_REQUIRE(ds && a && scale && shift && k1 && k2 && k3 && da && K > 0 && C > 0 && n > 0 && ld_a >= n && ld_ds >= n && ld_da >= n,
              "sar_gin_bwd_apply: bad arguments");
  if (vec4_ok(n, {ld_a, ld_ds, ld_da}, {a, ds, da}))
    hipLaunchKernelGGL((gin_bwd_kernel<4, 1>), dim3(row_blocks(n, 4), K * C), dim3(TPB), 0, as_stream(st_), ds, ld_ds, a, ld_a, scale,
                       shift, k1, k2, k3, C, n, da, ld_da);
  else
    hipLaunchKernelGGL((gin_bwd_kernel<1, 1>), dim3(row_blocks(n, 1), K * C), dim3(TPB), 0, as_stream(st_), ds, ld_ds, a, ld_a, scale,
                       shift, k1, k2, k3, C, n, da, ld_da);
  SAR_LAUNCH_CHECK("sar_gin_bwd_apply_f32");
  return 0;
}

extern "C" int sar_gin_adjacency_f32(const float* A, int Km1, int V, const float* eps, float* table, float* scale, int C,
                                     float* slice_scale, sar_stream_t st_) {
  SAR_REQUIRE((A || Km1 == 0 || !table) && Km1 >= 0 && Km1 < 8 && V > 0 && V <= 32 && eps && (scale || C == 0) && C >= 0,
              "sar_gin_adjacency: bad arguments");
  hipLaunchKernelGGL(gin_adjacency_kernel, dim3(8), dim3(256), 0, as_stream(st_), A, Km1, V, eps, table, scale, C, slice_scale);
  SAR_LAUNCH_CHECK("sar_gin_adjacency_f32");
  return 0;
}

extern "C" int sar_gin_eps_grad_f32(float* G, const float* W, int64_t n, const float* eps, float* deps, sar_stream_t st_) {
  SAR_REQUIRE(G && W && n > 0 && eps && deps, "sar_gin_eps_grad: bad arguments");
  hipLaunchKernelGGL(gin_eps_grad_kernel, dim3(1), dim3(1024), 0, as_stream(st_), G, W, n, eps, deps);
  SAR_LAUNCH_CHECK("sar_gin_eps_grad_f32");
  return 0;
}

static GGnz gg_nz(const int32_t* nz, int K) {
  GGnz r;
  for (int k = 0; k < GG_KMAX; ++k) r.nz[k] = (nz && k < K && nz[k] >= 1 && nz[k] <= GG_NZ) ? nz[k] : GG_NZ;
  return r;
}

static int gg_check(const char* who, const float* in, const int32_t* idx, const float* wt, float* out, int K, int F, int V, int64_t n,
                    int64_t ld_in, int64_t ld_out) {
  SAR_REQUIRE(in && idx && wt && out && K > 0 && K <= GG_KMAX && F > 0 && F <= 65535 && V > 0 && V <= GG_VMAX && n > 0 && n % V == 0 &&
                  ld_in >= n && ld_out >= n,
              "%s: bad arguments (K <= %d, V <= %d, whole frames)", who, GG_KMAX, GG_VMAX);
  return 0;
}

extern "C" int sar_graph_gather_sum_f32(const float* in, int64_t ld_in, const int32_t* idx, const float* wt, const int32_t* nz,
                                        const float* scale, int K, int F, int V, int64_t n, float* out, int64_t ld_out,
                                        const float* add, int64_t ld_add, sar_stream_t st_) {
  if (int rc = gg_check("sar_graph_gather_sum", in, idx, wt, out, K, F, V, n, ld_in, ld_out)) return rc;
  SAR_REQUIRE(!add || ld_add >= n, "sar_graph_gather_sum: bad add");
  hipLaunchKernelGGL(graph_gather_kernel<0>, dim3(row_blocks(n, 1), F), dim3(TPB), 0, as_stream(st_), in, ld_in, idx, wt, scale, K, F, V,
                     n, out, ld_out, add, ld_add, gg_nz(nz, K));
  SAR_LAUNCH_CHECK("sar_graph_gather_sum_f32");
  return 0;
}

extern "C" int sar_graph_gather_expand_f32(const float* in, int64_t ld_in, const int32_t* idx, const float* wt, const int32_t* nz, int K,
                                           int F, int V, int64_t n, float* out, int64_t ld_out, sar_stream_t st_) {
  if (int rc = gg_check("sar_graph_gather_expand", in, idx, wt, out, K, F, V, n, ld_in, ld_out)) return rc;
  hipLaunchKernelGGL(graph_gather_kernel<1>, dim3(row_blocks(n, 1), F), dim3(TPB), 0, as_stream(st_), in, ld_in, idx, wt,
                     (const float*)nullptr, K, F, V, n, out, ld_out, (const float*)nullptr, (int64_t)0, gg_nz(nz, K));
  SAR_LAUNCH_CHECK("sar_graph_gather_expand_f32");
  return 0;
}

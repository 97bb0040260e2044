// graph_dense.hip -- the adjacency contraction with a DENSE (trainable) adjacency:
//
//   out[m, (t,w)] = sum_k sum_v y[k F + m, (t,v)] A[k, v, w]                   einsum 'nkctv,kvw->nctw', models/gcn.py:207-208,
//   dy[k F + m, (t,v)] = sum_w dout[m, (t,w)] A[k, v, w]                       236-237 (AdjGraphConv: A is a trainable variable,
//   dA[k, v, w] = sum_{m, b, t} y[k F + m, (b,t,v)] dout[m, (b,t,w)]           main_gnn.py:228-232 un-freezes it by name)
// The same two contractions serve the graph isomorphism convolution (models/gcn.py:149-156, 'nctv,kvw->nkctw'): bwd_data
// with the transposed table of gin.hip produces the K stacked x . A_k, fwd (+ `add`, the skip-path gradient) its gradient.
//
// The fixed-adjacency kernels (conv_gemm.hip) fold A into the operand load as <= 4-entry gather lists and never
// materialise the 3F-channel tensor y; a trained adjacency is dense (625 entries per slice), so this path keeps the
// reference's order -- 1x1 convolution to 3F channels (sar_conv_gemm_f32, TEMPORAL, taps = 1), then the contraction here.
// fp32 CN layout.  One workgroup = one output row m x a tile of whole frames; the three y rows of the tile and A (K V V
// floats) sit in LDS, a thread owns one (frame, joint) and runs the K V-term dot product; BatchNorm partial sums of the
// result (sum, sum of squares per row and tile) come out of the same pass.  The dA reduction gives every workgroup one
// (k, row block) and a contiguous range of tiles; partial V x V blocks go to slabs reduced by sar_slab_reduce_f32 in a
// fixed order (deterministic, no atomics).
#include "sar_common.h"

namespace {

constexpr int TPB = 256;
constexpr int GD_VMAX = 32;      // joints per frame supported (NTU: 25)
constexpr int GD_FT = 8;         // frames per tile (FT * V <= TPB for V <= 32)

// out row m, tile of frames.  MODE 0: out[m] = sum_k y[kF+m] . A_k (+ stats);  MODE 1: dy[kF+m] = dout[m] . A_k^T
template <int MODE>
__global__ __launch_bounds__(TPB) void graph_dense_kernel(const float* __restrict__ in, int64_t ld_in, const float* __restrict__ A,
                                                          float* __restrict__ out, int64_t ld_out, int K, int F, int V,
                                                          int64_t nframes, float* __restrict__ partials, int nparts,
                                                          const float* __restrict__ add, int64_t ld_add) {
  extern __shared__ float sm[];
  float* As = sm;                       // [K][V][V]
  float* Ys = sm + K * V * V;           // MODE 0: [K][GD_FT * V]; MODE 1: [GD_FT * V]
  const int m = blockIdx.y;
  const int64_t f0 = (int64_t)blockIdx.x * GD_FT;
  const int nf = (int)((f0 + GD_FT <= nframes) ? GD_FT : nframes - f0);
  const int ncol = nf * V;
  for (int i = threadIdx.x; i < K * V * V; i += TPB) As[i] = A[i];
  const int64_t col0 = f0 * V;
  if (MODE == 0) {
    for (int i = threadIdx.x; i < K * ncol; i += TPB) {
      const int k = i / ncol, c = i - k * ncol;
      Ys[k * GD_FT * V + c] = in[(int64_t)(k * F + m) * ld_in + col0 + c];
    }
  } else {
    for (int i = threadIdx.x; i < ncol; i += TPB) Ys[i] = in[(int64_t)m * ld_in + col0 + i];
  }
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  if ((int)threadIdx.x < ncol) {
    const int fr = threadIdx.x / V, j = threadIdx.x - fr * V;
    if (MODE == 0) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) {
        const float* yr = Ys + k * GD_FT * V + fr * V;
        const float* ak = As + k * V * V + j;        // column w = j
        for (int v = 0; v < V; ++v) acc = fmaf(yr[v], ak[v * V], acc);
      }
      if (add) acc += add[(int64_t)m * ld_add + col0 + threadIdx.x];
      out[(int64_t)m * ld_out + col0 + threadIdx.x] = acc;
      s1 = acc;
      s2 = acc * acc;
    } else {
      const float* dr = Ys + fr * V;
      for (int k = 0; k < K; ++k) {
        const float* ak = As + k * V * V + j * V;    // row v = j
        float acc = 0.f;
        for (int w = 0; w < V; ++w) acc = fmaf(dr[w], ak[w], acc);
        out[(int64_t)(k * F + m) * ld_out + col0 + threadIdx.x] = acc;
      }
    }
  }
  if (MODE == 0 && partials) {
    __shared__ float red[8];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) {
      red[threadIdx.x >> 6] = s1;
      red[4 + (threadIdx.x >> 6)] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float* pp = partials + ((int64_t)m * nparts + blockIdx.x) * 2;
      pp[0] = (red[0] + red[1]) + (red[2] + red[3]);
      pp[1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
  }
}

// dA: workgroup (split, k, row block of RB rows); thread -> entries (v, w) = e, e + TPB, ... of the V x V block
constexpr int DA_RB = 8;
__global__ __launch_bounds__(TPB) void graph_dA_kernel(const float* __restrict__ y, int64_t ld_y, const float* __restrict__ dout,
                                                       int64_t ld_d, int K, int F, int V, int64_t nframes, int nsplit,
                                                       float* __restrict__ slab) {
  __shared__ float Ys[DA_RB][GD_FT * GD_VMAX], Ds[DA_RB][GD_FT * GD_VMAX];
  const int split = blockIdx.x, k = blockIdx.y, rb = blockIdx.z;
  const int VV = V * V;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};   // VV <= 4 * TPB
  const int64_t ntiles = (nframes + GD_FT - 1) / GD_FT;
  const int64_t per = (ntiles + nsplit - 1) / nsplit;
  const int64_t t_lo = split * per, t_hi = (t_lo + per < ntiles) ? t_lo + per : ntiles;
  const int m0 = rb * DA_RB;
  for (int64_t tile = t_lo; tile < t_hi; ++tile) {
    const int64_t f0 = tile * GD_FT;
    const int nf = (int)((f0 + GD_FT <= nframes) ? GD_FT : nframes - f0);
    const int ncol = nf * V;
    const int64_t col0 = f0 * V;
    __syncthreads();
    for (int i = threadIdx.x; i < DA_RB * ncol; i += TPB) {
      const int r = i / ncol, c = i - r * ncol;
      const bool ok = m0 + r < F;
      Ys[r][c] = ok ? y[(int64_t)(k * F + m0 + r) * ld_y + col0 + c] : 0.f;
      Ds[r][c] = ok ? dout[(int64_t)(m0 + r) * ld_d + col0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = threadIdx.x + q * TPB;
      if (e < VV) {
        const int v = e / V, w = e - v * V;
        float a = acc[q];
        for (int r = 0; r < DA_RB; ++r)
          for (int fr = 0; fr < nf; ++fr) a = fmaf(Ys[r][fr * V + v], Ds[r][fr * V + w], a);
        acc[q] = a;
      }
    }
  }
  float* out = slab + ((int64_t)split * gridDim.z + rb) * K * VV + (int64_t)k * VV;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = threadIdx.x + q * TPB;
    if (e < VV) out[e] = acc[q];
  }
}

}  // namespace

static int gd_check(const char* who, const float* in, const float* A, float* out, int K, int F, int V, int64_t nframes) {
  SAR_REQUIRE(in && A && out && K > 0 && K <= 8 && F > 0 && V > 0 && V <= GD_VMAX && nframes > 0, "%s: bad arguments (V <= %d, K <= 8)", who,
              GD_VMAX);
  return 0;
}

extern "C" int sar_graph_dense_nparts(int64_t nframes) { return (int)((nframes + GD_FT - 1) / GD_FT); }

extern "C" int sar_graph_dense_fwd_f32(const float* y, int64_t ld_y, const float* A, float* out, int64_t ld_out, int K, int F,
                                       int V, int64_t nframes, float* partials, const float* add, int64_t ld_add,
                                       sar_stream_t s) {
  if (int rc = gd_check("sar_graph_dense_fwd", y, A, out, K, F, V, nframes)) return rc;
  SAR_REQUIRE(ld_y >= nframes * V && ld_out >= nframes * V && (!add || ld_add >= nframes * V),
              "sar_graph_dense_fwd: leading dimension smaller than frames * V");
  const int ntiles = sar_graph_dense_nparts(nframes);
  const size_t lds = (size_t)(K * V * V + K * GD_FT * V) * 4;
  hipLaunchKernelGGL(graph_dense_kernel<0>, dim3(ntiles, F), dim3(TPB), lds, as_stream(s), y, ld_y, A, out, ld_out, K, F, V,
                     nframes, partials, ntiles, add, ld_add);
  SAR_LAUNCH_CHECK("sar_graph_dense_fwd_f32");
  return 0;
}

extern "C" int sar_graph_dense_bwd_data_f32(const float* dout, int64_t ld_d, const float* A, float* dy, int64_t ld_dy, int K, int F,
                                            int V, int64_t nframes, sar_stream_t s) {
  if (int rc = gd_check("sar_graph_dense_bwd_data", dout, A, dy, K, F, V, nframes)) return rc;
  SAR_REQUIRE(ld_d >= nframes * V && ld_dy >= nframes * V, "sar_graph_dense_bwd_data: leading dimension smaller than frames * V");
  const int ntiles = sar_graph_dense_nparts(nframes);
  const size_t lds = (size_t)(K * V * V + GD_FT * V) * 4;
  hipLaunchKernelGGL(graph_dense_kernel<1>, dim3(ntiles, F), dim3(TPB), lds, as_stream(s), dout, ld_d, A, dy, ld_dy, K, F, V,
                     nframes, (float*)nullptr, 0, (const float*)nullptr, (int64_t)0);
  SAR_LAUNCH_CHECK("sar_graph_dense_bwd_data_f32");
  return 0;
}

extern "C" int64_t sar_graph_dense_dadj_slab_floats(int K, int F, int V, int nsplit) {
  if (K <= 0 || F <= 0 || V <= 0 || nsplit <= 0) return SAR_E_ARG;
  return (int64_t)nsplit * ((F + DA_RB - 1) / DA_RB) * K * V * V;
}

extern "C" int sar_graph_dense_dadj_f32(const float* y, int64_t ld_y, const float* dout, int64_t ld_d, int K, int F, int V,
                                      int64_t nframes, int nsplit, float* slab, float* dA, sar_stream_t s) {
  SAR_REQUIRE(y && dout && slab && dA && K > 0 && K <= 8 && F > 0 && V > 0 && V <= GD_VMAX && nframes > 0 && nsplit > 0 &&
                  nsplit <= 65535,
              "sar_graph_dense_dA: bad arguments");
  const int nrb = (F + DA_RB - 1) / DA_RB;
  hipLaunchKernelGGL(graph_dA_kernel, dim3(nsplit, K, nrb), dim3(TPB), 0, as_stream(s), y, ld_y, dout, ld_d, K, F, V, nframes,
                     nsplit, slab);
  SAR_LAUNCH_CHECK("sar_graph_dense_dadj_f32");
  const int64_t n = (int64_t)K * V * V;
  return sar_slab_reduce_f32(slab, nsplit * nrb, n, n, dA, s);
}

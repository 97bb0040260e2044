// conv_gemm.hip -- fused graph / temporal convolution as an implicit GEMM on fp32 MFMA (gfx950).
//
//   out[m, n] = sum_tap sum_c W[tap][c][m] * OP_tap(pro(src))[c, n] (+ bias) ; epilogue
//
// Replaces (see include/sar_hip.h): GraphConvTD (models/gcn.py:199-209), the BN/ReLU/Conv2D 9x1
// chain (models/stgcn.py:27-36), the strided 1x1 residual conv (models/stgcn.py:47-54) and, with
// transposed weights / gather lists, their data gradients (tape.gradient, main_gnn.py:233).
//
// Design (MI355X):
//  * Activations are [C][B*T*V] matrices, so the GEMM N axis is contiguous in HBM and every global
//    access is a coalesced 128-B half-wave segment.
//  * A workgroup tile covers FT whole frames of ONE sequence (FT*V <= TILE_N columns: 125 of 128 or
//    250 of 256 for V = 25).  Tiles never straddle a sequence, so the temporal halo / TF-SAME zero
//    padding is materialised once by the LDS stager and the graph gather never leaves the tile:
//    the inner loop has no bounds logic.
//  * The B operand of v_mfma_f32_32x32x2_f32 is built straight from the staged src tile in LDS:
//    temporal taps are 9 shifted ds_read_b32 of the same row; the graph op is a <=4-entry weighted
//    gather per adjacency slice (25x25 adjacency has 73 non-zeros), i.e. x.A_k is applied on the
//    Cin side and never materialised (the reference materialises the 3F-channel intermediate).
//  * BatchNorm+ReLU of the producer is folded into the stager (pro_scale/pro_shift), BatchNorm
//    statistics of the result are reduced in the epilogue (half-wave shuffles -> per-tile partials,
//    no atomics: deterministic), so BN costs no extra pass over HBM.
//  * fp32 MFMA issues one 32x32x2 per 64 cycles per SIMD: 2 LDS dwords per MFMA per lane at most,
//    so the kernel is matrix-pipe bound, not LDS bound; occupancy (2-3 workgroups/CU) hides staging.
#include "sar_common.h"

namespace {

#ifndef SAR_KC
#define SAR_KC 4
#endif
constexpr int KC = SAR_KC;  // src channels staged per main-loop iteration (KC/2 MFMA k-steps)

struct ConvK {
  sar_conv_desc d;
  int FT, TPS, NF, RW, SROW, nparts;
  int w_vec;   // weight rows may be read as aligned float4
};

template <int MODE, int TRANSPOSED, int TAPS, int MS, int NS, int WM, int WN, int NZ0, int NZ1, int NZ2>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvK k) {
  constexpr int BM = 32 * MS * WM;
  constexpr int NZMAX = 4;
  constexpr int NZ[3] = {NZ0, NZ1, NZ2};
  // staging maps (no per-element division):
  //  W tile: TAPS*KC rows of BM floats, one float4 per lane, 1024/BM rows per pass
  //  S tile: KC rows, 256/KC consecutive lanes per row and pass, SJMAX passes cover RW columns
  constexpr int WROWS = TAPS * KC;
  constexpr int WRPP = 1024 / BM;                       // W rows per pass
  constexpr int WIT = (WROWS + WRPP - 1) / WRPP;        // W passes
  constexpr int SLPR = 256 / KC;                        // lanes per S row
  constexpr int SJMAX = ((MODE == SAR_CONV_GRAPH) ? 32 * NS * WN : (NS * WN == 4 ? 448 : 704)) / SLPR;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert((KC == 8 || KC == 4) && 256 % KC == 0, "S stager: 256/KC lanes per src channel row");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const sar_conv_desc& d = k.d;
  float* Wl = smem;                      // [TAPS][KC][BM]   (16-byte aligned rows)
  float* S = smem + WROWS * BM;          // [KC][SROW]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int V = d.V;
  const int tile = blockIdx.x;
  const int b = tile / k.TPS;
  const int t0 = (tile - b * k.TPS) * k.FT;
  const int m0 = blockIdx.y * BM;

  // ---- per-lane column geometry (fixed for the whole kernel)
  bool colok[NS];
  int64_t coln[NS];  // output column index n
  int off[TAPS][NS];           // TEMPORAL: LDS column offset per tap
  unsigned vmask[NS];          // TEMPORAL transposed: tap validity bits
  int goff[3][NS][NZMAX];      // GRAPH: LDS column offset of each gather entry
  float gw[3][NS][NZMAX];      // GRAPH: weight of each gather entry
  float gcs[3][NS];            // GRAPH: colsum(A_k)[v] for the bias term

  int t_lo;
  if (MODE == SAR_CONV_GRAPH) t_lo = t0;
  else if (!TRANSPOSED) t_lo = t0 * d.stride - d.pad;
  else t_lo = floordiv(t0 + d.pad - (TAPS - 1), d.stride);

#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    int fo = p / V;
    const int v = p - fo * V;
    colok[ns] = (fo < k.FT) && (t0 + fo < d.T_out);
    if (!colok[ns]) fo = 0;
    coln[ns] = ((int64_t)b * d.T_out + (t0 + fo)) * V + v;
    vmask[ns] = 0;
    if (MODE == SAR_CONV_TEMPORAL) {
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp) {
        if (!TRANSPOSED) {
          off[tp][ns] = (fo * d.stride + tp) * V + v;
        } else {
          const int q = t0 + fo + d.pad - tp;
          const int to = floordiv(q, d.stride);
          const bool ok = (q - to * d.stride) == 0;
          vmask[ns] |= (ok ? 1u : 0u) << tp;
          off[tp][ns] = (to - t_lo) * V + v;
        }
      }
    } else {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        gcs[tp][ns] = d.g_colsum ? d.g_colsum[tp * V + v] : 0.f;
#pragma unroll
        for (int j = 0; j < NZMAX; ++j) {
          if (j < NZ[tp]) {
            goff[tp][ns][j] = fo * V + d.g_idx[(tp * V + v) * NZMAX + j];
            gw[tp][ns][j] = d.g_wt[(tp * V + v) * NZMAX + j];
          }
        }
      }
    }
  }

  f32x16 acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

  const int seq_len = d.T_src * V;
  const float* src_b = d.src + (int64_t)b * seq_len;
  const bool has_pro = d.pro_scale != nullptr;

  // ---- staging state: the next stage's global loads are issued BEFORE the MFMA phase of the current
  // stage and land in registers while the matrix pipe works; they are written to LDS after the barrier.
  const int w_m4 = (tid % (BM / 4)) * 4;     // float4 column of this lane inside a W row
  const int w_r0 = tid / (BM / 4);           // first W row of this lane
  const bool w_vec = k.w_vec != 0;
  const int s_row = tid / SLPR;              // S row (src channel inside the stage) of this lane
  const int s_c0 = tid % SLPR;
  float4 wreg[WIT];
  float sreg[SJMAX];
  float psc = 1.f, psh = 0.f;

  // Loads are UNCONDITIONAL (out-of-range lanes read a clamped, valid address) and nothing consumes the
  // loaded registers until store_lds(): a predicated load or an early select would make the compiler wait
  // for the data at the issue point and serialise the whole stage on memory latency.
  auto issue_loads = [&](int c0) {
    if (w_vec) {
#pragma unroll
      for (int i = 0; i < WIT; ++i) {
        const int row = w_r0 + i * WRPP;
        const int tp = row / KC, c = row % KC;
        const int cg = c0 + c, mg = m0 + w_m4;
        const bool ok = row < WROWS && cg < d.Kc && mg < d.M;
        const float* wp = ok ? d.W + (int64_t)tp * d.w_stride_tap + (int64_t)cg * d.w_stride_c + mg : d.W;
        wreg[i] = *reinterpret_cast<const float4*>(wp);
      }
    }
    const int cg = c0 + s_row;
    const bool rowok = cg < d.Kc;
    const float* sp = src_b + (int64_t)(rowok ? cg : 0) * d.ld_src;
#pragma unroll
    for (int j = 0; j < SJMAX; ++j) {
      const int col = s_c0 + SLPR * j;
      const int rabs = t_lo * V + col;
      const bool ok = rowok && col < k.RW && (unsigned)rabs < (unsigned)seq_len;
      sreg[j] = sp[ok ? rabs : 0];
    }
    if (has_pro) {
      psc = d.pro_scale[rowok ? cg : 0];
      psh = d.pro_shift[rowok ? cg : 0];
    }
  };

  auto store_lds = [&](int c0) {
    if (w_vec) {
#pragma unroll
      for (int i = 0; i < WIT; ++i) {
        const int row = w_r0 + i * WRPP;
        const int c = row % KC;
        const bool ok = (c0 + c) < d.Kc && (m0 + w_m4) < d.M;
        if (row < WROWS) *reinterpret_cast<float4*>(Wl + row * BM + w_m4) = ok ? wreg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {  // unaligned / M % 4 != 0 weights (3-channel layers only): plain strided copy
      for (int idx = tid; idx < WROWS * BM; idx += 256) {
        const int m = idx % BM, row = idx / BM;
        const int tp = row / KC, cg = c0 + row % KC, mg = m0 + m;
        Wl[idx] = (cg < d.Kc && mg < d.M) ? d.W[(int64_t)tp * d.w_stride_tap + (int64_t)cg * d.w_stride_c + mg] : 0.f;
      }
    }
    const bool rowok = (c0 + s_row) < d.Kc;
#pragma unroll
    for (int j = 0; j < SJMAX; ++j) {
      const int col = s_c0 + SLPR * j;
      const int rabs = t_lo * V + col;
      if (col < k.RW) {
        float val = sreg[j];
        if (has_pro) {   // folded BN(+ReLU)
          val = fmaf(val, psc, psh);
          if (d.pro_relu) val = fmaxf(val, 0.f);
        }
        // everything outside the sequence (temporal zero padding) or beyond Kc is exactly 0
        S[s_row * k.SROW + col] = (rowok && (unsigned)rabs < (unsigned)seq_len) ? val : 0.f;
      }
    }
  };

  issue_loads(0);
  for (int c0 = 0; c0 < d.Kc; c0 += KC) {
    store_lds(c0);
    __syncthreads();
    if (c0 + KC < d.Kc) issue_loads(c0 + KC);

#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
#pragma unroll
      for (int cc = 0; cc < KC; cc += 2) {
        float a[MS], bv[NS];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) a[ms] = Wl[(tp * KC + cc + hi) * BM + (wm * MS + ms) * 32 + l31];
        const float* Srow = S + (cc + hi) * k.SROW;
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          if (MODE == SAR_CONV_TEMPORAL) {
            float x = Srow[off[tp][ns]];
            if (TRANSPOSED) x = ((vmask[ns] >> tp) & 1u) ? x : 0.f;
            bv[ns] = x;
          } else {
            float x = gw[tp][ns][0] * Srow[goff[tp][ns][0]];
#pragma unroll
            for (int j = 1; j < NZMAX; ++j)
              if (j < NZ[tp]) x = fmaf(gw[tp][ns][j], Srow[goff[tp][ns][j]], x);
            bv[ns] = x;
          }
        }
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms], bv[ns], acc[ms][ns], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: bias, mask / add, store, BN partial reductions
  const int part = tile * WN + wn;
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (wm * MS + ms) * 32 + mfma_row(r, hi);
      const bool rowok = row < d.M;  // uniform over the 32 lanes of a half-wave
      float s1 = 0.f, s2 = 0.f;
      float asc = 0.f, ash = 0.f, amu = 0.f;
      if (d.epi == SAR_EPI_MASK && rowok) {
        asc = d.aux_scale[row];
        ash = d.aux_shift[row];
        if (d.aux_mean) amu = d.aux_mean[row];
      }
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        float val = acc[ms][ns][r];
        if (rowok && colok[ns]) {
          if (d.bias) {
            if (MODE == SAR_CONV_TEMPORAL) {
              val += d.bias[row];
            } else {
#pragma unroll
              for (int tp = 0; tp < 3; ++tp) val = fmaf(d.bias[tp * d.M + row], gcs[tp][ns], val);
            }
          }
          if (d.epi == SAR_EPI_STATS) {
            s1 += val;
            s2 = fmaf(val, val, s2);
          } else if (d.epi == SAR_EPI_MASK) {
            const float ax = d.aux[(int64_t)row * d.ld_aux + coln[ns]];
            val = (fmaf(ax, asc, ash) > 0.f) ? val : 0.f;
            s1 += val;
            s2 = fmaf(val, ax - amu, s2);
          } else if (d.epi == SAR_EPI_ADD) {
            val += d.aux[(int64_t)row * d.ld_aux + coln[ns]];
          }
          d.out[(int64_t)row * d.ld_out + coln[ns]] = val;
        }
      }
      if (d.epi == SAR_EPI_STATS || d.epi == SAR_EPI_MASK) {
        s1 = half_wave_sum(s1);
        s2 = half_wave_sum(s2);
        if (l31 == 0 && rowok) {
          float* pp = d.partials + ((int64_t)row * k.nparts + part) * 2;
          pp[0] = s1;
          pp[1] = s2;
        }
      }
    }
  }
}

template <int WN>
int tile_geometry(const sar_conv_desc& d, int NSv, ConvK& k) {
  const int tile_n = 32 * NSv * WN;
  k.FT = tile_n / d.V;
  if (k.FT < 1) return -1;
  if (k.FT > d.T_out) k.FT = d.T_out;
  k.TPS = (d.T_out + k.FT - 1) / k.FT;
  if (d.mode == SAR_CONV_GRAPH) k.NF = k.FT;
  else if (!d.transposed) k.NF = (k.FT - 1) * d.stride + d.taps;
  else k.NF = (k.FT - 1 + d.taps - 1) / d.stride + 2;
  k.RW = k.NF * d.V;
  k.SROW = k.RW;
  k.nparts = d.B * k.TPS * WN;
  k.w_vec = ((d.M & 3) == 0 && (d.w_stride_c & 3) == 0 && (d.w_stride_tap & 3) == 0 && ((uintptr_t)d.W & 15) == 0) ? 1 : 0;
  const int rwmax = (d.mode == SAR_CONV_GRAPH) ? 32 * NSv * WN : (NSv * WN == 4 ? 448 : 704);
  if (k.RW > rwmax) return -2;   // staged row does not fit the register prefetch (V too large)
  return 0;
}

template <int MODE, int TRANSPOSED, int TAPS, int NZ0, int NZ1, int NZ2>
int launch_by_m(const sar_conv_desc& d, hipStream_t st, bool query_only, int* nparts_out) {
  ConvK k;
  k.d = d;
  if (d.M > 64) {
    constexpr int MS = 2, NS = 2, WM = 2, WN = 2;
    if (int g = tile_geometry<WN>(d, NS, k)) { sar_set_error("sar_conv_gemm: unsupported tile geometry (V=%d, stride=%d)", d.V, d.stride); return g == -2 ? SAR_E_UNSUP : SAR_E_ARG; }
    if (nparts_out) *nparts_out = k.nparts;
    if (query_only) return 0;
    const size_t lds = sizeof(float) * (KC * k.SROW + TAPS * KC * 32 * MS * WM);
    dim3 grid(d.B * k.TPS, (d.M + 32 * MS * WM - 1) / (32 * MS * WM));
    hipLaunchKernelGGL((conv_gemm_kernel<MODE, TRANSPOSED, TAPS, MS, NS, WM, WN, NZ0, NZ1, NZ2>), grid, dim3(256), lds,
                       st, k);
  } else if (d.M > 32) {
    constexpr int MS = 2, NS = 2, WM = 1, WN = 4;
    if (int g = tile_geometry<WN>(d, NS, k)) { sar_set_error("sar_conv_gemm: unsupported tile geometry (V=%d, stride=%d)", d.V, d.stride); return g == -2 ? SAR_E_UNSUP : SAR_E_ARG; }
    if (nparts_out) *nparts_out = k.nparts;
    if (query_only) return 0;
    const size_t lds = sizeof(float) * (KC * k.SROW + TAPS * KC * 32 * MS * WM);
    dim3 grid(d.B * k.TPS, (d.M + 32 * MS * WM - 1) / (32 * MS * WM));
    hipLaunchKernelGGL((conv_gemm_kernel<MODE, TRANSPOSED, TAPS, MS, NS, WM, WN, NZ0, NZ1, NZ2>), grid, dim3(256), lds,
                       st, k);
  } else {
    constexpr int MS = 1, NS = 2, WM = 1, WN = 4;
    if (int g = tile_geometry<WN>(d, NS, k)) { sar_set_error("sar_conv_gemm: unsupported tile geometry (V=%d, stride=%d)", d.V, d.stride); return g == -2 ? SAR_E_UNSUP : SAR_E_ARG; }
    if (nparts_out) *nparts_out = k.nparts;
    if (query_only) return 0;
    const size_t lds = sizeof(float) * (KC * k.SROW + TAPS * KC * 32 * MS * WM);
    dim3 grid(d.B * k.TPS, (d.M + 32 * MS * WM - 1) / (32 * MS * WM));
    hipLaunchKernelGGL((conv_gemm_kernel<MODE, TRANSPOSED, TAPS, MS, NS, WM, WN, NZ0, NZ1, NZ2>), grid, dim3(256), lds,
                       st, k);
  }
  return 0;
}

int validate(const sar_conv_desc* d) {
  SAR_REQUIRE(d != nullptr, "sar_conv_gemm: null descriptor");
  SAR_REQUIRE(d->mode == SAR_CONV_GRAPH || d->mode == SAR_CONV_TEMPORAL, "sar_conv_gemm: bad mode %d", d->mode);
  SAR_REQUIRE(d->B > 0 && d->V > 0 && d->V <= 64 && d->T_src > 0 && d->T_out > 0 && d->Kc > 0 && d->M > 0,
              "sar_conv_gemm: bad sizes B=%d V=%d T_src=%d T_out=%d Kc=%d M=%d", d->B, d->V, d->T_src, d->T_out, d->Kc,
              d->M);
  SAR_REQUIRE(d->src && d->out && d->W, "sar_conv_gemm: null src/out/W");
  SAR_REQUIRE(d->w_stride_c >= d->M && d->w_stride_tap >= 0, "sar_conv_gemm: bad weight strides");
  SAR_REQUIRE(d->ld_src >= (int64_t)d->B * d->T_src * d->V && d->ld_out >= (int64_t)d->B * d->T_out * d->V,
              "sar_conv_gemm: leading dimension smaller than B*T*V");
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv_gemm: pro_scale/pro_shift mismatch");
  SAR_REQUIRE(d->epi >= SAR_EPI_NONE && d->epi <= SAR_EPI_ADD, "sar_conv_gemm: bad epilogue %d", d->epi);
  if (d->epi == SAR_EPI_STATS || d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->partials, "sar_conv_gemm: partials required");
  if (d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD)
    SAR_REQUIRE(d->aux && d->ld_aux >= (int64_t)d->B * d->T_out * d->V, "sar_conv_gemm: aux required");
  if (d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->aux_scale && d->aux_shift, "sar_conv_gemm: aux affine required");
  if (d->mode == SAR_CONV_GRAPH) {
    SAR_REQUIRE(d->taps == 3, "sar_conv_gemm: graph mode needs 3 adjacency slices (got %d)", d->taps);
    SAR_REQUIRE(d->T_src == d->T_out, "sar_conv_gemm: graph mode keeps T");
    SAR_REQUIRE(d->g_idx && d->g_wt, "sar_conv_gemm: graph gather tables required");
    SAR_REQUIRE(!d->bias || d->g_colsum, "sar_conv_gemm: graph bias needs g_colsum");
    for (int i = 0; i < 3; ++i)
      if (d->nz[i] < 1 || d->nz[i] > 4) {
        sar_set_error("sar_conv_gemm: adjacency slice %d needs %d gather entries per column (max 4)", i, d->nz[i]);
        return SAR_E_UNSUP;
      }
  } else {
    SAR_REQUIRE(d->stride >= 1 && d->pad >= 0, "sar_conv_gemm: bad stride/pad");
    if (d->taps != 9 && d->taps != 1) {
      sar_set_error("sar_conv_gemm: temporal kernel size %d not built (1 and 9 are)", d->taps);
      return SAR_E_UNSUP;
    }
  }
  return 0;
}

int dispatch(const sar_conv_desc& d, hipStream_t st, bool query_only, int* nparts_out) {
  if (d.mode == SAR_CONV_GRAPH) {
    if (d.nz[0] == 1 && d.nz[1] == 1) return launch_by_m<SAR_CONV_GRAPH, 0, 3, 1, 1, 4>(d, st, query_only, nparts_out);
    if (d.nz[0] == 1 && d.nz[2] == 1) return launch_by_m<SAR_CONV_GRAPH, 0, 3, 1, 4, 1>(d, st, query_only, nparts_out);
    return launch_by_m<SAR_CONV_GRAPH, 0, 3, 4, 4, 4>(d, st, query_only, nparts_out);
  }
  if (!d.transposed) {
    if (d.taps == 9) return launch_by_m<SAR_CONV_TEMPORAL, 0, 9, 1, 1, 1>(d, st, query_only, nparts_out);
    return launch_by_m<SAR_CONV_TEMPORAL, 0, 1, 1, 1, 1>(d, st, query_only, nparts_out);
  }
  if (d.taps == 9) return launch_by_m<SAR_CONV_TEMPORAL, 1, 9, 1, 1, 1>(d, st, query_only, nparts_out);
  return launch_by_m<SAR_CONV_TEMPORAL, 1, 1, 1, 1, 1>(d, st, query_only, nparts_out);
}

}  // namespace

// Diagnostic: resident workgroups per CU the runtime predicts for the 9-tap temporal forward kernel
// (128x128 tile) at a given dynamic-LDS size.  which=1 selects the 64x256 tile.
extern "C" int sar_debug_occupancy(int which, int lds_bytes) {
  int n = -1;
  hipError_t e;
  if (which == 1)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_gemm_kernel<SAR_CONV_TEMPORAL, 0, 9, 2, 2, 1, 4, 1, 1, 1>, 256, lds_bytes);
  else
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_gemm_kernel<SAR_CONV_TEMPORAL, 0, 9, 2, 2, 2, 2, 1, 1, 1>, 256, lds_bytes);
  hipFuncAttributes at;
  const void* fp = which == 1 ? (const void*)conv_gemm_kernel<SAR_CONV_TEMPORAL, 0, 9, 2, 2, 1, 4, 1, 1, 1>
                              : (const void*)conv_gemm_kernel<SAR_CONV_TEMPORAL, 0, 9, 2, 2, 2, 2, 1, 1, 1>;
  if (hipFuncGetAttributes(&at, fp) == hipSuccess)
    fprintf(stderr, "[sar_debug] which=%d lds=%d: numRegs=%d sharedSizeBytes=%zu maxDynamicShared=%d localSizeBytes=%zu "
            "constSizeBytes=%zu maxThreadsPerBlock=%d -> blocks/CU %d\n", which, lds_bytes, at.numRegs, at.sharedSizeBytes,
            at.maxDynamicSharedSizeBytes, at.localSizeBytes, at.constSizeBytes, at.maxThreadsPerBlock, n);
  return e == hipSuccess ? n : -(int)e;
}

extern "C" int sar_conv_gemm_nparts(const sar_conv_desc* d) {
  if (!d || d->V <= 0 || d->T_out <= 0 || d->B <= 0 || d->M <= 0) return SAR_E_ARG;
  int np = 0;
  sar_conv_desc c = *d;
  if (c.mode == SAR_CONV_GRAPH) { c.nz[0] = c.nz[1] = 1; c.nz[2] = 4; c.taps = 3; }
  else if (c.taps != 1) c.taps = 9;
  int rc = dispatch(c, nullptr, true, &np);
  return rc ? rc : np;
}

extern "C" int sar_conv_gemm_f32(const sar_conv_desc* d, sar_stream_t s) {
  int rc = validate(d);
  if (rc) return rc;
  // nz lists shorter than the instantiated length are padded by the caller with zero weights;
  // the (1,1,4)/(1,4,1) fast paths require exact lengths, anything else takes the (4,4,4) path.
  rc = dispatch(*d, as_stream(s), false, nullptr);
  if (rc) return rc;
  SAR_LAUNCH_CHECK("sar_conv_gemm_f32");
  return 0;
}

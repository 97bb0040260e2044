// conv_gemm_bf16.hip -- the temporal convolution / its data gradient with bf16 MFMA operands (gfx950).
//
//   out[m, n] = sum_tap sum_c bf16(W[tap][c][m]) * bf16(pro(src))[c, n + shift(tap)] (+ bias) ; epilogue
//
// Same operator, tiling, prologue and epilogue as conv_gemm.hip (models/stgcn.py:27-36,47-54 and their data
// gradients); the difference is the arithmetic of the contraction: both operands are rounded to bfloat16
// (round-to-nearest-even) when they are staged, products are exact and are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  Activations stay fp32 in HBM, BatchNorm statistics are
// reduced from the fp32 accumulators, the parameters are the caller's fp32 master copy.
//
// Design (MI355X):
//  * The bf16 MFMA takes 8 consecutive k (= src channels) per lane, so the LDS image is k-innermost: one 16-byte
//    unit holds 8 channels of ONE column.  A temporal tap is a shift by whole units: every operand read is one
//    aligned ds_read_b128 whatever the tap (V = 25 makes the shifts odd, which rules out both a [c][n] bf16 image
//    read by ds_read_b64_tr_b16 -- 8-byte alignment -- and packed pairs along n).
//  * The stager does the transposition for free: a lane owns a column and loads it from 8 channel rows (each load
//    is a coalesced row segment across the wave), applies the folded BatchNorm + ReLU, converts
//    (v_cvt_pk_bf16_f32) and writes one ds_write_b128.
//  * The weights are packed once per call by a small kernel into the exact LDS image ([tap][c/8][m][8] bf16), so
//    their staging is a straight 16-byte copy.
//  * The bf16 MFMA is 16x faster than the fp32 one: the kernel is bound by HBM (fp32 activations in and out) and
//    by the L2 -> LDS traffic of the weights, not by the matrix pipe; one LDS buffer, two workgroups per CU.
#include "sar_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KC16 = 16;   // src channels per main-loop stage = one MFMA k-step

struct ConvKB {
  sar_conv_desc d;
  const uint4* wp;   // packed weights [taps][G][M] units of 8 bf16
  int G;             // channel groups of 8 (even)
  int FT, TPS, NF, RW, nparts, ntiles, ny;
};

template <int TAPS, int MS, int NS, int WM, int WN>
struct TileCfgB {
  static constexpr int BM = 32 * MS * WM;
  static constexpr int TN = 32 * NS * WN;
  static constexpr int RWMAX = (TN == 128 ? 448 : 704);   // staged columns (as conv_gemm.hip)
  static constexpr int SCOLS = RWMAX + 8;                 // + the always-zero column
  static constexpr int WUNITS = TAPS * 2 * BM;            // [tap][h][m]
  static constexpr int SUNITS = 2 * SCOLS;                // [h][col]
  static constexpr int CJ = (RWMAX + 255) / 256;          // S columns per lane
  static constexpr int WIT = (WUNITS + 255) / 256;        // W units per lane
  static constexpr int UNITS = WUNITS + SUNITS;
};

// fp32 weights (element (tap,c,m) at tap*st + c*sc + m) -> bf16 units [tap][g][m][8], zero beyond Kc
__global__ void pack_weights_kernel(const float* __restrict__ W, int64_t st, int64_t sc, int taps, int Kc, int M, int G,
                                    uint4* __restrict__ out) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (int64_t)taps * G * M) return;
  const int m = (int)(u % M);
  const int g = (int)((u / M) % G);
  const int tp = (int)(u / ((int64_t)M * G));
  bf16x8 p;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 8 * g + j;
    p[j] = (__bf16)(c < Kc ? W[tp * st + c * sc + m] : 0.f);
  }
  out[u] = *reinterpret_cast<uint4*>(&p);
}

// the same for many tensors at once (blockIdx.y = item): sar_pack_weights_bf16_batch
__global__ void pack_weights_batch_kernel(const float* __restrict__ base, const sar_pack_item* __restrict__ items,
                                          uint4* __restrict__ out) {
  const sar_pack_item it = items[blockIdx.y];
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (int64_t)it.taps * it.G * it.M) return;
  const int m = (int)(u % it.M);
  const int g = (int)((u / it.M) % it.G);
  const int tp = (int)(u / ((int64_t)it.M * it.G));
  const float* W = base + it.src_off + tp * it.st + m * it.sm;
  bf16x8 p;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 8 * g + j;
    p[j] = (__bf16)(c < it.Kc ? W[c * it.sc] : 0.f);
  }
  out[it.dst_unit + u] = *reinterpret_cast<uint4*>(&p);
}

#include "conv_epi_f32.h"

// TR: 0 forward; 1 data gradient, stride 1; 2 data gradient, generic stride (tap validity mask); 3 data gradient,
// stride 2, parity-split column map (see conv_gemm.hip)
template <int TR, int TAPS, int MS, int NS, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_gemm_bf16_kernel(const ConvKB k) {
  using TC = TileCfgB<TAPS, MS, NS, WM, WN>;
  constexpr int TRANSPOSED = TR != 0;
  constexpr int PAR = (TR == 3);
  constexpr int JT = PAR ? (TAPS + 1) / 2 : TAPS;
  constexpr int BM = TC::BM, SCOLS = TC::SCOLS, CJ = TC::CJ, WIT = TC::WIT;
  constexpr int ZCOL = TC::RWMAX;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int PAREA_U = 4 * 16 * 65 / 4;   // the epilogue's transpose area (floats / 4), aliases the operand image
  constexpr int IMG_U = TC::UNITS > PAREA_U ? TC::UNITS : PAREA_U;
  __shared__ uint4 smem_u[IMG_U + BM];       // image | per-row parameters (float4 per row)
  uint4* Wl = smem_u;
  uint4* Sl = smem_u + TC::WUNITS;
  float* smem = reinterpret_cast<float*>(smem_u);
  float4* rowp = reinterpret_cast<float4*>(smem_u + IMG_U);
  const sar_conv_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int V = d.V;
  // workgroup -> (tile, row block), XCD-aware (conv_gemm.hip)
  const int ny = k.ny, nwork = k.ntiles * ny;
  int w = blockIdx.x;
  {
    const int per = (nwork + 7) / 8;
    const int xcd = w & 7, slot = w >> 3;
    w = xcd * per + slot;
    if (w >= nwork || slot >= per) return;
  }
  const int tile = w / ny;
  const int b = tile / k.TPS;
  const int t0 = (tile - b * k.TPS) * k.FT;
  const int m0 = (w - tile * ny) * BM;

  // ---- per-lane column geometry
  bool colok[NS];
  int64_t coln[NS];
  int off[JT][NS];
  unsigned vmask[NS];
  int t_lo;
  if (!TRANSPOSED) t_lo = t0 * d.stride - d.pad;
  else t_lo = floordiv(t0 + d.pad - (TAPS - 1), d.stride);
  constexpr int HALFC = 16 * NS * WN;
  const int par = PAR ? (wn * NS * 32 >= HALFC ? 1 : 0) : 0;
  const int tp0 = PAR ? ((par + d.pad) & 1) : 0;
  const int ntap_w = PAR ? (TAPS - tp0 + 1) / 2 : TAPS;
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    int fo, v;
    if (PAR) {
      const int pp = p - par * HALFC;
      const int fh = pp / V;
      v = pp - fh * V;
      fo = 2 * fh + par;
    } else {
      fo = p / V;
      v = p - fo * V;
    }
    colok[ns] = (fo < k.FT) && (t0 + fo < d.T_out);
    if (!colok[ns]) fo = par;
    coln[ns] = ((int64_t)b * d.T_out + (t0 + fo)) * V + v;
    vmask[ns] = 0;
#pragma unroll
    for (int tp = 0; tp < JT; ++tp) {
      if (!TRANSPOSED) {
        off[tp][ns] = (fo * d.stride + tp) * V + v;
      } else if (PAR) {
        const int to = (t0 + fo + d.pad - (tp0 + 2 * tp)) >> 1;
        off[tp][ns] = (to - t_lo) * V + v;
      } else {
        const int q = t0 + fo + d.pad - tp;
        const int to = floordiv(q, d.stride);
        const bool ok = (q - to * d.stride) == 0;
        vmask[ns] |= (ok ? 1u : 0u) << tp;
        off[tp][ns] = (to - t_lo) * V + v;
      }
      if (!colok[ns]) off[tp][ns] = ZCOL;
      off[tp][ns] += hi * SCOLS;   // this lane's k half
    }
  }

  if (tid < BM) {
    const int row = m0 + tid;
    float4 bp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias && row < d.M) bp.x = d.bias[row];
    rowp[tid] = bp;
  }
  if (tid < 2) Sl[tid * SCOLS + ZCOL] = make_uint4(0u, 0u, 0u, 0u);
  f32x16 acc[MS][NS];

  const int seq_len = d.T_src * V;
  const float* src_b = d.src + (int64_t)b * seq_len;

  // ---- staging: per-lane offsets and masks once, scalar arithmetic per stage
  int svo[CJ];
  bool sok[CJ];
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int col = tid + 256 * j;
    const int rabs = t_lo * V + col;
    sok[j] = col < k.RW && (unsigned)rabs < (unsigned)seq_len;
    svo[j] = sok[j] ? rabs * 4 : 0;
  }
  unsigned wvo[WIT];
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const int u = tid + 256 * i;
    const int m = u % BM, h = (u / BM) & 1, tp = u / (2 * BM);
    const bool ok = u < TC::WUNITS && (m0 + m) < d.M;
    wvo[i] = ok ? (unsigned)((((int64_t)tp * k.G + h) * d.M + m0 + m) * 16) : 0x80000000u;   // rejected by the range check -> 0
  }
  const unsigned wbytes = (unsigned)((int64_t)TAPS * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  uint4 wreg[WIT];
  float sreg[2][CJ][8];

  auto issue_loads = [&](int c0) {
    const int wso = (c0 / 8) * d.M * 16;   // scalar: first channel group of the stage
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[i], wso, 0);
      wreg[i] = *reinterpret_cast<const uint4*>(&v);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + 8 * h + q;
        const int cg = c < d.Kc ? c : 0;   // wave-uniform
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, seq_len * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < CJ; ++j) sreg[h][j][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo[j], 0, 0));
      }
  };

  auto store_lds = [&](int c0) {
#pragma unroll
    for (int i = 0; i < WIT; ++i)
      if ((i + 1) * 256 <= TC::WUNITS || tid + 256 * i < TC::WUNITS) Wl[tid + 256 * i] = wreg[i];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float psc[8], psh[8];
      bool rok[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + 8 * h + q;
        rok[q] = c < d.Kc;
        const int cg = rok[q] ? c : 0;
        psc[q] = d.pro_scale ? d.pro_scale[cg] : 1.f;
        psh[q] = d.pro_scale ? d.pro_shift[cg] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < CJ; ++j) {
        bf16x8 p;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float val = fmaxf(fmaf(sreg[h][j][q], psc[q], psh[q]), relu_lo);
          p[q] = (__bf16)((sok[j] && rok[q]) ? val : 0.f);
        }
        if ((j + 1) * 256 <= TC::RWMAX || tid + 256 * j < TC::RWMAX) Sl[h * SCOLS + tid + 256 * j] = *reinterpret_cast<uint4*>(&p);
      }
    }
  };

  issue_loads(0);
  __syncthreads();   // rowp
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 bp = rowp[(wm * MS + ms) * 32 + mfma_row(r, hi)];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[ms][ns][r] = colok[ns] ? bp.x : 0.f;
    }

  const uint4* Wa = Wl + (tp0 * 2 + hi) * BM + wm * MS * 32 + l31;
  auto taps_mma = [&](int j) {
    const int tpw = PAR ? 2 * j : j;
    uint4 a[MS], bq[NS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) a[ms] = Wa[tpw * 2 * BM + ms * 32];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      bq[ns] = Sl[off[j][ns]];
      if (TR == 2 && !((vmask[ns] >> j) & 1u)) bq[ns] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a[ms]),
                                                              *reinterpret_cast<bf16x8*>(&bq[ns]), acc[ms][ns], 0, 0, 0);
  };

  for (int c0 = 0; c0 < d.Kc; c0 += KC16) {
    store_lds(c0);
    __syncthreads();
    if (c0 + KC16 < d.Kc) issue_loads(c0 + KC16);   // in flight during the MFMA phase
    constexpr int JSURE = PAR ? JT - 1 : JT;
#pragma unroll
    for (int j = 0; j < JSURE; ++j) taps_mma(j);
    if (PAR && ntap_w == JT) taps_mma(JT - 1);   // wave-uniform
    __syncthreads();   // every wave is done with the image (next store / the epilogue's transpose area)
  }

  epilogue_b<MS, NS, WN, BM>(d, k.nparts, tile, wm, wn, m0, colok, coln, acc, rowp, smem);
}

// ---- GraphConvTD (models/gcn.py:199-209) and its data gradient with bf16 MFMA operands:
//   out[m, (t,w)] = sum_k sum_c bf16(W_k[c][m]) * bf16(z_k)[c, (t,w)] + sum_k b_k[m] colsum(A_k)[w],
//   z_k[c, (t,w)] = sum_v pro(src)[c, (t,v)] A_k[v, w]      (fp32, <= 4 non-zeros per column of A_k)
// The adjacency is applied on the src side in fp32 and the result rounded once.  Stage = 16 src channels:
//   (1) raw fp32 rows -> LDS (coalesced row segments, folded BatchNorm/ReLU applied),
//   (2) every thread builds the three k-innermost 16-byte units of its (channel half, column): 8 channels x NZ gather
//       entries read back from the raw image, weighted in fp32, converted, one ds_write_b128 per slice,
//   (3) 3 slices x MS x NS MFMAs.
template <int MS, int NS, int WM, int WN, int NZ0, int NZ1, int NZ2>
__global__ __launch_bounds__(256, 2) void conv_graph_bf16_kernel(const ConvKB k) {
  constexpr int BM = 32 * MS * WM, TN = 32 * NS * WN;
  constexpr int NZ[3] = {NZ0, NZ1, NZ2};
  constexpr int XS = TN + 4;                 // raw row stride (floats)
  constexpr int CPT = TN / 128;              // columns per thread in the unit builder (thread = (half, column))
  constexpr int WUNITS = 3 * 2 * BM;         // [slice][h][m]
  constexpr int ZUNITS = 3 * 2 * TN;         // [slice][h][col]
  constexpr int WIT = (WUNITS + 255) / 256;
  constexpr int XJ = TN / 64;                // raw columns per lane and row (wave w stages rows w, w+4, ..)
  constexpr int PAREA_U = 4 * 16 * 65 / 4;
  constexpr int IMG_U = (WUNITS + ZUNITS) > PAREA_U ? (WUNITS + ZUNITS) : PAREA_U;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ uint4 smem_u[IMG_U + BM];
  __shared__ float XR[KC16 * XS];
  uint4* Wl = smem_u;
  uint4* Zl = smem_u + WUNITS;
  float* smem = reinterpret_cast<float*>(smem_u);
  float4* rowp = reinterpret_cast<float4*>(smem_u + IMG_U);
  const sar_conv_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int V = d.V;
  const int ny = k.ny, nwork = k.ntiles * ny;
  int w = blockIdx.x;
  {
    const int per = (nwork + 7) / 8;
    const int xcd = w & 7, slot = w >> 3;
    w = xcd * per + slot;
    if (w >= nwork || slot >= per) return;
  }
  const int tile = w / ny;
  const int b = tile / k.TPS;
  const int t0 = (tile - b * k.TPS) * k.FT;
  const int m0 = (w - tile * ny) * BM;
  const int ncols = ((t0 + k.FT <= d.T_out) ? k.FT : d.T_out - t0) * V;   // live columns of this tile

  bool colok[NS];
  int64_t coln[NS];
  float gcs[3][NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    colok[ns] = p < ncols;
    const int pv = colok[ns] ? p : 0;
    coln[ns] = ((int64_t)b * d.T_out + t0) * V + pv;
    const int v = pv % V;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) gcs[tp][ns] = (d.g_colsum && colok[ns]) ? d.g_colsum[tp * V + v] : 0.f;
  }
  if (tid < BM) {
    const int row = m0 + tid;
    float4 bp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias && row < d.M) {
      bp.x = d.bias[row];
      bp.y = d.bias[d.M + row];
      bp.z = d.bias[2 * d.M + row];
    }
    rowp[tid] = bp;
  }
  // unit builder geometry: this thread's channel half and columns, gather offsets (floats inside a raw row) and weights
  const int uh = tid >> 7;   // 0 / 1 (wave-uniform)
  int go[CPT][3][4];
  float gwt[CPT][3][4];
  bool ulive[CPT];
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int col = (tid & 127) + 128 * q;
    ulive[q] = col < ncols;
    const int cc = ulive[q] ? col : 0;
    const int fo = cc / V, v = cc - fo * V;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < NZ[tp]) {
          go[q][tp][j] = fo * V + d.g_idx[(tp * V + v) * 4 + j];
          gwt[q][tp][j] = ulive[q] ? d.g_wt[(tp * V + v) * 4 + j] : 0.f;
        }
  }

  f32x16 acc[MS][NS];
  const int seq_len = d.T_src * V;
  const float* src_b = d.src + ((int64_t)b * d.T_src + t0) * V;
  int svo[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) svo[j] = (lane + 64 * j) < ncols ? (lane + 64 * j) * 4 : 0x7fffffff;   // rejected -> 0
  unsigned wvo[WIT];
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const int u = tid + 256 * i;
    const int m = u % BM, h = (u / BM) & 1, tp = u / (2 * BM);
    const bool ok = u < WUNITS && (m0 + m) < d.M;
    wvo[i] = ok ? (unsigned)((((int64_t)tp * k.G + h) * d.M + m0 + m) * 16) : 0x80000000u;
  }
  const unsigned wbytes = (unsigned)((int64_t)3 * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  uint4 wreg[WIT];
  float sreg[4][XJ];

  auto issue_loads = [&](int c0) {
    const int wso = (c0 / 8) * d.M * 16;
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[i], wso, 0);
      wreg[i] = *reinterpret_cast<const uint4*>(&v);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = c0 + wave + 4 * r;
      const int cg = c < d.Kc ? c : 0;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0,
                                                                          (unsigned)(seq_len - t0 * V) * 4, 0x00020000);
#pragma unroll
      for (int j = 0; j < XJ; ++j) sreg[r][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo[j], 0, 0));
    }
  };
  auto store_raw = [&](int c0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = c0 + wave + 4 * r;
      const bool rok = c < d.Kc;
      const int cg = rok ? c : 0;
      const float psc = d.pro_scale ? d.pro_scale[cg] : 1.f, psh = d.pro_scale ? d.pro_shift[cg] : 0.f;
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        const float val = fmaxf(fmaf(sreg[r][j], psc, psh), relu_lo);
        XR[(wave + 4 * r) * XS + lane + 64 * j] = (rok && (lane + 64 * j) < ncols) ? val : 0.f;
      }
    }
  };
  auto build_units = [&]() {
#pragma unroll
    for (int i = 0; i < WIT; ++i)
      if ((i + 1) * 256 <= WUNITS || tid + 256 * i < WUNITS) Wl[tid + 256 * i] = wreg[i];
    const float* Xh = XR + uh * 8 * XS;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        bf16x8 p;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float z = gwt[q][tp][0] * Xh[c * XS + go[q][tp][0]];
#pragma unroll
          for (int j = 1; j < 4; ++j)
            if (j < NZ[tp]) z = fmaf(gwt[q][tp][j], Xh[c * XS + go[q][tp][j]], z);
          p[c] = (__bf16)z;
        }
        Zl[(tp * 2 + uh) * TN + (tid & 127) + 128 * q] = *reinterpret_cast<uint4*>(&p);
      }
    }
  };

  issue_loads(0);
  __syncthreads();   // rowp
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 bp = rowp[(wm * MS + ms) * 32 + mfma_row(r, hi)];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[ms][ns][r] = fmaf(bp.z, gcs[2][ns], fmaf(bp.y, gcs[1][ns], bp.x * gcs[0][ns]));
    }
  const uint4* Wa = Wl + hi * BM + wm * MS * 32 + l31;
  const uint4* Za = Zl + hi * TN + wn * NS * 32 + l31;
  for (int c0 = 0; c0 < d.Kc; c0 += KC16) {
    store_raw(c0);
    __syncthreads();   // raw image complete; every wave is past the MFMA phase of the previous stage
    build_units();
    __syncthreads();
    if (c0 + KC16 < d.Kc) issue_loads(c0 + KC16);
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) {
      uint4 a[MS], bq[NS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a[ms] = Wa[tp * 2 * BM + ms * 32];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) bq[ns] = Za[tp * 2 * TN + ns * 32];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a[ms]),
                                                                *reinterpret_cast<bf16x8*>(&bq[ns]), acc[ms][ns], 0, 0, 0);
    }
  }
  __syncthreads();   // the epilogue's transpose area aliases the image
  epilogue_b<MS, NS, WN, BM>(d, k.nparts, tile, wm, wn, m0, colok, coln, acc, rowp, smem);
}

template <int WN>
int tile_geometry_b(const sar_conv_desc& d, int NSv, bool parity, ConvKB& k) {
  const int tile_n = 32 * NSv * WN;
  if (parity) {
    k.FT = 2 * ((tile_n / 2) / d.V);
    const int t_even = d.T_out + (d.T_out & 1);
    if (k.FT > t_even) k.FT = t_even;
  } else {
    k.FT = tile_n / d.V;
    if (k.FT > d.T_out) k.FT = d.T_out;
  }
  if (k.FT < 1) return -1;
  k.TPS = (d.T_out + k.FT - 1) / k.FT;
  if (d.mode == SAR_CONV_GRAPH) k.NF = k.FT;
  else if (!d.transposed) k.NF = (k.FT - 1) * d.stride + d.taps;
  else k.NF = (k.FT - 1 + d.taps - 1) / d.stride + 2;
  k.RW = k.NF * d.V;
  k.nparts = d.B * k.TPS * WN;
  const int rwmax = (d.mode == SAR_CONV_GRAPH) ? tile_n : (NSv * WN == 4 ? 448 : 704);
  if (k.RW > rwmax) return -2;
  return 0;
}

template <int TR, int TAPS, int MS, int NS, int WM, int WN>
int launch_cfg_b(const sar_conv_desc& d, uint4* wp, hipStream_t st) {
  ConvKB k;
  k.d = d;
  k.wp = wp;
  k.G = 2 * ((d.Kc + 15) / 16);
  if (int g = tile_geometry_b<WN>(d, NS, TR == 3, k)) {
    sar_set_error("sar_conv_gemm_bf16: unsupported tile geometry (V=%d, stride=%d)", d.V, d.stride);
    return g == -2 ? SAR_E_UNSUP : SAR_E_ARG;
  }
  constexpr int BM = 32 * MS * WM;
  k.ntiles = d.B * k.TPS;
  k.ny = (d.M + BM - 1) / BM;
  const int nwork = k.ntiles * k.ny;
  const int64_t units = (int64_t)d.taps * k.G * d.M;
  if (d.W)   // else: `wp` already holds the packed image (sar_pack_weights_bf16_batch)
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, d.W, d.w_stride_tap,
                       d.w_stride_c, d.taps, d.Kc, d.M, k.G, wp);
  hipLaunchKernelGGL((conv_gemm_bf16_kernel<TR, TAPS, MS, NS, WM, WN>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, st, k);
  return 0;
}

template <int MS, int NS, int WM, int WN, int NZ0, int NZ1, int NZ2>
int launch_graph_cfg_b(const sar_conv_desc& d, uint4* wp, hipStream_t st) {
  ConvKB k;
  k.d = d;
  k.wp = wp;
  k.G = 2 * ((d.Kc + 15) / 16);
  if (int g = tile_geometry_b<WN>(d, NS, false, k)) {
    sar_set_error("sar_conv_gemm_bf16: unsupported tile geometry (V=%d)", d.V);
    return g == -2 ? SAR_E_UNSUP : SAR_E_ARG;
  }
  constexpr int BM = 32 * MS * WM;
  k.ntiles = d.B * k.TPS;
  k.ny = (d.M + BM - 1) / BM;
  const int nwork = k.ntiles * k.ny;
  const int64_t units = (int64_t)3 * k.G * d.M;
  if (d.W)
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, d.W, d.w_stride_tap,
                       d.w_stride_c, 3, d.Kc, d.M, k.G, wp);
  hipLaunchKernelGGL((conv_graph_bf16_kernel<MS, NS, WM, WN, NZ0, NZ1, NZ2>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, st, k);
  return 0;
}

template <int NZ0, int NZ1, int NZ2>
int launch_graph_by_m_b(const sar_conv_desc& d, uint4* wp, hipStream_t st) {
  if (d.M > 64) return launch_graph_cfg_b<2, 2, 2, 2, NZ0, NZ1, NZ2>(d, wp, st);
  if (d.M > 32) return launch_graph_cfg_b<2, 2, 1, 4, NZ0, NZ1, NZ2>(d, wp, st);
  return launch_graph_cfg_b<1, 2, 1, 4, NZ0, NZ1, NZ2>(d, wp, st);
}

// the same tile choice as conv_gemm.hip's launch_by_m: the partial-sum layout (sar_conv_gemm_nparts) is shared
template <int TR, int TAPS>
int launch_by_m_b(const sar_conv_desc& d, uint4* wp, hipStream_t st) {
  if constexpr (TR != 3)
    if (d.M > 64) return launch_cfg_b<TR, TAPS, 2, 2, 2, 2>(d, wp, st);
  // (32-row blocks for M = 64 -- 3 workgroups per CU -- measured slower: 31.3 vs 30.5 ms per step)
  if (d.M > 32) return launch_cfg_b<TR, TAPS, 2, 2, 1, 4>(d, wp, st);
  return launch_cfg_b<TR, TAPS, 1, 2, 1, 4>(d, wp, st);
}

int dispatch_b(const sar_conv_desc& d, uint4* wp, hipStream_t st) {
  if (d.mode == SAR_CONV_GRAPH) {
    if (d.nz[0] == 1 && d.nz[1] == 1) return launch_graph_by_m_b<1, 1, 4>(d, wp, st);
    if (d.nz[0] == 1 && d.nz[2] == 1) return launch_graph_by_m_b<1, 4, 1>(d, wp, st);
    return launch_graph_by_m_b<4, 4, 4>(d, wp, st);
  }
  if (!d.transposed) return d.taps == 9 ? launch_by_m_b<0, 9>(d, wp, st) : launch_by_m_b<0, 1>(d, wp, st);
  if (d.stride == 1) return d.taps == 9 ? launch_by_m_b<1, 9>(d, wp, st) : launch_by_m_b<1, 1>(d, wp, st);
  if (d.taps == 9) return d.stride == 2 ? launch_by_m_b<3, 9>(d, wp, st) : launch_by_m_b<2, 9>(d, wp, st);
  return launch_by_m_b<2, 1>(d, wp, st);
}

}  // namespace

extern "C" int64_t sar_conv_gemm_bf16_workspace_bytes(const sar_conv_desc* d) {
  if (!d || d->Kc <= 0 || d->M <= 0 || d->taps <= 0) return SAR_E_ARG;
  return (int64_t)d->taps * 2 * ((d->Kc + 15) / 16) * d->M * 16;
}

extern "C" int sar_conv_gemm_bf16(const sar_conv_desc* d, void* workspace, sar_stream_t s) {
  SAR_REQUIRE(d != nullptr && workspace != nullptr, "sar_conv_gemm_bf16: null descriptor / workspace");
  SAR_REQUIRE(((uintptr_t)workspace & 15) == 0, "sar_conv_gemm_bf16: workspace must be 16-byte aligned");
  SAR_REQUIRE(d->mode == SAR_CONV_TEMPORAL || d->mode == SAR_CONV_GRAPH, "sar_conv_gemm_bf16: bad mode %d", d->mode);
  SAR_REQUIRE(d->B > 0 && d->V > 0 && d->V <= 64 && d->T_src > 0 && d->T_out > 0 && d->Kc > 0 && d->M > 0,
              "sar_conv_gemm_bf16: bad sizes");
  SAR_REQUIRE((d->M & 7) == 0, "sar_conv_gemm_bf16: M must be a multiple of 8 (got %d)", d->M);
  SAR_REQUIRE(d->src && d->out, "sar_conv_gemm_bf16: null src/out");   /* W == NULL: workspace is already packed */
  if (d->mode == SAR_CONV_GRAPH) {
    SAR_REQUIRE(d->taps == 3 && d->T_src == d->T_out, "sar_conv_gemm_bf16: graph mode needs 3 adjacency slices and keeps T");
    SAR_REQUIRE(!(d->g_flags & SAR_GRAPH_AUX_EVEN_FRAMES), "sar_conv_gemm_bf16: SAR_GRAPH_AUX_EVEN_FRAMES is built for sar_conv_gemm_f32 / sar_conv_gemm_split");
    SAR_REQUIRE(d->g_idx && d->g_wt, "sar_conv_gemm_bf16: graph gather tables required");
    SAR_REQUIRE(!d->bias || d->g_colsum, "sar_conv_gemm_bf16: graph bias needs g_colsum");
    for (int i = 0; i < 3; ++i)
      SAR_REQUIRE(d->nz[i] >= 1 && d->nz[i] <= 4, "sar_conv_gemm_bf16: adjacency slice %d needs %d gather entries (max 4)", i, d->nz[i]);
  } else {
    SAR_REQUIRE(d->taps == 9 || d->taps == 1, "sar_conv_gemm_bf16: temporal kernel size %d not built (1 and 9 are)", d->taps);
    SAR_REQUIRE(d->stride >= 1 && d->pad >= 0, "sar_conv_gemm_bf16: bad stride/pad");
  }
  SAR_REQUIRE(d->ld_src >= (int64_t)d->B * d->T_src * d->V && d->ld_out >= (int64_t)d->B * d->T_out * d->V,
              "sar_conv_gemm_bf16: leading dimension smaller than B*T*V");
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv_gemm_bf16: pro_scale/pro_shift mismatch");
  SAR_REQUIRE((int64_t)d->T_src * d->V < (1 << 28), "sar_conv_gemm_bf16: sequence row too long");
  SAR_REQUIRE(d->ld_out < (1 << 22) && d->ld_aux < (1 << 22), "sar_conv_gemm_bf16: leading dimension too large (2^22 columns)");
  SAR_REQUIRE(sar_conv_gemm_bf16_workspace_bytes(d) < (1ll << 31), "sar_conv_gemm_bf16: weight tensor too large");
  SAR_REQUIRE(d->epi >= SAR_EPI_NONE && d->epi <= SAR_EPI_ADD, "sar_conv_gemm_bf16: bad epilogue %d", d->epi);
  if (d->epi == SAR_EPI_STATS || d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->partials, "sar_conv_gemm_bf16: partials required");
  if (d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD)
    SAR_REQUIRE(d->aux && d->ld_aux >= (int64_t)d->B * d->T_out * d->V, "sar_conv_gemm_bf16: aux required");
  if (d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->aux_scale && d->aux_shift, "sar_conv_gemm_bf16: aux affine required");
  int rc = dispatch_b(*d, (uint4*)workspace, as_stream(s));
  if (rc) return rc;
  SAR_LAUNCH_CHECK("sar_conv_gemm_bf16");
  return 0;
}

extern "C" int sar_pack_weights_bf16_batch(const float* base, const sar_pack_item* items, int nitems, int64_t max_units,
                                           void* out, sar_stream_t s) {
  SAR_REQUIRE(base && items && out && nitems > 0 && max_units > 0, "sar_pack_weights_bf16_batch: bad arguments");
  SAR_REQUIRE(((uintptr_t)out & 15) == 0, "sar_pack_weights_bf16_batch: out must be 16-byte aligned");
  SAR_REQUIRE(nitems <= 65535 && (max_units + 255) / 256 < (1ll << 31), "sar_pack_weights_bf16_batch: too many items / units");
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)((max_units + 255) / 256), nitems), dim3(256), 0, as_stream(s),
                     base, items, (uint4*)out);
  SAR_LAUNCH_CHECK("sar_pack_weights_bf16_batch");
  return 0;
}

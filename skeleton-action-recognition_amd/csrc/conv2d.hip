// conv2d.hip -- the reference's 1-channel ResNet-18 (models/resnet18.py:131-254) on gfx950: 2-D convolutions as
// implicit GEMMs on fp32 MFMA, plus the stem tail (BN+ReLU+MaxPool), Adam and the weight re-packing helper.
//
// Same design as conv_gemm.hip / conv_wgrad.hip (read those headers first), generalised from "frames x joints" to
// "image rows x image columns": activations are [C][B*H*W] matrices; a workgroup tile is TH whole output rows of
// ONE image; the source rows it needs (+ halo, zero padding on all four sides materialised, producer BN+ReLU
// folded) are staged in LDS as a padded image, so a tap (kh,kw) is a uniform offset kh*Wq+kw and the inner loops
// carry no bounds logic.  Strided data gradients keep a per-lane parity mask per tap.
//  * conv2d_gemm_kernel   : 3x3 / 1x1, stride 1|2, forward and data gradient (epilogues as in conv_gemm.hip)
//  * conv2d_stem_kernel   : the 7x7/2 conv on ONE input channel -- the 49 taps sit on the MFMA K axis
//  * conv2d_wgrad_kernel  : weight gradients; the two MFMA k-lanes reduce the same column of two consecutive
//                           output rows (frame-pair trick of conv_wgrad.hip); the stem variant puts the 49 taps
//                           on the lane (row) axis of the A operand.
#include "sar_common.h"
#include <type_traits>

namespace {

constexpr int KC2 = 4;   // src channels per main-loop stage

struct C2K {
  sar_conv2d_desc d;
  int TH, TPI, NR, Wq, RW, SROW, nparts, w_vec, col_lo;
  int NI, IRW, ntiles;   // images per tile (small feature maps: several whole images share a tile), staged elements per image
  float invWq;
};

// ------------------------------------------------------------------------------------------------ forward / dgrad
// Same machinery as conv_gemm.hip (read its header): buffer-load stagers with per-lane offsets / masks computed
// once, two LDS buffers and one barrier per stage, software-pipelined operand reads with immediate offsets, bias-
// free epilogue with buffer stores and the wave-private LDS transpose for the BatchNorm sums, 3 workgroups per CU
// where LDS allows.  TR: 0 forward, 1 data gradient at stride 1 (no tap mask), 2 data gradient, strided (per-lane
// tap validity mask).
template <int TAPS, int MS, int NS, int WM, int WN>
struct Tile2 {
  static constexpr int BM = 32 * MS * WM;
  static constexpr int WROWS = TAPS * KC2;
  static constexpr int WRPP = 1024 / BM;
  static constexpr int WIT = (WROWS + WRPP - 1) / WRPP;
  static constexpr int WPAD = ((WROWS + 3) / 4) * 4;
  static constexpr int RWMAX = (TAPS == 1 ? 512 : 640);       // staged elements per src channel row (bound)
  static constexpr int SJ = RWMAX / 64;
  static constexpr int SSTR = RWMAX + 8;
  static constexpr int BUF = WPAD * BM + KC2 * SSTR;
};

template <int TR, int KH, int KW, int MS, int NS, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv2d_gemm_kernel(const C2K k) {
  constexpr int TAPS = KH * KW;
  constexpr int TRANSPOSED = TR != 0;
  using TC = Tile2<TAPS, MS, NS, WM, WN>;
  constexpr int BM = TC::BM, WROWS = TC::WROWS, WRPP = TC::WRPP, WIT = TC::WIT, SJ = TC::SJ, SSTR = TC::SSTR;
  constexpr int ZCOL = TC::RWMAX;
  static_assert(WM * WN == 4 && KC2 == 4, "4 waves, one staged channel row per wave");
  constexpr int PAREA = 4 * 16 * 65;
  constexpr int ROWP_OFF = TC::BUF >= PAREA ? TC::BUF : (2 * TC::BUF > PAREA ? 2 * TC::BUF : PAREA);
  constexpr int LDS_FLOATS = 2 * TC::BUF > ROWP_OFF + 4 * BM ? 2 * TC::BUF : ROWP_OFF + 4 * BM;
  __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
  float4* rowp = reinterpret_cast<float4*>(smem + ROWP_OFF);   // aux affine per output row (MASK epilogue)
  const sar_conv2d_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int tile = blockIdx.x;
  // NI == 1: the tile is TH output rows of image b.  NI > 1 (feature maps of <= half a tile): NI whole images
  // b .. b + NI - 1, each with its own padded region of IRW staged elements.
  const int b = k.NI > 1 ? tile * k.NI : tile / k.TPI;
  const int h0 = k.NI > 1 ? 0 : (tile - b * k.TPI) * k.TH;
  const int m0 = blockIdx.y * BM;
  const int s = d.stride;
  const int opix = d.H_out * d.W_out;

  int row_lo;
  if (!TRANSPOSED) row_lo = h0 * s - d.pad;
  else row_lo = floordiv(h0 + d.pad - (KH - 1), s);

  // ---- per-lane column geometry; off-tile columns read the always-zero LDS column
  bool colok[NS];
  int64_t coln[NS];
  int off[TAPS][NS];
  unsigned vmask[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    const int im = k.NI > 1 ? p / opix : 0;            // image inside the tile
    const int pp = p - im * opix;
    int hl = pp / d.W_out;
    int wo = pp - hl * d.W_out;
    colok[ns] = im < k.NI && (b + im) < d.B && hl < k.TH && (h0 + hl) < d.H_out;
    if (!colok[ns]) { hl = 0; wo = 0; }
    coln[ns] = ((int64_t)(b + (colok[ns] ? im : 0)) * d.H_out + (h0 + hl)) * d.W_out + wo;
    vmask[ns] = 0;
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
      const int kh = tp / KW, kw = tp % KW;
      if (!TRANSPOSED) {
        off[tp][ns] = im * k.IRW + (hl * s + kh) * k.Wq + wo * s + kw;
      } else {
        const int qh = h0 + hl + d.pad - kh, qw = wo + d.pad - kw;
        const int ho = floordiv(qh, s), ws = floordiv(qw, s);
        const bool ok = (qh - ho * s == 0) && (qw - ws * s == 0);
        vmask[ns] |= (ok ? 1u : 0u) << tp;
        off[tp][ns] = im * k.IRW + (ho - row_lo) * k.Wq + (ws - k.col_lo);
      }
      if (!colok[ns]) off[tp][ns] = ZCOL;
    }
  }
  if (tid < 2 * KC2) smem[(tid / KC2) * TC::BUF + TC::WPAD * BM + (tid % KC2) * SSTR + ZCOL] = 0.f;

  f32x16 acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

  // ---- staging (everything stage-invariant is computed here once)
  const int img = d.H_src * d.W_src;
  const float* src_b = d.src + (int64_t)b * img;
  int svo[SJ];
  bool sok[SJ];
#pragma unroll
  for (int j = 0; j < SJ; ++j) {
    const int e = lane + 64 * j;                       // staged element -> image, (r, q) -> src pixel (row_lo + r, col_lo + q)
    const int im = e / k.IRW, ei = e - im * k.IRW;
    const int r = ei / k.Wq, q = ei - r * k.Wq;
    const int hs = row_lo + r, ws = k.col_lo + q;
    sok[j] = e < k.RW && (b + im) < d.B && (unsigned)hs < (unsigned)d.H_src && (unsigned)ws < (unsigned)d.W_src;   // else zero padding
    svo[j] = sok[j] ? (im * img + hs * d.W_src + ws) * 4 : 0;
  }
  const bool w_vec = k.w_vec != 0;
  const int w_m4 = (tid % (BM / 4)) * 4, w_r0 = tid / (BM / 4);
  int wvo[WIT];
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const int row = w_r0 + i * WRPP;
    const int tp = row / KC2, c = row % KC2;
    const bool ok = row < WROWS && (m0 + w_m4) < d.M;
    wvo[i] = ok ? (int)(((int64_t)tp * d.w_stride_tap + (int64_t)c * d.w_stride_c + m0 + w_m4) * 4) : 0;
  }
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  float4 wreg[WIT];
  float sreg[SJ];
  float psc = 1.f, psh = 0.f;

  auto issue_loads = [&](int c0) {
    if (w_vec) {
      const __amdgpu_buffer_rsrc_t rw =
          __builtin_amdgcn_make_buffer_rsrc((void*)(d.W + (int64_t)c0 * d.w_stride_c), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int i = 0; i < WIT; ++i) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[i], 0, 0);
        wreg[i] = *reinterpret_cast<const float4*>(&v);
      }
    }
    const int cg = (c0 + wave < d.Kc) ? c0 + wave : 0;   // wave-uniform
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, k.NI * img * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < SJ; ++j) sreg[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo[j], 0, 0));
    if (d.pro_scale) {
      psc = d.pro_scale[cg];
      psh = d.pro_shift[cg];
    }
  };

  auto store_lds = [&](int c0, float* buf) {
    float* Wl = buf;
    float* S = buf + TC::WPAD * BM;
    if (w_vec) {
#pragma unroll
      for (int i = 0; i < WIT; ++i) {
        const int row = w_r0 + i * WRPP;
        if ((i + 1) * WRPP <= TC::WPAD || row < TC::WPAD) *reinterpret_cast<float4*>(Wl + row * BM + w_m4) = wreg[i];
      }
    } else {   // unaligned weights or a Kc tail: plain strided copy with zero fill
      for (int idx = tid; idx < WROWS * BM; idx += 256) {
        const int m = idx % BM, row = idx / BM;
        const int tp = row / KC2, cg = c0 + row % KC2, mg = m0 + m;
        Wl[idx] = (cg < d.Kc && mg < d.M) ? d.W[(int64_t)tp * d.w_stride_tap + (int64_t)cg * d.w_stride_c + mg] : 0.f;
      }
    }
    const bool rowok = (c0 + wave) < d.Kc;
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      const float val = fmaxf(fmaf(sreg[j], psc, psh), relu_lo);
      S[wave * SSTR + lane + 64 * j] = (sok[j] && rowok) ? val : 0.f;   // zero padding on all four sides
    }
  };

  issue_loads(0);
  store_lds(0, smem);
  __syncthreads();
  auto stage = [&](int c0, auto IT) {
    constexpr int it = decltype(IT)::value;
    const bool more = c0 + KC2 < d.Kc;
    if (more) issue_loads(c0 + KC2);
    const float* Wl = smem + it * TC::BUF;
    const float* S = Wl + TC::WPAD * BM;
    constexpr int HS = KC2 / 2;
    const float* Sh = S + hi * SSTR;
    typedef const float __attribute__((address_space(3))) * lds_cptr;
    lds_cptr Wa[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      unsigned a = (unsigned)(uintptr_t)(Wl + hi * BM + (wm * MS + ms) * 32 + l31);   // LDS byte address, kept opaque
      asm volatile("" : "+v"(a));
      Wa[ms] = (lds_cptr)(uintptr_t)a;
    }
    auto fetch = [&](int st, float (&a)[MS], float (&r)[NS]) {
      const int tp = st / HS, cc = (st % HS) * 2;
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a[ms] = Wa[ms][(tp * KC2 + cc) * BM];
      const float* Srow = Sh + cc * SSTR;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) r[ns] = Srow[off[tp][ns]];
    };
    auto mma = [&](int st, const float (&a)[MS], const float (&r)[NS], bool have_next) {
      const int tp = st / HS;
      float bv[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) bv[ns] = (TR == 2) ? (((vmask[ns] >> tp) & 1u) ? r[ns] : 0.f) : r[ns];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms], bv[ns], acc[ms][ns], 0, 0, 0);
      constexpr int NM = MS * NS, RD = MS + NS;
      int done = 0;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        const int n = have_next ? (RD - done + (NM - i) - 1) / (NM - i) : 0;
        if (n == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if (n == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        done += n;
      }
    };
    constexpr int NSTEP = TAPS * HS;
    {
      float a0[MS], r0[NS], a1[MS], r1[NS];
      fetch(0, a0, r0);
#pragma unroll
      for (int st = 0; st < NSTEP; st += 2) {
        if (st + 1 < NSTEP) fetch(st + 1, a1, r1);
        mma(st, a0, r0, st + 1 < NSTEP);
        if (st + 1 < NSTEP) {
          if (st + 2 < NSTEP) fetch(st + 2, a0, r0);
          mma(st + 1, a1, r1, st + 2 < NSTEP);
        }
      }
    }
    if (more) store_lds(c0 + KC2, smem + (it ^ 1) * TC::BUF);
    __syncthreads();
  };
  for (int c0 = 0; c0 < d.Kc; c0 += 2 * KC2) {
    stage(c0, std::integral_constant<int, 0>());
    if (c0 + KC2 < d.Kc) stage(c0 + KC2, std::integral_constant<int, 1>());
  }

  // ---- epilogue (see conv_gemm.hip): off-tile columns hold exact zeros
  const int part = tile * WN + wn;
  auto fast_epilogue = [&](auto EPI_) {
    constexpr int EPI = decltype(EPI_)::value;
    constexpr bool stats = EPI == SAR_EPI_STATS || EPI == SAR_EPI_MASK;
    constexpr bool has_aux = EPI == SAR_EPI_MASK || EPI == SAR_EPI_ADD;
    if (EPI == SAR_EPI_MASK) {
      if (tid < BM) {
        const int row = m0 + tid;
        float4 ap = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < d.M) {
          ap.x = d.aux_scale[row];
          ap.y = d.aux_shift[row];
          if (d.aux_mean) ap.z = d.aux_mean[row];
        }
        rowp[tid] = ap;
      }
      __syncthreads();
    }
    const int rows_w = m0 + wm * MS * 32;
    // num_records = bytes up to the end of the tensor (rows >= M of a partial row block must not be touched: the
    // range check sees voffset + soffset), capped at 2^31 so that the off-tile marker offset stays out of range
    auto rows_bytes = [&](int64_t ld) {
      const int64_t n = (int64_t)(d.M - rows_w) * ld * 4;
      return (unsigned)(n <= 0 ? 0 : (n > 0x80000000ll ? 0x80000000ll : n));
    };
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.out + (int64_t)rows_w * d.ld_out), 0, rows_bytes(d.ld_out), 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_aux ? d.aux + (int64_t)rows_w * d.ld_aux : d.out), 0, has_aux ? rows_bytes(d.ld_aux) : 0u, 0x00020000);
    unsigned vo_out[NS], vo_aux[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      vo_out[ns] = colok[ns] ? (unsigned)((coln[ns] + 4 * hi * d.ld_out) * 4) : 0x80000000u;
      vo_aux[ns] = colok[ns] ? (unsigned)((coln[ns] + 4 * hi * d.ld_aux) * 4) : 0x80000000u;
    }
    const int so_out = (int)(d.ld_out * 4), so_aux = (int)(d.ld_aux * 4);
    float* P = smem + wave * (16 * 65);
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        float ax[NS][16];
        if (has_aux) {
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
              const int r = rb * 8 + r8;
              ax[ns][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                  ra, vo_aux[ns], (ms * 32 + (r & 3) + 8 * (r >> 2)) * so_aux, 0));
            }
        }
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = rb * 8 + r8;
          const bool grp_ok = rows_w + ms * 32 + 8 * (r >> 2) < d.M;
          float s1 = 0.f, s2 = 0.f;
          float4 ap = make_float4(0.f, 0.f, 0.f, 0.f);
          if (EPI == SAR_EPI_MASK) ap = rowp[(wm * MS + ms) * 32 + mfma_row(r, hi)];
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) {
            float val = acc[ms][ns][r];
            if (EPI == SAR_EPI_STATS) {
              s1 += val;
              s2 = fmaf(val, val, s2);
            } else if (EPI == SAR_EPI_MASK) {
              val = (fmaf(ax[ns][r], ap.x, ap.y) > 0.f) ? val : 0.f;
              s1 += val;
              s2 = fmaf(val, ax[ns][r] - ap.z, s2);
            } else if (EPI == SAR_EPI_ADD) {
              val += ax[ns][r];
            }
            if (grp_ok)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), ro, vo_out[ns],
                                                    (ms * 32 + (r & 3) + 8 * (r >> 2)) * so_out, 0);
          }
          if (stats) {
            P[(2 * r8) * 65 + lane] = s1;
            P[(2 * r8 + 1) * 65 + lane] = s2;
          }
        }
        if (stats) {
          __builtin_amdgcn_wave_barrier();
          const int q = lane & 15, sub = (lane >> 4) & 1;
          const float* pr = P + q * 65 + hi * 32 + sub * 16;
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) t += pr[i];
          t += __shfl_xor(t, 16);
          __builtin_amdgcn_wave_barrier();
          const int r = rb * 8 + (q >> 1);
          const int row = rows_w + ms * 32 + mfma_row(r, hi);
          if (sub == 0 && row < d.M) d.partials[((int64_t)row * k.nparts + part) * 2 + (q & 1)] = t;
        }
      }
    }
  };
  if ((d.M & 7) == 0) {
    switch (d.epi) {
      case SAR_EPI_STATS: fast_epilogue(std::integral_constant<int, SAR_EPI_STATS>()); break;
      case SAR_EPI_MASK: fast_epilogue(std::integral_constant<int, SAR_EPI_MASK>()); break;
      case SAR_EPI_ADD: fast_epilogue(std::integral_constant<int, SAR_EPI_ADD>()); break;
      default: fast_epilogue(std::integral_constant<int, SAR_EPI_NONE>()); break;
    }
    return;
  }
  // generic path (M % 8 != 0)
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (wm * MS + ms) * 32 + mfma_row(r, hi);
      const bool rowok = row < d.M;
      float s1 = 0.f, s2 = 0.f, asc = 0.f, ash = 0.f, amu = 0.f;
      if (d.epi == SAR_EPI_MASK && rowok) {
        asc = d.aux_scale[row];
        ash = d.aux_shift[row];
        if (d.aux_mean) amu = d.aux_mean[row];
      }
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        float val = acc[ms][ns][r];
        if (rowok && colok[ns]) {
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

// ------------------------------------------------------------------------------------------------ 7x7/2 stem (Kc == 1)
// K axis = taps: k-step ks multiplies taps (2ks, 2ks+1) of the single input channel.  Tile 64 x 256 (TH rows).
template <int KH, int KW>
__global__ __launch_bounds__(256, 2) void conv2d_stem_kernel(const C2K k) {
  constexpr int TAPS = KH * KW, KS = (TAPS + 1) / 2;
  constexpr int MS = 2, NS = 2, WN = 4, BM = 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const sar_conv2d_desc& d = k.d;
  float* Wl = smem;                  // [2*KS][BM] (tap-major; the odd pad tap is zero)
  float* S = smem + 2 * KS * BM;     // [RW] padded image rows of the single channel
  const int tid = threadIdx.x;
  const int lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int tile = blockIdx.x;
  const int b = tile / k.TPI;
  const int h0 = (tile - b * k.TPI) * k.TH;
  const int m0 = blockIdx.y * BM;
  const int s = d.stride;
  const int row_lo = h0 * s - d.pad;

  for (int idx = tid; idx < 2 * KS * BM; idx += 256) {
    const int m = idx % BM, tp = idx / BM;
    Wl[idx] = (tp < TAPS && m0 + m < d.M) ? d.W[(int64_t)tp * d.w_stride_tap + m0 + m] : 0.f;
  }
  const float* src_b = d.src + (int64_t)b * d.H_src * d.W_src;
  for (int e = tid; e < k.RW; e += 256) {
    const int r = (int)(((float)e + 0.5f) * k.invWq);
    const int q = e - r * k.Wq;
    const int hs = row_lo + r, ws = k.col_lo + q;
    S[e] = ((unsigned)hs < (unsigned)d.H_src && (unsigned)ws < (unsigned)d.W_src) ? src_b[hs * d.W_src + ws] : 0.f;
  }
  __syncthreads();

  bool colok[NS];
  int64_t coln[NS];
  int base[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    int hl = p / d.W_out, wo = p - (p / d.W_out) * d.W_out;
    colok[ns] = hl < k.TH && (h0 + hl) < d.H_out;
    if (!colok[ns]) { hl = 0; wo = 0; }
    coln[ns] = ((int64_t)b * d.H_out + (h0 + hl)) * d.W_out + wo;
    base[ns] = hl * s * k.Wq + wo * s;
  }
  f32x16 acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int t0 = 2 * ks, t1 = (2 * ks + 1 < TAPS) ? 2 * ks + 1 : 0;   // pad tap: weight row is zero
    const int o0 = (t0 / KW) * k.Wq + (t0 % KW), o1 = (t1 / KW) * k.Wq + (t1 % KW);
    const int to = hi ? o1 : o0;
    float a[MS], bv[NS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) a[ms] = Wl[(2 * ks + hi) * BM + ms * 32 + l31];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) bv[ns] = S[base[ns] + to];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms], bv[ns], acc[ms][ns], 0, 0, 0);
  }
  const int part = tile * WN + wn;
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + ms * 32 + mfma_row(r, hi);
      const bool rowok = row < d.M;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const float val = acc[ms][ns][r];
        if (rowok && colok[ns]) {
          s1 += val;
          s2 = fmaf(val, val, s2);
          d.out[(int64_t)row * d.ld_out + coln[ns]] = val;
        }
      }
      if (d.epi == SAR_EPI_STATS) {
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

// ------------------------------------------------------------------------------------------------ weight gradients
struct W2K {
  sar_conv2d_desc d;
  int TH, RP, TPI, NT, NR, Wq, RW, SP, NPOS, DP, col_lo;
  float invWq, invWo;
};

// STEM: the A-operand rows are the TAPS of the single input channel (lane-constant tap offset), one MFMA tile per wave.
// Fixed-geometry reduction of one tile (compile-time output width WO, row pairs RPC, conv stride STRIDEC, padded
// source width WQ): fully unrolled, every LDS address is per-lane base + immediate, operands of step s+1 are read
// while the MFMAs of step s issue (see temporal_tile_fixed in conv_wgrad.hip).  Tap i of this wave is tap t0 + i.
template <int TPW, int KW, int WO, int STRIDEC, int RPC, int WQ>
__device__ __forceinline__ void wgrad2d_tile_fixed(f32x16 (&acc)[TPW], const float* Dbase, const float* Sbase, int t0) {
  constexpr int NSTEP = RPC * WO;
  // the tap offsets depend on the wave's first tap (wave-uniform, two values): fold it into the base pointer
  auto fetch = [&](int st, float& dv, float (&sv)[TPW], const float* const (&Sb)[TPW]) {
    const int rp = st / WO, w = st % WO;
    dv = Dbase[rp * 2 * WO + w];
#pragma unroll
    for (int i = 0; i < TPW; ++i) sv[i] = Sb[i][rp * 2 * STRIDEC * WQ + w * STRIDEC];
  };
  const float* Sb[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tp = t0 + i;
    Sb[i] = Sbase + (tp / KW) * WQ + (tp % KW);
  }
  auto mma = [&](float dv, const float (&sv)[TPW], bool have_next) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[i], dv, acc[i], 0, 0, 0);
    int done = 0;
    const int rd = have_next ? TPW + 1 : 0;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      const int n = (rd - done + (TPW - i) - 1) / (TPW - i);
      if (n == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      else if (n == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      done += n;
    }
  };
  float d0, s0[TPW], d1, s1[TPW];
  fetch(0, d0, s0, Sb);
#pragma unroll
  for (int st = 0; st < NSTEP; st += 2) {
    if (st + 1 < NSTEP) fetch(st + 1, d1, s1, Sb);
    mma(d0, s0, st + 1 < NSTEP);
    if (st + 1 < NSTEP) {
      if (st + 2 < NSTEP) fetch(st + 2, d0, s0, Sb);
      mma(d1, s1, st + 2 < NSTEP);
    }
  }
}

// WO / STRIDEC / RPC != 0: output width, conv stride and row pairs per tile are compile-time (resnet18's layer shapes)
template <int KH, int KW, int STEM, int WF, int WC, int WT, int TPW, int DJ, int SJMAX, int WO = 0, int STRIDEC = 0, int RPC = 0>
__global__ __launch_bounds__(256, 2) void conv2d_wgrad_kernel(const W2K k) {
  constexpr int TAPS = KH * KW;
  constexpr int BF = 32 * WF, CT = STEM ? 1 : 32 * WC;
  constexpr int DI = BF / 8, SI = STEM ? 1 : CT / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const sar_conv2d_desc& d = k.d;
  float* D = smem;                      // [BF][DP]
  float* S = D + BF * k.DP;             // [CT][SP]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wf = wave % WF, wc = (wave / WF) % WC, wt = wave / (WF * WC);
  const int f0 = blockIdx.y * BF, c0 = STEM ? 0 : blockIdx.z * CT;
  const int s = d.stride;

  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int img_s = d.H_src * d.W_src, img_o = d.H_out * d.W_out;
  const bool has_pro = d.pro_scale != nullptr;
  const int r8 = tid >> 5, c32 = tid & 31;
  float dreg[DI][DJ];
  float sreg[SI][SJMAX];
  float psc[SI], psh[SI];
#pragma unroll
  for (int i = 0; i < SI; ++i) {
    const int cg = c0 + (STEM ? 0 : r8 + 8 * i);
    psc[i] = (has_pro && cg < d.Kc) ? d.pro_scale[cg] : 1.f;
    psh[i] = (has_pro && cg < d.Kc) ? d.pro_shift[cg] : 0.f;
  }
  // staged src element e -> pixel; STEM stages its single row with all 256 threads
  auto src_index = [&](int e, int row_lo, bool& ok) -> int {
    const int r = (int)(((float)e + 0.5f) * k.invWq);
    const int q = e - r * k.Wq;
    const int hs = row_lo + r, ws = k.col_lo + q;
    ok = e < k.RW && (unsigned)hs < (unsigned)d.H_src && (unsigned)ws < (unsigned)d.W_src;
    return hs * d.W_src + ws;
  };

  auto issue_loads = [&](int tile) {
    const int b = tile / k.TPI;
    const int h0 = (tile - b * k.TPI) * k.TH;
    const int row_lo = h0 * s - d.pad;
    const float* dout_b = d.dout + (int64_t)b * img_o + (int64_t)h0 * d.W_out;
    const int lim = (d.H_out - h0) * d.W_out;   // positions of this image still inside
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int f = f0 + r8 + 8 * i;
      const float* rowp = dout_b + (int64_t)(f < d.M ? f : 0) * d.ld_dout;
#pragma unroll
      for (int j = 0; j < DJ; ++j) {
        const int p = c32 + 32 * j;
        const bool ok = f < d.M && p < k.NPOS && p < lim;
        dreg[i][j] = rowp[ok ? p : 0];
      }
    }
    const float* src_b = d.src + (int64_t)b * img_s;
#pragma unroll
    for (int i = 0; i < SI; ++i) {
      const int cg = c0 + (STEM ? 0 : r8 + 8 * i);
      const float* rowp = src_b + (int64_t)(cg < d.Kc ? cg : 0) * d.ld_src;
#pragma unroll
      for (int j = 0; j < SJMAX; ++j) {
        bool ok;
        const int idx = src_index(STEM ? tid + 256 * j : c32 + 32 * j, row_lo, ok);
        sreg[i][j] = rowp[(ok && cg < d.Kc) ? idx : 0];
      }
    }
  };

  auto store_lds = [&](int tile) {
    const int b = tile / k.TPI;
    const int h0 = (tile - b * k.TPI) * k.TH;
    const int row_lo = h0 * s - d.pad;
    const int lim = (d.H_out - h0) * d.W_out;
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int fr = r8 + 8 * i;
#pragma unroll
      for (int j = 0; j < DJ; ++j) {
        const int p = c32 + 32 * j;
        const bool ok = (f0 + fr) < d.M && p < k.NPOS && p < lim;
        if (p < k.NPOS) D[fr * k.DP + p] = ok ? dreg[i][j] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < SI; ++i) {
      const int cr = STEM ? 0 : r8 + 8 * i;
#pragma unroll
      for (int j = 0; j < SJMAX; ++j) {
        const int e = STEM ? tid + 256 * j : c32 + 32 * j;
        if (e < k.RW) {
          bool ok;
          (void)src_index(e, row_lo, ok);
          float val = sreg[i][j];
          if (has_pro) {
            val = fmaf(val, psc[i], psh[i]);
            if (d.pro_relu) val = fmaxf(val, 0.f);
          }
          S[cr * k.SP + e] = (ok && (c0 + cr) < d.Kc) ? val : 0.f;
        }
      }
    }
  };

  // operand bases: lanes 0-31 reduce over the even output row of a pair, lanes 32-63 over the odd one
  const int sfs = s * k.Wq;                                        // src elements per output row
  const float* Dbase = D + (wf * 32 + l31) * k.DP + hi * d.W_out;
  const float* Sbase;
  if (STEM) {
    int j = wc * 32 + l31;                                         // this lane's tap
    if (j >= TAPS) j = 0;                                          // rows >= TAPS are never stored
    Sbase = S + (j / KW) * k.Wq + (j % KW) + hi * sfs;
  } else {
    Sbase = S + (wc * 32 + l31) * k.SP + hi * sfs;
  }

  int tile = blockIdx.x;
  if (tile < k.NT) issue_loads(tile);
  for (; tile < k.NT; tile += gridDim.x) {
    __syncthreads();
    store_lds(tile);
    __syncthreads();
    if (tile + (int)gridDim.x < k.NT) issue_loads(tile + gridDim.x);
    if constexpr (WO != 0 && !STEM) {
      constexpr int WQ = WO * STRIDEC + 2 * (KH / 2);
      wgrad2d_tile_fixed<TPW, KW, WO, STRIDEC, RPC, WQ>(acc, Dbase, Sbase, wt * TPW);
      continue;
    }
    for (int rp = 0; rp < k.RP; ++rp) {
      const float* Dq = Dbase + rp * 2 * d.W_out;
      const float* Sq = Sbase + rp * 2 * sfs;
      float dn = Dq[0], sn[TPW];
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int tp = STEM ? 0 : wt * TPW + i;
        sn[i] = Sq[(tp / KW) * k.Wq + (tp % KW)];
      }
      for (int w = 0; w < d.W_out; ++w) {
        const float dc = dn;
        float sc[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) sc[i] = sn[i];
        const int wnx = (w + 1 < d.W_out) ? w + 1 : w;
        dn = Dq[wnx];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int tp = STEM ? 0 : wt * TPW + i;
          sn[i] = Sq[(tp / KW) * k.Wq + (tp % KW) + wnx * s];
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[i], dc, acc[i], 0, 0, 0);
      }
    }
  }

  // slab layout (tap, c, m) with m contiguous
  float* slab = d.slab + (int64_t)blockIdx.x * ((int64_t)TAPS * d.Kc * d.M);
  const int f = f0 + wf * 32 + l31;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (STEM) {
        const int tp = wc * 32 + mfma_row(r, hi);
        if (tp < TAPS && f < d.M) slab[(int64_t)tp * d.M + f] = acc[i][r];
      } else {
        const int tp = wt * TPW + i;
        const int c = c0 + wc * 32 + mfma_row(r, hi);
        if (tp < TAPS && c < d.Kc && f < d.M) slab[((int64_t)tp * d.Kc + c) * d.M + f] = acc[i][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ small kernels
__global__ void permute3_kernel(const float* __restrict__ in, float* __restrict__ out, int d0, int d1, int d2, int64_t s0,
                                int64_t s1, int64_t s2) {
  const int64_t n = (int64_t)d0 * d1 * d2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i % d2);
    const int64_t r = i / d2;
    const int j = (int)(r % d1), ii = (int)(r / d1);
    out[i] = in[ii * s0 + j * s1 + kk * s2];
  }
}

__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, float* __restrict__ y,
                                                                  int B, int H, int W, int Ho, int Wo, int64_t ld_x,
                                                                  int64_t ld_y) {
  const int c = blockIdx.y;
  const float a = scale[c], bsh = shift[c];
  const int64_t n = (int64_t)B * Ho * Wo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int wo = (int)(i % Wo);
    const int64_t r = i / Wo;
    const int ho = (int)(r % Ho), b = (int)(r / Ho);
    const float* xp = x + (int64_t)c * ld_x + (int64_t)b * H * W;
    float m = -INFINITY;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int h = ho * 2 - 1 + kh, w = wo * 2 - 1 + kw;
        if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) m = fmaxf(m, fmaxf(fmaf(xp[h * W + w], a, bsh), 0.f));
      }
    y[(int64_t)c * ld_y + i] = m;
  }
}

// Backward of BN+ReLU+MaxPool(3, 2, 1) in gather form, tiled: a workgroup owns a 32 x 32 block of input pixels of one
// (channel, image).  It activates the 35 x 35 pixels its 17 x 17 candidate windows touch ONCE into LDS, finds every
// window's first maximum in row-major scan order (torch's max_pool2d tie rule) ONCE, then every owned pixel looks up
// the <= 4 windows that contain it.  (The first version re-scanned 4 windows x 9 pixels per input pixel: 848 us for
// the resnet18 stem at bs = 32.)  Deterministic: no atomics; BN partial sums per workgroup.
constexpr int MPB_T = 32, MPB_A = MPB_T + 3, MPB_W = MPB_T / 2 + 1;
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  const float* __restrict__ mean, const float* __restrict__ dy,
                                                                  float* __restrict__ dz, float* __restrict__ partials,
                                                                  int B, int H, int W, int Ho, int Wo, int tiles_h, int tiles_w,
                                                                  int64_t ld_x, int64_t ld_y) {
  __shared__ float act[MPB_A * MPB_A];          // activated pixels, rows h0-1 .. h0+33
  __shared__ short wmax[MPB_W * MPB_W];         // argmax of each window as local pixel index (or -1)
  __shared__ float wdy[MPB_W * MPB_W];
  const int c = blockIdx.y;
  const int tile = blockIdx.x;
  const int b = tile / (tiles_h * tiles_w), tr = tile - b * tiles_h * tiles_w;
  const int h0 = (tr / tiles_w) * MPB_T, w0 = (tr % tiles_w) * MPB_T;
  const float a = scale[c], bsh = shift[c], mu = mean ? mean[c] : 0.f;
  const float* xp = x + (int64_t)c * ld_x + (int64_t)b * H * W;
  for (int i = threadIdx.x; i < MPB_A * MPB_A; i += 256) {
    const int lh = i / MPB_A, lw = i - lh * MPB_A;
    const int h = h0 - 1 + lh, w = w0 - 1 + lw;
    // outside the image: -inf so that it never wins a window (MaxPool pads with -inf)
    act[i] = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? fmaxf(fmaf(xp[h * W + w], a, bsh), 0.f) : -INFINITY;
  }
  __syncthreads();
  const int ho0 = h0 / 2, wo0 = w0 / 2;
  for (int i = threadIdx.x; i < MPB_W * MPB_W; i += 256) {
    const int lho = i / MPB_W, lwo = i - lho * MPB_W;
    const int ho = ho0 + lho, wo = wo0 + lwo;
    short am = -1;
    float g = 0.f;
    if (ho < Ho && wo < Wo) {
      float m = -INFINITY;
      // window (ho, wo) covers rows 2ho-1 .. 2ho+1 = local rows 2 lho .. 2 lho + 2
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int li = (2 * lho + kh) * MPB_A + 2 * lwo + kw;
          const float v = act[li];
          if (v > m) { m = v; am = (short)li; }
        }
      g = dy[(int64_t)c * ld_y + ((int64_t)b * Ho + ho) * Wo + wo];
    }
    wmax[i] = am;
    wdy[i] = g;
  }
  __syncthreads();
  float accs[2] = {0.f, 0.f};
  for (int i = threadIdx.x; i < MPB_T * MPB_T; i += 256) {
    const int lh = i / MPB_T, lw = i - lh * MPB_T;
    const int h = h0 + lh, w = w0 + lw;
    if (h >= H || w >= W) continue;
    const int li = (lh + 1) * MPB_A + lw + 1;
    float g = 0.f;
    if (act[li] > 0.f) {      // ReLU mask: an inactive pixel receives nothing
      // an even row lies in one window, an odd row in two: ho = h/2 .. (h+1)/2
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
          const int lho = (lh + dh) / 2, lwo = (lw + dw) / 2;
          if ((dh == 1 && !(lh & 1)) || (dw == 1 && !(lw & 1))) continue;   // even index: a single window
          if (wmax[lho * MPB_W + lwo] == (short)li) g += wdy[lho * MPB_W + lwo];
        }
    }
    const float xv = xp[h * W + w];
    dz[(int64_t)c * ld_x + (int64_t)b * H * W + h * W + w] = g;
    accs[0] += g;
    accs[1] = fmaf(g, xv - mu, accs[1]);
  }
  __shared__ float red[2][4];
  const float s1 = wave_sum(accs[0]), s2 = wave_sum(accs[1]);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* pp = partials + ((int64_t)c * gridDim.x + blockIdx.x) * 2;
    pp[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    pp[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, int64_t n, const float* __restrict__ lr_dev,
                                                   const float* __restrict__ step_dev, float b1, float b2, float eps) {
  const float lr = lr_dev[0], t = step_dev[0];
  const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w[i] = w[i] - (lr / bc1) * mi / (sqrtf(vi) / bc2s + eps);
  }
}

// ------------------------------------------------------------------------------------------------ host side
int gemm_geometry(const sar_conv2d_desc& d, int tile_n, int wn, C2K& k, int rwmax = 1 << 30, bool multi_image = false) {
  if (d.W_out > tile_n) return -2;
  int th = tile_n / d.W_out;
  if (th > d.H_out) th = d.H_out;
  for (;; --th) {   // as many whole output rows as fit the tile AND the staged-image budget
    k.TH = th;
    if (!d.transposed) {
      k.NR = (th - 1) * d.stride + d.KH;
      k.Wq = d.W_src + 2 * d.pad;
      k.col_lo = -d.pad;
    } else {
      k.NR = (th - 1 + d.KH - 1) / d.stride + 2;
      const int lo = d.pad - (d.KW - 1);
      const int clo = lo >= 0 ? lo / d.stride : -((-lo + d.stride - 1) / d.stride);
      const int chi = (d.W_out - 1 + d.pad) / d.stride;
      k.col_lo = clo;
      k.Wq = chi - clo + 1;
    }
    k.RW = k.NR * k.Wq;
    if (k.RW <= rwmax || th == 1) break;
  }
  k.TPI = (d.H_out + k.TH - 1) / k.TH;
  k.IRW = k.RW;
  k.NI = 1;
  if (multi_image && k.TH == d.H_out && 2 * d.H_out * d.W_out <= tile_n) {   // several whole images per tile
    int ni = tile_n / (d.H_out * d.W_out);
    if (ni > d.B) ni = d.B;
    while (ni > 1 && ni * k.IRW > rwmax) --ni;
    k.NI = ni;
    k.RW = ni * k.IRW;
  }
  k.ntiles = k.NI > 1 ? (d.B + k.NI - 1) / k.NI : d.B * k.TPI;
  k.SROW = k.RW;
  k.invWq = 1.0f / (float)k.Wq;
  k.nparts = k.ntiles * wn;
  k.w_vec = ((d.Kc % KC2) == 0 && (d.M & 3) == 0 && (d.w_stride_c & 3) == 0 && (d.w_stride_tap & 3) == 0 && ((uintptr_t)d.W & 15) == 0) ? 1 : 0;
  return 0;
}

template <int TR, int KH, int KW>
int launch_gemm_tr(const sar_conv2d_desc& d, hipStream_t st, bool query, int* nparts_out) {
  C2K k;
  k.d = d;
  constexpr int TAPS = KH * KW;
  const int rwmax = (TAPS == 1 ? 512 : 640);
  bool small_grid = false;
  if (d.M > 64) {   // the deep resnet layers at small batch: 128 x 128 tiles would leave most CUs without a workgroup
    C2K kk;
    kk.d = d;
    if (gemm_geometry(d, 128, 2, kk, rwmax, true) == 0) small_grid = (int64_t)kk.ntiles * ((d.M + 127) / 128) < 384;
  }
  bool tiny_grid = false;
  if (small_grid) {
    C2K kk;
    kk.d = d;
    if (gemm_geometry(d, 128, 4, kk, rwmax, true) == 0) tiny_grid = (int64_t)kk.ntiles * ((d.M + 63) / 64) < 192;
  }
  if (d.M > 64 && tiny_grid) {
    constexpr int MS = 1, NS = 1, WM = 1, WN = 4;   // 32 x 128 tile: four times the workgroups
    if (int g = gemm_geometry(d, 32 * NS * WN, WN, k, rwmax, true)) return g;
    if (nparts_out) *nparts_out = k.nparts;
    if (query) return 0;
    if (k.RW > rwmax) return -2;
    dim3 grid(k.ntiles, (d.M + 32 * MS * WM - 1) / (32 * MS * WM));
    hipLaunchKernelGGL((conv2d_gemm_kernel<TR, KH, KW, MS, NS, WM, WN>), grid, dim3(256), 0, st, k);
  } else if (d.M > 64 && small_grid) {
    constexpr int MS = 2, NS = 1, WM = 1, WN = 4;   // 64 x 128 tile: twice the workgroups
    if (int g = gemm_geometry(d, 32 * NS * WN, WN, k, rwmax, true)) return g;
    if (nparts_out) *nparts_out = k.nparts;
    if (query) return 0;
    if (k.RW > rwmax) return -2;
    dim3 grid(k.ntiles, (d.M + 32 * MS * WM - 1) / (32 * MS * WM));
    hipLaunchKernelGGL((conv2d_gemm_kernel<TR, KH, KW, MS, NS, WM, WN>), grid, dim3(256), 0, st, k);
  } else if (d.M > 64) {
    constexpr int MS = 2, NS = 2, WM = 2, WN = 2;
    if (int g = gemm_geometry(d, 32 * NS * WN, WN, k, rwmax, true)) return g;
    if (nparts_out) *nparts_out = k.nparts;
    if (query) return 0;
    if (k.RW > rwmax) return -2;
    dim3 grid(k.ntiles, (d.M + 32 * MS * WM - 1) / (32 * MS * WM));
    hipLaunchKernelGGL((conv2d_gemm_kernel<TR, KH, KW, MS, NS, WM, WN>), grid, dim3(256), 0, st, k);
  } else {
    constexpr int MS = 2, NS = 2, WM = 1, WN = 4;
    if (int g = gemm_geometry(d, 32 * NS * WN, WN, k, rwmax, true)) return g;
    if (nparts_out) *nparts_out = k.nparts;
    if (query) return 0;
    if (k.RW > rwmax) return -2;
    dim3 grid(k.ntiles, (d.M + 32 * MS * WM - 1) / (32 * MS * WM));
    hipLaunchKernelGGL((conv2d_gemm_kernel<TR, KH, KW, MS, NS, WM, WN>), grid, dim3(256), 0, st, k);
  }
  return 0;
}

template <int TRANSPOSED, int KH, int KW>
int launch_gemm(const sar_conv2d_desc& d, hipStream_t st, bool query, int* nparts_out) {
  if (!TRANSPOSED) return launch_gemm_tr<0, KH, KW>(d, st, query, nparts_out);
  if (d.stride == 1) return launch_gemm_tr<1, KH, KW>(d, st, query, nparts_out);
  return launch_gemm_tr<2, KH, KW>(d, st, query, nparts_out);
}

int launch_stem(const sar_conv2d_desc& d, hipStream_t st, bool query, int* nparts_out) {
  C2K k;
  k.d = d;
  if (int g = gemm_geometry(d, 256, 4, k)) return g;
  if (nparts_out) *nparts_out = k.nparts;
  if (query) return 0;
  const size_t lds = sizeof(float) * (50 * 64 + (size_t)k.RW);
  if (lds > 64 * 1024) return -2;
  dim3 grid(d.B * k.TPI, (d.M + 63) / 64);
  hipLaunchKernelGGL((conv2d_stem_kernel<7, 7>), grid, dim3(256), lds, st, k);
  return 0;
}

int dispatch_gemm(const sar_conv2d_desc& d, hipStream_t st, bool query, int* np) {
  const bool stem = d.KH == 7 && d.KW == 7 && d.Kc == 1 && !d.transposed;
  if (stem) return launch_stem(d, st, query, np);
  if (d.KH == 3 && d.KW == 3) return d.transposed ? launch_gemm<1, 3, 3>(d, st, query, np) : launch_gemm<0, 3, 3>(d, st, query, np);
  if (d.KH == 1 && d.KW == 1) return d.transposed ? launch_gemm<1, 1, 1>(d, st, query, np) : launch_gemm<0, 1, 1>(d, st, query, np);
  return -2;
}

int check_common(const sar_conv2d_desc* d, const char* who) {
  SAR_REQUIRE(d != nullptr, "%s: null descriptor", who);
  SAR_REQUIRE(d->B > 0 && d->Kc > 0 && d->M > 0 && d->H_src > 0 && d->W_src > 0 && d->H_out > 0 && d->W_out > 0,
              "%s: bad sizes", who);
  SAR_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride >= 1 && d->pad >= 0, "%s: bad kernel/stride/pad", who);
  SAR_REQUIRE(d->src && d->ld_src >= (int64_t)d->B * d->H_src * d->W_src, "%s: bad src", who);
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "%s: pro_scale/pro_shift mismatch", who);
  return 0;
}

template <int KH, int KW, int STEM, int WF, int WC, int WT, int TPW, int DJ, int SJMAX, int WO = 0, int STRIDEC = 0, int RPC = 0>
int launch_wgrad(const sar_conv2d_desc& d, hipStream_t st) {
  constexpr int TAPS = KH * KW, BF = 32 * WF, CT = STEM ? 1 : 32 * WC;
  W2K k;
  k.d = d;
  int th = (32 * DJ / d.W_out) & ~1;
  if (th < 2) return -2;
  const int h_even = (d.H_out + 1) & ~1;
  if (th > h_even) th = h_even;
  size_t lds = 0;
  for (;; th -= 2) {
    k.TH = th;
    k.RP = th / 2;
    k.TPI = (d.H_out + th - 1) / th;
    k.NT = d.B * k.TPI;
    k.NPOS = th * d.W_out;
    k.DP = k.NPOS | 1;
    k.NR = (th - 1) * d.stride + d.KH;
    k.Wq = d.W_src + 2 * d.pad;
    k.col_lo = -d.pad;
    k.RW = k.NR * k.Wq;
    k.SP = (k.RW + (STEM ? 0 : (WT * TPW - TAPS) * (k.Wq + KW))) | 1;
    lds = sizeof(float) * ((size_t)BF * k.DP + (size_t)CT * k.SP);
    if ((lds <= 78 * 1024 && k.RW <= (STEM ? 256 : 32) * SJMAX && k.NPOS <= 32 * DJ) || th == 2) break;
  }
  k.invWq = 1.0f / (float)k.Wq;
  k.invWo = 1.0f / (float)d.W_out;
  if (k.NPOS > 32 * DJ || k.RW > (STEM ? 256 : 32) * SJMAX || lds > 150 * 1024) return -2;
  if constexpr (WO != 0) {   // the fixed-geometry loop needs exactly this tile shape, else the generic loop
    if (d.W_out != WO || d.stride != STRIDEC || k.RP != RPC || d.pad != KH / 2 || d.W_src + 2 * d.pad != WO * STRIDEC + 2 * (KH / 2))
      return launch_wgrad<KH, KW, STEM, WF, WC, WT, TPW, DJ, SJMAX>(d, st);
  }
  auto kern = conv2d_wgrad_kernel<KH, KW, STEM, WF, WC, WT, TPW, DJ, SJMAX, WO, STRIDEC, RPC>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid(d.nsplit, (d.M + BF - 1) / BF, STEM ? 1 : (d.Kc + CT - 1) / CT);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, k);
  return 0;
}

// Data gradient of the one-input-channel stem conv (needed only when something upstream of the image trains:
// the radar location / wavelength, main_spectrogram.py:133-136).  Cin = 1, so this is 51 MMAC per image of plain
// FMA work: one thread per image pixel walks the <= ceil(KH/s) x ceil(KW/s) taps that reach it and the M filters;
// the packed weights [tap][m] sit in LDS.
__global__ __launch_bounds__(256) void conv2d_stem_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ Wt,
                                                                int B, int H, int W, int Ho, int Wo, int M, int KH, int KW,
                                                                int stride, int pad, int64_t ld_dout,
                                                                float* __restrict__ dx) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [KH*KW][M]
  for (int i = threadIdx.x; i < KH * KW * M; i += blockDim.x) wl[i] = Wt[i];
  __syncthreads();
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * H * W) return;
  const int w = (int)(gid % W);
  const int h = (int)((gid / W) % H);
  const int b = (int)(gid / ((int64_t)W * H));
  float acc = 0.f;
  for (int kh = 0; kh < KH; ++kh) {
    const int hh = h + pad - kh;
    if (hh < 0 || hh % stride != 0) continue;
    const int ho = hh / stride;
    if (ho >= Ho) continue;
    for (int kw = 0; kw < KW; ++kw) {
      const int ww = w + pad - kw;
      if (ww < 0 || ww % stride != 0) continue;
      const int wo = ww / stride;
      if (wo >= Wo) continue;
      const float* dp = dout + ((int64_t)b * Ho + ho) * Wo + wo;
      const float* wp = wl + (kh * KW + kw) * M;
      for (int m = 0; m < M; ++m) acc = fmaf(dp[(int64_t)m * ld_dout], wp[m], acc);
    }
  }
  dx[gid] = acc;
}

}  // namespace

extern "C" int sar_conv2d_nparts(const sar_conv2d_desc* d) {
  if (!d || d->W_out <= 0 || d->H_out <= 0 || d->B <= 0 || d->M <= 0) return SAR_E_ARG;
  int np = 0;
  int rc = dispatch_gemm(*d, nullptr, true, &np);
  return rc ? SAR_E_UNSUP : np;
}

extern "C" int sar_conv2d_gemm_f32(const sar_conv2d_desc* d, sar_stream_t s) {
  int rc = check_common(d, "sar_conv2d_gemm");
  if (rc) return rc;
  SAR_REQUIRE(d->out && d->W && d->ld_out >= (int64_t)d->B * d->H_out * d->W_out, "sar_conv2d_gemm: bad out/W");
  SAR_REQUIRE(d->epi >= SAR_EPI_NONE && d->epi <= SAR_EPI_ADD, "sar_conv2d_gemm: bad epilogue");
  if (d->epi == SAR_EPI_STATS || d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->partials, "sar_conv2d_gemm: partials required");
  if (d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD) SAR_REQUIRE(d->aux, "sar_conv2d_gemm: aux required");
  if (d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->aux_scale && d->aux_shift, "sar_conv2d_gemm: aux affine required");
  rc = dispatch_gemm(*d, as_stream(s), false, nullptr);
  if (rc == -2) {
    sar_set_error("sar_conv2d_gemm: %dx%d stride %d (Kc=%d, W_out=%d) is not built / does not fit a tile", d->KH, d->KW,
                  d->stride, d->Kc, d->W_out);
    return SAR_E_UNSUP;
  }
  if (rc) return rc;
  SAR_LAUNCH_CHECK("sar_conv2d_gemm_f32");
  return 0;
}

extern "C" int sar_conv2d_wgrad_f32(const sar_conv2d_desc* d, sar_stream_t s) {
  int rc = check_common(d, "sar_conv2d_wgrad");
  if (rc) return rc;
  SAR_REQUIRE(d->dout && d->slab && d->nsplit >= 1 && d->nsplit <= 65535 && !d->transposed, "sar_conv2d_wgrad: bad arguments");
  SAR_REQUIRE(d->ld_dout >= (int64_t)d->B * d->H_out * d->W_out, "sar_conv2d_wgrad: bad dout leading dimension");
  hipStream_t st = as_stream(s);
  if (d->KH == 7 && d->KW == 7 && d->Kc == 1) rc = launch_wgrad<7, 7, 1, 2, 2, 1, 1, 8, 10>(*d, st);
  else if (d->KH == 3 && d->KW == 3) {
    // resnet18's 3x3 layers at 256x256 input: (W_out, stride, row pairs per tile)
    const int wo = d->W_out, sd = d->stride;
    if (wo == 64 && sd == 1) rc = launch_wgrad<3, 3, 0, 2, 1, 2, 5, 4, 12, 64, 1, 1>(*d, st);
    else if (wo == 32 && sd == 1) rc = launch_wgrad<3, 3, 0, 2, 1, 2, 5, 4, 12, 32, 1, 2>(*d, st);
    else if (wo == 32 && sd == 2) rc = launch_wgrad<3, 3, 0, 2, 1, 2, 5, 4, 12, 32, 2, 1>(*d, st);
    else if (wo == 16 && sd == 1) rc = launch_wgrad<3, 3, 0, 2, 1, 2, 5, 4, 12, 16, 1, 4>(*d, st);
    else if (wo == 16 && sd == 2) rc = launch_wgrad<3, 3, 0, 2, 1, 2, 5, 4, 12, 16, 2, 2>(*d, st);
    else if (wo == 8 && sd == 1) rc = launch_wgrad<3, 3, 0, 2, 1, 2, 5, 4, 12, 8, 1, 4>(*d, st);
    else if (wo == 8 && sd == 2) rc = launch_wgrad<3, 3, 0, 2, 1, 2, 5, 4, 12, 8, 2, 4>(*d, st);
    else rc = launch_wgrad<3, 3, 0, 2, 1, 2, 5, 4, 12>(*d, st);
  }
  else if (d->KH == 1 && d->KW == 1) rc = launch_wgrad<1, 1, 0, 2, 2, 1, 1, 4, 8>(*d, st);
  else rc = -2;
  if (rc == -2) {
    sar_set_error("sar_conv2d_wgrad: %dx%d stride %d (Kc=%d, W_out=%d) is not built / does not fit a tile", d->KH, d->KW,
                  d->stride, d->Kc, d->W_out);
    return SAR_E_UNSUP;
  }
  if (rc) return rc;
  SAR_LAUNCH_CHECK("sar_conv2d_wgrad_f32");
  return 0;
}

extern "C" int sar_permute3_f32(const float* in, float* out, int d0, int d1, int d2, int64_t s0, int64_t s1, int64_t s2,
                                sar_stream_t s) {
  SAR_REQUIRE(in && out && d0 > 0 && d1 > 0 && d2 > 0, "sar_permute3: bad arguments");
  const int64_t n = (int64_t)d0 * d1 * d2;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(permute3_kernel, dim3(blocks), dim3(256), 0, as_stream(s), in, out, d0, d1, d2, s0, s1, s2);
  SAR_LAUNCH_CHECK("sar_permute3_f32");
  return 0;
}

extern "C" int sar_bn_relu_maxpool_fwd_f32(const float* x, const float* scale, const float* shift, float* y, int C, int B,
                                           int H, int W, int64_t ld_x, int64_t ld_y, sar_stream_t s) {
  SAR_REQUIRE(x && scale && shift && y && C > 0 && B > 0 && H > 1 && W > 1, "sar_bn_relu_maxpool_fwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  SAR_REQUIRE(ld_x >= (int64_t)B * H * W && ld_y >= (int64_t)B * Ho * Wo, "sar_bn_relu_maxpool_fwd: bad leading dimensions");
  int blocks = (int)(((int64_t)B * Ho * Wo + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3(blocks, C), dim3(256), 0, as_stream(s), x, scale, shift, y, B, H, W, Ho,
                     Wo, ld_x, ld_y);
  SAR_LAUNCH_CHECK("sar_bn_relu_maxpool_fwd_f32");
  return 0;
}

extern "C" int sar_bn_relu_maxpool_bwd_nparts(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return SAR_E_ARG;
  return B * ((H + MPB_T - 1) / MPB_T) * ((W + MPB_T - 1) / MPB_T);
}

extern "C" int sar_bn_relu_maxpool_bwd_f32(const float* x, const float* scale, const float* shift, const float* mean,
                                           const float* dy, float* dz, float* partials, int nparts, int C, int B, int H, int W,
                                           int64_t ld_x, int64_t ld_y, sar_stream_t s) {
  SAR_REQUIRE(x && scale && shift && dy && dz && partials && nparts > 0 && nparts <= 65535 && C > 0 && B > 0 && H > 1 && W > 1,
              "sar_bn_relu_maxpool_bwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  SAR_REQUIRE(ld_x >= (int64_t)B * H * W && ld_y >= (int64_t)B * Ho * Wo, "sar_bn_relu_maxpool_bwd: bad leading dimensions");
  const int th = (H + MPB_T - 1) / MPB_T, tw = (W + MPB_T - 1) / MPB_T;
  SAR_REQUIRE(nparts == B * th * tw, "sar_bn_relu_maxpool_bwd: nparts must be sar_bn_relu_maxpool_bwd_nparts(B, H, W) = %d",
              B * th * tw);
  hipLaunchKernelGGL(bn_relu_maxpool_bwd_kernel, dim3(nparts, C), dim3(256), 0, as_stream(s), x, scale, shift, mean, dy, dz,
                     partials, B, H, W, Ho, Wo, th, tw, ld_x, ld_y);
  SAR_LAUNCH_CHECK("sar_bn_relu_maxpool_bwd_f32");
  return 0;
}

extern "C" int sar_adam_f32(float* w, float* m, float* v, const float* g, int64_t n, const float* lr_dev,
                            const float* step_dev, float beta1, float beta2, float eps, sar_stream_t s) {
  SAR_REQUIRE(w && m && v && g && lr_dev && step_dev && n > 0, "sar_adam: bad arguments");
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, as_stream(s), w, m, v, g, n, lr_dev, step_dev, beta1, beta2, eps);
  SAR_LAUNCH_CHECK("sar_adam_f32");
  return 0;
}

extern "C" int sar_conv2d_stem_dgrad_f32(const float* dout, int64_t ld_dout, const float* w_packed, int B, int H, int W,
                                         int H_out, int W_out, int M, int KH, int KW, int stride, int pad, float* dx,
                                         sar_stream_t s) {
  SAR_REQUIRE(dout && w_packed && dx, "sar_conv2d_stem_dgrad: null pointer");
  SAR_REQUIRE(B > 0 && H > 0 && W > 0 && H_out > 0 && W_out > 0 && M > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
              "sar_conv2d_stem_dgrad: bad sizes");
  SAR_REQUIRE(ld_dout >= (int64_t)B * H_out * W_out, "sar_conv2d_stem_dgrad: leading dimension smaller than B*Ho*Wo");
  const size_t lds = sizeof(float) * KH * KW * M;
  SAR_REQUIRE(lds <= 64 * 1024, "sar_conv2d_stem_dgrad: weight tile too large");
  const int64_t n = (int64_t)B * H * W;
  hipLaunchKernelGGL(conv2d_stem_dgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, as_stream(s), dout,
                     w_packed, B, H, W, H_out, W_out, M, KH, KW, stride, pad, ld_dout, dx);
  SAR_LAUNCH_CHECK("sar_conv2d_stem_dgrad_f32");
  return 0;
}

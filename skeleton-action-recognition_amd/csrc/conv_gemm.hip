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
#include <type_traits>

namespace {

// Diagnostic build -DSAR_FP32_TL (tools/fp32_timeline.sh): wave 0 of every workgroup of the LAST launch writes one
// row -- start / end in 100 MHz ticks, HW_ID, XCC_ID, the shader-clock cycles it spent in each phase (summed over the stages).
// No stamp executes in the product build.
#ifdef SAR_FP32_TL
constexpr int FP32_TL_WG = 16384;
__device__ unsigned g_fp32_tl[FP32_TL_WG][16];
#define FP32_TL_BEGIN()                                                 \
  const unsigned long long tl_rt0 = __builtin_amdgcn_s_memrealtime();   \
  unsigned long long tl_last = __builtin_amdgcn_s_memtime();            \
  unsigned tl_acc[10] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}
#define FP32_TL(i)                                                \
  do {                                                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
    tl_acc[i] += (unsigned)(t_ - tl_last);                        \
    tl_last = t_;                                                 \
  } while (0)
#define FP32_TL_END()                                                                        \
  do {                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < FP32_TL_WG) {                                      \
      unsigned* row = g_fp32_tl[blockIdx.x];                                                \
      row[0] = (unsigned)tl_rt0, row[1] = (unsigned)__builtin_amdgcn_s_memrealtime();       \
      row[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                   \
      row[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                                  \
      for (int i_ = 0; i_ < 10; ++i_) row[4 + i_] = tl_acc[i_];                             \
    }                                                                                       \
  } while (0)
#else
#define FP32_TL_BEGIN()
#define FP32_TL(i)
#define FP32_TL_END()
#endif

constexpr int KC = 4;  // src channels staged per main-loop iteration (2 MFMA k-steps); one S row per wave

#ifndef SAR_XCD_MAP
#define SAR_XCD_MAP 1
#endif
struct ConvK {
  sar_conv_desc d;
  int FT, TPS, NF, RW, nparts, ntiles, ny;
  int w_vec;   // weight rows may be read as aligned float4
};

// compile-time tile description shared by the kernel and the host launcher
template <int MODE, int TAPS, int MS, int NS, int WM, int WN>
struct TileCfg {
  static constexpr int BM = 32 * MS * WM;
  static constexpr int TN = 32 * NS * WN;
  static constexpr int WROWS = TAPS * KC;
  static constexpr int WRPP = 1024 / BM;                        // W rows staged per pass (one float4 per lane)
  static constexpr int WIT = (WROWS + WRPP - 1) / WRPP;         // W passes
  static constexpr int WSTR = BM;                               // W row stride
  static constexpr int RWMAX = (MODE == SAR_CONV_GRAPH) ? TN : (TN == 128 ? 448 : 704);
  static constexpr int SJ = RWMAX / 64;                         // S columns per lane
  static constexpr int SSTR = RWMAX + 8;                        // S row stride (room for the zero column; RWMAX + 32 would put the two k rows of a read
                                                                // fall into different bank halves)
  static constexpr int WPAD = ((WROWS + 3) / 4) * 4;            // W rows kept in LDS (the last stager pass is masked)
  static constexpr int BUF = WPAD * WSTR + KC * SSTR;           // floats per LDS buffer (3 workgroups per CU fit 160 KB)
};

// TR selects the temporal variant: 0 forward; 1 data gradient at stride 1 (every tap valid); 2 data gradient,
// generic stride (per-lane tap validity mask); 3 data gradient at stride 2 with the PARITY-SPLIT column map: the
// first half of the tile's columns holds the even frames and the second half the odd ones, so a wave's columns
// share the frame parity and only the <= 5 taps that reach a real source frame are issued at all (the masked
// variant spends half of its MFMAs on zeros).
#ifndef SAR_OCC3
#define SAR_OCC3 1   // 3 workgroups per CU (LDS <= 53 KB and <= 168 VGPR per workgroup)
#endif
template <int MODE, int TR, int TAPS, int MS, int NS, int WM, int WN, int NZ0, int NZ1, int NZ2>
__global__ __launch_bounds__(256, (SAR_OCC3 && TR != 2) ? 3 : 2) void conv_gemm_kernel(const ConvK k) {
  using TC = TileCfg<MODE, TAPS, MS, NS, WM, WN>;
  constexpr int TRANSPOSED = TR != 0;
  constexpr int PAR = (TR == 3);
  constexpr int JT = PAR ? (TAPS + 1) / 2 : TAPS;   // tap slots a wave iterates
  constexpr int BM = TC::BM;
  constexpr int NZMAX = 4;
  constexpr int NZ[3] = {NZ0, NZ1, NZ2};
  constexpr int WROWS = TC::WROWS, WRPP = TC::WRPP, WIT = TC::WIT, WSTR = TC::WSTR, SJ = TC::SJ, SSTR = TC::SSTR;
  constexpr int ZCOL = TC::RWMAX;   // first padding column of an S row: written once with 0, never staged
  static_assert(WM * WN == 4, "4 waves per workgroup");
  // two LDS buffers: the stage being multiplied and the stage being written (one barrier per stage)
  // per-output-row parameters (bias in the prologue, the aux affine in the epilogue) live behind buffer 0 -- in
  // buffer 1, which is not written before the end of stage 0 and not read after the last stage -- and behind the
  // epilogue's transpose area (4 waves x 16 x 65 floats at the start of smem)
  constexpr int PAREA = 4 * 16 * 65;
  static_assert(TC::BUF < PAREA || 4 * BM <= TC::WPAD * WSTR, "rowp must fit the W region of buffer 1");
  constexpr int ROWP_OFF = TC::BUF >= PAREA ? TC::BUF : (2 * TC::BUF > PAREA ? 2 * TC::BUF : PAREA);   // small tiles: own space
  constexpr int LDS_FLOATS = 2 * TC::BUF > ROWP_OFF + 4 * BM ? 2 * TC::BUF : ROWP_OFF + 4 * BM;
  __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
  float4* rowp = reinterpret_cast<float4*>(smem + ROWP_OFF);
  const sar_conv_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int V = d.V;
  // Workgroup -> (tile, row block).  Dispatch order is 1-D: consecutive ids go to consecutive XCDs (8, each with
  // its own L2).  The row blocks of one tile are adjacent ids' worth of work on the SAME XCD (they share the staged
  // src rows), and each XCD walks a contiguous range of tiles, so the temporal halo a tile shares with its
  // neighbour is an L2 hit instead of a second HBM fetch.  id = slot*8 + xcd ; work item w = xcd*per + slot.
  const int ny = k.ny, nwork = k.ntiles * ny;
  int w = blockIdx.x;
  if (SAR_XCD_MAP) {
    const int per = (nwork + 7) / 8;
    const int xcd = w & 7, slot = w >> 3;
    w = xcd * per + slot;
    if (w >= nwork || slot >= per) return;   // the padded tail of the id space (whole workgroup: before any barrier)
  }
  const int tile = w / ny;
  const int b = tile / k.TPS;
  const int t0 = (tile - b * k.TPS) * k.FT;
  const int m0 = (w - tile * ny) * BM;
  FP32_TL_BEGIN();

  // ---- per-lane column geometry (fixed for the whole kernel)
  bool colok[NS];
  int64_t coln[NS];  // output column index n
  int64_t colna[NS]; // the aux tensor's column: coln, or (SAR_GRAPH_AUX_EVEN_FRAMES) the even-frame index / -1 on an odd frame
  const bool aux_even = MODE == SAR_CONV_GRAPH && (d.g_flags & SAR_GRAPH_AUX_EVEN_FRAMES) != 0;
  int off[JT][NS];             // TEMPORAL: LDS column offset per tap slot
  unsigned vmask[NS];          // TEMPORAL TR==2: tap validity bits
  int goff[3][NS][NZMAX];      // GRAPH: LDS column offset of each gather entry
  float gw[3][NS][NZMAX];      // GRAPH: weight of each gather entry
  float gcs[3][NS];            // GRAPH: colsum(A_k)[v] for the bias term

  int t_lo;
  if (MODE == SAR_CONV_GRAPH) t_lo = t0;
  else if (!TRANSPOSED) t_lo = t0 * d.stride - d.pad;
  else t_lo = floordiv(t0 + d.pad - (TAPS - 1), d.stride);

  // parity split (TR==3): t0 is even (FT even), so the frame parity of a column is that of its half-tile
  constexpr int HALFC = 16 * NS * WN;
  const int par = PAR ? (wn * NS * 32 >= HALFC ? 1 : 0) : 0;
  const int tp0 = PAR ? ((par + d.pad) & 1) : 0;          // first valid tap of this wave; then every other one
  const int ntap_w = PAR ? (TAPS - tp0 + 1) / 2 : TAPS;   // wave-uniform
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    int fo, v;
    if (PAR) {
      const int pp = p - par * HALFC;
      const int fh = pp / V;
      v = pp - fh * V;
      fo = 2 * fh + par;
      colok[ns] = (fo < k.FT) && (t0 + fo < d.T_out);
    } else {
      fo = p / V;
      v = p - fo * V;
      colok[ns] = (fo < k.FT) && (t0 + fo < d.T_out);
    }
    // a column outside the tile reads the always-zero LDS column ZCOL in every tap: its accumulators stay
    // exactly 0, so the epilogue needs no column predicate for the BatchNorm sums
    if (!colok[ns]) fo = par;
    coln[ns] = ((int64_t)b * d.T_out + (t0 + fo)) * V + v;
    colna[ns] = aux_even ? (((t0 + fo) & 1) ? -1 : ((int64_t)b * ((d.T_out + 1) >> 1) + ((t0 + fo) >> 1)) * V + v) : coln[ns];
    vmask[ns] = 0;
    if (MODE == SAR_CONV_TEMPORAL) {
#pragma unroll
      for (int tp = 0; tp < JT; ++tp) {
        if (!TRANSPOSED) {
          off[tp][ns] = (fo * d.stride + tp) * V + v;
        } else if (PAR) {
          const int to = (t0 + fo + d.pad - (tp0 + 2 * tp)) >> 1;   // exact: the numerator is even
          off[tp][ns] = (to - t_lo) * V + v;
        } else {
          const int q = t0 + fo + d.pad - tp;
          const int to = floordiv(q, d.stride);
          const bool ok = (q - to * d.stride) == 0;
          vmask[ns] |= (ok ? 1u : 0u) << tp;
          off[tp][ns] = (to - t_lo) * V + v;
        }
        if (!colok[ns]) off[tp][ns] = ZCOL;
      }
    } else {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        gcs[tp][ns] = (d.g_colsum && colok[ns]) ? d.g_colsum[tp * V + v] : 0.f;
#pragma unroll
        for (int j = 0; j < NZMAX; ++j) {
          if (j < NZ[tp]) {
            goff[tp][ns][j] = colok[ns] ? fo * V + d.g_idx[(tp * V + v) * NZMAX + j] : ZCOL;
            gw[tp][ns][j] = d.g_wt[(tp * V + v) * NZMAX + j];
          }
        }
      }
    }
  }

  // per-row bias -> LDS (read back after the first barrier as the initial accumulator value); zero columns
  if (tid < BM) {
    const int row = m0 + tid;
    float4 bp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias && row < d.M) {
      bp.x = d.bias[row];
      if (MODE == SAR_CONV_GRAPH) {
        bp.y = d.bias[d.M + row];
        bp.z = d.bias[2 * d.M + row];
      }
    }
    rowp[tid] = bp;
  }
  if (tid < 2 * KC) smem[(tid / KC) * TC::BUF + TC::WPAD * WSTR + (tid % KC) * SSTR + ZCOL] = 0.f;
  f32x16 acc[MS][NS];

  const int seq_len = d.T_src * V;
  const float* src_b = d.src + (int64_t)b * seq_len;

  // ---- staging.  Everything that does not change from stage to stage is computed here once: per-lane byte
  // offsets, validity masks, LDS addresses.  The loads are buffer loads (wave-uniform base in SGPRs, advanced by
  // scalar arithmetic), so that a stage costs the vector ALU -- which the fp32 MFMA shares -- 3 instructions
  // per staged src element (BN scale/shift, ReLU, zero-pad select) and none per weight.
  //   S: wave w stages src channel c0 + w; lane -> columns lane + 64 j
  //   W: one float4 per lane and pass, WRPP rows per pass
  int svo[SJ];
  bool sok[SJ];
#pragma unroll
  for (int j = 0; j < SJ; ++j) {
    const int col = lane + 64 * j;
    const int rabs = t_lo * V + col;
    sok[j] = col < k.RW && (unsigned)rabs < (unsigned)seq_len;   // else: temporal zero padding / lane padding
    svo[j] = sok[j] ? rabs * 4 : 0;
  }
  const bool w_vec = k.w_vec != 0;
  const int w_m4 = (tid % (BM / 4)) * 4;     // float4 column of this lane inside a W row
  const int w_r0 = tid / (BM / 4);           // first W row of this lane
  int wvo[WIT];
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const int row = w_r0 + i * WRPP;
    const int tp = row / KC, c = row % KC;
    const bool ok = row < WROWS && (m0 + w_m4) < d.M;   // rows >= M only feed output rows that are never stored
    wvo[i] = ok ? (int)(((int64_t)tp * d.w_stride_tap + (int64_t)c * d.w_stride_c + m0 + w_m4) * 4) : 0;
  }
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  float4 wreg[WIT];
  float sreg[SJ];
  float psc = 1.f, psh = 0.f;

  // nothing consumes the loaded registers until store_lds(): the loads of stage i+1 are in flight during
  // the MFMA phase of stage i
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
        __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, seq_len * 4, 0x00020000);
#pragma unroll
    for (int j = 0; j < SJ; ++j) sreg[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo[j], 0, 0));
    if (d.pro_scale) {
      psc = d.pro_scale[cg];
      psh = d.pro_shift[cg];
    }
  };

  auto store_lds = [&](int c0, float* buf) {
    float* Wl = buf;
    float* S = buf + TC::WPAD * WSTR;
    if (w_vec) {
#pragma unroll
      for (int i = 0; i < WIT; ++i) {
        const int row = w_r0 + i * WRPP;
        if ((i + 1) * WRPP <= TC::WPAD || row < TC::WPAD) *reinterpret_cast<float4*>(Wl + row * WSTR + w_m4) = wreg[i];
      }
    } else {  // unaligned / M % 4 != 0 weights (3-channel layers only): plain strided copy
      for (int idx = tid; idx < WROWS * BM; idx += 256) {
        const int m = idx % BM, row = idx / BM;
        const int tp = row / KC, cg = c0 + row % KC, mg = m0 + m;
        Wl[row * WSTR + m] = (cg < d.Kc && mg < d.M) ? d.W[(int64_t)tp * d.w_stride_tap + (int64_t)cg * d.w_stride_c + mg] : 0.f;
      }
    }
    const bool rowok = (c0 + wave) < d.Kc;   // wave-uniform
#pragma unroll
    for (int j = 0; j < SJ; ++j) {
      // folded BN(+ReLU); everything outside the sequence (temporal zero padding) or beyond Kc is exactly 0
      const float val = fmaxf(fmaf(sreg[j], psc, psh), relu_lo);
      S[wave * SSTR + lane + 64 * j] = (sok[j] && rowok) ? val : 0.f;
    }
  };

#ifndef SAR_ABLATE
#define SAR_ABLATE 0   // diagnostic builds only (tools/ablate.sh): 1 no epilogue, 2 stage once, 4 no barriers, 16 / 32 see below
#endif
  FP32_TL(0);   // geometry, gather tables, stager set-up
  issue_loads(0);
  store_lds(0, smem);
  __syncthreads();
  FP32_TL(1);   // first stage: loads -> LDS (exposed round trip) + barrier
  // accumulators start at the bias term: b[m] (temporal) or sum_k b_k[m] colsum(A_k)[v] (graph), 0 off-tile
  SAR_LDS_SKEW();   // last read of the bias rows (overlaid on buffer 1's weight region)
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 bp = rowp[(wm * MS + ms) * 32 + mfma_row(r, hi)];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        if (MODE == SAR_CONV_TEMPORAL) acc[ms][ns][r] = colok[ns] ? bp.x : 0.f;
        else acc[ms][ns][r] = fmaf(bp.z, gcs[2][ns], fmaf(bp.y, gcs[1][ns], bp.x * gcs[0][ns]));
      }
    }
  // The bias rows live in buffer 1, which the FIRST wave to finish stage 0 overwrites: every wave must have read them before
  // any wave gets there.  (Without this barrier a wave that the SIMD arbitration held back for a whole MFMA phase initialised
  // its accumulators from the next stage's weights: one wrong tile in ~1 of 1 000 launches, found by tools/trace_divergence.py
  // in round 3 -- the fp32 step was not repeatable in 1-4 % of its runs, in round 2 as well.)
  // HAPPENS-BEFORE of the LDS regions of this kernel:
  //  * bias rows (rowp, in buffer 1's W region when the tile is large): written before the barrier above, last read by the
  //    accumulator initialisation, first overwritten by store_lds(KC, buffer 1) at the end of stage 0 -- ordered by THIS barrier;
  //  * buffer it ^ 1 is rewritten at the end of stage s by a wave that has passed the closing barrier of stage s - 1, which every
  //    wave joins after its MFMA phase of stage s - 1 = its last read of that buffer;
  //  * rowp is rewritten with the MASK parameters in the epilogue by waves that passed the closing barrier of the LAST stage
  //    (nobody reads buffer 1 / rowp between that barrier and the barrier behind the rewrite);
  //  * the epilogue's transpose area (start of smem) is wave-private, written after the same closing barrier.
#ifndef SAR_DEBUG_LDS_DROP_BIAS_BARRIER
  __syncthreads();
#endif
  FP32_TL(2);   // accumulator initialisation + barrier
  // one main-loop stage on LDS buffer IT (compile-time, so every LDS address is register + immediate)
  auto stage = [&](int c0, auto IT) {
    constexpr int it = decltype(IT)::value;
    const bool more = c0 + KC < d.Kc && !(SAR_ABLATE & 2);
    if (more && !(SAR_ABLATE & 16)) issue_loads(c0 + KC);   // 16: LDS stores of stale registers only
    FP32_TL(3);   // load issue
    SAR_LDS_SKEW();   // this wave's reads of buffer IT start late: the other waves may only refill IT ^ 1 meanwhile
    const float* Wl = smem + it * TC::BUF;
    const float* S = Wl + TC::WPAD * WSTR;

    // ---- MFMA phase.  The fp32 MFMA runs on the vector ALU of the SIMD (no co-issue with VALU work; measured:
    // tools/mfma_fill.hip), so the loop carries LDS reads and MFMAs only, and the operands of step s+1 are read
    // while the MFMAs of step s issue (register double buffer) so that no MFMA waits on LDS latency.
    constexpr int HS = KC / 2;                       // k-steps per tap
    constexpr int RZ = (MODE == SAR_CONV_GRAPH) ? NZMAX : 1;
    const float* Sh = S + hi * SSTR;
    // one base register per 32-row block, made opaque to the optimiser: left alone it fuses the MS reads of a
    // step into ds_read2_b32, whose 8-bit offsets then need one v_add per step for a new base -- vector-ALU
    // work that the MFMA pipe pays for; a plain ds_read_b32 carries the whole offset as a 16-bit immediate.
    typedef const float __attribute__((address_space(3))) * lds_cptr;
    lds_cptr Wa[MS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
      unsigned a = (unsigned)(uintptr_t)(Wl + (tp0 * KC + hi) * WSTR + (wm * MS + ms) * 32 + l31);   // LDS byte address
      asm volatile("" : "+v"(a));
      Wa[ms] = (lds_cptr)(uintptr_t)a;
    }
    auto fetch = [&](int st, float (&a)[MS], float (&r)[NS][RZ]) {
      const int j = st / HS, cc = (st % HS) * 2;
      const int tpw = PAR ? 2 * j : j;
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a[ms] = Wa[ms][(tpw * KC + cc) * WSTR];
      const float* Srow = Sh + cc * SSTR;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        if (MODE == SAR_CONV_TEMPORAL) {
          r[ns][0] = Srow[off[j][ns]];
        } else {
#pragma unroll
          for (int q = 0; q < NZMAX; ++q)
            if (q < NZ[j]) r[ns][q] = Srow[goff[j][ns][q]];
        }
      }
    };
    auto mma = [&](int st, const float (&a)[MS], const float (&r)[NS][RZ]) {
      const int j = st / HS;
      float bv[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        if (MODE == SAR_CONV_TEMPORAL) {
          bv[ns] = (TR == 2) ? (((vmask[ns] >> j) & 1u) ? r[ns][0] : 0.f) : r[ns][0];
        } else {
          float x = gw[j][ns][0] * r[ns][0];
#pragma unroll
          for (int q = 1; q < NZMAX; ++q)
            if (q < NZ[j]) x = fmaf(gw[j][ns][q], r[ns][q], x);
          bv[ns] = x;
        }
      }
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms], bv[ns], acc[ms][ns], 0, 0, 0);
    };
    constexpr int NSTEP_ALL = JT * HS;
    constexpr int NSTEP_SURE = PAR ? (JT - 1) * HS : NSTEP_ALL;   // the last parity slot exists for tp0 == 0 only
    // issue order inside one step: the MFMAs of step st alternate with the LDS reads of step st+1 (the
    // compiler's own schedule sinks every read to just before its use and waits for it there)
    auto order = [&](int st_next, bool have_next) {
      constexpr int NM = MS * NS;
      const int jn = st_next / HS;
      const int rd = have_next ? MS + NS * (MODE == SAR_CONV_GRAPH ? NZ[jn < 3 ? jn : 0] : 1) : 0;
      int done = 0;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        const int n = (rd - done + (NM - i) - 1) / (NM - i);
        if (n == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if (n == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        else if (n == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        else if (n >= 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        done += n;
      }
    };
    {
      float a0[MS], r0[NS][RZ], a1[MS], r1[NS][RZ];
      fetch(0, a0, r0);
#pragma unroll
      for (int st = 0; st < NSTEP_SURE; st += 2) {
        if (st + 1 < NSTEP_SURE) fetch(st + 1, a1, r1);
        mma(st, a0, r0);
        order(st + 1, st + 1 < NSTEP_SURE);
        if (st + 1 < NSTEP_SURE) {
          if (st + 2 < NSTEP_SURE) fetch(st + 2, a0, r0);
          mma(st + 1, a1, r1);
          order(st + 2, st + 2 < NSTEP_SURE);
        }
      }
      if (PAR && ntap_w == JT) {   // wave-uniform
        fetch(NSTEP_SURE, a0, r0);
#pragma unroll
        for (int st = NSTEP_SURE; st < NSTEP_ALL; st += 2) {
          if (st + 1 < NSTEP_ALL) fetch(st + 1, a1, r1);
          mma(st, a0, r0);
          order(st + 1, st + 1 < NSTEP_ALL);
          if (st + 1 < NSTEP_ALL) {
            if (st + 2 < NSTEP_ALL) fetch(st + 2, a0, r0);
            mma(st + 1, a1, r1);
            order(st + 2, st + 2 < NSTEP_ALL);
          }
        }
      }
    }
    FP32_TL(4);   // MFMA phase (operand reads + MFMAs)
    if (more && !(SAR_ABLATE & 32)) store_lds(c0 + KC, smem + (it ^ 1) * TC::BUF);
    FP32_TL(5);   // wait for the next stage's loads + LDS stores
    if (more && (SAR_ABLATE & 32)) {   // 32: global loads only (wait for them, keep them alive, no LDS store)
#pragma unroll
      for (int j = 0; j < SJ; ++j) asm volatile("" ::"v"(sreg[j]));
#pragma unroll
      for (int i = 0; i < WIT; ++i) asm volatile("" ::"v"(wreg[i].x), "v"(wreg[i].w));
    }
    if (!(SAR_ABLATE & 4)) __syncthreads();
    FP32_TL(6);   // stage barrier
  };
  for (int c0 = 0; c0 < d.Kc; c0 += 2 * KC) {
    stage(c0, std::integral_constant<int, 0>());
    if (c0 + KC < d.Kc) stage(c0 + KC, std::integral_constant<int, 1>());
  }
  if (SAR_ABLATE & 1) {
    float t = 0.f;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[ms][ns][r];
    if (t == 123.456f) d.out[tid] = t;
    return;
  }

  // ---- epilogue: mask / add, store, BatchNorm partial sums.
  // (the bias is already in the accumulators; off-tile columns hold exact zeros)
  const int part = tile * WN + wn;
  const bool stats = d.epi == SAR_EPI_STATS || d.epi == SAR_EPI_MASK;
  auto fast_epilogue = [&](auto EPI_) {
    constexpr int EPI = decltype(EPI_)::value;
    constexpr bool gate = EPI == SAR_EPI_ADD_GATE;   // out = gate(acc + aux), sums of the gated values (include/sar_hip.h)
    constexpr bool stats = EPI == SAR_EPI_STATS || EPI == SAR_EPI_MASK || gate;
    // Fast path.  A lane's 16 accumulator rows of one 32-row block are rows (r&3) + 8 (r>>2) + 4 hi: with
    // M % 8 == 0 validity is uniform per group of 4 registers (scalar branch), the row part of every address is
    // a scalar offset and the column part one per-lane byte offset: buffer loads / stores, no address VALU.
    constexpr bool has_aux = EPI == SAR_EPI_MASK || EPI == SAR_EPI_ADD || gate;
    if (EPI == SAR_EPI_MASK || gate) {   // per-row affine of the mask / centre of the second sum -> LDS (all waves are past the main loop)
      if (tid < BM) {
        const int row = m0 + tid;
        float4 ap = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < d.M) {
          if (EPI == SAR_EPI_MASK) {
            ap.x = d.aux_scale[row];
            ap.y = d.aux_shift[row];
          }
          if (d.aux_mean) ap.z = d.aux_mean[row];
        }
        rowp[tid] = ap;
      }
      __syncthreads();
    }
    const int rows_w = m0 + wm * MS * 32;   // first row of this wave
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
    for (int ns = 0; ns < NS; ++ns) {   // off-tile columns: an offset the range check rejects
      vo_out[ns] = colok[ns] ? (unsigned)((coln[ns] + 4 * hi * d.ld_out) * 4) : 0x80000000u;
      vo_aux[ns] = (colok[ns] && colna[ns] >= 0) ? (unsigned)((colna[ns] + 4 * hi * d.ld_aux) * 4) : 0x80000000u;   // (rejected: 0)
    }
    const int so_out = (int)(d.ld_out * 4), so_aux = (int)(d.ld_aux * 4);   // bytes per row
    // SAR_EPI_ADD_GATE: aux2 [M][ld_aux2] fp32 and its gate bytes [M][ld_aux2 / 4] (bit j of byte i = column 4 i + j, the layout
    // sar_bn_add_relu_fwd_mask_f32 writes)
    const float* u2 = reinterpret_cast<const float*>(d.aux2);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(gate ? u2 + (int64_t)rows_w * d.ld_aux2 : d.out), 0, gate ? rows_bytes(d.ld_aux2) : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(gate ? d.aux_mask + (int64_t)rows_w * (d.ld_aux2 >> 2) : (const unsigned char*)d.out), 0,
        gate ? rows_bytes(d.ld_aux2) >> 4 : 0u, 0x00020000);
    unsigned vo_u[NS], vo_m[NS], cbit[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      vo_u[ns] = colok[ns] ? (unsigned)((coln[ns] + 4 * hi * d.ld_aux2) * 4) : 0x80000000u;
      vo_m[ns] = colok[ns] ? (unsigned)((coln[ns] >> 2) + hi * d.ld_aux2) : 0x80000000u;
      cbit[ns] = (unsigned)(coln[ns] & 3);
    }
    const int so_u = (int)(d.ld_aux2 * 4), so_m = (int)(d.ld_aux2 >> 2);
    float* P = smem + wave * (16 * 65);   // wave-private transpose area for the sums (16 sums x 64 lanes, stride 65)
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {   // 8 registers = 16 sums per transpose round
        float ax[NS][16], ux[NS][16];
        unsigned gm[NS][16];
        if (has_aux) {   // the aux loads of this half row block are issued before the first use
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
              const int r = rb * 8 + r8;
              ax[ns][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                  ra, vo_aux[ns], (ms * 32 + (r & 3) + 8 * (r >> 2)) * so_aux, 0));
              if (gate) {
                ux[ns][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    ru, vo_u[ns], (ms * 32 + (r & 3) + 8 * (r >> 2)) * so_u, 0));
                gm[ns][r] = (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(
                    rm, vo_m[ns], (ms * 32 + (r & 3) + 8 * (r >> 2)) * so_m, 0);
              }
            }
        }
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = rb * 8 + r8;
          const bool grp_ok = rows_w + ms * 32 + 8 * (r >> 2) < d.M;   // wave-uniform
          float s1 = 0.f, s2 = 0.f;
          float4 ap = make_float4(0.f, 0.f, 0.f, 0.f);
          if (EPI == SAR_EPI_MASK || gate) ap = rowp[(wm * MS + ms) * 32 + mfma_row(r, hi)];
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
            } else if (gate) {   // replaces, for the block below, bn_add_relu_bwd_reduce and the masked-gradient write of the apply pass
              val += ax[ns][r];
              val = ((gm[ns][r] >> cbit[ns]) & 1u) ? val : 0.f;
              s1 += val;
              s2 = fmaf(val, ux[ns][r] - ap.z, s2);
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
          // sum q of half-wave h = sum over 32 lanes; lane L adds 16 of them: q = L & 15, lanes 16 (L>>4 & 1) ..,
          // half h = L >> 5; bank = (q + 16 (L>>4&1) + i) % 32 is distinct over each group of 32 lanes
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
      case SAR_EPI_ADD_GATE: fast_epilogue(std::integral_constant<int, SAR_EPI_ADD_GATE>()); break;
      default: fast_epilogue(std::integral_constant<int, SAR_EPI_NONE>()); break;
    }
    FP32_TL(7);   // epilogue
    FP32_TL_END();
    return;
  }
  // Generic path (M % 8 != 0: the 3-channel input layer's data gradient)
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
          if (d.epi == SAR_EPI_STATS) {
            s1 += val;
            s2 = fmaf(val, val, s2);
          } else if (d.epi == SAR_EPI_MASK) {
            const float ax = colna[ns] >= 0 ? d.aux[(int64_t)row * d.ld_aux + colna[ns]] : 0.f;
            val = (fmaf(ax, asc, ash) > 0.f) ? val : 0.f;
            s1 += val;
            s2 = fmaf(val, ax - amu, s2);
          } else if (d.epi == SAR_EPI_ADD) {
            val += colna[ns] >= 0 ? d.aux[(int64_t)row * d.ld_aux + colna[ns]] : 0.f;
          }
          d.out[(int64_t)row * d.ld_out + coln[ns]] = val;
        }
      }
      if (stats) {
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
int tile_geometry(const sar_conv_desc& d, int NSv, bool parity, ConvK& k) {
  const int tile_n = 32 * NSv * WN;
  if (parity) {   // frames per parity half, FT even so that every tile starts on an even frame
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
  // (the vector path stages whole KC-channel slabs: a Kc tail takes the scalar path, which zero-fills it)
  k.w_vec = ((d.Kc % KC) == 0 && (d.M & 3) == 0 && (d.w_stride_c & 3) == 0 && (d.w_stride_tap & 3) == 0 && ((uintptr_t)d.W & 15) == 0) ? 1 : 0;
  const int rwmax = (d.mode == SAR_CONV_GRAPH) ? 32 * NSv * WN : (NSv * WN == 4 ? 448 : 704);
  if (k.RW > rwmax) return -2;   // staged row does not fit the register prefetch (V too large)
  return 0;
}

template <int MODE, int TR, int TAPS, int MS, int NS, int WM, int WN, int NZ0, int NZ1, int NZ2>
int launch_cfg(const sar_conv_desc& d, hipStream_t st, bool query_only, int* nparts_out) {
  ConvK k;
  k.d = d;
  if (int g = tile_geometry<WN>(d, NS, TR == 3, k)) {
    sar_set_error("sar_conv_gemm: unsupported tile geometry (V=%d, stride=%d)", d.V, d.stride);
    return g == -2 ? SAR_E_UNSUP : SAR_E_ARG;
  }
  if (nparts_out) *nparts_out = k.nparts;
  if (query_only) return 0;
  constexpr int BM = 32 * MS * WM;
  k.ntiles = d.B * k.TPS;
  k.ny = (d.M + BM - 1) / BM;
  const int nwork = k.ntiles * k.ny;
  dim3 grid(SAR_XCD_MAP ? ((nwork + 7) / 8) * 8 : nwork);
  hipLaunchKernelGGL((conv_gemm_kernel<MODE, TR, TAPS, MS, NS, WM, WN, NZ0, NZ1, NZ2>), grid, dim3(256), 0, st, k);
  return 0;
}

// tile by output rows: M > 64 -> 128 x 128 (2x2 waves of 64x64); 32 < M <= 64 -> 64 x 256; M <= 32 -> 32 x 256.
// The parity-split variant needs the 4-waves-along-N layouts (a wave's 64 columns lie in one parity half).
template <int MODE, int TR, int TAPS, int NZ0, int NZ1, int NZ2>
int launch_by_m(const sar_conv_desc& d, hipStream_t st, bool query_only, int* nparts_out) {
  if constexpr (TR != 3)
    if (d.M > 64) return launch_cfg<MODE, TR, TAPS, 2, 2, 2, 2, NZ0, NZ1, NZ2>(d, st, query_only, nparts_out);
  if (d.M > 32) return launch_cfg<MODE, TR, TAPS, 2, 2, 1, 4, NZ0, NZ1, NZ2>(d, st, query_only, nparts_out);
  return launch_cfg<MODE, TR, TAPS, 1, 2, 1, 4, NZ0, NZ1, NZ2>(d, st, query_only, nparts_out);
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
  // the stagers address one sequence row / the weight tensor with 32-bit byte offsets
  SAR_REQUIRE((int64_t)d->T_src * d->V < (1 << 28), "sar_conv_gemm: sequence row too long");
  SAR_REQUIRE(d->ld_out < (1 << 22) && d->ld_aux < (1 << 22), "sar_conv_gemm: leading dimension too large (2^22 columns)");
  SAR_REQUIRE(((int64_t)(d->taps - 1) * d->w_stride_tap + (int64_t)4 * d->w_stride_c + d->M) < (1 << 28),
              "sar_conv_gemm: weight tensor too large for 32-bit offsets");
  SAR_REQUIRE(d->epi >= SAR_EPI_NONE && d->epi <= SAR_EPI_ADD_GATE, "sar_conv_gemm: bad epilogue %d", d->epi);
  if (d->epi == SAR_EPI_STATS || d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD_GATE)
    SAR_REQUIRE(d->partials, "sar_conv_gemm: partials required");
  if (d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD || d->epi == SAR_EPI_ADD_GATE)
    SAR_REQUIRE(d->aux && d->ld_aux >= (int64_t)d->B * (aux_even_frames(*d) ? (d->T_out + 1) / 2 : d->T_out) * d->V, "sar_conv_gemm: aux required");
  SAR_REQUIRE(!aux_even_frames(*d) || (d->mode == SAR_CONV_GRAPH && (d->epi == SAR_EPI_ADD || d->epi == SAR_EPI_ADD_GATE)),
              "sar_conv_gemm: SAR_GRAPH_AUX_EVEN_FRAMES goes with the GRAPH operator and the ADD / ADD_GATE epilogues");
  if (d->epi == SAR_EPI_ADD_GATE) {   // fp32: aux2 is [M][ld_aux2] floats, aux_mask [M][ld_aux2 / 4] bytes (one bit per column)
    SAR_REQUIRE(d->mode == SAR_CONV_GRAPH && (d->M & 7) == 0, "sar_conv_gemm: SAR_EPI_ADD_GATE is built for the graph data gradient with M %% 8 == 0");
    SAR_REQUIRE(d->aux2 && d->aux_mask && d->ld_aux2 >= (int64_t)d->B * d->T_out * d->V && (d->ld_aux2 & 3) == 0 && d->ld_aux2 < (1 << 22),
                "sar_conv_gemm: SAR_EPI_ADD_GATE needs aux2, aux_mask and ld_aux2 %% 4 == 0");
  }
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
  if (d.stride == 1) {   // every tap of every column reaches a real source frame: no validity mask
    if (d.taps == 9) return launch_by_m<SAR_CONV_TEMPORAL, 1, 9, 1, 1, 1>(d, st, query_only, nparts_out);
    return launch_by_m<SAR_CONV_TEMPORAL, 1, 1, 1, 1, 1>(d, st, query_only, nparts_out);
  }
  if (d.taps == 9) {
    if (d.stride == 2) return launch_by_m<SAR_CONV_TEMPORAL, 3, 9, 1, 1, 1>(d, st, query_only, nparts_out);
    return launch_by_m<SAR_CONV_TEMPORAL, 2, 9, 1, 1, 1>(d, st, query_only, nparts_out);
  }
  return launch_by_m<SAR_CONV_TEMPORAL, 2, 1, 1, 1, 1>(d, st, query_only, nparts_out);
}

}  // namespace

#ifdef SAR_DEBUG
#include "../../include/sar_hip_debug.h"
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
#endif

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

#ifdef SAR_FP32_TL
extern "C" int sar_debug_fp32_timeline(unsigned* out, int nwg, int reset) {   // out: [nwg][16] (host memory)
  if (nwg > FP32_TL_WG) nwg = FP32_TL_WG;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fp32_tl), (size_t)nwg * 16 * sizeof(unsigned)) != hipSuccess) return -1;
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fp32_tl)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned) * 16 * FP32_TL_WG) != hipSuccess) return -1;
  }
  return nwg;
}
#endif

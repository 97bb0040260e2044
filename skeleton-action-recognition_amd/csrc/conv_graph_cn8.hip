// conv_graph_cn8.hip -- GraphConvTD (models/gcn.py:199-209) and its data gradient on bf16 CN8 activations with the
// adjacency applied WHEN THE MATRIX OPERAND IS READ (round 4; replaces the unit builder of conv_graph_cn8_kernel in
// conv_gemm_cn8.hip for adjacencies with few non-trivial gather lists -- SAR_GRAPH_FEW_DENSE, the NTU graph).
//
//   out[m, (t,w)] = sum_k sum_c bf16(W_k[c][m]) * bf16(z_k)[c, (t,w)] + sum_k b_k[m] colsum(A_k)[w]
//   z_k[c, (t,w)] = sum_j wt_kj(w) * x[c, (t, idx_kj(w))]                    (<= 4 entries per column of A_k)
//
// conv_graph_cn8_kernel BUILDS the three z_k tiles in LDS every stage (gather, weight in fp32, round, store, barrier): 1 750
// of the 3 360 cycles of a 16-channel stage (profiles/r03_bf16_phase_stamps.txt).  But of the 75 (slice, joint) gather lists
// of the NTU graph 25 are the identity, 44 are ONE entry of weight 1 (z_k[., w] IS the raw column of another joint), 4 are
// empty and only 2 (fwd) / 8 (transposed) need arithmetic.  So:
//  * a B fragment of slice k for column (t, w) is read STRAIGHT from the raw tile at the per-lane address of joint idx_k(w)
//    (a ds_read_b128 with a per-lane address costs what the un-gathered read costs), an empty list reads an always-zero unit;
//  * the few lists that need arithmetic become VIRTUAL joints V .. V + NV - 1 appended to every frame of the raw tile, built
//    by a mini-builder (<= 1 unit per thread and stage: the same fp32 fma chain in table order, rounded once -- the values the
//    unit builder produces, so the two kernels agree BIT FOR BIT);
//  * no z image: the LDS it frees double-buffers the raw tile and the weights, which removes the stage's closing barrier.
// Stage = 16 source channels: W units [slice][plane][BM] + raw tile [plane][frame][V + NV] -> LDS buffer s & 1 (register
// staging: the loads of stage s + 1 are in flight during the MFMA phase of stage s), barrier, mini-builder, barrier,
// 3 x MS x NS MFMAs per wave; epilogue = conv_cn8_common.h (STATS / ADD / NONE).
#include "conv_cn8_common.h"

namespace {

constexpr bool SAR_GRAPH2_WG4_DEFAULT = true;
constexpr int NVMAX = 16;   // virtual joints per frame the tables may ask for

#ifndef SAR_G2_PF2
#define SAR_G2_PF2 0      // operand prefetch distance 2 (two staging register sets); 0 = one stage ahead (A/B builds)
#endif
// SAR_G2_WG4 (template parameter; SAR_GRAPH2_WG4=0|1 at run time): four workgroups per CU (<= 128 VGPRs: one fragment set, aux
// half units loaded inside the epilogue) or three (148-152 VGPRs: two fragment sets, aux half units requested before the last
// MFMA phase).  Alone the four-workgroup kernel is 4 % faster; beside the weight-gradient stream the three-workgroup one leaves
// room for the other stream's workgroups (profiles/r04_bf16_instep_ab.txt).
#ifndef SAR_G2_WG4_PREAUX
#define SAR_G2_WG4_PREAUX 0   // 1: the ADD epilogue's aux half units requested before the last MFMA phase (5 spilled registers at 128: measured 1-3 % slower)
#endif
#ifndef SAR_G2_PRE_HALF
#define SAR_G2_PRE_HALF 0   // 1: half block 0 of the epilogue requested before the last MFMA phase -- measured: no effect (0.94 vs 0.95 ms per step), the epilogue is VALU work
#endif
#ifndef SAR_G2_BIAS_CHUNK
#define SAR_G2_BIAS_CHUNK 4
#endif
#ifndef SAR_G2_ABLATE
#define SAR_G2_ABLATE 0   // diagnostic builds only (tools/ablate_g2.sh): 1 no MFMA, 2 global loads of stage 0 only, 4 no epilogue, 8 no mini-builder, 16 LDS stores of stage 0 only, 32 no classification of the gather lists
#endif

// Diagnostic build -DSAR_G2_TIMELINE (tools/g2_timeline.sh): wave 0 of every workgroup writes one row -- start / end in 100 MHz
// ticks (s_memrealtime), HW_ID, XCC_ID, shader-clock cycles of the prologue (up to its barrier / the wait there / the geometry + accumulator
// initialisation behind it) / the stages / the epilogue -- so that the
// launch's timeline (workgroups resident per CU, lifetimes, dispatch rate) can be read back.
#ifdef SAR_G2_TIMELINE
constexpr int G2_TL_WG = 16384;
__device__ unsigned g_g2_tl[G2_TL_WG][12];
#define G2_TL(i)                                                  \
  do {                                                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
    tl_acc[i] = (unsigned)(t_ - tl_last);                         \
    tl_last = t_;                                                 \
  } while (0)
#else
#define G2_TL(i)
#endif

template <int MS, int NS, int WM, int WN, int SAR_G2_WG4>
__global__ __launch_bounds__(256, SAR_G2_WG4 ? 4 : 3) void conv_graph2_cn8_kernel(const ConvK8 k, const int nv_asserted) {
  constexpr int BM = 32 * MS * WM, TN = 32 * NS * WN;
  constexpr int PL = 2;                      // CN8 planes per stage (16 source channels = one MFMA k-step)
  constexpr int XS = TN + TN / 2 + 8;        // plane stride (units) >= FT * (V + NV) + 1; the LAST unit of a plane is never written: zero
  constexpr int ZUNIT = XS - 1;
  constexpr int WUNITS = 3 * PL * BM;        // [slice][plane][m]
  constexpr int RUNITS = PL * XS;            // [plane][frame][V + NV]
  constexpr int BUF = WUNITS + RUNITS;
  constexpr int WIT = (WUNITS + 255) / 256;
  constexpr int XJ = (PL * TN) / 256;        // raw units per thread and stage
  constexpr int PAREA_U = 4 * 16 * 65 / 4;   // the epilogue's transpose area aliases the image
  constexpr int IMG_U = 2 * BUF > PAREA_U ? 2 * BUF : PAREA_U;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ uint4 smem_u[IMG_U + BM];
  __shared__ int vmap[3 * 64];               // (slice, joint) -> raw joint | V + virtual id | -1 (empty list)
  __shared__ int vl_idx[NVMAX][4];           // gather entries of the virtual joints
  __shared__ float vl_wt[NVMAX][4];
  __shared__ int nv_s;
  float* smem = reinterpret_cast<float*>(smem_u);
  float4* rowp = reinterpret_cast<float4*>(smem_u + IMG_U);
  const sar_conv_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int V = d.V;
  const int ny = k.ny;
  const int w = xcd_work(k.ntiles * ny);
  if (w < 0) return;
#ifdef SAR_G2_TIMELINE
  const unsigned long long tl_rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long tl_last = __builtin_amdgcn_s_memtime();
  unsigned tl_acc[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#endif
  const int tile = w / ny;
  const int b = tile / k.TPS;
  const int t0 = (tile - b * k.TPS) * k.FT;
  const int m0 = (w - tile * ny) * BM;
  const int nfr = (t0 + k.FT <= d.T_out) ? k.FT : d.T_out - t0;   // live frames of this tile
  const int ncols = nfr * V;

  // ---- prologue: everything that comes from global memory is ISSUED first (one round trip)
  f32x16 acc[MS][NS];
  const int seq_left = (d.T_src - t0) * V;   // columns from the tile start to the end of the sequence
  const char* src_b = (const char*)d.src + ((int64_t)b * d.T_src + t0) * V * 16;
  int xvo[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int u = tid + 256 * j;
    const int xc = u % TN;
    xvo[j] = xc < ncols ? xc * 16 : 0x7fffffff;   // rejected by the range check -> 0
  }
  unsigned wvo[WIT];
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const int u = tid + 256 * i;
    const int m = u % BM, p = (u / BM) % PL, tp = u / (BM * PL);
    const bool ok = u < WUNITS && (m0 + m) < d.M;
    wvo[i] = ok ? (unsigned)((((int64_t)tp * k.G + p) * d.M + m0 + m) * 16) : 0x80000000u;
  }
  const unsigned wbytes = (unsigned)((int64_t)3 * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  // TWO staging register sets: the operands of stage s + 2 are requested while stage s is multiplied.  One stage ahead the
  // workgroup ran at one memory round trip per 16-channel stage (~3 900 cycles for 384 cycles of matrix work per wave,
  // phases additive: profiles/r04_bf16_graph_ablation.txt) -- with 12 MFMAs per stage the round trip is the stage.
  uint4 wregA[WIT], wregB[WIT];
  uint4 xregA[XJ], xregB[XJ];
  auto issue_loads = [&](int c0, uint4 (&wreg)[WIT], uint4 (&xreg)[XJ]) {
    const int wso = (c0 / 8) * d.M * 16;   // planes beyond G lie past the image: range check -> 0
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo[i], wso, 0);
      wreg[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int g = c0 / 8 + (tid + 256 * j) / TN;   // wave-uniform (TN is a multiple of 64)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(src_b + (int64_t)(g < k.Gs ? g : 0) * d.ld_src * 16), 0, g < k.Gs ? (unsigned)seq_left * 16u : 0u, 0x00020000);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, xvo[j], 0, 0);
      xreg[j] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  };
  issue_loads(0, wregA, xregA);
  if (SAR_G2_PF2 && KC16 < d.Kc) issue_loads(KC16, wregB, xregB);

  // Every table the prologue needs is REQUESTED before the first of them is used (vmcnt retires in order, so the first use waits
  // for the stage-0 operands anyway: one round trip for everything; with the classification below reading its lists pass by
  // pass, then the column sums, then the bias, wave 0 spent four dependent round trips -- 8 500-10 400 cycles -- before the
  // workgroup's first barrier: tools/g2_timeline.sh)
  float gcs[3][NS];   // column sums of A_k at this lane's joints (bias term; masked by colok at the use)
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    const int v = (p < ncols ? p : 0) % V;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) gcs[tp][ns] = d.g_colsum ? d.g_colsum[tp * V + v] : 0.f;
  }
  float4 bias_row = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < BM && d.bias && m0 + tid < d.M) {
    bias_row.x = d.bias[m0 + tid];
    bias_row.y = d.bias[d.M + m0 + tid];
    bias_row.z = d.bias[2 * d.M + m0 + tid];
  }
  // classify the 3 V gather lists (wave 0; list p = lane + 64 pass, V <= 64: at most three passes): a list with ONE entry of
  // weight 1 is the raw column of that joint, an empty list the zero unit, everything else a virtual joint (ranked in (slice,
  // joint) order)
  if (SAR_G2_ABLATE & 32) {   // diagnostic: no classification (every list reads its own joint, no virtual joints: WRONG results)
    if (tid < 3 * V) vmap[tid] = tid % V;
    if (tid == 0) nv_s = 0;
  } else if (wave == 0) {
    constexpr int NPASS = 3;
    int ei[NPASS][4];
    float ew[NPASS][4];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int p = ps * 64 + lane;
      const int pc = p < 3 * V ? p : 0;   // clamped: the loads are unconditional, the pass is masked below
#pragma unroll
      for (int j = 0; j < 4; ++j) ei[ps][j] = d.g_idx[pc * 4 + j], ew[ps][j] = d.g_wt[pc * 4 + j];
    }
    int base = 0;
    const int nz0 = d.nz[0], nz1 = d.nz[1], nz2 = d.nz[2];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int p = ps * 64 + lane;
      const bool in = p < 3 * V;
      const int tp = in ? p / V : 0;
      const int nzl = tp == 0 ? nz0 : (tp == 1 ? nz1 : nz2);   // (d.nz[tp] with a per-lane tp is a LOAD from the kernel arguments: one more round trip per pass)
      int cnt = 0, first = 0;
      float wfirst = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool live = in && j < nzl;
        if (!live) ei[ps][j] = 0, ew[ps][j] = 0.f;
        if (live && ew[ps][j] != 0.f) {
          if (cnt == 0) first = ei[ps][j], wfirst = ew[ps][j];
          ++cnt;
        }
      }
      const bool virt = in && (cnt > 1 || (cnt == 1 && wfirst != 1.0f));
      const unsigned long long bal = __ballot(virt);
      const int rank = base + __popcll(bal & ((1ull << lane) - 1ull));
      if (in) vmap[p] = virt ? (rank < nv_asserted ? V + rank : -1) : (cnt == 1 ? first : -1);
      if (virt && rank < nv_asserted && rank < NVMAX) {
#pragma unroll
        for (int j = 0; j < 4; ++j) vl_idx[rank][j] = ei[ps][j], vl_wt[rank][j] = ew[ps][j];
      }
      base += __popcll(bal);
    }
    if (lane == 0) nv_s = base < nv_asserted ? base : nv_asserted;
  }
  asm volatile("" ::: "memory");   // the bias rows' wait + LDS store stay BEHIND the classification's loads (hoisted above them, the store's
  if (tid < BM) rowp[tid] = bias_row;   // wait for the bias separated two round trips that now are one)
  if (tid < 2 * PL) smem_u[(tid >> 1) * BUF + WUNITS + (tid & 1) * XS + ZUNIT] = make_uint4(0u, 0u, 0u, 0u);   // the zero units
  G2_TL(0);
  __syncthreads();   // rowp, vmap, nv_s
  G2_TL(1);

  bool colok[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) colok[ns] = (wn * NS + ns) * 32 + l31 < ncols;
  const int NV = nv_s;
  const int FS = V + NV;   // units per frame of the raw tile
  int xdst[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int u = tid + 256 * j;
    const int xh = u / TN, xc = u - xh * TN;
    const int f = xc / V, v = xc - f * V;
    xdst[j] = xc < k.FT * V ? WUNITS + xh * XS + f * FS + v : -1;   // columns beyond the tile's frames are not stored
  }
  unsigned vo[NS];
  int boff[3][NS];   // unit (inside a buffer) of this lane's B fragment of slice tp for column block ns
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    const int pv = colok[ns] ? p : 0;
    vo[ns] = colok[ns] ? (unsigned)((((int64_t)b * d.T_out + t0) * V + pv) * 16 + 8 * hi) : 0x80000000u;
    const int fo = pv / V, v = pv - fo * V;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) {
      const int vm = vmap[tp * V + v];
      boff[tp][ns] = WUNITS + hi * XS + ((colok[ns] && vm >= 0) ? fo * FS + vm : ZUNIT);
    }
  }
  // bias term sum_k b_k[m] colsum(A_k)[w] (forward only: a data gradient has none and starts from zero).  The bias rows are read
  // BC at a time, unconditionally, BEFORE the arithmetic: written as `colok ? f(rowp[..]) : 0` per accumulator the compiler emitted
  // 64 branches, each with its own LDS read and wait -- 64 dependent LDS round trips, about half of a 64-channel workgroup's
  // prologue (tools/g2_timeline.sh)
  if (d.bias) {   // uniform
    constexpr int BC = SAR_G2_BIAS_CHUNK;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += BC) {
        float4 bp[BC];
#pragma unroll
        for (int r = 0; r < BC; ++r) bp[r] = rowp[(wm * MS + ms) * 32 + mfma_row(r0 + r, hi)];
#pragma unroll
        for (int r = 0; r < BC; ++r) asm volatile("" : "+v"(bp[r].x), "+v"(bp[r].y), "+v"(bp[r].z));   // the reads stay where they are
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
          for (int r = 0; r < BC; ++r) {
            const float v = fmaf(bp[r].z, gcs[2][ns], fmaf(bp[r].y, gcs[1][ns], bp[r].x * gcs[0][ns]));
            acc[ms][ns][r0 + r] = colok[ns] ? v : 0.f;
          }
      }
  } else {
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;
  }
  // mini-builder geometry: item = (plane, frame, virtual joint), at most ONE per thread (PL * FT * NV <= 256 is checked by the host)
  const int nitems = PL * k.FT * NV;
  int vb_dst = -1, vb_src[4] = {0, 0, 0, 0};
  float vb_wt[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid < nitems) {
    const int pl = tid / (k.FT * NV), rem = tid - pl * (k.FT * NV);
    const int f = rem / NV, r = rem - f * NV;
    vb_dst = WUNITS + pl * XS + f * FS + V + r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vb_src[j] = WUNITS + pl * XS + f * FS + vl_idx[r][j];
      vb_wt[j] = vl_wt[r][j];
    }
  }

  auto store_images = [&](uint4* buf, const uint4 (&wreg)[WIT], const uint4 (&xreg)[XJ]) {
#pragma unroll
    for (int i = 0; i < WIT; ++i)
      if ((i + 1) * 256 <= WUNITS || tid + 256 * i < WUNITS) buf[tid + 256 * i] = wreg[i];
#pragma unroll
    for (int j = 0; j < XJ; ++j)
      if (xdst[j] >= 0) buf[xdst[j]] = xreg[j];
  };
  auto build_virtual = [&](uint4* buf) {
    if (vb_dst >= 0) {
      float z[8], x[8];
      cn8_unpack(buf[vb_src[0]], x);
#pragma unroll
      for (int c = 0; c < 8; ++c) z[c] = vb_wt[0] * x[c];
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        cn8_unpack(buf[vb_src[j]], x);
#pragma unroll
        for (int c = 0; c < 8; ++c) z[c] = vb_wt[j] != 0.f ? fmaf(vb_wt[j], x[c], z[c]) : z[c];   // a missing entry adds nothing, whatever its source holds
      }
      buf[vb_dst] = cn8_pack(z);
    }
  };
  auto mma_phase = [&](const uint4* buf) {
    SAR_LDS_SKEW();   // this wave reads buffer s & 1 late: the others may only fill the OTHER buffer before the next barrier A
    const uint4* Wa = buf + hi * BM + wm * MS * 32 + l31;
    uint4 fa[2][MS], fb[2][NS];
    auto frag_load = [&](int tp, uint4 (&a)[MS], uint4 (&bq)[NS]) {
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a[ms] = Wa[tp * PL * BM + ms * 32];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) bq[ns] = buf[boff[tp][ns]];
    };
    if (SAR_G2_WG4) {   // four waves per SIMD cover the LDS round trip: one fragment set
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        frag_load(tp, fa[0], fb[0]);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&fa[0][ms]),
                                                                  *reinterpret_cast<bf16x8*>(&fb[0][ns]), acc[ms][ns], 0, 0, 0);
      }
      return;
    }
    frag_load(0, fa[0], fb[0]);
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) {
      if (tp + 1 < 3) frag_load(tp + 1, fa[(tp + 1) & 1], fb[(tp + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // the reads of the next slice stay AHEAD of this slice's MFMAs
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&fa[tp & 1][ms]),
                                                                *reinterpret_cast<bf16x8*>(&fb[tp & 1][ns]), acc[ms][ns], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the ADD epilogue's aux half units (skip-path / residual gradient of the data gradient) are loaded before the last MFMA
  // phase, not inside the epilogue (conv_gemm_cn8_kernel)
  const bool epi_gate = d.epi == SAR_EPI_ADD_GATE;
  const bool pre_aux = (!SAR_G2_WG4 || SAR_G2_WG4_PREAUX) && (d.epi == SAR_EPI_ADD || epi_gate);
  // four workgroups per CU: no room for every aux half unit during the MFMA phase -- the epilogue reads its operands one half
  // block ahead, and the first half block is requested here, before the last MFMA phase
  const bool pre_half = SAR_G2_PRE_HALF && !pre_aux && (d.epi == SAR_EPI_ADD || epi_gate);
  Epi8Half<NS> pre0;
  u32x2 axr[MS * 4 * NS];
  auto issue_aux = [&]() {
    const Epi8Desc e8 = epi8_desc<MS>(k, wm, m0, true);
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns)
            axr[((ms * 2 + rb) * 2 + q2) * NS + ns] =
                __builtin_amdgcn_raw_buffer_load_b64(e8.ra, vo[ns], (4 * ms + 2 * rb + q2) * e8.so_aux, 0);
  };
  // Happens-before of the two buffers (no closing barrier): stage s stores into buffer s & 1, last READ by the MFMA phase of
  // stage s - 2; a wave reaches store(s) only after barrier B of stage s - 1, which every wave joins after ITS MFMA phase of
  // stage s - 2 (program order: MFMA(s-2), store(s-1), A(s-1), build, B(s-1)).  The mini-builder of stage s reads raw units
  // of buffer s & 1 written before barrier A(s) and writes virtual units nobody reads before barrier B(s).
  const int nst = (d.Kc + KC16 - 1) / KC16;
  auto stage = [&](int s, uint4 (&wreg)[WIT], uint4 (&xreg)[XJ], bool last) {
    uint4* buf = smem_u + (s & 1) * BUF;
    if (!(SAR_G2_ABLATE & 16) || s == 0) store_images(buf, wreg, xreg);
    __syncthreads();         // A: raw tile + weights of this stage complete
    const int sn = s + (SAR_G2_PF2 ? 2 : 1);
    if (!(SAR_G2_ABLATE & 2) && sn < nst) issue_loads(sn * KC16, wreg, xreg);   // uniform; in flight for TWO stages
    if (last && pre_aux) issue_aux();   // uniform (last stage: the aux half units land during the builder + MFMA phase)
    if (last && epi_gate && tid < BM) {   // SAR_EPI_ADD_GATE: the centre of the second reduction replaces the bias rows (every wave
      float4 ap = make_float4(0.f, 0.f, 0.f, 0.f);   // is past its accumulator initialisation: >= 1 barrier ago)
      if (m0 + tid < d.M && d.aux_mean) ap.z = d.aux_mean[m0 + tid];
      rowp[tid] = ap;
    }
    if (!(SAR_G2_ABLATE & 8)) build_virtual(buf);
    __syncthreads();         // B: virtual joints complete
    if (last && pre_half) {  // uniform: the epilogue's first half block lands during the last MFMA phase (the staging registers are free)
      const Epi8Desc e8 = epi8_desc<MS>(k, wm, m0, true);
      if (epi_gate) epi8_half_loads<NS, true, true>(e8, gate8_desc(k, e8.g_w), vo, 0, pre0);
      else epi8_half_loads<NS, true, false>(e8, Gate8Desc(), vo, 0, pre0);
    }
    if (!(SAR_G2_ABLATE & 1)) mma_phase(buf);
  };
  G2_TL(2);
  int s_ = 0;
  if (SAR_G2_PF2) {
    for (; s_ + 2 < nst; s_ += 2) {
      stage(s_, wregA, xregA, false);
      stage(s_ + 1, wregB, xregB, false);
    }
    if (nst - s_ == 2) {
      stage(s_, wregA, xregA, false);
      stage(s_ + 1, wregB, xregB, true);
    } else {
      stage(s_, wregA, xregA, true);
    }
  } else {
    for (; s_ + 1 < nst; ++s_) stage(s_, wregA, xregA, false);
    stage(s_, wregA, xregA, true);
  }
  __syncthreads();           // the epilogue's transpose area aliases the image
  G2_TL(3);
  if (SAR_G2_ABLATE & 4) {   // every accumulator stays live
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[ms][ns][r]));
    return;
  }
  if (pre_aux) epilogue8<MS, NS, WN, BM, true>(k, tile, wm, wn, m0, vo, acc, rowp, smem, axr);
  else if (pre_half) epilogue8<MS, NS, WN, BM, false>(k, tile, wm, wn, m0, vo, acc, rowp, smem, nullptr, &pre0);
  else epilogue8<MS, NS, WN, BM, false>(k, tile, wm, wn, m0, vo, acc, rowp, smem);
#ifdef SAR_G2_TIMELINE
  G2_TL(4);
  if (tid == 0 && blockIdx.x < G2_TL_WG) {
    unsigned* row = g_g2_tl[blockIdx.x];
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    row[0] = (unsigned)tl_rt0, row[1] = (unsigned)rt1;
    row[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
    row[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    row[4] = tl_acc[0], row[5] = tl_acc[1], row[6] = tl_acc[2], row[7] = tl_acc[3], row[8] = tl_acc[4], row[9] = (unsigned)w;
  }
#endif
}

template <int MS, int NS, int WM, int WN, int WG4>
int launch_graph2_cfg(const sar_conv_desc& d, const uint4* wp, hipStream_t st, int* nparts_only, int nv) {
  ConvK8 k;
  fill_common(d, wp, k);
  if (int g = tile_geometry8<WN>(d, NS, false, k)) {
    sar_set_error("sar_conv_gemm_cn8: unsupported tile geometry (V=%d)", d.V);
    return g == -2 ? SAR_E_UNSUP : SAR_E_ARG;
  }
  constexpr int BM = 32 * MS * WM, TN = 32 * NS * WN;
  // the raw tile with its virtual joints must fit a plane, the mini-builder handles one unit per thread
  if (k.FT * (d.V + nv) + 1 > TN + TN / 2 + 8 || 2 * k.FT * nv > 256) return SAR_GRAPH2_NOT_APPLICABLE;
  if (nparts_only) {
    *nparts_only = k.nparts;
    return 0;
  }
  k.ntiles = d.B * k.TPS;
  k.ny = (d.M + BM - 1) / BM;
  const int nwork = k.ntiles * k.ny;
  hipLaunchKernelGGL((conv_graph2_cn8_kernel<MS, NS, WM, WN, WG4>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, st, k, nv);
  return 0;
}

}  // namespace

#ifdef SAR_G2_TIMELINE
extern "C" int sar_debug_g2_timeline(unsigned* out, int nwg, int reset) {   // out: [nwg][12]
  if (nwg > G2_TL_WG) nwg = G2_TL_WG;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_g2_tl), (size_t)nwg * 12 * sizeof(unsigned)) != hipSuccess) return -1;
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_g2_tl)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned) * 12 * G2_TL_WG) != hipSuccess) return -1;
  }
  return nwg;
}
#endif

// Called by sar_conv_gemm_cn8 / sar_conv_gemm_cn8_nparts (conv_gemm_cn8.hip) for SAR_CONV_GRAPH descriptors.  Returns
// SAR_GRAPH2_NOT_APPLICABLE when the caller has not asserted SAR_GRAPH_FEW_DENSE or the tile cannot hold the virtual joints
// (the caller then keeps the unit builder; both kernels use the same tiles, i.e. the same partial-sum geometry).
int sar_graph2_cn8_dispatch(const sar_conv_desc& d, const void* wp_, hipStream_t st, int* np) {
  static const bool on = [] {
    const char* e = getenv("SAR_GRAPH_READ_GATHER");
    return !(e && e[0] == '0');
  }();
  if (!on || !(d.g_flags & SAR_GRAPH_FEW_DENSE) || d.V > 64) return SAR_GRAPH2_NOT_APPLICABLE;
  const int nv = (d.g_flags >> SAR_GRAPH_FEW_DENSE_SHIFT) & 0xff;
  if (nv > NVMAX) return SAR_GRAPH2_NOT_APPLICABLE;
  const uint4* wp = (const uint4*)wp_;
  static const bool wg4 = [] {
    const char* e = getenv("SAR_GRAPH2_WG4");
    return e ? e[0] == '1' : SAR_GRAPH2_WG4_DEFAULT;
  }();
  if (wg4) {
    if (d.M > 64) return launch_graph2_cfg<2, 2, 2, 2, 1>(d, wp, st, np, nv);
    if (d.M > 32) return launch_graph2_cfg<2, 2, 1, 4, 1>(d, wp, st, np, nv);
    return launch_graph2_cfg<1, 2, 1, 4, 1>(d, wp, st, np, nv);
  }
  if (d.M > 64) return launch_graph2_cfg<2, 2, 2, 2, 0>(d, wp, st, np, nv);
  if (d.M > 32) return launch_graph2_cfg<2, 2, 1, 4, 0>(d, wp, st, np, nv);
  return launch_graph2_cfg<1, 2, 1, 4, 0>(d, wp, st, np, nv);
}

// conv_gemm_cn8.hip -- graph / temporal / residual convolutions and their data gradients on bf16 CN8 activations
// (cn8.h): SURVEY.md 8(d) config 3 -- bf16 activations in HBM, bf16 MFMA operands, fp32 accumulation, fp32 BatchNorm
// statistics (reduced from the accumulators), fp32 master weights.
//
//   out[m, n] = sum_tap sum_c bf16(W[tap][c][m]) * OP_tap(pro(src))[c, n] (+ bias) ; epilogue ; out rounded to bfloat16
//
// Operators, prologue and epilogue semantics are those of sar_conv_gemm_f32 (include/sar_hip.h; models/stgcn.py:27-36,
// 47-54, models/gcn.py:199-209 and their data gradients).  What the layout buys (cn8.h):
//  * src -> LDS is a straight copy of 16-byte units (8 channels of one column) into the k-innermost operand image of
//    v_mfma_f32_32x32x16_bf16; TF-SAME zero padding and everything outside the sequence come from the hardware range
//    check of a per-(sequence, channel group) buffer descriptor (a negative or past-the-end offset reads 0);
//  * a folded BatchNorm + ReLU prologue (temporal forward) touches the unit in registers: unpack, fma, max, repack;
//  * the graph gather z_k = x . A_k reads whole units (8 channels per ds_read_b128) of the raw tile;
//  * the epilogue stores the 4 consecutive channels a lane holds in registers 4q .. 4q+3 as ONE 8-byte half unit:
//    512 contiguous bytes per wave instruction, 4 stores per 32x32 accumulator tile (fp32 CN layout: 16).
#include "conv_cn8_common.h"

namespace {

// TR: 0 forward; 1 data gradient, stride 1; 2 data gradient, generic stride (tap validity mask); 3 data gradient,
// stride 2, parity-split column map (see conv_gemm.hip)
template <int TR, int TAPS, int MS, int NS, int WM, int WN>
__global__ __launch_bounds__(256, MS * NS > 4 ? 2 : 3) void conv_gemm_cn8_kernel(const ConvK8 k) {
  using TC = TileCfg8<TAPS, MS, NS, WM, WN>;
  constexpr int TRANSPOSED = TR != 0;
  constexpr int PAR = (TR == 3);
  constexpr int JT = PAR ? (TAPS + 1) / 2 : TAPS;
  constexpr int BM = TC::BM, SCOLS = TC::SCOLS, CJ = TC::CJ, WIT = TC::WIT;
  constexpr int ZCOL = TC::RWMAX;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int PAREA_U = 4 * 16 * 65 / 4;   // the epilogue's transpose area (floats / 4), aliases the operand image
  constexpr int IMG_U = TC::UNITS > PAREA_U ? TC::UNITS : PAREA_U;
  __shared__ uint4 smem_u[IMG_U + BM];       // image | per-row parameters (float4 per row)
  // Folded BatchNorm scale / shift of ALL src channels, staged once: fetched per stage in the store phase (cn8_params8) they were
  // vector loads with a wait each -- scale, wait, shift, wait, per plane: four dependent L2 round trips in EVERY stage, most of
  // the 2 700-3 500 cycles the phase stamps attributed to "store" (found in the ISA after tools/g2_timeline.sh)
  constexpr int PMAX = 512;
  __shared__ float pro_l[2 * PMAX];
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
  const int ny = k.ny;
  const int w = xcd_work(k.ntiles * ny);
  if (w < 0) return;
#ifdef SAR_CN8_STAMPS
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_last = st_t0;
#endif
  if (k.stagger && blockIdx.x < 768) {   // de-phase the co-resident workgroups (they would otherwise run their phases in lockstep)
    const int ph = (blockIdx.x >> k.stagger_shift) % 3;
    for (int i = 0; i < ph * k.stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  const int tile = w / ny;
  const int b = tile / k.TPS;
  const int t0 = (tile - b * k.TPS) * k.FT;
  const int m0 = (w - tile * ny) * BM;

  // ---- per-lane column geometry
  bool colok[NS];
  unsigned vo[NS];   // epilogue: byte offset of the lane's half unit in a plane of out / aux
  // LDS unit of tap slot j for column block ns: off0[ns] + j * ostep[ns] (forward: +V per tap; data gradients: -V per tap
  // slot; off-tile columns: the zero column with step 0).  TR == 2 (generic stride) keeps an explicit table.
  int off0[NS], ostep[NS];
  int offt[TR == 2 ? JT : 1][NS];
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
    vo[ns] = colok[ns] ? (unsigned)((((int64_t)b * d.T_out + (t0 + fo)) * V + v) * 16 + 8 * hi) : 0x80000000u;
    vmask[ns] = 0;
    if (TR == 0) {
      off0[ns] = fo * d.stride * V + v;
      ostep[ns] = V;
    } else if (TR == 1) {
      off0[ns] = (t0 + fo + d.pad - t_lo) * V + v;                       // tap 0; tap tp reads tp frames earlier
      ostep[ns] = -V;
    } else if (PAR) {
      off0[ns] = (((t0 + fo + d.pad - tp0) >> 1) - t_lo) * V + v;        // exact: the numerator is even
      ostep[ns] = -V;
    } else {
      off0[ns] = 0;
      ostep[ns] = 0;
#pragma unroll
      for (int tp = 0; tp < JT; ++tp) {
        const int q = t0 + fo + d.pad - tp;
        const int to = floordiv(q, d.stride);
        const bool ok = (q - to * d.stride) == 0;
        vmask[ns] |= (ok ? 1u : 0u) << tp;
        offt[tp][ns] = (colok[ns] ? (to - t_lo) * V + v : ZCOL) + hi * SCOLS;
      }
    }
    if (!colok[ns]) {
      off0[ns] = ZCOL;
      ostep[ns] = 0;
    }
    off0[ns] += hi * SCOLS;   // this lane's k half
  }

  // the bias row is REQUESTED here and stored to LDS behind the stage-0 loads: stored here, its wait (a memory round trip) stood in
  // front of the issue of the stage's operands (tools/g2_timeline.sh found the same pattern in the graph kernel)
  float bias_v = 0.f;
  if (tid < BM && d.bias && m0 + tid < d.M) bias_v = d.bias[m0 + tid];
  if (tid < 2) Sl[tid * SCOLS + ZCOL] = make_uint4(0u, 0u, 0u, 0u);
  f32x16 acc[MS][NS];

  // ---- staging.  S: unit (h, col) <- plane c0/8 + h, column t_lo V + col of sequence b; the descriptor spans exactly the
  // sequence, so the temporal zero padding (negative / past-the-end columns) is the range check's 0
  const int seq_len = d.T_src * V;
  const char* src_b = (const char*)d.src + (int64_t)b * seq_len * 16;
  // unit tid + 256 j of a plane: column t_lo V + tid + 256 j of the sequence (byte offset svo0 + 4096 j); sbits: bit j set =
  // inside the staged width AND inside the sequence (the prologue must keep everything else exactly 0)
  const int svo0 = (t_lo * V + tid) * 16;
  unsigned sbits = 0, swidth = 0;
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int col = tid + 256 * j;
    const int rabs = t_lo * V + col;
    swidth |= (col < k.RW ? 1u : 0u) << j;
    sbits |= ((col < k.RW && (unsigned)rabs < (unsigned)seq_len) ? 1u : 0u) << j;
  }
  // W unit tid + 256 i = (tap, half, row): 256 = (256 / (2 BM)) whole taps, so the lane's (half, row) is fixed and the tap
  // advances by 256 / (2 BM) per i: ONE per-lane offset, the rest is a scalar stride
  static_assert(256 % (2 * BM) == 0, "a W pass must cover whole taps");
  constexpr int TPI = 256 / (2 * BM);
  const int wm_ = tid % BM, wh_ = (tid / BM) & 1, wt_ = tid / (2 * BM);
  const unsigned wvo0 = (m0 + wm_) < d.M ? (unsigned)((((int64_t)wt_ * k.G + wh_) * d.M + m0 + wm_) * 16)
                                         : 0x80000000u;   // rows beyond M: rejected by the range check -> 0
  const int wstep = TPI * k.G * d.M * 16;   // bytes per W pass (scalar)
  const unsigned wbytes = (unsigned)((int64_t)TAPS * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  const bool has_pro = d.pro_scale != nullptr;
  const bool pro_relu = d.pro_relu != 0;
  const bool pro_lds = has_pro && d.Kc <= PMAX;   // uniform
  const int ncj = (k.RW + 255) >> 8;   // live 256-column chunks of the staged window (wave-uniform, <= CJ)
  uint4 wreg[WIT];
  uint4 sreg[2][CJ];

  auto issue_loads = [&](int c0) {
    const int wso = (c0 / 8) * d.M * 16;   // scalar: first channel group of the stage
#pragma unroll
    for (int i = 0; i < WIT; ++i) {   // taps beyond TAPS (last, partial pass) lie past the image: range check -> 0
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo0, wso + i * wstep, 0);
      wreg[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int g = c0 / 8 + h;   // wave-uniform
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(src_b + (int64_t)(g < k.Gs ? g : 0) * d.ld_src * 16), 0, g < k.Gs ? (unsigned)seq_len * 16u : 0u, 0x00020000);
#pragma unroll
      for (int j = 0; j < CJ; ++j)
        if (j < ncj) {   // wave-uniform: chunks beyond the staged width are never loaded, transformed, stored or read
          // the offset is formed in the vector ALU (32-bit wrap: a negative lane base plus 4096 j is the right non-negative
          // offset); as a scalar offset the hardware range check would see the un-wrapped sum and reject it
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ((swidth >> j) & 1u) ? svo0 + j * 4096 : 0x7fffffff, 0, 0);
          sreg[h][j] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
  };

  auto store_lds = [&](int c0) {
#pragma unroll
    for (int i = 0; i < WIT; ++i)
      if ((i + 1) * 256 <= TC::WUNITS || tid + 256 * i < TC::WUNITS) Wl[tid + 256 * i] = wreg[i];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (has_pro) {   // uniform: BatchNorm + ReLU of the producer folded into the operand (models/stgcn.py:27-28)
        float psc[8], psh[8];
        if (pro_lds) {   // uniform: four broadcast LDS reads (channels beyond Kc: zeros, as cn8_params8)
          const float4* ps = reinterpret_cast<const float4*>(pro_l + c0 + 8 * h);
          const float4* pt = reinterpret_cast<const float4*>(pro_l + PMAX + c0 + 8 * h);
          const float4 a = ps[0], bq = ps[1], c = pt[0], e = pt[1];
          psc[0] = a.x, psc[1] = a.y, psc[2] = a.z, psc[3] = a.w, psc[4] = bq.x, psc[5] = bq.y, psc[6] = bq.z, psc[7] = bq.w;
          psh[0] = c.x, psh[1] = c.y, psh[2] = c.z, psh[3] = c.w, psh[4] = e.x, psh[5] = e.y, psh[6] = e.z, psh[7] = e.w;
        } else {
          cn8_params8(d.pro_scale, c0 + 8 * h, d.Kc, psc);
          cn8_params8(d.pro_shift, c0 + 8 * h, d.Kc, psh);
        }
#pragma unroll
        for (int j = 0; j < CJ; ++j)
          if (j < ncj)   // padding / columns outside the sequence stay exactly 0 (keep mask)
            sreg[h][j] = cn8_bn_relu_unit(sreg[h][j], psc, psh, pro_relu, ((sbits >> j) & 1u) ? 0xffffffffu : 0u);
      }
#pragma unroll
      for (int j = 0; j < CJ; ++j)
        if (j < ncj && ((j + 1) * 256 <= TC::RWMAX || tid + 256 * j < TC::RWMAX)) Sl[h * SCOLS + tid + 256 * j] = sreg[h][j];
    }
  };

  issue_loads(0);
  if (pro_lds) {   // uniform; requested behind the stage-0 operands, visible behind the barrier below
    float pv[2][2 * PMAX / 256];
#pragma unroll
    for (int i = 0; i < PMAX / 256; ++i) {
      const int c = tid + 256 * i;
      pv[0][i] = c < d.Kc ? d.pro_scale[c] : 0.f;
      pv[1][i] = c < d.Kc ? d.pro_shift[c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PMAX / 256; ++i) pro_l[tid + 256 * i] = pv[0][i], pro_l[PMAX + tid + 256 * i] = pv[1][i];
  }
  asm volatile("" ::: "memory");
  if (tid < BM) rowp[tid] = make_float4(bias_v, 0.f, 0.f, 0.f);
  __syncthreads();   // rowp, pro_l
  // HAPPENS-BEFORE of the LDS regions of this kernel (single image, two barriers per stage):
  //  * image (W | S): written by store_lds(s) behind the closing barrier of stage s - 1, which every wave joins after its MFMA phase
  //    of stage s - 1 (its last read of the image); read behind the opening barrier of stage s;
  //  * rowp holds the bias rows until every wave has initialised its accumulators (below, before the opening barrier of stage 0)
  //    and is rewritten with the MASK parameters behind the opening barrier of the LAST stage (>= 1 barrier later); the epilogue
  //    reads them behind the closing barrier of the last stage;
  //  * the epilogue's transpose area aliases the image and is wave-private; it is written behind that same closing barrier.
  SAR_LDS_SKEW();   // last read of the bias rows
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 bp = rowp[(wm * MS + ms) * 32 + mfma_row(r, hi)];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[ms][ns][r] = colok[ns] ? bp.x : 0.f;
    }

  const uint4* Wa = Wl + (tp0 * 2 + hi) * BM + wm * MS * 32 + l31;
  // Operand fragments of tap slot j.  The tap loop is software-pipelined BY HAND over two fragment sets: the four
  // ds_read_b128 of slot j + 1 are issued before the four MFMAs of slot j.  (Left to the compiler, every slot re-used ONE
  // fragment set -- read, wait for the LDS round trip, multiply, read ... : ~270 cycles per slot for 128 cycles of matrix
  // work, which is what a wave's MFMA phase cost whenever its co-resident waves were not multiplying themselves.)
  // The B addresses are formed per use (off0 + j * ostep): kept loop-invariant they were 18 VGPRs.
  auto frag_load = [&](int j, const int (&ost)[NS], uint4 (&a)[MS], uint4 (&bq)[NS]) {
    const int tpw = PAR ? 2 * j : j;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) a[ms] = Wa[tpw * 2 * BM + ms * 32];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      bq[ns] = Sl[TR == 2 ? offt[TR == 2 ? j : 0][ns] : off0[ns] + j * ost[ns]];
      if (TR == 2 && !((vmask[ns] >> j) & 1u)) bq[ns] = make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto frag_mma = [&](uint4 (&a)[MS], uint4 (&bq)[NS]) {
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a[ms]),
                                                              *reinterpret_cast<bf16x8*>(&bq[ns]), acc[ms][ns], 0, 0, 0);
  };

  auto mma_phase = [&]() {
    constexpr int JSURE = PAR ? JT - 1 : JT;
    SAR_LDS_SKEW();   // this wave reads the image late: nobody may restage it before the closing barrier
    if (!(SAR_ABLATE8 & 1)) {
#ifdef SAR_CN8_SETPRIO
      __builtin_amdgcn_s_setprio(1);   // experiment: the multiplying wave wins issue arbitration against co-resident staging waves
#endif
      int ost[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        ost[ns] = ostep[ns];
        asm volatile("" : "+v"(ost[ns]));   // not loop-invariant any more: the addresses are re-formed, not kept
      }
      uint4 fa[2][MS], fb[2][NS];
      frag_load(0, ost, fa[0], fb[0]);
#pragma unroll
      for (int j = 0; j < JSURE; ++j) {
        if (j + 1 < JSURE) frag_load(j + 1, ost, fa[(j + 1) & 1], fb[(j + 1) & 1]);
        else if (PAR && ntap_w == JT) frag_load(JT - 1, ost, fa[(j + 1) & 1], fb[(j + 1) & 1]);   // wave-uniform
        __builtin_amdgcn_sched_barrier(0);   // the reads of the next slot stay AHEAD of this slot's MFMAs
        frag_mma(fa[j & 1], fb[j & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (PAR && ntap_w == JT) frag_mma(fa[JSURE & 1], fb[JSURE & 1]);
#ifdef SAR_CN8_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
  };
  // aux half units of the whole wave tile (ReLU-mask source / residual gradient): MS x 2 x 2 x NS loads of 8 bytes issued
  // BEFORE the last MFMA phase.  Loaded inside the epilogue they were four dependent HBM round trips per workgroup (one per
  // 32-row half block): 12 us of a 25 us workgroup on the 64-channel data gradient.
  const bool epi_mask = d.epi == SAR_EPI_MASK;
  const bool has_aux = epi_mask || d.epi == SAR_EPI_ADD;
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

  int c0 = 0;
  STAMP8(4);
  for (; c0 + KC16 < d.Kc; c0 += KC16) {
    if (!(SAR_ABLATE8 & 8) || c0 == 0) store_lds(c0);
    STAMP8(0);
    __syncthreads();
    STAMP8(1);
    if (!(SAR_ABLATE8 & 2)) issue_loads(c0 + KC16);   // in flight during the MFMA phase
    mma_phase();
    STAMP8(2);
    __syncthreads();   // every wave is done with the image (next store)
    STAMP8(3);
  }
  // last stage (peeled: the aux registers take the place of the staging registers)
  if (!(SAR_ABLATE8 & 8) || c0 == 0) store_lds(c0);
  STAMP8(0);
  __syncthreads();
  STAMP8(1);
  if (has_aux) issue_aux();   // uniform
  if (epi_mask && tid < BM) {   // MASK parameters replace the bias rows (every wave is past its accumulator initialisation)
    const int row = m0 + tid;
    float4 ap = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < d.M) {
      ap.x = d.aux_scale[row];
      ap.y = d.aux_shift[row];
      if (d.aux_mean) ap.z = d.aux_mean[row];
    }
    rowp[tid] = ap;
  }
  mma_phase();
  STAMP8(2);
  __syncthreads();   // the epilogue's transpose area aliases the image; rowp is complete
  STAMP8(3);
  if (SAR_ABLATE8 & 4) {   // every accumulator stays live (an un-used one would take its MFMAs with it)
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[ms][ns][r]));
    return;
  }
  if (has_aux) epilogue8<MS, NS, WN, BM, true>(k, tile, wm, wn, m0, vo, acc, rowp, smem, axr);
  else epilogue8<MS, NS, WN, BM, false>(k, tile, wm, wn, m0, vo, acc, rowp, smem);
#ifdef SAR_CN8_STAMPS
  __builtin_amdgcn_s_waitcnt(0);   // the epilogue's stores have left
  STAMP8(5);
  if (tid == 0 && blockIdx.x < STAMP_WG) {
    unsigned* row = g_stamps8[blockIdx.x];
    for (int i = 0; i < 6; ++i) row[i] = (unsigned)st_acc[i];
    row[6] = 1u;
    row[7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
    row[8] = (unsigned)(__builtin_amdgcn_s_memtime() - st_t0);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Deep-prefetch 9-tap kernel (experiment switch SAR_CN8_DB=1): stride-1 forward (TR 0), stride-1 data gradient (TR 1),
// parity-split stride-2 data gradient (TR 3) on 64 x 256 tiles.  Same arithmetic, operand image and epilogue as
// conv_gemm_cn8_kernel; the src units of stage s+2 are loaded (into a second register set) while stage s is multiplied, so
// that an HBM round trip has TWO MFMA phases to land (the weights of stage s+1, which come from L2, stay one stage ahead).
// The staged window is capped at 512 columns (18-19 frames): the image is 35 KB.
// (The first variant tried here kept two LDS images with one barrier per stage: only two workgroups fit a CU and it was
// 5-10 % slower than three single-image workgroups -- forward 1.81 vs 1.73 ms per step, data gradient 2.10 vs 1.89.)
template <int TR, int MS, int NS, int WM, int WN>
__global__ __launch_bounds__(256, 3) void conv_gemm_cn8_db_kernel(const ConvK8 k) {
  constexpr int TAPS = 9;
  constexpr int PAR = (TR == 3);
  constexpr int JT = PAR ? (TAPS + 1) / 2 : TAPS;
  constexpr int BM = 32 * MS * WM;
  constexpr int RWMAX = 512, SCOLS = RWMAX + 8, ZCOL = RWMAX, CJ = RWMAX / 256;
  constexpr int WUNITS = TAPS * 2 * BM, SUNITS = 2 * SCOLS, BUF = WUNITS + SUNITS;
  constexpr int WIT = (WUNITS + 255) / 256;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(TR == 0 || TR == 1 || TR == 3, "forward, stride-1 and parity-split data gradients");
  static_assert(BUF >= 4 * 16 * 65 / 4, "the epilogue's transpose area aliases the image");
  __shared__ uint4 smem_u[BUF + BM];   // image | per-row parameters
  uint4* Wl = smem_u;
  uint4* Sl = smem_u + WUNITS;
  float* smem = reinterpret_cast<float*>(smem_u);
  float4* rowp = reinterpret_cast<float4*>(smem_u + BUF);
  const sar_conv_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int V = d.V;
  const int ny = k.ny;
  const int w = xcd_work(k.ntiles * ny);
  if (w < 0) return;
  const int tile = w / ny;
  const int b = tile / k.TPS;
  const int t0 = (tile - b * k.TPS) * k.FT;
  const int m0 = (w - tile * ny) * BM;

  // ---- per-lane column geometry (conv_gemm_cn8_kernel)
  bool colok[NS];
  unsigned vo[NS];
  int off0[NS], ostep[NS];
  int t_lo;
  if (TR == 0) t_lo = t0 * d.stride - d.pad;
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
    vo[ns] = colok[ns] ? (unsigned)((((int64_t)b * d.T_out + (t0 + fo)) * V + v) * 16 + 8 * hi) : 0x80000000u;
    if (TR == 0) {
      off0[ns] = fo * d.stride * V + v;
      ostep[ns] = V;
    } else if (TR == 1) {
      off0[ns] = (t0 + fo + d.pad - t_lo) * V + v;
      ostep[ns] = -V;
    } else {
      off0[ns] = (((t0 + fo + d.pad - tp0) >> 1) - t_lo) * V + v;
      ostep[ns] = -V;
    }
    if (!colok[ns]) {
      off0[ns] = ZCOL;
      ostep[ns] = 0;
    }
    off0[ns] += hi * SCOLS;
  }

  const bool epi_mask = d.epi == SAR_EPI_MASK;
  const bool has_aux = epi_mask || d.epi == SAR_EPI_ADD;
  // the bias row is REQUESTED here and stored to LDS behind the stage-0 loads: stored here, its wait (a memory round trip) stood in
  // front of the issue of the stage's operands (tools/g2_timeline.sh found the same pattern in the graph kernel)
  float bias_v = 0.f;
  if (tid < BM && d.bias && m0 + tid < d.M) bias_v = d.bias[m0 + tid];
  if (tid < 2) Sl[tid * SCOLS + ZCOL] = make_uint4(0u, 0u, 0u, 0u);
  f32x16 acc[MS][NS];

  // ---- staging (conv_gemm_cn8_kernel)
  const int seq_len = d.T_src * V;
  const char* src_b = (const char*)d.src + (int64_t)b * seq_len * 16;
  const int svo0 = (t_lo * V + tid) * 16;
  unsigned sbits = 0, swidth = 0;
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int col = tid + 256 * j;
    const int rabs = t_lo * V + col;
    swidth |= (col < k.RW ? 1u : 0u) << j;
    sbits |= ((col < k.RW && (unsigned)rabs < (unsigned)seq_len) ? 1u : 0u) << j;
  }
  static_assert(256 % (2 * BM) == 0, "a W pass must cover whole taps");
  constexpr int TPI = 256 / (2 * BM);
  const int wm_ = tid % BM, wh_ = (tid / BM) & 1, wt_ = tid / (2 * BM);
  const unsigned wvo0 = (m0 + wm_) < d.M ? (unsigned)((((int64_t)wt_ * k.G + wh_) * d.M + m0 + wm_) * 16) : 0x80000000u;
  const int wstep = TPI * k.G * d.M * 16;
  const unsigned wbytes = (unsigned)((int64_t)TAPS * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  const bool has_pro = d.pro_scale != nullptr;
  const bool pro_relu = d.pro_relu != 0;
  const int ncj = (k.RW + 255) >> 8;
  uint4 wreg[WIT];
  uint4 sregA[2][CJ], sregB[2][CJ];

  auto issue_w = [&](int c0) {
    const int wso = (c0 / 8) * d.M * 16;
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, wvo0, wso + i * wstep, 0);
      wreg[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  };
  auto issue_s = [&](int c0, uint4 (&sreg)[2][CJ]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int g = c0 / 8 + h;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(src_b + (int64_t)(g < k.Gs ? g : 0) * d.ld_src * 16), 0, g < k.Gs ? (unsigned)seq_len * 16u : 0u, 0x00020000);
#pragma unroll
      for (int j = 0; j < CJ; ++j)
        if (j < ncj) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, ((swidth >> j) & 1u) ? svo0 + j * 4096 : 0x7fffffff, 0, 0);
          sreg[h][j] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
  };
  auto store_lds = [&](int c0, uint4 (&sreg)[2][CJ]) {
#pragma unroll
    for (int i = 0; i < WIT; ++i)
      if ((i + 1) * 256 <= WUNITS || tid + 256 * i < WUNITS) Wl[tid + 256 * i] = wreg[i];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (has_pro) {
        float psc[8], psh[8];
        cn8_params8(d.pro_scale, c0 + 8 * h, d.Kc, psc);
        cn8_params8(d.pro_shift, c0 + 8 * h, d.Kc, psh);
#pragma unroll
        for (int j = 0; j < CJ; ++j)
          if (j < ncj) sreg[h][j] = cn8_bn_relu_unit(sreg[h][j], psc, psh, pro_relu, ((sbits >> j) & 1u) ? 0xffffffffu : 0u);
      }
#pragma unroll
      for (int j = 0; j < CJ; ++j)
        if (j < ncj) Sl[h * SCOLS + tid + 256 * j] = sreg[h][j];
    }
  };

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

  const uint4* Wa = Wl + (tp0 * 2 + hi) * BM + wm * MS * 32 + l31;
  auto mma_phase = [&]() {
    auto taps_mma = [&](int j) {
      const int tpw = PAR ? 2 * j : j;
      uint4 a[MS], bq[NS];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a[ms] = Wa[tpw * 2 * BM + ms * 32];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) bq[ns] = Sl[off0[ns] + j * ostep[ns]];
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a[ms]),
                                                                *reinterpret_cast<bf16x8*>(&bq[ns]), acc[ms][ns], 0, 0, 0);
    };
    constexpr int JSURE = PAR ? JT - 1 : JT;
#pragma unroll
    for (int j = 0; j < JSURE; ++j) taps_mma(j);
    if (PAR && ntap_w == JT) taps_mma(JT - 1);   // wave-uniform
  };

  issue_w(0);
  issue_s(0, sregA);
  if (KC16 < d.Kc) issue_s(KC16, sregB);
  asm volatile("" ::: "memory");
  if (tid < BM) rowp[tid] = make_float4(bias_v, 0.f, 0.f, 0.f);
  __syncthreads();   // rowp / zero column
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 bp = rowp[(wm * MS + ms) * 32 + mfma_row(r, hi)];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[ms][ns][r] = colok[ns] ? bp.x : 0.f;
    }

  // stage s consumes register set s & 1 and refills it with the src units of stage s + 2
  const int nst = (d.Kc + KC16 - 1) / KC16;
  auto stage = [&](int s, uint4 (&sreg)[2][CJ]) {
    store_lds(s * KC16, sreg);
    __syncthreads();
    issue_w((s + 1) * KC16);
    if (s + 2 < nst) issue_s((s + 2) * KC16, sreg);   // uniform
    mma_phase();
    __syncthreads();
  };
  auto last_stage = [&](int s, uint4 (&sreg)[2][CJ]) {
    store_lds(s * KC16, sreg);
    __syncthreads();
    if (epi_mask && tid < BM) {   // MASK parameters replace the bias rows
      const int row = m0 + tid;
      float4 ap = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < d.M) {
        ap.x = d.aux_scale[row];
        ap.y = d.aux_shift[row];
        if (d.aux_mean) ap.z = d.aux_mean[row];
      }
      rowp[tid] = ap;
    }
    mma_phase();
    __syncthreads();
  };
  int s = 0;
  for (; s + 2 < nst; s += 2) {
    stage(s, sregA);
    stage(s + 1, sregB);
  }
  if (s + 1 < nst) {
    stage(s, sregA);
    ++s;
  }
  if (s & 1) last_stage(s, sregB);
  else last_stage(s, sregA);
  if (has_aux) {   // every aux half unit in ONE round trip (the second register set leaves no room to prefetch them earlier)
    issue_aux();
    epilogue8<MS, NS, WN, BM, true>(k, tile, wm, wn, m0, vo, acc, rowp, smem, axr);
  } else {
    epilogue8<MS, NS, WN, BM, false>(k, tile, wm, wn, m0, vo, acc, rowp, smem);
  }
}

// ---- GraphConvTD (models/gcn.py:199-209) and its data gradient:
//   out[m, (t,w)] = sum_k sum_c bf16(W_k[c][m]) * bf16(z_k)[c, (t,w)] + sum_k b_k[m] colsum(A_k)[w],
//   z_k[c, (t,w)] = sum_v src[c, (t,v)] A_k[v, w]      (fp32, <= 4 non-zeros per column of A_k)
// Stage = 16 src channels: (1) the tile's raw units -> LDS (straight copy), (2) every thread builds the three z_k
// units of its (channel half, column): NZ gather entries, each ONE ds_read_b128 of 8 channels, weighted in fp32 and
// rounded once (an entry list {(v, 1.0)} is a plain copy), (3) 3 slices x MS x NS MFMAs.
// MATRIX-CORE GATHER (SAR_GRAPH_WT_BF16_EXACT, V <= 32, one dense slice): step (2) of the slice with 4-entry lists costs
// ~70 vector instructions per unit (unpack, 32 fma, pack) -- a third of the kernel's time, measured by replacing it with a
// copy -- while the matrix pipe is idle (12 MFMAs per stage).  With gather weights that are exact in bfloat16 the slice is
// instead ONE small matrix product per frame: Z_f[16 channels x V] = X_f[16 x 32 joints] . A_k[32 x 32]
// (two v_mfma_f32_16x16x32_bf16; products exact, fp32 accumulation, rounded once like the vector path).  The A operand --
// 8 consecutive joints of one channel -- is the TRANSPOSE of the unit image and comes from ds_read_b64_tr_b16; A_k lives
// in 8 registers per lane for the whole kernel; a result register quad is 4 consecutive channels of one joint = half a
// unit of the z image (one 8-byte LDS store).  Frames are dealt round-robin to the four waves.
template <int MS, int NS, int WM, int WN, int NZ0, int NZ1, int NZ2>
__global__ __launch_bounds__(256, (NZ0 + NZ1 + NZ2 > 6) ? 2 : 3) void conv_graph_cn8_kernel(const ConvK8 k) {
  constexpr int BM = 32 * MS * WM, TN = 32 * NS * WN;
  constexpr int NZ[3] = {NZ0, NZ1, NZ2};
  constexpr int DENSE = ((NZ0 > 1) + (NZ1 > 1) + (NZ2 > 1) == 1) ? (NZ0 > 1 ? 0 : NZ1 > 1 ? 1 : 2) : -1;   // the slice the matrix cores may gather
  constexpr int XS = pad_stride8(TN + 8);    // raw plane stride (units): 4 (mod 16) for the transposed reads, 8 columns of slack behind the tile
  constexpr int CPT = TN / 128;              // columns per thread in the unit builder (thread = (half, column))
  constexpr int WUNITS = 3 * 2 * BM;         // [slice][h][m]
  constexpr int ZUNITS = 3 * 2 * TN;         // [slice][h][col]
  constexpr int WIT = (WUNITS + 255) / 256;
  constexpr int XJ = (2 * TN) / 256;         // raw units per thread and stage
  constexpr int PAREA_U = 4 * 16 * 65 / 4;
  constexpr int IMG_U = (WUNITS + ZUNITS) > PAREA_U ? (WUNITS + ZUNITS) : PAREA_U;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ uint4 smem_u[IMG_U + BM + 2 * XS];
  uint4* Wl = smem_u;
  uint4* Zl = smem_u + WUNITS;
  uint4* XR = smem_u + IMG_U + BM;
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
#ifdef SAR_CN8_STAMPS   // graph kernel: [0] store_raw, [1] barrier, [2] build_units, [3] barrier, [4] prologue, [5] epilogue; loads + MFMA in g_stamps8g
  unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0};
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_last = st_t0;
#endif
  const int tile = w / ny;
  const int b = tile / k.TPS;
  const int t0 = (tile - b * k.TPS) * k.FT;
  const int m0 = (w - tile * ny) * BM;
  const int ncols = ((t0 + k.FT <= d.T_out) ? k.FT : d.T_out - t0) * V;   // live columns of this tile
  const bool mg = DENSE >= 0 && (d.g_flags & SAR_GRAPH_WT_BF16_EXACT) && V <= 32;   // uniform: dense slice on the matrix cores

  // ---- prologue.  Everything that comes from global memory -- the first stage's operands, the gather tables, the column
  // sums, the bias -- is ISSUED first and consumed afterwards: one memory round trip, not one per table (as a sequence of
  // load -> use blocks the prologue was 8 700 cycles, a quarter of a 64-channel workgroup's life)
  f32x16 acc[MS][NS];
  const int seq_left = (d.T_src - t0) * V;   // columns from the tile start to the end of the sequence
  const char* src_b = (const char*)d.src + ((int64_t)b * d.T_src + t0) * V * 16;
  // raw stager: thread -> (plane xh, column xc) for XJ units
  int xvo[XJ], xdst[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int u = tid + 256 * j;
    const int xh = u / TN, xc = u - xh * TN;
    xvo[j] = xc < ncols ? xc * 16 : 0x7fffffff;   // rejected -> 0
    xdst[j] = xh * XS + xc;
  }
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
  uint4 wreg[WIT];
  uint4 xreg[XJ];
  auto issue_loads = [&](int c0) {
    const int wso = (c0 / 8) * d.M * 16;
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
  issue_loads(0);
  // table loads (clamped indices, no branches), all in flight together
  bool colok[NS];
  unsigned vo[NS];
  float gcs[3][NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    colok[ns] = p < ncols;
    const int pv = colok[ns] ? p : 0;
    vo[ns] = colok[ns] ? (unsigned)((((int64_t)b * d.T_out + t0) * V + pv) * 16 + 8 * hi) : 0x80000000u;
    const int v = pv % V;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) gcs[tp][ns] = d.g_colsum ? d.g_colsum[tp * V + v] : 0.f;   // masked by colok at the use
  }
  float4 bias_row = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < BM && d.bias && m0 + tid < d.M) {
    bias_row.x = d.bias[m0 + tid];
    bias_row.y = d.bias[d.M + m0 + tid];
    bias_row.z = d.bias[2 * d.M + m0 + tid];
  }
  // unit builder geometry: this thread's channel half and columns, gather offsets (units inside a raw plane) and weights
  const int uh = tid >> 7;   // 0 / 1 (wave-uniform)
  int go[CPT][3][4];
  float gwt[CPT][3][4];
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int col = (tid & 127) + 128 * q;
    const bool live = col < ncols;
    const int cc = live ? col : 0;
    const int fo = cc / V, v = cc - fo * V;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < NZ[tp] && !(mg && tp == DENSE)) {
          go[q][tp][j] = fo * V + d.g_idx[(tp * V + v) * 4 + j];
          gwt[q][tp][j] = live ? d.g_wt[(tp * V + v) * 4 + j] : 0.f;
        }
  }
  // matrix-core gather: the table entries of this lane's B fragments (A_k[v = 8 G + j][w = (lane & 15) + 16 nb], j = 0..7)
  const int gG = lane >> 4, gi = lane & 15;
  constexpr int DS = DENSE >= 0 ? DENSE : 0;
  int bvi[2][4];
  float bwt[2][4];
  if (mg) {   // uniform
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int wv = gi + 16 * nb, wvc = wv < V ? wv : 0;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < NZ[DS]) {
          bvi[nb][e] = d.g_idx[(DS * V + wvc) * 4 + e];
          bwt[nb][e] = wv < V ? d.g_wt[(DS * V + wvc) * 4 + e] : 0.f;
        }
    }
  }
  STAMP8P(0);
  // ---- consumers
  if (tid < BM) rowp[tid] = bias_row;
  bf16x8 bfr[2];
  unsigned tr_addr = 0;
  int zst_unit = 0;
  // A-operand keep mask: element e of this lane's fragment is joint 8 G + e.  The 32-joint read of a V-joint frame reaches into
  // the NEXT frame's first 32 - V columns; their B rows are zero, but 0 x Inf = NaN, so those elements are cleared before the
  // product (what the vector gather never touched must not reach the result either).
  unsigned keep[4] = {0u, 0u, 0u, 0u};
  if (mg) {
#pragma unroll
    for (int dd = 0; dd < 4; ++dd)
      keep[dd] = ((8 * gG + 2 * dd < V) ? 0xffffu : 0u) | ((8 * gG + 2 * dd + 1 < V) ? 0xffff0000u : 0u);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < NZ[DS]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] += (bvi[nb][e] == 8 * gG + j) ? bwt[nb][e] : 0.f;
        }
      const uint4 pk = cn8_pack(a);
      bfr[nb] = *reinterpret_cast<const bf16x8*>(&pk);
    }
    // transposed read: lane 4 q + p of a 16-lane group supplies joint 8 G + q, channels 4 p .. 4 p + 3 (plane p >> 1)
    const int q = gi >> 2, p = gi & 3;
    tr_addr = (unsigned)(uintptr_t)XR + (unsigned)((((p >> 1) * XS + 8 * gG + q) * 16) + 8 * (p & 1));
    // result quad = channels 4 G .. 4 G + 3 of joint gi (+ 16): half (G & 1) of the unit in plane G >> 1 of the dense z slice
    zst_unit = (DS * 2 + (gG >> 1)) * TN + gi;
    // the dense z slice beyond the last whole frame and the slack columns of the raw image are read but never written
    for (int i = tid; i < 2 * TN; i += 256) Zl[DS * 2 * TN + i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < 2 * (XS - TN); i += 256) XR[(i / (XS - TN)) * XS + TN + i % (XS - TN)] = make_uint4(0u, 0u, 0u, 0u);
  }
  STAMP8P(1);
  STAMP8P(2);
  auto store_raw = [&]() {
#pragma unroll
    for (int j = 0; j < XJ; ++j) XR[xdst[j]] = xreg[j];
  };
  auto build_units = [&]() {
#pragma unroll
    for (int i = 0; i < WIT; ++i)
      if ((i + 1) * 256 <= WUNITS || tid + 256 * i < WUNITS) Wl[tid + 256 * i] = wreg[i];
    const uint4* Xh = XR + uh * XS;
    if (mg) {   // uniform: the dense slice as one 16 x V x 32 matrix product per frame, frames dealt to the waves
      // three frames per pass, software-pipelined by hand: all transposed reads, then all MFMAs, then the stores (one frame at
      // a time is a dependent read -> MFMA -> convert -> store chain per frame: as slow as the vector gather it replaces)
      for (int f0 = wave; f0 < k.FT; f0 += 12) {
        bf16x8 afr[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int f = (f0 + 4 * i < k.FT) ? f0 + 4 * i : f0;   // a frame beyond the tile repeats frame f0 (its result is dropped)
          const unsigned ra = tr_addr + (unsigned)(f * V * 16);
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)ra);
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(ra + 64));
          const s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          uint4 am = *reinterpret_cast<const uint4*>(&av);
          am.x &= keep[0], am.y &= keep[1], am.z &= keep[2], am.w &= keep[3];
          afr[i] = *reinterpret_cast<const bf16x8*>(&am);
        }
        f32x4 z[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            z[i][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[i], bfr[nb], zero, 0, 0, 0);
          }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int f = f0 + 4 * i;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            if (f < k.FT && gi + 16 * nb < V)
              reinterpret_cast<uint2*>(Zl + zst_unit + f * V + 16 * nb)[gG & 1] =
                  make_uint2(cn8_pack2(z[i][nb][0], z[i][nb][1]), cn8_pack2(z[i][nb][2], z[i][nb][3]));
        }
      }
    }
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        if (mg && tp == DENSE) continue;
        uint4 zu;
        if (NZ[tp] == 1 && gwt[q][tp][0] == 1.0f) {
          zu = Xh[go[q][tp][0]];                       // a {(v, 1.0)} list: bf16(1.0 * x) = x
        } else {
          float z[8], x[8];
          cn8_unpack(Xh[go[q][tp][0]], x);
#pragma unroll
          for (int c = 0; c < 8; ++c) z[c] = gwt[q][tp][0] * x[c];
#pragma unroll
          for (int j = 1; j < 4; ++j)
            if (j < NZ[tp]) {
              cn8_unpack(Xh[go[q][tp][j]], x);
#pragma unroll
              for (int c = 0; c < 8; ++c) z[c] = fmaf(gwt[q][tp][j], x[c], z[c]);
            }
          zu = cn8_pack(z);
        }
        Zl[(tp * 2 + uh) * TN + (tid & 127) + 128 * q] = zu;
      }
    }
  };

  __syncthreads();   // rowp
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 bp = rowp[(wm * MS + ms) * 32 + mfma_row(r, hi)];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
        acc[ms][ns][r] = colok[ns] ? fmaf(bp.z, gcs[2][ns], fmaf(bp.y, gcs[1][ns], bp.x * gcs[0][ns])) : 0.f;
    }
  STAMP8P(3);
  const uint4* Wa = Wl + hi * BM + wm * MS * 32 + l31;
  const uint4* Za = Zl + hi * TN + wn * NS * 32 + l31;
  // (the reads of slice tp + 1 ahead of the MFMAs of slice tp -- the hand pipelining of conv_gemm_cn8_kernel -- needs 16 more
  // VGPRs: 172-184, i.e. two waves per SIMD, or 44-148 bytes of scratch at three; measured neutral, not kept)
  auto mma_phase = [&]() {
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
  };
  // the ADD epilogue's aux half units (skip-path / residual gradient of the data gradient) are loaded while the last stage
  // is built and multiplied, not inside the epilogue (conv_gemm_cn8_kernel)
  const bool pre_aux = d.epi == SAR_EPI_ADD;
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
  int c0 = 0;
  STAMP8(4);
  for (; c0 + KC16 < d.Kc; c0 += KC16) {
    store_raw();
    STAMP8(0);
    __syncthreads();   // raw image complete; every wave is past the MFMA phase of the previous stage
    STAMP8(1);
    build_units();
    STAMP8(2);
    __syncthreads();
    STAMP8(3);
    issue_loads(c0 + KC16);
    mma_phase();
    STAMP8(6);
  }
  store_raw();         // last stage (peeled: the aux registers take the place of the staging registers)
  STAMP8(0);
  __syncthreads();
  STAMP8(1);
  if (pre_aux) issue_aux();   // uniform
  build_units();
  STAMP8(2);
  __syncthreads();
  STAMP8(3);
  mma_phase();
  STAMP8(6);
  __syncthreads();   // the epilogue's transpose area aliases the image
  STAMP8(3);
  if (pre_aux) epilogue8<MS, NS, WN, BM, true>(k, tile, wm, wn, m0, vo, acc, rowp, smem, axr);
  else epilogue8<MS, NS, WN, BM, false>(k, tile, wm, wn, m0, vo, acc, rowp, smem);
#ifdef SAR_CN8_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  STAMP8(5);
  if (tid == 0 && blockIdx.x < STAMP_WG) {
    unsigned* row = g_stamps8[blockIdx.x];
    for (int i = 0; i < 6; ++i) row[i] = (unsigned)st_acc[i];
    row[6] = 1u;
    row[7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
    row[8] = (unsigned)(__builtin_amdgcn_s_memtime() - st_t0);
    row[9] = (unsigned)st_acc[6];
  }
#endif
}

template <int TR, int TAPS, int MS, int NS, int WM, int WN>
int launch_cfg8(const sar_conv_desc& d, const uint4* wp, hipStream_t st, int* nparts_only) {
  ConvK8 k;
  fill_common(d, wp, k);
  if (int g = tile_geometry8<WN>(d, NS, TR == 3, k)) {
    sar_set_error("sar_conv_gemm_cn8: unsupported tile geometry (V=%d, stride=%d)", d.V, d.stride);
    return g == -2 ? SAR_E_UNSUP : SAR_E_ARG;
  }
  if (nparts_only) {
    *nparts_only = k.nparts;
    return 0;
  }
  constexpr int BM = 32 * MS * WM;
  k.ntiles = d.B * k.TPS;
  k.ny = (d.M + BM - 1) / BM;
  const int nwork = k.ntiles * k.ny;
  hipLaunchKernelGGL((conv_gemm_cn8_kernel<TR, TAPS, MS, NS, WM, WN>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, st, k);
  return 0;
}

// the deep-prefetch kernel (9 taps; stride-1 forward, stride-1 / parity-split data gradients): experiment switch SAR_CN8_DB=1
bool db_enabled() {
  static const bool v = [] {
    const char* e = getenv("SAR_CN8_DB");
    return e && e[0] == '1';
  }();
  return v;
}

template <int TR, int MS, int NS, int WM, int WN>
int launch_db8(const sar_conv_desc& d, const uint4* wp, hipStream_t st, int* nparts_only) {
  ConvK8 k;
  fill_common(d, wp, k);
  if (int g = tile_geometry8<WN>(d, NS, TR == 3, k, 512)) {
    sar_set_error("sar_conv_gemm_cn8: unsupported tile geometry (V=%d, stride=%d)", d.V, d.stride);
    return g == -2 ? SAR_E_UNSUP : SAR_E_ARG;
  }
  if (nparts_only) {
    *nparts_only = k.nparts;
    return 0;
  }
  constexpr int BM = 32 * MS * WM;
  k.ntiles = d.B * k.TPS;
  k.ny = (d.M + BM - 1) / BM;
  const int nwork = k.ntiles * k.ny;
  hipLaunchKernelGGL((conv_gemm_cn8_db_kernel<TR, MS, NS, WM, WN>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, st, k);
  return 0;
}

template <int MS, int NS, int WM, int WN, int NZ0, int NZ1, int NZ2>
int launch_graph_cfg8(const sar_conv_desc& d, const uint4* wp, hipStream_t st, int* nparts_only) {
  ConvK8 k;
  fill_common(d, wp, k);
  if (int g = tile_geometry8<WN>(d, NS, false, k)) {
    sar_set_error("sar_conv_gemm_cn8: unsupported tile geometry (V=%d)", d.V);
    return g == -2 ? SAR_E_UNSUP : SAR_E_ARG;
  }
  if (nparts_only) {
    *nparts_only = k.nparts;
    return 0;
  }
  constexpr int BM = 32 * MS * WM;
  k.ntiles = d.B * k.TPS;
  k.ny = (d.M + BM - 1) / BM;
  const int nwork = k.ntiles * k.ny;
  hipLaunchKernelGGL((conv_graph_cn8_kernel<MS, NS, WM, WN, NZ0, NZ1, NZ2>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, st, k);
  return 0;
}

template <int NZ0, int NZ1, int NZ2>
int launch_graph_by_m8(const sar_conv_desc& d, const uint4* wp, hipStream_t st, int* np) {
  if (d.M > 64) return launch_graph_cfg8<2, 2, 2, 2, NZ0, NZ1, NZ2>(d, wp, st, np);
  if (d.M > 32) return launch_graph_cfg8<2, 2, 1, 4, NZ0, NZ1, NZ2>(d, wp, st, np);
  return launch_graph_cfg8<1, 2, 1, 4, NZ0, NZ1, NZ2>(d, wp, st, np);
}

// tile of the temporal / residual GEMMs for M > 64 (experiment switch SAR_CN8_TILE): 0 = 128 x 128 (7 % slower step),
// 1 = 128 x 256 (256 VGPRs + scratch: 7x slower, kept only as the measured counter-example), 2 = 64 x 256 row blocks
// (default; the tile of the M <= 64 layers), 3 = 64 x 128 (12 % slower), 4 = 64 x 512 for the stride-1 9-tap launches (wave
// tile 64 x 128, 0.75 LDS operand reads per MFMA, 254 VGPRs: 8-15 % faster per kernel in isolation -- 3.85 -> 3.54 ms per
// step for forward + data gradient -- but 2 waves per SIMD co-reside worse with the weight-gradient stream: step 1-3 % slower)
int tile_choice() {
  static const int v = [] {
    const char* e = getenv("SAR_CN8_TILE");
    return e ? atoi(e) : 2;
  }();
  return v;
}

template <int TR, int TAPS>
int launch_by_m8(const sar_conv_desc& d, const uint4* wp, hipStream_t st, int* np) {
  // M > 64: 128 x 256 tiles (wave tile 64 x 128): per output element the weight image -- re-read from L2 by every tile and
  // stage -- is fetched half as often as with 128 x 128 tiles (measured: SAR_CN8_TN128=1 restores the small tile)
  if constexpr (TAPS == 9 && (TR == 1 || TR == 3)) {
    const int rc = sar_cn8_dma_dispatch(TR, d, wp, st, np);   // LDS-DMA operand staging (SAR_CN8_DMA=1)
    if (rc != SAR_CN8_DMA_NOT_APPLICABLE) return rc;
  }
  if constexpr (TAPS == 9 && (TR == 0 || TR == 1 || TR == 3))
    if (d.M > 32 && (TR != 0 || d.stride == 1) && tile_choice() == 2 && db_enabled() && d.V * 19 <= 512)
      return launch_db8<TR, 2, 2, 1, 4>(d, wp, st, np);
  if constexpr (TR != 3)
    if (d.M > 64) {
      const int t = tile_choice();
      if (t == 1) return launch_cfg8<TR, TAPS, 2, 4, 2, 2>(d, wp, st, np);
      if (t == 0) return launch_cfg8<TR, TAPS, 2, 2, 2, 2>(d, wp, st, np);
    }
  if constexpr (TR != 3)
    if (d.M > 32 && tile_choice() == 3) return launch_cfg8<TR, TAPS, 2, 1, 1, 4>(d, wp, st, np);   // 64 x 128: 4 workgroups per CU
  if constexpr (TR == 0 || TR == 1)
    if (d.M > 32 && tile_choice() == 4 && d.stride == 1 && TAPS == 9)   // 64 x 512: wave tile 64 x 128 (0.75 LDS reads per MFMA)
      return launch_cfg8<TR, TAPS, 2, 4, 1, 4>(d, wp, st, np);
  if (d.M > 32) return launch_cfg8<TR, TAPS, 2, 2, 1, 4>(d, wp, st, np);
  return launch_cfg8<TR, TAPS, 1, 2, 1, 4>(d, wp, st, np);
}

int dispatch8(const sar_conv_desc& d, const uint4* wp, hipStream_t st, int* np) {
  if (d.mode == SAR_CONV_GRAPH) {
    const int rc2 = sar_graph2_cn8_dispatch(d, wp, st, np);   // gather at operand-read time (power-of-two gather weights)
    if (rc2 != SAR_GRAPH2_NOT_APPLICABLE) return rc2;
    if (d.epi == SAR_EPI_ADD_GATE) {
      sar_set_error("sar_conv_gemm_cn8: SAR_EPI_ADD_GATE needs the read-gather graph kernel (SAR_GRAPH_READ_GATHER=0 or an unsupported tile)");
      return SAR_E_UNSUP;
    }
    if (d.nz[0] == 1 && d.nz[1] == 1) return launch_graph_by_m8<1, 1, 4>(d, wp, st, np);
    if (d.nz[0] == 1 && d.nz[2] == 1) return launch_graph_by_m8<1, 4, 1>(d, wp, st, np);
    return launch_graph_by_m8<4, 4, 4>(d, wp, st, np);
  }
  if (!d.transposed) return d.taps == 9 ? launch_by_m8<0, 9>(d, wp, st, np) : launch_by_m8<0, 1>(d, wp, st, np);
  if (d.stride == 1) return d.taps == 9 ? launch_by_m8<1, 9>(d, wp, st, np) : launch_by_m8<1, 1>(d, wp, st, np);
  if (d.taps == 9) return d.stride == 2 ? launch_by_m8<3, 9>(d, wp, st, np) : launch_by_m8<2, 9>(d, wp, st, np);
  return launch_by_m8<2, 1>(d, wp, st, np);
}

int check8(const sar_conv_desc* d) {
  SAR_REQUIRE(d != nullptr, "sar_conv_gemm_cn8: null descriptor");
  SAR_REQUIRE(d->mode == SAR_CONV_TEMPORAL || d->mode == SAR_CONV_GRAPH, "sar_conv_gemm_cn8: bad mode %d", d->mode);
  SAR_REQUIRE(d->B > 0 && d->V > 0 && d->V <= 64 && d->T_src > 0 && d->T_out > 0 && d->Kc > 0 && d->M > 0,
              "sar_conv_gemm_cn8: bad sizes");
  if (d->mode == SAR_CONV_GRAPH) {
    SAR_REQUIRE(d->taps == 3 && d->T_src == d->T_out, "sar_conv_gemm_cn8: graph mode needs 3 adjacency slices and keeps T");
    for (int i = 0; i < 3; ++i)
      SAR_REQUIRE(d->nz[i] >= 1 && d->nz[i] <= 4, "sar_conv_gemm_cn8: adjacency slice %d needs %d gather entries (max 4)", i, d->nz[i]);
  } else {
    SAR_REQUIRE(d->taps == 9 || d->taps == 1, "sar_conv_gemm_cn8: temporal kernel size %d not built (1 and 9 are)", d->taps);
    SAR_REQUIRE(d->stride >= 1 && d->pad >= 0, "sar_conv_gemm_cn8: bad stride/pad");
  }
  SAR_REQUIRE(d->epi >= SAR_EPI_NONE && d->epi <= SAR_EPI_ADD_GATE, "sar_conv_gemm_cn8: bad epilogue %d", d->epi);
  if (d->epi == SAR_EPI_ADD_GATE)
    SAR_REQUIRE(d->mode == SAR_CONV_GRAPH && (d->g_flags & SAR_GRAPH_FEW_DENSE),
                "sar_conv_gemm_cn8: SAR_EPI_ADD_GATE is built for the graph data gradient with SAR_GRAPH_FEW_DENSE tables");
  return 0;
}

}  // namespace

#ifdef SAR_CN8_STAMPS
extern "C" int sar_debug_cn8_stamps(unsigned long long* out10, int reset) {
  static unsigned host[STAMP_WG][10];
  if (out10) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps8), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 10; ++i) out10[i] = 0;
    for (int w = 0; w < STAMP_WG; ++w)
      for (int i = 0; i < 10; ++i) out10[i] += host[w][i];
  }
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps8)) != hipSuccess || hipMemset(p, 0, sizeof(host)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

extern "C" int sar_conv_gemm_cn8_nparts(const sar_conv_desc* d) {
  if (int rc = check8(d)) return rc;
  int np = 0;
  const int rc = dispatch8(*d, nullptr, nullptr, &np);
  return rc ? rc : np;
}

extern "C" int sar_conv_gemm_cn8(const sar_conv_desc* d, const void* packed_w, sar_stream_t s) {
  if (int rc = check8(d)) return rc;
  SAR_REQUIRE(packed_w && ((uintptr_t)packed_w & 15) == 0, "sar_conv_gemm_cn8: packed weights must be 16-byte aligned");
  SAR_REQUIRE(d->src && d->out && (((uintptr_t)d->src | (uintptr_t)d->out | (uintptr_t)d->aux) & 15) == 0,
              "sar_conv_gemm_cn8: null / misaligned src, out or aux (CN8 tensors are 16-byte aligned)");
  if (d->mode == SAR_CONV_GRAPH) {
    SAR_REQUIRE(d->g_idx && d->g_wt, "sar_conv_gemm_cn8: graph gather tables required");
    SAR_REQUIRE(!d->bias || d->g_colsum, "sar_conv_gemm_cn8: graph bias needs g_colsum");
    if (d->pro_scale) {
      sar_set_error("sar_conv_gemm_cn8: a folded prologue is not built for graph mode");
      return SAR_E_UNSUP;
    }
  }
  SAR_REQUIRE(d->ld_src >= (int64_t)d->B * d->T_src * d->V && d->ld_out >= (int64_t)d->B * d->T_out * d->V,
              "sar_conv_gemm_cn8: leading dimension smaller than B*T*V");
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv_gemm_cn8: pro_scale/pro_shift mismatch");
  SAR_REQUIRE((int64_t)d->T_src * d->V < (1 << 26), "sar_conv_gemm_cn8: sequence row too long");
  SAR_REQUIRE(d->ld_out < (1 << 26) && d->ld_aux < (1 << 26), "sar_conv_gemm_cn8: leading dimension too large (2^26 columns)");
  SAR_REQUIRE((int64_t)d->B * d->T_out * d->V < (1 << 27), "sar_conv_gemm_cn8: more than 2^27 output columns");
  SAR_REQUIRE((int64_t)d->taps * 2 * ((d->Kc + 15) / 16) * d->M * 16 < (1ll << 31), "sar_conv_gemm_cn8: weight tensor too large");
  if (d->epi == SAR_EPI_STATS || d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD_GATE) SAR_REQUIRE(d->partials, "sar_conv_gemm_cn8: partials required");
  if (d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD || d->epi == SAR_EPI_ADD_GATE)
    SAR_REQUIRE(d->aux && d->ld_aux >= (int64_t)d->B * d->T_out * d->V, "sar_conv_gemm_cn8: aux required");
  if (d->epi == SAR_EPI_ADD_GATE) {
    SAR_REQUIRE(d->aux2 && d->aux_mask && d->aux_mean && d->partials && d->ld_aux2 >= (int64_t)d->B * d->T_out * d->V && d->ld_aux2 < (1 << 26),
                "sar_conv_gemm_cn8: SAR_EPI_ADD_GATE needs aux2, aux_mask, aux_mean, partials and ld_aux2 >= B*T*V");
    SAR_REQUIRE(((uintptr_t)d->aux2 & 15) == 0, "sar_conv_gemm_cn8: aux2 must be 16-byte aligned");
  }
  if (d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->aux_scale && d->aux_shift, "sar_conv_gemm_cn8: aux affine required");
  int rc = dispatch8(*d, (const uint4*)packed_w, as_stream(s), nullptr);
  if (rc) return rc;
  SAR_LAUNCH_CHECK("sar_conv_gemm_cn8");
  return 0;
}

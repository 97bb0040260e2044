// conv_wgrad.hip -- weight gradients of the fused graph / temporal convolutions (fp32 MFMA, gfx950).
//
//   dW[tap][c][m] = sum_n dout[m, n] * OP_tap(pro(src))[c, n]        (reduction over ALL positions)
//
// Replaces tape.gradient (main_gnn.py:233) for the GraphConvTD kernel (models/gcn.py:192-196), the
// 9x1 temporal conv (models/stgcn.py:29-36) and the strided 1x1 residual conv (models/stgcn.py:47-54).
//
// Design (MI355X): the MFMA reduction axis is the position axis.  A workgroup owns a (32*WF) x
// (32*WC) x taps block of the weight tensor and walks its share of the sequence-aligned position
// tiles (an EVEN number FT of whole frames): per tile it stages the dout rows and the src rows
// (+ temporal halo, BN+ReLU folded, zero padding materialised) in LDS with ODD row strides so that
// the per-lane-row operand reads are bank-conflict free.  Each v_mfma_f32_32x32x2_f32 reduces over
// the SAME joint v of two consecutive frames (k = 0 on lanes 0-31, k = 1 on lanes 32-63), so every
// LDS address is "per-lane constant + wave-uniform scalar": no position tables, no dependent reads,
// and the graph gather lists (<=4 entries per joint and adjacency slice) are wave-uniform and come
// through scalar loads.  One MFMA per tap per joint per frame pair.  Partial results go to
// per-split slabs (plain coalesced stores; summed in split order by sar_slab_reduce_f32), so the
// result is deterministic -- no float atomics.  MFMA orientation: A = src operand (rows c),
// B = dout (cols m), so the accumulator columns are m and slab stores are 128-B contiguous.
#include "sar_common.h"

namespace {

struct WgradK {
  sar_wgrad_desc d;
  int FT, FP, TPS, NT, NF, RW, SP, NPOS, NP, DP;
  int gy, gz;   // weight blocks along m and c (the launch is 1-D: see wg_decode)
};

// Workgroup id -> (split, m block, c block).  Ids are dispatched round-robin over the 8 XCDs (one L2 each): the
// gy*gz weight blocks that reduce the SAME tiles (same split) get adjacent slots of ONE XCD, so the tile data is
// fetched from HBM once per split instead of once per weight block.
struct WgId { int split, y, z; bool live; };
__device__ __forceinline__ WgId wg_decode(int id, int nsplit, int gy, int gz) {
  const int nyz = gy * gz, nwork = nsplit * nyz;
  const int per = (nwork + 7) / 8;
  const int xcd = id & 7, slot = id >> 3;
  const int w = xcd * per + slot;
  WgId r;
  r.live = slot < per && w < nwork;
  r.split = w / nyz;
  const int yz = w - r.split * nyz;
  r.z = yz / gy;
  r.y = yz - r.z * gy;
  return r;
}

// One tile of the temporal reduction for a wave that owns NT consecutive taps starting at Sbase's tap.
// No per-MFMA conditions: NT is a compile-time count, every address is lane-constant + uniform, and the
// operands of joint v+1 are read from LDS while the MFMAs of joint v issue (register double buffer).
template <int NT, int TPW>
__device__ __forceinline__ void temporal_tile(f32x16 (&acc)[TPW], float& bsum, const float* Dbase, const float* Sbase,
                                              int FP, int V, int sfs) {
  for (int fp = 0; fp < FP; ++fp) {
    const float* Dq = Dbase + fp * 2 * V;
    const float* Sq = Sbase + fp * 2 * sfs;
    float dn = Dq[0], sn[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) sn[i] = Sq[i * V];
    for (int v = 0; v < V; ++v) {
      const float dc = dn;
      float sc[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) sc[i] = sn[i];
      const int vn = (v + 1 < V) ? v + 1 : v;
      dn = Dq[vn];
#pragma unroll
      for (int i = 0; i < NT; ++i) sn[i] = Sq[i * V + vn];
      bsum += dc;
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(sc[i], dc, acc[i], 0, 0, 0);
    }
  }
}

// The same reduction with the geometry known at compile time (VC joints, FPC frame pairs per tile, conv stride
// STRIDE): fully unrolled, every LDS address is "per-lane base register + immediate", the loop carries no vector
// ALU work besides the optional bias sum (the fp32 MFMA shares the vector ALU with everything else: tools/mfma_fill.hip),
// and the operands of step s+1 are read while the MFMAs of step s issue.
template <int TPW, int VC, int STRIDE, int FPC, bool BIAS>
__device__ __forceinline__ void temporal_tile_fixed(f32x16 (&acc)[TPW], float& bsum, const float* Dbase, const float* Sbase) {
  constexpr int NSTEP = FPC * VC;
  auto fetch = [&](int st, float& dv, float (&sv)[TPW]) {
    const int fp = st / VC, v = st % VC;
    dv = Dbase[fp * 2 * VC + v];
#pragma unroll
    for (int i = 0; i < TPW; ++i) sv[i] = Sbase[fp * 2 * STRIDE * VC + i * VC + v];
  };
  auto mma = [&](float dv, const float (&sv)[TPW], bool have_next) {
    if (BIAS) bsum += dv;
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[i], dv, acc[i], 0, 0, 0);
    // issue order: MFMA, then the next step's reads spread behind the MFMAs
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
  fetch(0, d0, s0);
#pragma unroll
  for (int st = 0; st < NSTEP; st += 2) {
    if (st + 1 < NSTEP) fetch(st + 1, d1, s1);
    mma(d0, s0, st + 1 < NSTEP);
    if (st + 1 < NSTEP) {
      if (st + 2 < NSTEP) fetch(st + 2, d0, s0);
      mma(d1, s1, st + 2 < NSTEP);
    }
  }
}

// 9 taps on two wave groups WITHOUT a phantom tap slot: each group owns four whole taps (0-3 / 5-8) and HALF of the middle
// tap 4 -- its reduction over the tile's two frame pairs is split, group g takes frame pair g.  Every wave runs the same
// code: its "first" frame pair (pair g) with 5 MFMAs per step, the other one with 4 -- 225 MFMAs per tile instead of the
// 250 of the 2 x 5 slot scheme (one slot of which multiplied the zero padding).  The caller adds the two partial
// accumulators of tap 4 through LDS after the tile loop.  D1 / S1: this wave's first frame pair, D2 / S2: the second;
// Sf*: first of the four whole taps, Ssh1: tap 4.
template <int VC, bool BIAS>
__device__ __forceinline__ void temporal_tile_shared(f32x16 (&acc)[5], float& bsum, const float* D1, const float* D2,
                                                     const float* Sf1, const float* Sf2, const float* Ssh1) {
  constexpr int NSTEP = 2 * VC;
  auto fetch = [&](int st, float& dv, float (&sv)[5]) {
    const int half = st / VC, v = st % VC;
    dv = (half ? D2 : D1)[v];
#pragma unroll
    for (int i = 0; i < 4; ++i) sv[i] = (half ? Sf2 : Sf1)[i * VC + v];
    if (!half) sv[4] = Ssh1[v];
  };
  auto mma = [&](int st, float dv, const float (&sv)[5], bool have_next) {
    const int nm = (st / VC) ? 4 : 5;
    if (BIAS) bsum += dv;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (i < nm) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[i], dv, acc[i], 0, 0, 0);
    // issue order: MFMA, then the next step's reads spread behind the MFMAs (temporal_tile_fixed)
    int done = 0;
    const int rd = have_next ? 1 + (((st + 1) / VC) ? 4 : 5) : 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (i < nm) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        const int n = (rd - done + (nm - i) - 1) / (nm - i);
        if (n == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if (n == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        done += n;
      }
  };
  float d0, s0[5], d1, s1[5];
  fetch(0, d0, s0);
#pragma unroll
  for (int st = 0; st < NSTEP; st += 2) {
    if (st + 1 < NSTEP) fetch(st + 1, d1, s1);
    mma(st, d0, s0, st + 1 < NSTEP);
    if (st + 1 < NSTEP) {
      if (st + 2 < NSTEP) fetch(st + 2, d0, s0);
      mma(st + 1, d1, s1, st + 2 < NSTEP);
    }
  }
}

// Graph reduction: per joint v one packed LDS table row {gather joints[E], weights[E], colsum[3]} read with a
// wave-uniform address.  Three-stage register pipeline: table row of joint v+2, gathered x values of joint
// v+1 and the MFMAs of joint v are in flight together.
template <int NZ0, int NZ1, int NZ2>
struct GraphTab {
  static constexpr int E = NZ0 + NZ1 + NZ2;
  static constexpr int ROW = (2 * E + 3 + 3) / 4 * 4;
};

template <int NZ0, int NZ1, int NZ2>
__device__ __forceinline__ void graph_tile(f32x16 (&acc)[3], float (&bsum)[3], const float* Dbase, const float* Sbase,
                                           const float* T, int FP, int V) {
  using GT = GraphTab<NZ0, NZ1, NZ2>;
  constexpr int E = GT::E, ROW = GT::ROW, Q = ROW / 4;
  constexpr int NZ[3] = {NZ0, NZ1, NZ2};
  auto load_row = [&](float4 (&r)[Q], int v) {
#pragma unroll
    for (int q = 0; q < Q; ++q) r[q] = reinterpret_cast<const float4*>(T + v * ROW)[q];
  };
  auto elem = [&](const float4 (&r)[Q], int i) -> float {
    const float4 x = r[i >> 2];
    return (i & 3) == 0 ? x.x : (i & 3) == 1 ? x.y : (i & 3) == 2 ? x.z : x.w;
  };
  for (int fp = 0; fp < FP; ++fp) {
    const float* Dq = Dbase + fp * 2 * V;
    const float* Sq = Sbase + fp * 2 * V;
    float4 tc[Q], tn[Q];          // table rows of joint v and v+1
    float xn[E], dn;              // gathered x values / dout value of joint v (loaded one iteration early)
    load_row(tc, 0);
    load_row(tn, V > 1 ? 1 : 0);
#pragma unroll
    for (int e = 0; e < E; ++e) xn[e] = Sq[__float_as_int(elem(tc, e))];
    dn = Dq[0];
    for (int v = 0; v < V; ++v) {
      float xc[E];
#pragma unroll
      for (int e = 0; e < E; ++e) xc[e] = xn[e];
      const float dc = dn;
      float4 tcur[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) tcur[q] = tc[q];
      // stage joint v+1 (its table row is already in tn) and fetch the row of joint v+2
      const int v1 = (v + 1 < V) ? v + 1 : v;
      const int v2 = (v + 2 < V) ? v + 2 : V - 1;
#pragma unroll
      for (int e = 0; e < E; ++e) xn[e] = Sq[__float_as_int(elem(tn, e))];
      dn = Dq[v1];
#pragma unroll
      for (int q = 0; q < Q; ++q) tc[q] = tn[q];
      load_row(tn, v2);
      int e0 = 0;
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        float sval = elem(tcur, E + e0) * xc[e0];
#pragma unroll
        for (int j = 1; j < 4; ++j)
          if (j < NZ[tp]) sval = fmaf(elem(tcur, E + e0 + j), xc[e0 + j], sval);
        bsum[tp] = fmaf(dc, elem(tcur, 2 * E + tp), bsum[tp]);
        acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(sval, dc, acc[tp], 0, 0, 0);
        e0 += NZ[tp];
      }
    }
  }
}

// VC / STRIDEC != 0: V, the conv stride and FP == 2 frame pairs per tile are compile-time (NTU: V = 25)
template <int MODE, int TAPS, int WF, int WC, int WT, int TPW, int NZ0, int NZ1, int NZ2, int VC, int STRIDEC>
__global__ __launch_bounds__(64 * WF * WC * WT, (WF * WC * WT == 6) ? 3 : 2) void conv_wgrad_kernel(const WgradK k) {
  // NW waves per workgroup: 4, or 6 for the 9-tap kernel (2 m-blocks x 3 tap groups of 3: no phantom tap slot)
  constexpr int NW = WF * WC * WT, NHW = 2 * NW, NTH = 64 * NW;
  static_assert(NW == 4 || NW == 6, "4 or 6 waves per workgroup");
#ifndef SAR_WGRAD_SHARED_TAP
#define SAR_WGRAD_SHARED_TAP 1   // 0: the 2 x 5 tap-slot scheme with one phantom slot (A/B builds)
#endif
  // 9 taps on 2 wave groups at fixed geometry: no phantom tap slot (temporal_tile_shared)
  constexpr bool SHARED_TAP = SAR_WGRAD_SHARED_TAP && MODE == SAR_CONV_TEMPORAL && TAPS == 9 && WT == 2 && TPW == 5 && WC == 1 && VC != 0;
  static_assert(WT * TPW >= TAPS, "taps must be covered");
  constexpr int BF = 32 * WF, CT = 32 * WC;
  constexpr int NZMAX = 4;
  // staging maps: a half-wave owns one row and reads 32 consecutive columns per pass (128-B segments)
  constexpr int DI = (BF + NHW - 1) / NHW, DJ = 4;             // dout tile: BF rows x <=128 positions
  constexpr int SI = (CT + NHW - 1) / NHW;                     // src tile: CT rows x RW columns
  constexpr int SJMAX = (MODE == SAR_CONV_GRAPH) ? 4 : (TAPS == 1 ? 6 : 12);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const sar_wgrad_desc& d = k.d;
  const int V = d.V;
  float* D = smem;                          // [BF][DP]
  // LDS row strides: compile-time in the fixed-geometry variants (every LDS address is then register + immediate;
  // the S rows are as wide as the column passes of the stager, so its stores need no lane mask)
  const int DP = (VC != 0) ? (4 * VC) | 1 : k.DP;
  const int SP = (VC != 0) ? 32 * SJMAX + 1 : k.SP;
  float* S = D + BF * DP;                   // [CT][SP]
  float* T = S + CT * SP + ((4 - ((BF * DP + CT * SP) & 3)) & 3);   // GRAPH: [V][ROW] packed gather rows (16-B aligned)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction
  const int l31 = lane & 31, hi = lane >> 5;
  const int wf = wave % WF, wc = (wave / WF) % WC, wt = wave / (WF * WC);
  const WgId wg = wg_decode(blockIdx.x, d.nsplit, k.gy, k.gz);
  if (!wg.live) return;   // padded tail of the id space (whole workgroup, before any barrier)
  const int f0 = wg.y * BF, c0 = wg.z * CT;

  if (MODE == SAR_CONV_GRAPH) {   // pack {joints[E], weights[E], colsum[3]} per joint
    using GT = GraphTab<NZ0, NZ1, NZ2>;
    constexpr int NZc[3] = {NZ0, NZ1, NZ2};
    for (int v = tid; v < V; v += NTH) {
      int e = 0;
      for (int tp = 0; tp < 3; ++tp) {
        for (int j = 0; j < NZc[tp]; ++j, ++e) {
          T[v * GT::ROW + e] = __int_as_float(d.g_idx[(tp * V + v) * NZMAX + j]);
          T[v * GT::ROW + GT::E + e] = d.g_wt[(tp * V + v) * NZMAX + j];
        }
        T[v * GT::ROW + 2 * GT::E + tp] = d.g_colsum ? d.g_colsum[tp * V + v] : 0.f;
      }
    }
  }

  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float bsum[3] = {0.f, 0.f, 0.f};

  const int seq_src = d.T_src * V, seq_out = d.T_out * V;
  const bool has_pro = d.pro_scale != nullptr;
  const bool do_bias = (wc == 0) && (MODE == SAR_CONV_GRAPH || wt == 0) && wg.z == 0;

  // ---- register prefetch of the next tile (unconditional loads from clamped addresses; predicates are
  // applied when the registers are written to LDS, see conv_gemm.hip)
  const int r8 = tid >> 5, c32 = tid & 31;
  float dreg[DI][DJ];
  float sreg[SI][SJMAX];
  float psc[SI], psh[SI];
#pragma unroll
  for (int i = 0; i < SI; ++i) {
    const int cg = c0 + r8 + NHW * i;
    psc[i] = (has_pro && cg < d.Kc) ? d.pro_scale[cg] : 1.f;
    psh[i] = (has_pro && cg < d.Kc) ? d.pro_shift[cg] : 0.f;
  }

  // Tile classes (uniform over the workgroup).  An INTERIOR tile lies inside its sequence with every row valid:
  // its loads are buffer loads whose row / column-pass parts are a scalar offset and an immediate (one per-lane
  // offset register per tensor, no address or predicate VALU -- the fp32 MFMA shares the vector ALU), and its
  // LDS stores need the folded BN+ReLU only.  EDGE tiles (temporal zero padding, ragged last tile, M / Kc
  // tails) take the clamped-address + predicate path.
  const bool rows_full = (f0 + BF <= d.M) && (c0 + CT <= d.Kc);
  const unsigned dvo = (unsigned)(((int64_t)hi * d.ld_dout + c32) * 4);   // hi = second row of the wave's row pair
  const unsigned svo = (unsigned)(((int64_t)hi * d.ld_src + c32) * 4);
  bool dlane[DJ], slane[SJMAX];   // tile-invariant lane masks of the column passes (stay inside the LDS row)
#pragma unroll
  for (int j = 0; j < DJ; ++j) dlane[j] = (VC != 0 && j < 3) || c32 + 32 * j < k.NP;
#pragma unroll
  for (int j = 0; j < SJMAX; ++j) slane[j] = (VC != 0) || c32 + 32 * j < SP;
  auto bytes_to_end = [](int64_t elems) { return (unsigned)(elems <= 0 ? 0 : elems * 4 > 0xFFFFFFFFll ? 0xFFFFFFFFll : elems * 4); };
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  auto is_edge = [&](int t0, int t_lo) {
    return !rows_full || t0 * V + k.NPOS > seq_out || t_lo < 0 || t_lo * V + k.RW > seq_src;
  };

  // Loads are the same for every tile: buffer loads from the start of the sequence row; the tile offset is added
  // to the per-lane offset (one v_add per tensor and tile).  What a tile must not see reads as 0 or is masked at
  // store time: a negative offset (temporal padding before the sequence) and anything past the end of the
  // tensor fail the range check (which sees voffset + soffset + imm: tools/buffer_range.hip) and return 0;
  // positions past the end of the sequence and rows past M / Kc are zeroed by the edge-tile store path.
  auto issue_loads = [&](int tile) {
    const int b = tile / k.TPS;
    const int t0 = (tile - b * k.TPS) * k.FT;
    const int t_lo = (MODE == SAR_CONV_GRAPH) ? t0 : t0 * d.stride - d.pad;
    // the tile offset goes into the (scalar) base address; a tile that starts in the temporal padding before
    // the sequence (t_lo < 0, first tile only) keeps the base at the sequence start and shifts the per-lane
    // offsets instead: a negative offset must reach the range check as ONE out-of-range value (the hardware
    // does not wrap voffset + imm), so it is resolved in the vector ALU there
    const int t_base = t_lo > 0 ? t_lo : 0;
    const int64_t do_ = (int64_t)b * seq_out + t0 * V, so_ = (int64_t)b * seq_src + t_base * V;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(d.dout + (int64_t)(f0 + 2 * wave) * d.ld_dout + do_), 0,
        bytes_to_end((int64_t)(d.M - f0 - 2 * wave) * d.ld_dout - do_), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(d.src + (int64_t)(c0 + 2 * wave) * d.ld_src + so_), 0,
        bytes_to_end((int64_t)(d.Kc - c0 - 2 * wave) * d.ld_src - so_), 0x00020000);
    const int drow = (int)(d.ld_dout * 4 * NHW), srow = (int)(d.ld_src * 4 * NHW);   // NHW rows, bytes
    // (rows past the tile in the last row pass -- BF or CT not a multiple of NHW -- are skipped: wave-uniform)
#pragma unroll
    for (int i = 0; i < DI; ++i)
      if (BF % NHW == 0 || 2 * wave + NHW * i < BF) {
#pragma unroll
        for (int j = 0; j < DJ; ++j)
          dreg[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dvo + 128 * j, i * drow, 0));
      }
    if (t_lo >= 0) {
#pragma unroll
      for (int i = 0; i < SI; ++i)
        if (CT % NHW == 0 || 2 * wave + NHW * i < CT) {
#pragma unroll
          for (int j = 0; j < SJMAX; ++j)
            sreg[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo + 128 * j, i * srow, 0));
        }
    } else {
      const int shift = t_lo * V * 4;
#pragma unroll
      for (int j = 0; j < SJMAX; ++j) {
        const int o = (int)(c32 * 4) + 128 * j + shift;                    // offset inside the row
        const unsigned vo = o < 0 ? 0x80000000u : svo + (unsigned)(128 * j + shift);
#pragma unroll
        for (int i = 0; i < SI; ++i)
          if (CT % NHW == 0 || 2 * wave + NHW * i < CT) sreg[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, i * srow, 0));
      }
    }
  };

  auto store_lds = [&](int tile) {
    const int b = tile / k.TPS;
    const int t0 = (tile - b * k.TPS) * k.FT;
    const int t_lo = (MODE == SAR_CONV_GRAPH) ? t0 : t0 * d.stride - d.pad;
    if (!is_edge(t0, t_lo)) {
      float* Dw = D + r8 * DP + c32;
      float* Sw = S + r8 * SP + c32;
#pragma unroll
      for (int i = 0; i < DI; ++i)
        if (BF % NHW == 0 || 2 * wave + NHW * i < BF) {
#pragma unroll
          for (int j = 0; j < DJ; ++j)
            if (dlane[j]) Dw[i * NHW * DP + 32 * j] = dreg[i][j];
        }
#pragma unroll
      for (int i = 0; i < SI; ++i)
        if (CT % NHW == 0 || 2 * wave + NHW * i < CT) {
#pragma unroll
          for (int j = 0; j < SJMAX; ++j)   // columns in [RW, SP) only feed the phantom taps
            if (slane[j]) Sw[i * NHW * SP + 32 * j] = fmaxf(fmaf(sreg[i][j], psc[i], psh[i]), relu_lo);
        }
      return;
    }
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      const int fr = r8 + NHW * i;
#pragma unroll
      for (int j = 0; j < DJ; ++j) {
        const int p = c32 + 32 * j;
        const bool ok = (f0 + fr) < d.M && p < k.NPOS && (t0 * V + p) < seq_out;
        if (p < k.NP && (BF % NHW == 0 || fr < BF)) D[fr * DP + p] = ok ? dreg[i][j] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < SI; ++i) {
      const int cr = r8 + NHW * i;
#pragma unroll
      for (int j = 0; j < SJMAX; ++j) {
        const int col = c32 + 32 * j;
        const int rabs = t_lo * V + col;
        if (col < k.RW && (CT % NHW == 0 || cr < CT)) {
          const float val = fmaxf(fmaf(sreg[i][j], psc[i], psh[i]), relu_lo);
          S[cr * SP + col] = ((c0 + cr) < d.Kc && (unsigned)rabs < (unsigned)seq_src) ? val : 0.f;
        }
      }
    }
  };

  // per-lane operand bases: lanes 0-31 reduce over the even frame of a pair, lanes 32-63 over the odd one
  const int sfs = (MODE == SAR_CONV_GRAPH) ? V : d.stride * V;      // src columns per output frame
  const float* Dbase = D + (wf * 32 + l31) * DP + hi * V;
  const float* Sbase = S + (wc * 32 + l31) * SP + hi * sfs;

#ifndef SAR_ABLATE
#define SAR_ABLATE 0   // diagnostic builds only (tools/ablate.sh): 2 = stage the first tile only
#endif
  int tile = wg.split;
  if (tile < k.NT) issue_loads(tile);
  for (; tile < k.NT; tile += d.nsplit) {
    __syncthreads();  // previous tile's LDS reads done
    if (!(SAR_ABLATE & 2) || tile == wg.split) store_lds(tile);
    __syncthreads();
    if (!(SAR_ABLATE & 2) && tile + d.nsplit < k.NT) issue_loads(tile + d.nsplit);

    if constexpr (MODE == SAR_CONV_TEMPORAL) {
      // every wave runs TPW taps: taps beyond TAPS (last wave class when WT*TPW > TAPS) read the padded
      // tail of the S rows and land in accumulators that are never stored -- those SIMDs would otherwise
      // idle at the tile barrier, and one code path keeps the accumulators in AGPRs.
      if constexpr (SHARED_TAP) {
        // wave group wt: whole taps 5 wt .. 5 wt + 3 and frame pair wt of tap 4 (temporal_tile_shared)
        const int fa = wt * 2 * VC, fb = (1 - wt) * 2 * VC;
        temporal_tile_shared<VC, true>(acc, bsum[0], Dbase + fa, Dbase + fb, Sbase + fa * STRIDEC + wt * 5 * VC,
                                       Sbase + fb * STRIDEC + wt * 5 * VC, Sbase + fa * STRIDEC + 4 * VC);
      } else if constexpr (VC != 0) {
        // (one code path for every wave: a do_bias / no-bias pair of unrolled loops makes the register allocator
        // keep two copies of the accumulators)
        temporal_tile_fixed<TPW, VC, STRIDEC, 2, true>(acc, bsum[0], Dbase, Sbase + wt * TPW * VC);
      } else {
        temporal_tile<TPW, TPW>(acc, bsum[0], Dbase, Sbase + wt * TPW * V, k.FP, V, sfs);
      }
    } else {
      graph_tile<NZ0, NZ1, NZ2>(acc, bsum, Dbase, Sbase, T, k.FP, V);
    }
  }

  if constexpr (SHARED_TAP) {   // tap 4: group 1's half of the reduction is added to group 0's (through LDS, fixed order)
    __syncthreads();             // every wave is done with the last tile's operands
    float* X = smem + wf * (16 * 64);
    if (wt == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) X[r * 64 + lane] = acc[4][r];
    }
    __syncthreads();
    if (wt == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[4][r] += X[r * 64 + lane];
    }
  }
  // ---- write this split's slab
  float* slab = d.slab + (int64_t)wg.split * (d.wsize + d.bsize);
  const int f = f0 + wf * 32 + l31;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tp = SHARED_TAP ? (wt ? 5 + i : i) : wt * TPW + i;
    if (tp < TAPS && !(SHARED_TAP && wt == 1 && i == 4)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wc * 32 + mfma_row(r, hi);
        if (c < d.Kc && f < d.M) slab[(int64_t)tp * d.w_stride_tap + (int64_t)c * d.w_stride_c + f] = acc[i][r];
      }
    }
  }
  if (d.bsize > 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float t = bsum[i] + __shfl_xor(bsum[i], 32);
      const bool mine = (MODE == SAR_CONV_TEMPORAL) ? (i == 0) : (i < TAPS);
      if (mine && do_bias && hi == 0 && f < d.M) slab[d.wsize + (int64_t)i * d.M + f] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Graph weight gradient, fixed geometry (V = 25 joints, tiles of 2 frames = one frame pair).
//
// The gather  z_k[c, (t,v)] = sum_j w_kj(v) x[c, (t, idx_kj(v))]  is applied ONCE per tile while the tile is
// staged (LDS -> LDS, by the half-wave that stored the row: no extra barrier), not per MFMA operand by every
// wave: the MFMA loop then reads plain operands with immediate offsets (3 z + MB dout reads per 3 MB MFMAs, no
// vector-ALU work -- the fp32 MFMA shares the vector ALU).  Workgroup = (64 MB) m x 64 c x 3 slices, waves =
// 2 (m) x 2 (c), a wave owns MB x 3 accumulator tiles.  LDS: D (64 MB) x 51, Z 3 x 64 x 51 (Z[0] holds the raw
// x rows until the transform overwrites them in place).
#ifdef SAR_ABLATE
#define SAR_ABLATE_G SAR_ABLATE
#else
#define SAR_ABLATE_G 0   // diagnostic builds only (tools/ablate.sh): 2 = stage the first tile only, 8 = no gather transform
#endif
template <int NZ0, int NZ1, int NZ2, int MB>
__global__ __launch_bounds__(256, 2) void graph_wgrad_fixed_kernel(const WgradK k) {
  constexpr int VC = 25, NPOS = 50, BF = 64 * MB, CT = 64, DP = 51, ZP = 51, ZS = CT * ZP;
  constexpr int NZMAX = 4, E = NZ0 + NZ1 + NZ2;
  constexpr int NZ[3] = {NZ0, NZ1, NZ2};
  constexpr int DI = BF / 8, XI = CT / 8;   // rows per half-wave; two column passes (32 + 18 lanes)
  __shared__ float D[BF * DP];
  __shared__ float Z[3 * ZS];
  __shared__ float CS[3 * VC];   // colsum(A_k)[v] for the bias sums
  const sar_wgrad_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wf = wave & 1, wc = wave >> 1;
  const int r8 = tid >> 5, c32 = tid & 31;
  const WgId wg = wg_decode(blockIdx.x, d.nsplit, k.gy, k.gz);
  if (!wg.live) return;
  const int f0 = wg.y * BF, c0 = wg.z * CT;
  // the bias sums (sum_n dout * colsum(A_k)) are needed once per m: the c-block 0 waves of the z == 0 slice carry them
  const bool do_bias = d.bsize > 0 && wg.z == 0 && wc == 0;
  const int seq = d.T_src * VC;   // T_src == T_out

  if (tid < 3 * VC) CS[tid] = d.g_colsum ? d.g_colsum[tid] : 0.f;   // visible after the first tile barrier
  // gather lists of joint v = c32 (transform lanes): source pointers into this half-wave's first row, weights
  const int vq = c32 < VC ? c32 : 0;
  unsigned gp[E];   // LDS byte addresses (the low 32 bits of a shared-memory pointer are its LDS offset)
  float gwt[E];
  {
    int e = 0;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int j = 0; j < NZMAX; ++j)
        if (j < NZ[tp]) {
          gp[e] = (unsigned)(uintptr_t)(Z + r8 * ZP + d.g_idx[(tp * VC + vq) * NZMAX + j]);
          gwt[e] = d.g_wt[(tp * VC + vq) * NZMAX + j];
          ++e;
        }
  }

  f32x16 acc[MB][3];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;
  float bsum[MB][3];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) bsum[mb][0] = bsum[mb][1] = bsum[mb][2] = 0.f;

  const bool rows_full = (f0 + BF <= d.M) && (c0 + CT <= d.Kc);
  const unsigned dvo = (unsigned)(((int64_t)hi * d.ld_dout + c32) * 4);
  const unsigned svo = (unsigned)(((int64_t)hi * d.ld_src + c32) * 4);
  const bool lane1 = c32 + 32 < NPOS;   // the second column pass is partial (18 lanes)
  auto bytes_to_end = [](int64_t elems) { return (unsigned)(elems <= 0 ? 0 : elems * 4 > 0xFFFFFFFFll ? 0xFFFFFFFFll : elems * 4); };
  float dreg[DI][2], xreg[XI][2];

  // same loads for every tile (see conv_wgrad_kernel): what lies past the end of the tensor reads as 0, positions
  // past the end of the sequence and rows past M / Kc are zeroed by the edge-tile store path
  auto issue_loads = [&](int tile) {
    const int b = tile / k.TPS;
    const int t0 = (tile - b * k.TPS) * 2;
    const int64_t o = (int64_t)b * seq + t0 * VC;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(d.dout + (int64_t)(f0 + 2 * wave) * d.ld_dout + o), 0,
        bytes_to_end((int64_t)(d.M - f0 - 2 * wave) * d.ld_dout - o), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(d.src + (int64_t)(c0 + 2 * wave) * d.ld_src + o), 0,
        bytes_to_end((int64_t)(d.Kc - c0 - 2 * wave) * d.ld_src - o), 0x00020000);
    const int drow = (int)(d.ld_dout * 32), srow = (int)(d.ld_src * 32);   // 8 rows, bytes
#pragma unroll
    for (int i = 0; i < DI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        dreg[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dvo + 128 * j, i * drow, 0));
#pragma unroll
    for (int i = 0; i < XI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        xreg[i][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo + 128 * j, i * srow, 0));
  };

  auto store_lds = [&](int tile) {
    const int b = tile / k.TPS;
    const int t0 = (tile - b * k.TPS) * 2;
    float* Dw = D + r8 * DP + c32;
    float* Xw = Z + r8 * ZP + c32;
    if (!(rows_full && t0 * VC + NPOS <= seq)) {   // edge tile: rows >= M / Kc and positions past the sequence are 0
#pragma unroll
      for (int i = 0; i < DI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          dreg[i][j] = ((f0 + r8 + 8 * i) < d.M && t0 * VC + c32 + 32 * j < seq) ? dreg[i][j] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < DI; ++i) {
      Dw[i * 8 * DP] = dreg[i][0];
      if (lane1) Dw[i * 8 * DP + 32] = dreg[i][1];
    }
    if (!(rows_full && t0 * VC + NPOS <= seq)) {
#pragma unroll
      for (int i = 0; i < XI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          xreg[i][j] = ((c0 + r8 + 8 * i) < d.Kc && t0 * VC + c32 + 32 * j < seq) ? xreg[i][j] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      Xw[i * 8 * ZP] = xreg[i][0];
      if (lane1) Xw[i * 8 * ZP + 32] = xreg[i][1];
    }
    // in-place gather transform of the rows this half-wave just stored (lane = joint v).  Software-pipelined
    // over the rows: the gathered values of row i+1 are read before the results of row i are written
    // (different rows never overlap, so the in-place update is safe).
    __builtin_amdgcn_wave_barrier();
    if (c32 < VC && !(SAR_ABLATE_G & 8)) {
      const unsigned Zw = (unsigned)(uintptr_t)(Z + r8 * ZP + c32);
      float xs[2][E];
      // ds_read_b32 with a 16-bit immediate: written as asm because the compiler pairs the gathers into
      // ds_read2_b32 (8-bit offsets) and then spends one v_add per pair on new base addresses -- vector-ALU
      // work the MFMAs of the co-resident workgroup pay for.  The waits are explicit (counted lgkmcnt).
      // Order per row: wait for the row's gathers, compute its 6 results, issue the next row's gathers into
      // the same registers, then write the results (the writes queue behind the gathers in the LDS pipe).
      auto gather = [&](int i) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < E; ++e)
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(xs[t][e]) : "v"(gp[e]), "n"((i * 8 * ZP + t * VC) * 4));
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this half-wave's x rows are in LDS
      gather(0);
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        // the wait carries the gathered registers as in/out operands: their uses cannot be scheduled above it
        static_assert(E == 6, "wait operand list below");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xs[0][0]), "+v"(xs[0][1]), "+v"(xs[0][2]), "+v"(xs[0][3]), "+v"(xs[0][4]), "+v"(xs[0][5]),
                       "+v"(xs[1][0]), "+v"(xs[1][1]), "+v"(xs[1][2]), "+v"(xs[1][3]), "+v"(xs[1][4]), "+v"(xs[1][5])
                     :
                     : "memory");
        float zk[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          int e0 = 0;
#pragma unroll
          for (int tp = 0; tp < 3; ++tp) {
            float zz = gwt[e0] * xs[t][e0];
#pragma unroll
            for (int j = 1; j < NZMAX; ++j)
              if (j < NZ[tp]) zz = fmaf(gwt[e0 + j], xs[t][e0 + j], zz);
            zk[t][tp] = zz;
            e0 += NZ[tp];
          }
        }
        // the results are pinned before the gathers may overwrite their inputs
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int tp = 0; tp < 3; ++tp) asm volatile("" : "+v"(zk[t][tp]));
        if (i + 1 < XI) gather(i + 1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int tp = 0; tp < 3; ++tp)
            asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(Zw), "v"(zk[t][tp]), "n"((tp * ZS + i * 8 * ZP + t * VC) * 4) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };

  const float* Dq = D + (wf * 32 * MB + l31) * DP + hi * VC;
  const float* Zq = Z + (wc * 32 + l31) * ZP + hi * VC;
  auto fetch = [&](int v, float (&dv)[MB], float (&zv)[3]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) dv[mb] = Dq[mb * 32 * DP + v];
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) zv[tp] = Zq[tp * ZS + v];
  };
  auto mma = [&](int v, const float (&dv)[MB], const float (&zv)[3], bool have_next) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int tp = 0; tp < 3; ++tp)
        acc[mb][tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(zv[tp], dv[mb], acc[mb][tp], 0, 0, 0);
    // issue order: the next step's 3 + MB reads spread behind this step's MFMAs
    constexpr int NM = 3 * MB, RD = 3 + MB;
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

  int tile = wg.split;
  if (tile < k.NT) issue_loads(tile);
  for (; tile < k.NT; tile += d.nsplit) {
    __syncthreads();   // previous tile's operand reads done
    if (!(SAR_ABLATE_G & 2) || tile == wg.split) store_lds(tile);
    __syncthreads();
    if (!(SAR_ABLATE_G & 2) && tile + d.nsplit < k.NT) issue_loads(tile + d.nsplit);
    float d0[MB], z0[3], d1[MB], z1[3];
    fetch(0, d0, z0);
#pragma unroll
    for (int v = 0; v < VC; v += 2) {
      if (v + 1 < VC) fetch(v + 1, d1, z1);
      mma(v, d0, z0, v + 1 < VC);
      if (v + 1 < VC) {
        if (v + 2 < VC) fetch(v + 2, d0, z0);
        mma(v + 1, d1, z1, v + 2 < VC);
      }
    }
    if (do_bias) {   // wave-uniform; kept out of the MFMA loop so that its issue order and registers stay untouched
#pragma unroll
      for (int v = 0; v < VC; ++v)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float dv = Dq[mb * 32 * DP + v];
#pragma unroll
          for (int tp = 0; tp < 3; ++tp) bsum[mb][tp] = fmaf(dv, CS[tp * VC + v], bsum[mb][tp]);
        }
    }
  }

  // ---- write this split's slab
  float* slab = d.slab + (int64_t)wg.split * (d.wsize + d.bsize);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int f = f0 + (wf * MB + mb) * 32 + l31;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wc * 32 + mfma_row(r, hi);
        if (c < d.Kc && f < d.M) slab[(int64_t)tp * d.w_stride_tap + (int64_t)c * d.w_stride_c + f] = acc[mb][tp][r];
      }
    if (do_bias) {
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        const float t = bsum[mb][tp] + __shfl_xor(bsum[mb][tp], 32);
        if (hi == 0 && f < d.M) slab[d.wsize + (int64_t)tp * d.M + f] = t;
      }
    }
  }
}

// out[i] = sum over the nsplit slabs, in a fixed order (deterministic): wave g of a workgroup adds the slabs
// g, g+4, g+8, ... for 64 consecutive outputs (4 loads in flight per lane), then the four partial sums are
// combined in wave order through LDS.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nsplit,
                                                          int64_t slab_stride, int64_t n, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    const float* p = slab + i;
    int k = g;
    for (; k + 12 < nsplit; k += 16) {
      s0 += p[(int64_t)k * slab_stride];
      s1 += p[(int64_t)(k + 4) * slab_stride];
      s2 += p[(int64_t)(k + 8) * slab_stride];
      s3 += p[(int64_t)(k + 12) * slab_stride];
    }
    for (; k < nsplit; k += 4) s0 += p[(int64_t)k * slab_stride];
  }
  part[g][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && i < n) out[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// the same sums for a batch of slab sets: blockIdx.y = item (device table), blockIdx.x = 64 outputs of it; per element the
// additions of slab_reduce_kernel in the same order (bit-identical)
__global__ __launch_bounds__(256) void slab_reduce_batch_kernel(const sar_slab_item* __restrict__ items) {
  __shared__ float part[4][64];
  const sar_slab_item it = items[blockIdx.y];
  if ((int64_t)blockIdx.x * 64 >= it.n) return;   // uniform
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const int nsplit = it.nsplit;
  const int64_t slab_stride = it.slab_stride;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < it.n) {
    const float* p = it.slab + i;
    int k = g;
    for (; k + 12 < nsplit; k += 16) {
      s0 += p[(int64_t)k * slab_stride];
      s1 += p[(int64_t)(k + 4) * slab_stride];
      s2 += p[(int64_t)(k + 8) * slab_stride];
      s3 += p[(int64_t)(k + 12) * slab_stride];
    }
    for (; k < nsplit; k += 4) s0 += p[(int64_t)k * slab_stride];
  }
  part[g][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && i < it.n) it.out[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

size_t lds_bytes(const sar_wgrad_desc& d, const WgradK& k, int BF, int CT) {
  size_t lds = sizeof(float) * ((size_t)BF * k.DP + (size_t)CT * k.SP);
  if (d.mode == SAR_CONV_GRAPH) lds += sizeof(float) * (4 + (size_t)d.V * 28);   // packed gather rows (ROW <= 28)
  return lds;
}

// Frame tile: an even number of whole frames (frame PAIRS feed the two MFMA k-lanes) with at most 128
// positions, shrunk until two workgroups fit the CU's LDS.
int geometry(const sar_wgrad_desc& d, WgradK& k, int BF, int CT, int extra_taps) {
  int ft = (128 / d.V) & ~1;
  if (ft < 2) return -1;
  const int t_even = (d.T_out + 1) & ~1;
  if (ft > t_even) ft = t_even;
  for (;; ft -= 2) {
    k.FT = ft;
    k.FP = ft / 2;
    k.TPS = (d.T_out + k.FT - 1) / k.FT;
    k.NT = d.B * k.TPS;
    k.NPOS = k.FT * d.V;
    k.NP = k.NPOS;
    k.DP = k.NP | 1;
    k.NF = (d.mode == SAR_CONV_GRAPH) ? k.FT : (k.FT - 1) * d.stride + d.taps;
    k.RW = k.NF * d.V;
    k.SP = (k.RW + extra_taps * d.V) | 1;   // room for the phantom taps of the last wave class
    if (lds_bytes(d, k, BF, CT) <= 78 * 1024 || ft == 2) break;
  }
  const int sjmax = (d.mode == SAR_CONV_GRAPH) ? 4 : (d.taps == 1 ? 6 : 12);
  if (k.RW > 32 * sjmax || k.NP > 128) return -2;
  return 0;
}

template <int MODE, int TAPS, int WF, int WC, int WT, int TPW, int NZ0, int NZ1, int NZ2, int VC = 0, int STRIDEC = 0>
int launch(const sar_wgrad_desc& d, hipStream_t st) {
  WgradK k;
  k.d = d;
  constexpr int BF = 32 * WF, CT = 32 * WC;
  if (int g = geometry(d, k, BF, CT, WT * TPW - TAPS)) {
    sar_set_error("sar_conv_wgrad: unsupported tile geometry (V=%d, stride=%d)", d.V, d.stride);
    return g == -2 ? SAR_E_UNSUP : SAR_E_ARG;
  }
  if constexpr (VC != 0) {   // the fixed-geometry loop needs exactly two frame pairs per tile
    if (k.FP != 2) return launch<MODE, TAPS, WF, WC, WT, TPW, NZ0, NZ1, NZ2, 0, 0>(d, st);
    k.DP = (4 * VC) | 1;
    k.SP = 32 * ((MODE == SAR_CONV_GRAPH) ? 4 : (TAPS == 1 ? 6 : 12)) + 1;   // = 32 SJMAX + 1 in the kernel
  }
  const size_t lds = lds_bytes(d, k, BF, CT) + 256;   // the phantom taps of the last row may read a few floats on
  auto kern = conv_wgrad_kernel<MODE, TAPS, WF, WC, WT, TPW, NZ0, NZ1, NZ2, VC, STRIDEC>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      sar_set_error("sar_conv_wgrad: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
      return (int)e;
    }
  }
  k.gy = (d.M + BF - 1) / BF;
  k.gz = (d.Kc + CT - 1) / CT;
  dim3 grid((unsigned)(((int64_t)d.nsplit * k.gy * k.gz + 7) / 8 * 8));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WF * WC * WT), lds, st, k);
  return 0;
}

bool graph_fixed_supported(const sar_wgrad_desc& d) {
  return (d.nz[0] == 1 && d.nz[1] == 1 && d.nz[2] == 4) || (d.nz[0] == 1 && d.nz[2] == 1 && d.nz[1] == 4);
}

// ------------------------------------------------------------------------------------------------ graph weight gradient, Kc <= 4
// The first layer's graph convolution has 3 input channels: 9 (c, k) weight rows per output channel, 3 bias rows.  As an MFMA
// reduction (32-row blocks, 3 live) it ran at 3-7 TF and 400 us; it is 1.5 GFLOP of plain FMA work behind 246 MB of dout, i.e.
// a streaming kernel: one workgroup = 32 output channels (wave w: channels 8 w .. 8 w + 7) x its share of the columns,
// lanes = 64 consecutive columns (dout rows read as 256-byte segments), per lane the 3 x Kc gathered z values of its column
// (<= 4-entry lists, the 46 MB src stays in L2) and 8 x (3 Kc + 3) accumulators (three waves per SIMD), reduced over the lanes once at the end.
// Slabs as in every weight-gradient kernel (deterministic: fixed column order per lane, fixed lane tree, slabs in order).
template <int KC>
__global__ __launch_bounds__(256, KC <= 3 ? 3 : 2) void graph_wgrad_small_kernel(const sar_wgrad_desc d, int ncol, int cols_per_split) {
  constexpr int MW = 8;   // output channels per wave
  __shared__ int t_idx[3 * 64 * 4];     // gather tables [k][w][j] (V <= 64) and column sums: read per chunk, loaded once
  __shared__ float t_wt[3 * 64 * 4];
  __shared__ float t_cs[3 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x, m0 = blockIdx.y * 4 * MW + wave * MW;
  const int V = d.V;
  const bool do_bias = d.bsize > 0;
  for (int i = tid; i < 3 * V * 4; i += 256) {
    const int k = i / (V * 4), j = i & 3;
    t_idx[i] = d.g_idx[i];
    t_wt[i] = j < d.nz[k] ? d.g_wt[i] : 0.f;
  }
  for (int i = tid; i < 3 * V; i += 256) t_cs[i] = (do_bias && d.g_colsum) ? d.g_colsum[i] : 0.f;
  __syncthreads();
  const int col_lo = split * cols_per_split;
  const int col_hi = (col_lo + cols_per_split < ncol) ? col_lo + cols_per_split : ncol;
  float acc[MW][3 * KC], bacc[MW][3];
#pragma unroll
  for (int i = 0; i < MW; ++i) {
#pragma unroll
    for (int j = 0; j < 3 * KC; ++j) acc[i][j] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) bacc[i][j] = 0.f;
  }
  for (int c0 = col_lo; c0 < col_hi; c0 += 64) {
    const int col = c0 + lane;
    const bool live = col < col_hi;
    const int cc = live ? col : col_lo;
    const int fr = cc / V, w = cc - fr * V;
    // dout first (the long-latency stream), then z_k[c] of this column: the table-order fma chain of every gathering kernel
    float dv[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) dv[i] = (live && m0 + i < d.M) ? d.dout[(int64_t)(m0 + i) * d.ld_dout + cc] : 0.f;
    float z[3][KC], cs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      cs[k] = live ? t_cs[k * V + w] : 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) z[k][c] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < d.nz[k]) {   // uniform
          const int vi = t_idx[(k * V + w) * 4 + j];
          const float wt = live ? t_wt[(k * V + w) * 4 + j] : 0.f;
#pragma unroll
          for (int c = 0; c < KC; ++c)
            if (c < d.Kc) {   // uniform
              const float xv = d.src[(int64_t)c * d.ld_src + (int64_t)fr * V + vi];
              z[k][c] = j == 0 ? wt * xv : fmaf(wt, xv, z[k][c]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MW; ++i) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int c = 0; c < KC; ++c) acc[i][k * KC + c] = fmaf(z[k][c], dv[i], acc[i][k * KC + c]);
        bacc[i][k] = fmaf(cs[k], dv[i], bacc[i][k]);
      }
    }
  }
  float* slab = d.slab + (int64_t)split * (d.wsize + d.bsize);
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    const int m = m0 + i;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float t = wave_sum(acc[i][k * KC + c]);
        if (lane == 0 && m < d.M && c < d.Kc) slab[k * d.w_stride_tap + c * d.w_stride_c + m] = t;
      }
      if (do_bias) {   // uniform
        const float t = wave_sum(bacc[i][k]);
        if (lane == 0 && m < d.M) slab[d.wsize + (int64_t)k * d.M + m] = t;
      }
    }
  }
}

// Round 6: the same reduction with FOUR consecutive columns per lane (tools/leftover_bench.py: the kernel above moves its 257 MB at
// 0.8 TB/s -- 313 us against an HBM time of 43 -- because an iteration has 2 KB per wave in flight and exposes a full memory round
// trip): dout rows are read as 16-byte pieces (1 KB per wave instruction, 8 KB per wave and iteration in flight), the z values of
// the lane's four columns are built from the same gather tables (per-column frame / joint: a column quad may straddle a frame),
// accumulators and lane tree as above.  Needs 16-byte aligned dout rows and ncol % 4 == 0 (else the kernel above); two waves per SIMD.
template <int KC>
__global__ __launch_bounds__(256, 2) void graph_wgrad_small4_kernel(const sar_wgrad_desc d, int ncol, int cols_per_split) {
  constexpr int MW = 8;            // output channels per wave
  constexpr int NZV = 3 * KC + 3;  // values per column: z_k[c] (k-major) and colsum_k
  constexpr int XS = 256 + 64;     // <= 256 columns + one frame of slack (V <= 32)
  __shared__ int4 t_idx[3 * 32];   // gather tables [k][w] -> 4 entries (weights of unused entries are 0: no branches in the builder)
  __shared__ float4 t_wt[3 * 32];
  __shared__ float t_cs[3 * 32];
  __shared__ float xs[KC][XS];                                       // the src frames of the iteration's columns
  __shared__ __attribute__((aligned(16))) float zs[NZV][256];        // the iteration's z / colsum values, one row per value
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x, m0 = blockIdx.y * 4 * MW + wave * MW;
  const int V = d.V;
  const bool do_bias = d.bsize > 0;
  if (tid < 3 * V) {
    const int k = tid / V;
    int4 ix = *reinterpret_cast<const int4*>(d.g_idx + tid * 4);
    float4 wt = *reinterpret_cast<const float4*>(d.g_wt + tid * 4);
    const int nzk = d.nz[k];
    if (nzk < 1) wt.x = 0.f, ix.x = 0;
    if (nzk < 2) wt.y = 0.f, ix.y = 0;
    if (nzk < 3) wt.z = 0.f, ix.z = 0;
    if (nzk < 4) wt.w = 0.f, ix.w = 0;
    const int w = tid - k * V;
    t_idx[k * 32 + w] = ix;
    t_wt[k * 32 + w] = wt;
    t_cs[k * 32 + w] = (do_bias && d.g_colsum) ? d.g_colsum[tid] : 0.f;
  }
  const int col_lo = split * cols_per_split;   // multiples of 256
  const int col_hi = (col_lo + cols_per_split < ncol) ? col_lo + cols_per_split : ncol;
  float acc[MW][NZV];
#pragma unroll
  for (int i = 0; i < MW; ++i)
#pragma unroll
    for (int j = 0; j < NZV; ++j) acc[i][j] = 0.f;
  // software pipeline: the dout quads and the src frames of iteration i + 1 are requested during iteration i
  float4 dvn[MW];
  float xr[KC][2];
  auto request_dout = [&](int c0) {
    const int col = c0 + 4 * lane;
    const bool live = col < col_hi;          // col_hi % 4 == 0: a quad is live or dead as a whole
#pragma unroll
    for (int i = 0; i < MW; ++i)
      dvn[i] = (live && m0 + i < d.M) ? *reinterpret_cast<const float4*>(d.dout + (int64_t)(m0 + i) * d.ld_dout + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto request_src = [&](int c0) {
    const int f_lo = c0 / V;
    const int c_last = (c0 + 255 < ncol ? c0 + 255 : ncol - 1);
    const int nst = (c_last / V - f_lo + 1) * V;   // whole frames: every gather of a live column lies inside
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int c = 0; c < KC; ++c)
        xr[c][q] = (c < d.Kc && tid + 256 * q < nst) ? d.src[(int64_t)c * d.ld_src + (int64_t)f_lo * V + tid + 256 * q] : 0.f;
  };
  if (col_lo < col_hi) {
    request_src(col_lo);
    request_dout(col_lo);
  }
  for (int c0 = col_lo; c0 < col_hi; c0 += 256) {
    const int f_lo = c0 / V;
    const bool more = c0 + 256 < col_hi;   // uniform
    __syncthreads();   // (A) every wave has consumed zs / xs of the previous iteration (and the tables are written)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (tid + 256 * q < XS)
#pragma unroll
        for (int c = 0; c < KC; ++c) xs[c][tid + 256 * q] = xr[c][q];
    __syncthreads();     // (B) xs complete
    if (more) request_src(c0 + 256);
    {   // thread = column c0 + tid: its NZV values, branch-free (an unused entry has weight 0 and index 0)
      const int ce = c0 + tid;
      const bool lv = ce < col_hi;
      const int cq = lv ? ce : c0;
      const int fr = cq / V, w = cq - fr * V;
      const int fo = (fr - f_lo) * V;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int4 ix = t_idx[k * 32 + w];
        float4 wt = t_wt[k * 32 + w];
        if (!lv) wt = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          float zz = wt.x * xs[c][fo + ix.x];
          zz = fmaf(wt.y, xs[c][fo + ix.y], zz);
          zz = fmaf(wt.z, xs[c][fo + ix.z], zz);
          zz = fmaf(wt.w, xs[c][fo + ix.w], zz);
          zs[k * KC + c][tid] = zz;
        }
        zs[3 * KC + k][tid] = lv ? t_cs[k * 32 + w] : 0.f;
      }
    }
    __syncthreads();     // (C) zs complete
    float4 dv[MW];
#pragma unroll
    for (int i = 0; i < MW; ++i) dv[i] = dvn[i];
    if (more) request_dout(c0 + 256);
#pragma unroll
    for (int j = 0; j < NZV; ++j) {
      const float4 zq = *reinterpret_cast<const float4*>(&zs[j][4 * lane]);
#pragma unroll
      for (int i = 0; i < MW; ++i)
        acc[i][j] = fmaf(zq.w, dv[i].w, fmaf(zq.z, dv[i].z, fmaf(zq.y, dv[i].y, fmaf(zq.x, dv[i].x, acc[i][j]))));
    }
  }
  float* slab = d.slab + (int64_t)split * (d.wsize + d.bsize);
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    const int m = m0 + i;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const float t = wave_sum_dpp(acc[i][k * KC + c]);
        if (lane == 0 && m < d.M && c < d.Kc) slab[k * d.w_stride_tap + c * d.w_stride_c + m] = t;
      }
      if (do_bias) {   // uniform
        const float t = wave_sum_dpp(acc[i][3 * KC + k]);
        if (lane == 0 && m < d.M) slab[d.wsize + (int64_t)k * d.M + m] = t;
      }
    }
  }
}

int launch_graph_small(const sar_wgrad_desc& d, hipStream_t st) {
  const int ncol = d.B * d.T_out * d.V;
  static const bool quad = [] { const char* e = getenv("SAR_WGRAD_SMALL4"); return !(e && e[0] == '0'); }();
  if (quad && d.V <= 32 && ncol % 4 == 0 && d.ld_dout % 4 == 0 && ((uintptr_t)d.dout & 15) == 0 &&
      (((uintptr_t)d.g_idx | (uintptr_t)d.g_wt) & 15) == 0) {   // (V <= 32: an iteration's frames fit the staging buffer; tables read as 16-byte entries)
    const int cps4 = ((ncol + d.nsplit - 1) / d.nsplit + 255) / 256 * 256;   // whole 256-column chunks per slab
    dim3 grid4(d.nsplit, (d.M + 31) / 32);
    if (d.Kc <= 3) hipLaunchKernelGGL(graph_wgrad_small4_kernel<3>, grid4, dim3(256), 0, st, d, ncol, cps4);
    else hipLaunchKernelGGL(graph_wgrad_small4_kernel<4>, grid4, dim3(256), 0, st, d, ncol, cps4);
    return 0;
  }
  const int cps = ((ncol + d.nsplit - 1) / d.nsplit + 63) / 64 * 64;   // whole 64-column chunks per slab
  dim3 grid(d.nsplit, (d.M + 31) / 32);
  if (d.Kc <= 3) hipLaunchKernelGGL(graph_wgrad_small_kernel<3>, grid, dim3(256), 0, st, d, ncol, cps);
  else hipLaunchKernelGGL(graph_wgrad_small_kernel<4>, grid, dim3(256), 0, st, d, ncol, cps);
  return 0;
}

template <int NZ0, int NZ1, int NZ2, int MB>
int launch_graph_fixed_mb(const sar_wgrad_desc& d, hipStream_t st) {
  WgradK k;
  k.d = d;
  k.FT = 2;
  k.FP = 1;
  k.TPS = (d.T_out + 1) / 2;
  k.NT = d.B * k.TPS;
  const int gy = (d.M + 64 * MB - 1) / (64 * MB), gz = (d.Kc + 63) / 64;
  k.gy = gy;
  k.gz = gz;
  hipLaunchKernelGGL((graph_wgrad_fixed_kernel<NZ0, NZ1, NZ2, MB>), dim3((unsigned)(((int64_t)d.nsplit * gy * gz + 7) / 8 * 8)),
                     dim3(256), 0, st, k);
  return 0;
}

template <int NZ0, int NZ1, int NZ2>
int launch_graph_fixed_nz(const sar_wgrad_desc& d, hipStream_t st) {
  if (d.M > 64) return launch_graph_fixed_mb<NZ0, NZ1, NZ2, 2>(d, st);
  return launch_graph_fixed_mb<NZ0, NZ1, NZ2, 1>(d, st);
}

int launch_graph_fixed(const sar_wgrad_desc& d, hipStream_t st) {
  if (d.nz[0] == 1 && d.nz[1] == 1) return launch_graph_fixed_nz<1, 1, 4>(d, st);
  return launch_graph_fixed_nz<1, 4, 1>(d, st);
}

}  // namespace

extern "C" int sar_conv_wgrad_f32(const sar_wgrad_desc* d, sar_stream_t s) {
  SAR_REQUIRE(d != nullptr, "sar_conv_wgrad: null descriptor");
  SAR_REQUIRE(d->mode == SAR_CONV_GRAPH || d->mode == SAR_CONV_TEMPORAL, "sar_conv_wgrad: bad mode %d", d->mode);
  SAR_REQUIRE(d->B > 0 && d->V > 0 && d->V <= 64 && d->T_src > 0 && d->T_out > 0 && d->Kc > 0 && d->M > 0,
              "sar_conv_wgrad: bad sizes");
  SAR_REQUIRE(d->src && d->dout && d->slab, "sar_conv_wgrad: null src/dout/slab");
  SAR_REQUIRE(d->nsplit >= 1 && d->nsplit <= 65535, "sar_conv_wgrad: nsplit %d out of range", d->nsplit);
  SAR_REQUIRE(d->ld_src >= (int64_t)d->B * d->T_src * d->V && d->ld_dout >= (int64_t)d->B * d->T_out * d->V,
              "sar_conv_wgrad: leading dimension smaller than B*T*V");
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv_wgrad: pro_scale/pro_shift mismatch");
  SAR_REQUIRE(d->wsize > 0 && d->bsize >= 0, "sar_conv_wgrad: bad slab sizes");
  int rc = 0;
  hipStream_t st = as_stream(s);
  if (d->mode == SAR_CONV_GRAPH) {
    SAR_REQUIRE(d->taps == 3 && d->T_src == d->T_out, "sar_conv_wgrad: graph mode needs 3 slices and equal T");
    SAR_REQUIRE(d->g_idx && d->g_wt, "sar_conv_wgrad: graph gather tables required");
    SAR_REQUIRE(d->bsize == 0 || (d->bsize == 3 * (int64_t)d->M && d->g_colsum), "sar_conv_wgrad: graph bias slab is [3][M]");
    for (int i = 0; i < 3; ++i)
      if (d->nz[i] < 1 || d->nz[i] > 4) {
        sar_set_error("sar_conv_wgrad: adjacency slice %d needs %d gather entries per column (max 4)", i, d->nz[i]);
        return SAR_E_UNSUP;
      }
    if (d->Kc <= 4 && !d->pro_scale) rc = launch_graph_small(*d, st);
    else if (d->V == 25 && d->Kc >= 32 && !d->pro_scale && graph_fixed_supported(*d)) rc = launch_graph_fixed(*d, st);
    else if (d->nz[0] == 1 && d->nz[1] == 1) rc = launch<SAR_CONV_GRAPH, 3, 2, 2, 1, 3, 1, 1, 4>(*d, st);
    else if (d->nz[0] == 1 && d->nz[2] == 1) rc = launch<SAR_CONV_GRAPH, 3, 2, 2, 1, 3, 1, 4, 1>(*d, st);
    else rc = launch<SAR_CONV_GRAPH, 3, 2, 2, 1, 3, 4, 4, 4>(*d, st);
  } else {
    SAR_REQUIRE(d->stride >= 1 && d->pad >= 0, "sar_conv_wgrad: bad stride/pad");
    SAR_REQUIRE(d->bsize == 0 || d->bsize == d->M, "sar_conv_wgrad: temporal bias slab is [M]");
    const bool ntu = d->V == 25;
    if (d->taps == 9) {
      // 4 waves = 2 m-blocks x 2 groups of 5 tap slots (one phantom).  The 6-wave split <9,2,1,3,3> (3 groups of 3
      // taps, no phantom slot) is supported by the kernel but measured slower: 82 TF at 202 VGPR (one workgroup per
      // CU) and 75 TF squeezed to 168 VGPR (spills) against 105 TF here.
      if (ntu && d->stride == 1) rc = launch<SAR_CONV_TEMPORAL, 9, 2, 1, 2, 5, 1, 1, 1, 25, 1>(*d, st);
      else if (ntu && d->stride == 2) rc = launch<SAR_CONV_TEMPORAL, 9, 2, 1, 2, 5, 1, 1, 1, 25, 2>(*d, st);
      else rc = launch<SAR_CONV_TEMPORAL, 9, 2, 1, 2, 5, 1, 1, 1>(*d, st);
    } else if (d->taps == 1) {
      if (ntu && d->stride == 1) rc = launch<SAR_CONV_TEMPORAL, 1, 2, 2, 1, 1, 1, 1, 1, 25, 1>(*d, st);
      else if (ntu && d->stride == 2) rc = launch<SAR_CONV_TEMPORAL, 1, 2, 2, 1, 1, 1, 1, 1, 25, 2>(*d, st);
      else rc = launch<SAR_CONV_TEMPORAL, 1, 2, 2, 1, 1, 1, 1, 1>(*d, st);
    }
    else {
      sar_set_error("sar_conv_wgrad: temporal kernel size %d not built (1 and 9 are)", d->taps);
      return SAR_E_UNSUP;
    }
  }
  if (rc) return rc;
  SAR_LAUNCH_CHECK("sar_conv_wgrad_f32");
  return 0;
}

extern "C" int sar_slab_reduce_f32(const float* slab, int nsplit, int64_t slab_stride, int64_t n, float* out,
                                   sar_stream_t s) {
  SAR_REQUIRE(slab && out && nsplit >= 1 && n > 0 && slab_stride >= n, "sar_slab_reduce: bad arguments");
  const int64_t blocks = (n + 63) / 64;
  SAR_REQUIRE(blocks < (1ll << 31), "sar_slab_reduce: n too large");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(s), slab, nsplit, slab_stride, n, out);
  SAR_LAUNCH_CHECK("sar_slab_reduce_f32");
  return 0;
}

extern "C" int sar_slab_reduce_batch_f32(const sar_slab_item* items, int nitems, int64_t max_n, sar_stream_t s) {
  SAR_REQUIRE(items && nitems >= 1 && nitems <= 65535 && max_n > 0, "sar_slab_reduce_batch: bad arguments");
  const int64_t blocks = (max_n + 63) / 64;
  SAR_REQUIRE(blocks < (1ll << 31), "sar_slab_reduce_batch: n too large");
  hipLaunchKernelGGL(slab_reduce_batch_kernel, dim3((unsigned)blocks, (unsigned)nitems), dim3(256), 0, as_stream(s), items);
  SAR_LAUNCH_CHECK("sar_slab_reduce_batch_f32");
  return 0;
}

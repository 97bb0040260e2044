// conv_wgrad_split.hip -- weight / bias gradient of the 9-tap temporal convolution (stride 1; stride 2 through parity images) and of
// the graph convolution with fp32 results on the fp16 /
// bf16 matrix pipe (the "split" arithmetic of conv_gemm_split.hip; gfx950).
//
//   dW[tap][c][m] = sum_n pro(src)[c, n + (tap - pad) V] * dout[m, n]        (tf.GradientTape of the Conv2D [9,1],
//   dbias[m]      = sum_n dout[m, n]   (fp32 sums of the fp32 values)         main_gnn.py:233, models/stgcn.py:29-36)
//
// Same slab contract as sar_conv_wgrad_f32 (include/sar_hip.h): slab[s][wsize + bsize], summed by sar_slab_reduce_f32 in slab
// order (deterministic, no atomics).  Every fp32 operand element enters the matrix pipe as two fp16 terms of its scaled value
// (f16x3a: three products per fp32 product, conv_gemm_split.hip) or three bfloat16 terms (bf16x6: six products); products are exact, accumulation fp32.
//
// Design (MI355X):
//  * the contraction runs over positions n = (t, v) of one sequence, the contiguous axis of both operands in the CN layout: a
//    lane's MFMA fragment is 8 consecutive positions of one row.  A temporal tap shifts the src window by tap * V positions,
//    whatever the tile's alignment to frames: a tile is KT consecutive positions (a multiple of 16, not of V), its src window
//    KT + 8 V positions.
//  * workgroup = 32 src channels x 128 (M >= 128) or 64 (M = 64) dout channels x all nine taps: each wave owns a 32 x 32 block
//    of every tap (9 accumulators = 144 registers); at M = 64 the two wave pairs split the tile's k-steps and write two slabs.
//  * only the src window goes through LDS ([term][32 rows][window] 2-byte elements, split behind the folded BatchNorm + ReLU by
//    the stager): V = 25 makes odd taps start on an odd element, so a window is read as aligned dwords and funnel-shifted by two
//    bytes (v_alignbyte_b32) for odd taps.  Row stride / 2 is odd: the 32 rows of a fragment read hit distinct banks.
//  * the dout fragment of a k-step is the same for all nine taps: each wave loads its 32 rows x 16 positions straight from
//    global memory into registers one k-step ahead (two 16-byte loads per lane), sums them for the bias gradient, splits them
//    in registers.  No LDS for dout: the three src images of a 192-position tile take 76 KB -- two workgroups per CU.
//  * per k-step and wave: 27 MFMAs (f16x3a), ~80 vector instructions (split of dout, funnel shifts, bias sums), 81 dword LDS
//    reads.
#include "sar_common.h"
#include <stdlib.h>
#include <type_traits>

// This file is compiled twice: as itself (part 0: the ST-GCN operators and their entry points) and through conv2d_wgrad_split.hip
// (part 1: the same temporal kernel instantiated for the 3x3 convolutions of the resnet, W2 > 0, and its entry point).
#ifndef SAR_WSPLIT_PART
#define SAR_WSPLIT_PART 0
#endif

struct WgradKS {     // kernel argument (the same definition in every part)
  sar_wgrad_desc d;
  int TPS, ntiles, gy, gz;
  int H2, seq2;      // W2 kernels: image height, positions of the flattened batch (B H W2)
  float invH2;
  const unsigned* src_bound;
  const unsigned* dout_bound;
  int ablate;        // -DSAR_GW_ABLATE builds only (tools/gw_ablate.sh): phases of graph_wgrad_split_kernel switched off by SAR_GW_ABLATE_BITS
};
// bit 0: no gathered slices (1, 2), 1: no slice-0 images, 2: dout fragments not converted (zeros), 3: no MFMA, 4: no slab stores.  Results are
// then WRONG; only the launch time is read.
#ifdef SAR_GW_ABLATE
#define GW_ON(bit) (!(k.ablate & (1 << (bit))))
#else
#define GW_ON(bit) true
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build -DSAR_SPLIT_TL (tools/split_timeline.sh): wave 0 of every workgroup of the LAST launch writes one row -- start /
// end in 100 MHz ticks (s_memrealtime), HW_ID, XCC_ID and the shader-clock cycles it spent in each phase (summed over the stages) --
// DESIGN 3.9h's instrument for the split kernels.  No stamp executes in the product build.
#ifdef SAR_SPLIT_TL
constexpr int WSPLIT_TL_WG = 16384;
__device__ unsigned g_wsplit_tl[WSPLIT_TL_WG][16];
#define SPLIT_TL_BEGIN()                                                \
  const unsigned long long tl_rt0 = __builtin_amdgcn_s_memrealtime();   \
  unsigned long long tl_last = __builtin_amdgcn_s_memtime();            \
  unsigned tl_acc[10] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}
#define SPLIT_TL(i)                                               \
  do {                                                            \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
    tl_acc[i] += (unsigned)(t_ - tl_last);                        \
    tl_last = t_;                                                 \
  } while (0)
#define SPLIT_TL_END(wid)                                                                   \
  do {                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < WSPLIT_TL_WG) {                                     \
      unsigned* row = g_wsplit_tl[blockIdx.x];                                               \
      row[0] = (unsigned)tl_rt0, row[1] = (unsigned)__builtin_amdgcn_s_memrealtime();       \
      row[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                   \
      row[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                                  \
      for (int i_ = 0; i_ < 10; ++i_) row[4 + i_] = tl_acc[i_];                             \
      row[14] = (unsigned)(wid);                                                            \
    }                                                                                       \
  } while (0)
#else
#define SPLIT_TL_BEGIN()
#define SPLIT_TL(i)
#define SPLIT_TL_END(wid)
#endif

constexpr int VJ = 25, TAPS = 9, CB = 32;
constexpr int AR_B6 = SAR_SPLIT_BF16X6, AR_H3A = SAR_SPLIT_F16X3A;
constexpr float H3_LO = 2048.f;
constexpr bool ar_f16(int ar) { return ar == AR_H3A; }
// f16x3a (conv_gemm_split.hip): the well-conditioned operand -- here the src activation behind its BatchNorm + ReLU, scaled from
// the Samuelson bound -- carries three images (a0, a1, a0 2^-11), the wide-range operand -- dout, a gradient -- two (d0, d1 2^11)
constexpr int ar_nta(int ar) { return 3; }
constexpr int ar_ntb(int ar) { return ar == AR_H3A ? 2 : 3; }
constexpr int ar_nprod(int ar) { return ar == AR_H3A ? 3 : 6; }
// product p: (src term, dout term), smallest magnitude first (conv_gemm_split.hip)
constexpr int ar_pi(int ar, int p) {
  constexpr int i6[6] = {0, 2, 1, 0, 1, 0};
  return ar == AR_H3A ? 2 - p : i6[p];
}
constexpr int ar_pj(int ar, int p) {
  constexpr int j6[6] = {2, 0, 1, 1, 0, 0};
  return ar == AR_H3A ? (p == 0 ? 1 : 0) : j6[p];
}
// S2 (stride 2, TF-SAME pad 3 of an even T: models/stgcn.py:117,120): out frame t reads src frame 2 t + tap - 3.  The src frames are
// de-interleaved by parity into two images in the coordinates of the dout positions -- E: frame 2 i at (i, v), O: frame 2 i + 1 --
// in which a tap is again a shift by whole frames: odd taps read E at n + 25 (tap - 3) / 2, even taps O at n + 25 (tap - 4) / 2.
// Both images cover [n0 - 50, n0 + KT + 50): with KT = 96 the row is as long as stride 1's (392 elements).
// W2 > 0 (part 1): the 3x3 / stride-1 / pad-1 convolution on images of width W2 (a power of two >= 8), the whole batch flattened
// into ONE sequence of B H W2 positions.  Tap (kh, kw) is the shift (kh - 1) W2 + (kw - 1) of the flat position -- exact except at
// the image borders, where the shifted read lands in a neighbouring row / image instead of the zero padding: a chunk of 8 positions
// lies inside one image row (W2 % 8 == 0), so the left / right border costs one masked element of the dout fragment (kw = 0: the
// chunk's first element when it starts a row; kw = 2: its last when it ends one) and the top / bottom border redirects the src
// fragment read of the chunk to an always-zero row of the LDS image (kh = 0 in the first image row, kh = 2 in the last).
template <int AR, int S2 = 0, int W2 = 0> struct Cfg {
  static constexpr int NT = ar_nta(AR);                 // src images in LDS
  static constexpr int KT = W2 ? 256 : (S2 ? 96 : 192);  // dout positions per tile (multiple of 16): 3 x 32 x 394 x 2 B = 75.6 KB
  static constexpr int KS = KT / 16;
  static constexpr int WIN2 = KT + 4 * VJ;               // S2: positions per parity image
  static constexpr int WIN = W2 ? KT + 2 * W2 + 2 : (S2 ? 2 * WIN2 : KT + (TAPS - 1) * VJ);   // src window (S2: E | O)
  static constexpr int RS = ((WIN + 2 + 1) / 2 * 2) + ((((WIN + 2 + 1) / 2) & 1) ? 0 : 2);   // row stride (elements): >= WIN + 2, RS / 2 odd
  static constexpr int NCH = (WIN + 127) / 128;          // stager chunks of 128 positions (a lane owns two adjacent positions)
  static constexpr int ROWS = CB + (W2 ? 1 : 0);         // rows per term image (W2: + the always-zero row)
  static_assert((RS / 2) % 2 == 1 && RS >= WIN + 2, "row stride");
  static_assert(W2 == 0 || (W2 >= 8 && W2 <= 64 && (W2 & (W2 - 1)) == 0 && S2 == 0), "image width: a power of two in [8, 64]");
};

#include "split_scale.h"   // scale_exp, bound_nonfinite, split_unscale (shared with split_terms.h)

__device__ __forceinline__ unsigned pk_bf16(float x, float y) {
  bf16x2 p;
  p[0] = (__bf16)x;
  p[1] = (__bf16)y;
  return *reinterpret_cast<unsigned*>(&p);
}
__device__ __forceinline__ unsigned pk_f16(float x, float y) {
  f16x2 p;
  p[0] = (_Float16)x;
  p[1] = (_Float16)y;
  return *reinterpret_cast<unsigned*>(&p);
}
// f16x3a: the conditioned operand's third image is (first image) x 2^-11.  It is neither written to LDS nor read: a fragment of it is four
// v_pk_mul_f16 on the fragment of the first image -- an exact power-of-two scaling with the same fp16 rounding split2 applies -- so a
// third of the stager's LDS stores, one conversion per element pair and a third of the k-steps' operand reads are gone (round 6;
// bit-identical results).  ntl<AR>() = images that live in LDS.
template <int AR> constexpr int ntl() { return AR == AR_H3A ? ar_nta(AR) - 1 : ar_nta(AR); }
__device__ __forceinline__ u32x4 third_image(const u32x4& a0) {
  const f16x8 w0 = *reinterpret_cast<const f16x8*>(&a0);
  const f16x8 w2 = w0 * (_Float16)(1.f / H3_LO);
  return *reinterpret_cast<const u32x4*>(&w2);
}
// two adjacent (already scaled) values -> one dword per term.  WSIDE: the conditioned operand's images (src), else dout's
template <int AR, bool WSIDE>
__device__ __forceinline__ void split2(float x, float y, unsigned (&w)[WSIDE ? ar_nta(AR) : ar_ntb(AR)]) {
  constexpr int NT = WSIDE ? ar_nta(AR) : ar_ntb(AR);
  if constexpr (ar_f16(AR)) {
    x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    y = __builtin_amdgcn_fmed3f(y, -65504.f, 65504.f);
    f16x2 h;
    h[0] = (_Float16)x;
    h[1] = (_Float16)y;
    w[0] = *reinterpret_cast<unsigned*>(&h);
    if constexpr (WSIDE) {
      w[1] = pk_f16(x - (float)h[0], y - (float)h[1]);
      w[2] = pk_f16((float)h[0] * (1.f / H3_LO), (float)h[1] * (1.f / H3_LO));
    } else {
      w[1] = pk_f16((x - (float)h[0]) * H3_LO, (y - (float)h[1]) * H3_LO);
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const unsigned b = pk_bf16(x, y);
      w[t] = b;
      if (t + 1 < NT) {
        x -= __uint_as_float(b << 16);
        y -= __uint_as_float(b & 0xffff0000u);
      }
    }
  }
}


// WK = 1: four waves side by side along m (128 dout channels); WK = 2: two along m, the pairs split the k-steps (M <= 64)
template <int AR, int WK, int S2, int W2 = 0>
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_kernel(const WgradKS k) {
  using C = Cfg<AR, S2, W2>;
  constexpr int NT = C::NT, NTB = ar_ntb(AR), NPROD = ar_nprod(AR), KT = C::KT, KS = C::KS, WIN = C::WIN, RS = C::RS, NCH = C::NCH, V = VJ;
  constexpr int ROWS = C::ROWS;
  constexpr int WMM = 4 / WK, MBLK = 32 * WMM;
  __shared__ __attribute__((aligned(16))) unsigned short Hs[NT * ROWS * RS];
  __shared__ float2 bnp[256];
  const sar_wgrad_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wmm = wave % WMM, kh = wave / WMM;

  // workgroup -> (split group, m block, c block); the blocks of one split are adjacent slots of one XCD (L2 reuse of the tiles)
  int sg, by, bz;
  {
    const int nyz = k.gy * k.gz, ngrp = d.nsplit / WK, nwork = ngrp * nyz;
    const int per = (nwork + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int w = xcd * per + slot;
    if (slot >= per || w >= nwork) return;
    sg = w / nyz;
    const int yz = w - sg * nyz;
    bz = yz / k.gy;
    by = yz - bz * k.gy;
  }
  const int m0 = by * MBLK + wmm * 32, c0 = bz * CB;
  const int ngrp = d.nsplit / WK;
  SPLIT_TL_BEGIN();

  int ea = 0, eb = 0;
  bool nonfin = false;   // an operand bound holds Inf / NaN bits: every output of the launch is NaN (split_scale.h)
  if (ar_f16(AR)) {
    ea = scale_exp(*k.src_bound);
    eb = scale_exp(*k.dout_bound);
    nonfin = bound_nonfinite(*k.src_bound) || bound_nonfinite(*k.dout_bound);
  }
  const float sa = __builtin_ldexpf(1.f, ea), sb = __builtin_ldexpf(1.f, eb);
  {   // the folded prologue of this block's 32 channels, the src scale folded in (power of two: exact)
    float2 p = make_float2(sa, 0.f);
    if (tid < CB && d.pro_scale && c0 + tid < d.Kc) p = make_float2(d.pro_scale[c0 + tid] * sa, d.pro_shift[c0 + tid] * sa);
    if (tid < CB) bnp[tid] = p;
  }
  if constexpr (W2 > 0) {   // the always-zero row of every term image
    for (int i = tid; i < NT * (RS / 2); i += 256)
      *reinterpret_cast<unsigned*>(&Hs[((i / (RS / 2)) * ROWS + CB) * RS + 2 * (i % (RS / 2))]) = 0u;
  }

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = d.bsize > 0 && bz == 0;   // uniform

  const int tps = (k.ntiles + ngrp - 1) / ngrp;
  const int tile_lo = sg * tps;
  const int tile_hi = (tile_lo + tps < k.ntiles) ? tile_lo + tps : k.ntiles;
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  const int seq = W2 ? k.seq2 : d.T_out * V;        // dout positions per sequence
  const int seq_src = W2 ? k.seq2 : d.T_src * V;    // (stride 1: the same)

  // LDS read base of this lane: row l31, element 8 * hi
  const unsigned a_base = (unsigned)(uintptr_t)Hs + (unsigned)((l31 * RS + 8 * hi) * 2);
  typedef const unsigned __attribute__((address_space(3))) * lds_u32;
  // dout rows of this wave: one descriptor over the whole tensor, per-lane offsets
  const bool mrow_ok = (m0 + l31) < d.M;
  constexpr unsigned REJECT = 0xf0000000u;   // beyond num_records, and + 16 does not wrap
  const int64_t dbytes = (int64_t)d.M * d.ld_dout * 4;
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.dout, 0, (unsigned)(dbytes < (int64_t)REJECT ? dbytes : (int64_t)REJECT), 0x00020000);
  const unsigned drow = (unsigned)(((int64_t)(mrow_ok ? m0 + l31 : 0) * d.ld_dout) * 4);

  // ---- the stager.  A tile's window is staged in two passes of four rows per wave.  (Requesting the first pass of tile i + 1 in
  // front of the MFMA phase of tile i -- 32 registers across the k-loop -- gained 5 %; the registers buy more as the per-tap fragment
  // pipeline of the k-loop.)  Geometry (this lane's window columns and their src positions, the same for every row): stride 1 -- consecutive
  // positions; S2 -- column q of image E / O is dout-space position n0 - 50 + q = (i, v), i.e. src frame 2 i (+ 1).
  auto geometry = [&](int n0, int (&spos)[NCH][2], bool (&sok)[NCH][2]) {
    const int p_lo = W2 ? n0 - (W2 + 1) : n0 - d.pad * V;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int col = 2 * lane + 128 * j + e;
        if (S2) {
          const int img = col >= C::WIN2 ? 1 : 0, q = col - img * C::WIN2;
          const int nn = n0 - 2 * V + q;
          const int i = floordiv(nn, V), v = nn - i * V;
          const int fr = 2 * i + img;
          sok[j][e] = col < WIN && fr >= 0 && fr < d.T_src;
          spos[j][e] = sok[j][e] ? fr * V + v : -1;
        } else {
          const int pa = p_lo + col;
          sok[j][e] = col < WIN && (unsigned)pa < (unsigned)seq_src;
          spos[j][e] = pa;
        }
      }
  };
  auto issue_pass = [&](int b, int rh, const int (&spos)[NCH][2], float (&x)[4][NCH][2]) {
    const float* src_b = d.src + (int64_t)b * seq_src;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + wave + 4 * (rh * 4 + q);
      const int cg = c < d.Kc ? c : 0;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, seq_src * 4, 0x00020000);
#pragma unroll
      for (int j = 0; j < NCH; ++j) {   // negative / past-the-end offsets: rejected by the range check -> 0
        x[q][j][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, spos[j][0] * 4, 0, 0));
        x[q][j][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, spos[j][1] * 4, 0, 0));
      }
    }
  };
  auto convert_pass = [&](int rh, const bool (&sok)[NCH][2], const float (&x)[4][NCH][2]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = wave + 4 * (rh * 4 + q);
      const bool rok = c0 + row < d.Kc;
      const float2 ps = bnp[row];
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int col = 2 * lane + 128 * j;
        const float v0 = (rok && sok[j][0]) ? fmaxf(fmaf(x[q][j][0], ps.x, ps.y), relu_lo) : 0.f;   // TF-SAME padding stays exactly 0
        const float v1 = (rok && sok[j][1]) ? fmaxf(fmaf(x[q][j][1], ps.x, ps.y), relu_lo) : 0.f;
        unsigned w[NT];
        split2<AR, true>(v0, v1, w);
        if (col < RS) {
#pragma unroll
          for (int t = 0; t < ntl<AR>(); ++t) *reinterpret_cast<unsigned*>(&Hs[(t * ROWS + row) * RS + col]) = w[t];
        }
      }
    }
  };
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int b = tile / k.TPS;
    const int n0 = (tile - b * k.TPS) * KT;
    SPLIT_TL(0);   // prologue / loop overhead
    __syncthreads();   // closing: every wave has read its last fragment of the previous tile (and bnp is written)
    SPLIT_TL(1);   // closing barrier
    // dout fragments come straight from global memory, one k-step ahead; the first one is requested here, in front of the stager.
    // (Three k-steps ahead in a ring of four register sets: measured equal -- the fragment's latency is not what a k-step waits for.)
    const bool ragged = n0 + KT > seq;   // the tile reaches past the end of the sequence: mask dout per element
    auto load_dout = [&](int ks, u32x4 (&raw)[2]) {
      const int pos = n0 + 16 * ks + 8 * hi;
      const unsigned vo = (mrow_ok && pos < seq) ? drow + (unsigned)(((int64_t)b * seq + pos) * 4) : REJECT;   // rejected -> 0
      raw[0] = __builtin_amdgcn_raw_buffer_load_b128(rdo, vo, 0, 0);
      raw[1] = __builtin_amdgcn_raw_buffer_load_b128(rdo, vo, 16, 0);
    };
    u32x4 raw[2][2];
    load_dout(kh, raw[0]);
    {
      int spos[NCH][2];
      bool sok[NCH][2];
      geometry(n0, spos, sok);
#pragma unroll 1
      for (int rh = 0; rh < 2; ++rh) {
        float x[4][NCH][2];
        issue_pass(b, rh, spos, x);
        convert_pass(rh, sok, x);
      }
    }
    SPLIT_TL(2);   // stager: requests, wait, convert, LDS stores
    __syncthreads();   // opening: the window is complete
    SPLIT_TL(3);   // opening barrier

    // ---- k-steps of this wave (WK = 2: the second wave pair takes the odd ones)
    SAR_LDS_SKEW();   // this wave reads the window late: the next tile's stager must wait at the closing barrier
    // The nine taps' fragments run through a ring of THREE register sets: the dwords of tap t + 2 (of the next k-step behind tap 7)
    // are requested before the MFMAs of tap t issue, the order pinned with sched_barrier.  tools/wsplit_timeline.py: a k-step took
    // 2 780 cycles for 864 of matrix work -- left to the compiler every tap was read -> s_waitcnt -> multiply (35 waits per k-step);
    // one tap ahead (3 MFMAs + 12 funnel shifts = ~150 cycles) is less than the LDS round trip with eight waves reading.
    auto tap_elem = [&](int t) {   // window start in elements: stride 1 -- t V (V odd: the parity of t); S2 -- image E for odd
      // taps, O for even ones, shifted by whole frames: (floor((t - 3) / 2) + 2) V inside the image
      return W2 ? (t / 3) * W2 + (t % 3) : S2 ? ((t & 1) ? 0 : C::WIN2) + ((t - 3 - ((t & 1) ? 0 : 1)) / 2 + 2) * V : t * V;
    };
    // W2: which LDS address the src fragment of tap t is read from for the chunk of k-step kc_ -- the window, or the zero row when
    // the tap leaves the image through its top (kh = 0, first row) / bottom (kh = 2, last row) border
    const unsigned a_zero = (unsigned)(uintptr_t)Hs + (unsigned)(CB * RS * 2);
    auto chunk_edges = [&](int kc_, bool& top, bool& bot) {
      const int row = (n0 + 16 * kc_ + 8 * hi) / (W2 ? W2 : 1);
      const int img = (int)(((float)row + 0.5f) * k.invH2);
      const int h = row - img * k.H2;
      top = h == 0, bot = h == k.H2 - 1;
    };
    unsigned fw[3][NT][5];
    auto load_tap = [&](unsigned a_ks, int t, unsigned (&w)[NT][5]) {
      const int e = tap_elem(t);
#pragma unroll
      for (int tm = 0; tm < ntl<AR>(); ++tm) {
        lds_u32 p = (lds_u32)(uintptr_t)(a_ks + tm * (ROWS * RS * 2) + (e & ~1) * 2);
#pragma unroll
        for (int i = 0; i < 4 + (e & 1); ++i) w[tm][i] = p[i];
      }
    };
    auto tap_base = [&](unsigned a_ks, int t, bool top, bool bot) {
      if constexpr (W2 > 0) {
        if (t < 3) return top ? a_zero : a_ks;
        if (t >= 6) return bot ? a_zero : a_ks;
      }
      return a_ks;
    };
    bool top_n = false, bot_n = false;   // W2: border flags of the chunk whose first taps are in the ring
    if (kh < KS) {
      if constexpr (W2 > 0) chunk_edges(kh, top_n, bot_n);
      load_tap(tap_base(a_base + kh * 32, 0, top_n, bot_n), 0, fw[0]);
      load_tap(tap_base(a_base + kh * 32, 1, top_n, bot_n), 1, fw[1]);
    }
#pragma unroll 1
    for (int ks = kh; ks < KS; ks += 2 * WK) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {   // two k-steps per iteration: static double buffer
        const int kc = ks + half * WK;
        if (kc < KS) {
          if (kc + WK < KS) load_dout(kc + WK, raw[half ^ 1]);
          // split the dout fragment (8 consecutive positions of row m0 + l31)
          float dv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) dv[j] = __uint_as_float(raw[half][j >> 2][j & 3]);
          if (ragged) {
            const int nv = seq - (n0 + 16 * kc + 8 * hi);
#pragma unroll
            for (int j = 0; j < 8; ++j) dv[j] = j < nv ? dv[j] : 0.f;
          }
          if (do_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += dv[j];
          }
          unsigned bw[NTB][4];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            unsigned w[NTB];
            split2<AR, false>(dv[2 * p] * sb, dv[2 * p + 1] * sb, w);
#pragma unroll
            for (int t = 0; t < NTB; ++t) bw[t][p] = w[t];
          }
          const unsigned a_ks = a_base + kc * 32;
          const bool more = kc + WK < KS;   // wave-uniform
          const bool top_c = top_n, bot_c = bot_n;
          unsigned bwl0[NTB], bwr3[NTB];   // W2: dword 0 / 3 of the dout fragment with the row's first / last element masked
          if constexpr (W2 > 0) {
            const int pos = n0 + 16 * kc + 8 * hi;
            const unsigned ml = (pos & (W2 - 1)) == 0 ? 0xffff0000u : 0xffffffffu;
            const unsigned mr = ((pos + 8) & (W2 - 1)) == 0 ? 0x0000ffffu : 0xffffffffu;
#pragma unroll
            for (int t = 0; t < NTB; ++t) bwl0[t] = bw[t][0] & ml, bwr3[t] = bw[t][3] & mr;
            if (more) chunk_edges(kc + WK, top_n, bot_n);
          }
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            if (t + 2 < TAPS) load_tap(tap_base(a_ks, t + 2, top_c, bot_c), t + 2, fw[(t + 2) % 3]);
            else if (more) load_tap(tap_base(a_ks + WK * 32, t + 2 - TAPS, top_n, bot_n), t + 2 - TAPS, fw[(t + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            const int e = tap_elem(t);
            u32x4 aq[NT];
#pragma unroll
            for (int tm = 0; tm < ntl<AR>(); ++tm) {
              const unsigned (&w)[5] = fw[t % 3][tm];
              if ((e & 1) == 0) aq[tm] = u32x4{w[0], w[1], w[2], w[3]};
              else
                aq[tm] = u32x4{__builtin_amdgcn_alignbyte(w[1], w[0], 2), __builtin_amdgcn_alignbyte(w[2], w[1], 2),
                               __builtin_amdgcn_alignbyte(w[3], w[2], 2), __builtin_amdgcn_alignbyte(w[4], w[3], 2)};
            }
            if constexpr (ntl<AR>() < NT) aq[NT - 1] = third_image(aq[0]);
#pragma unroll
            for (int p = 0; p < NPROD; ++p) {
              const int i = ar_pi(AR, p), j = ar_pj(AR, p);
              const u32x4 bq = u32x4{(W2 > 0 && t % 3 == 0) ? bwl0[j] : bw[j][0], bw[j][1], bw[j][2], (W2 > 0 && t % 3 == 2) ? bwr3[j] : bw[j][3]};
              if (ar_f16(AR))
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&aq[i]),
                                                                *reinterpret_cast<const f16x8*>(&bq), acc[t], 0, 0, 0);
              else
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&aq[i]),
                                                                 *reinterpret_cast<const bf16x8*>(&bq), acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    SPLIT_TL(4);   // k-steps
  }

  SPLIT_TL(4);   // the last tile's k-steps (the others are booked with the loop overhead 0)
  // ---- this wave's slab: rows c (registers), columns m (lanes: contiguous)
  float* slab = d.slab + (int64_t)(sg * WK + kh) * (d.wsize + d.bsize);
  const float unscale = (nonfin ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + eb)));
  const int m = m0 + l31;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + mfma_row(r, hi);
      if (c < d.Kc && m < d.M) slab[(int64_t)t * d.w_stride_tap + (int64_t)c * d.w_stride_c + m] = acc[t][r] * unscale;
    }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32);
    if (hi == 0 && m < d.M) slab[d.wsize + m] = bsum;
  }
  SPLIT_TL(5);   // slab stores
  SPLIT_TL_END(tile_hi - tile_lo);
}

#if SAR_WSPLIT_PART == 0
// ---- GraphConvTD (models/gcn.py:199-209) weight / bias gradient in the split arithmetics:
//   dW_k[c][m] = sum_n z_k[c, n] dout[m, n],  z_k[c, (t,w)] = sum_v x[c, (t,v)] A_k[v, w];   db_k[m] = sum_n dout[m, n] colsum(A_k)[w(n)]
// Same structure as the temporal kernel with the three adjacency slices in the place of the nine taps: a tile = 4 frames (100
// positions, 7 k-steps of 16), the three gathered images z_k are built by the stager (<= 4 gathered loads per element straight from
// global memory -- the frame is in L1 / L2 --, the fp32 kernel's fma chain, then the split into the conditioned operand's three
// images), [slice][term][32 rows][120] 2-byte elements = 69 KB: two workgroups per CU.  A wave owns 32 src channels x 64 dout channels
// x 3 slices (96 accumulator registers); at M <= 128 / 64 the wave groups split the tile's k-steps (WK = 2 / 4 slabs per group).
constexpr int GFT = 4, GKP = GFT * VJ, GKS = (GKP + 15) / 16, GRS = 120, GRAW = 101;   // raw tile: 32 rows x 101 floats (odd stride)
template <int AR, int WK, int NZ0, int NZ1, int NZ2, bool S0ID>
__global__ __launch_bounds__(256, 2) void graph_wgrad_split_kernel(const WgradKS k) {
  constexpr int NT = ar_nta(AR), NTB = ar_ntb(AR), NPROD = ar_nprod(AR), V = VJ, KS = GKS, RS = GRS;
  constexpr int NZ[3] = {NZ0, NZ1, NZ2};
  constexpr int WMM = 4 / WK, MBLK = 64 * WMM;
  constexpr int RPP = (NZ0 + NZ1 + NZ2 > 6) ? 1 : 2;   // generic stager: rows per pass
  static_assert(RS >= KS * 16 && (RS * 2) % 16 == 0 && ((RS / 2) / 4) % 2 == 1, "row stride: 16-byte rows, conflict-free 16-byte reads");
  static_assert(CB * GRAW * 4 <= NT * CB * RS * 2, "the raw tile fits the area of slice 0's images");
  __shared__ __attribute__((aligned(16))) unsigned short Zs[3 * NT * CB * RS];
  __shared__ __attribute__((aligned(16))) float csl[3 * KS * 16];   // colsum(A_k) of every tile position (0 beyond the live ones)
  const sar_wgrad_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wmm = wave % WMM, kh = wave / WMM;

  int sg, by, bz;
  {
    const int nyz = k.gy * k.gz, ngrp = d.nsplit / WK, nwork = ngrp * nyz;
    const int per = (nwork + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int w = xcd * per + slot;
    if (slot >= per || w >= nwork) return;
    sg = w / nyz;
    const int yz = w - sg * nyz;
    bz = yz / k.gy;
    by = yz - bz * k.gy;
  }
  const int m0 = by * MBLK + wmm * 64, c0 = bz * CB;
  const int ngrp = d.nsplit / WK;

  int ea = 0, eb = 0;
  bool nonfin = false;   // an operand bound holds Inf / NaN bits: every output of the launch is NaN (split_scale.h)
  if (ar_f16(AR)) {   // a gathered value is a weighted sum of <= 4 source values: bound(src) x the largest sum of |weights| (<= 4 lists x 3 V)
    float gmax = 1.f;
    for (int i = 0; i < 3 * V; ++i) {
      float sm = 0.f;
      for (int j = 0; j < 4; ++j) sm += fabsf(d.g_wt[i * 4 + j]);
      gmax = fmaxf(gmax, sm);
    }
    ea = scale_exp(__float_as_uint(__uint_as_float(*k.src_bound) * gmax));
    eb = scale_exp(*k.dout_bound);
    nonfin = bound_nonfinite(*k.src_bound) || bound_nonfinite(*k.dout_bound);
  }
  const float sa = __builtin_ldexpf(1.f, ea), sb = __builtin_ldexpf(1.f, eb);
  for (int i = tid; i < 3 * KS * 16; i += 256) {
    const int kk = i / (KS * 16), p = i - kk * (KS * 16);
    csl[i] = (p < GKP && d.g_colsum) ? d.g_colsum[kk * V + p % V] : 0.f;
  }

  // this lane's two adjacent tile positions and their gather entries (offsets inside the tile, floats)
  int goff[3][2][4];
  float gwt[3][2][4];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int p = 2 * lane + e;
    const bool live = p < GKP;
    const int fo = live ? p / V : 0, w = live ? p - fo * V : 0;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < NZ[kk]) {
          goff[kk][e][j] = fo * V + d.g_idx[(kk * V + w) * 4 + j];
          gwt[kk][e][j] = live ? d.g_wt[(kk * V + w) * 4 + j] : 0.f;
        }
  }

  f32x16 acc[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][mb][r] = 0.f;
  float bsum[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  const bool do_bias = d.bsize > 0 && bz == 0;   // uniform

  const int tps = (k.ntiles + ngrp - 1) / ngrp;
  const int tile_lo = sg * tps;
  const int tile_hi = (tile_lo + tps < k.ntiles) ? tile_lo + tps : k.ntiles;
  const int seq = d.T_src * V;
  const unsigned a_base = (unsigned)(uintptr_t)Zs + (unsigned)((l31 * RS + 8 * hi) * 2);
  typedef const u32x4 __attribute__((address_space(3))) * lds_u128;
  constexpr unsigned REJECT = 0xf0000000u;
  const int64_t dbytes = (int64_t)d.M * d.ld_dout * 4;
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.dout, 0, (unsigned)(dbytes < (int64_t)REJECT ? dbytes : (int64_t)REJECT), 0x00020000);
  unsigned drow[2];
  bool mrow_ok[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    mrow_ok[mb] = (m0 + 32 * mb + l31) < d.M;
    drow[mb] = (unsigned)(((int64_t)(mrow_ok[mb] ? m0 + 32 * mb + l31 : 0) * d.ld_dout) * 4);
  }

  // the raw tile of a tile: rows c0 .. c0 + 31 (wave w takes rows w, w + 4, ..), this lane's two adjacent positions
  auto issue_raw = [&](int tile_, float (&x)[8][2]) {
    const int b_ = tile_ / k.TPS;
    const int t0_ = (tile_ - b_ * k.TPS) * GFT;
    const int nlive_ = ((t0_ + GFT <= d.T_out) ? GFT : d.T_out - t0_) * V;
    const int n0_ = t0_ * V;
    const float* src_t = d.src + (int64_t)b_ * seq + n0_;
    const unsigned tb = (unsigned)(seq - n0_) * 4;   // the sequence's remaining bytes: positions beyond it are rejected -> 0
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = c0 + wave + 4 * q;   // wave-uniform: the row part of the address is a scalar descriptor
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src_t + (int64_t)(c < d.Kc ? c : 0) * d.ld_src), 0,
                                                                          c < d.Kc ? tb : 0u, 0x00020000);
#pragma unroll
      for (int e = 0; e < 2; ++e)
        x[q][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (2 * lane + e < nlive_) ? (2 * lane + e) * 4 : 0x7fffffff, 0, 0));
    }
  };
  float xn[8][2];          // S0ID: the NEXT tile's raw values, requested in front of this tile's k-steps (one exposed round trip less per tile)
  bool have_next = false;
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int b = tile / k.TPS;
    const int t0 = (tile - b * k.TPS) * GFT;
    const int nlive = ((t0 + GFT <= d.T_out) ? GFT : d.T_out - t0) * V;   // live positions of this tile
    const int n0 = t0 * V;
    __syncthreads();   // closing: every wave has read its last fragment of the previous tile (and csl is written)
    // the dout fragments of this wave's first two k-steps are requested here, in front of the stager: they land while the images
    // are built (conv_wgrad_split_kernel: a fragment requested one k-step ahead stalled every k-step)
    auto load_dout = [&](int ks, u32x4 (&raw)[2][2]) {
      const int pos = 16 * ks + 8 * hi;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const unsigned vo = (mrow_ok[mb] && pos < nlive) ? drow[mb] + (unsigned)(((int64_t)b * seq + n0 + pos) * 4) : REJECT;
        raw[mb][0] = __builtin_amdgcn_raw_buffer_load_b128(rdo, vo, 0, 0);
        raw[mb][1] = __builtin_amdgcn_raw_buffer_load_b128(rdo, vo, 16, 0);
      }
    };
    u32x4 raw[3][2][2];
    load_dout(kh, raw[0]);
    if (kh + WK < KS) load_dout(kh + WK, raw[1]);
    if constexpr (S0ID) {
    // ---- raw tile: rows c0 .. c0 + 31 (wave w takes rows w, w + 4, ..), this lane's two adjacent positions; every load first.  It
    // lives in the LDS area of slice 0's images: slice 0 is the identity (SAR_GRAPH_SLICE0_IDENTITY), so its images are the split of
    // the lane's OWN raw values, which stay in registers across the barrier that frees the raw tile.
    float* rawt = reinterpret_cast<float*>(&Zs[0]);
    float xr[8][2];
    {
      if (have_next) {   // requested in front of the previous tile's k-steps
#pragma unroll
        for (int q = 0; q < 8; ++q) xr[q][0] = xn[q][0], xr[q][1] = xn[q][1];
      } else {
        issue_raw(tile, xr);
      }
      if (2 * lane < GKP) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          rawt[(wave + 4 * q) * GRAW + 2 * lane] = xr[q][0];
          rawt[(wave + 4 * q) * GRAW + 2 * lane + 1] = xr[q][1];
        }
      }
    }
    __syncthreads();   // the raw tile is complete
    // ---- slices 1 and 2: gathered from the raw tile (a wave builds the rows it staged; the barrier above covers all rows)
    if (GW_ON(0))
#pragma unroll 1
    for (int q = 0; q < 8; ++q) {
      const int row = wave + 4 * q;
      const float* rr = rawt + row * GRAW;
#pragma unroll
      for (int kk = 1; kk < 3; ++kk) {
        float z[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float zz = gwt[kk][e][0] * rr[goff[kk][e][0]];
#pragma unroll
          for (int j = 1; j < 4; ++j)
            if (j < NZ[kk]) zz = fmaf(gwt[kk][e][j], rr[goff[kk][e][j]], zz);
          z[e] = (2 * lane + e < nlive) ? (ar_f16(AR) ? zz * sa : zz) : 0.f;   // (rows beyond Kc are zero in the raw tile)
        }
        unsigned w[NT];
        split2<AR, true>(z[0], z[1], w);
        if (2 * lane < RS) {
#pragma unroll
          for (int t = 0; t < ntl<AR>(); ++t) *reinterpret_cast<unsigned*>(&Zs[((kk * NT + t) * CB + row) * RS + 2 * lane]) = w[t];
        }
      }
    }
    __syncthreads();   // every wave has gathered from the raw tile: its area becomes slice 0's images
    if (2 * lane < RS && GW_ON(1)) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        unsigned w[NT];
        split2<AR, true>(ar_f16(AR) ? xr[q][0] * sa : xr[q][0], ar_f16(AR) ? xr[q][1] * sa : xr[q][1], w);
#pragma unroll
        for (int t = 0; t < ntl<AR>(); ++t) *reinterpret_cast<unsigned*>(&Zs[(t * CB + wave + 4 * q) * RS + 2 * lane]) = w[t];
      }
    }
    } else {
    // ---- build the three gathered images of rows c0 .. c0 + 31: wave w takes rows w, w + 4, ..; two rows per pass
    {
      const float* src_t = d.src + (int64_t)b * seq + n0;
#pragma unroll 1
      for (int rp = 0; rp < 8 / RPP; ++rp) {
        float x[RPP][3][2][4];
#pragma unroll
        for (int q = 0; q < RPP; ++q) {
          const int row = wave + 4 * (rp * RPP + q);
          const int c = c0 + row;
          const float* rowp_ = src_t + (int64_t)(c < d.Kc ? c : 0) * d.ld_src;
#pragma unroll
          for (int kk = 0; kk < 3; ++kk)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (j < NZ[kk]) x[q][kk][e][j] = (2 * lane + e < nlive) ? rowp_[goff[kk][e][j]] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < RPP; ++q) {
          const int row = wave + 4 * (rp * RPP + q);
          const bool rok = c0 + row < d.Kc;
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            float z[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              float zz = gwt[kk][e][0] * x[q][kk][e][0];
#pragma unroll
              for (int j = 1; j < 4; ++j)
                if (j < NZ[kk]) zz = fmaf(gwt[kk][e][j], x[q][kk][e][j], zz);
              z[e] = (rok && 2 * lane + e < nlive) ? (ar_f16(AR) ? zz * sa : zz) : 0.f;
            }
            unsigned w[NT];
            split2<AR, true>(z[0], z[1], w);
            if (2 * lane < RS) {
#pragma unroll
              for (int t = 0; t < ntl<AR>(); ++t) *reinterpret_cast<unsigned*>(&Zs[((kk * NT + t) * CB + row) * RS + 2 * lane]) = w[t];
            }
          }
        }
      }
    }
    }
    __syncthreads();   // opening: the images are complete
    have_next = false;
    if constexpr (S0ID) {
      if (tile + 1 < tile_hi) {
        issue_raw(tile + 1, xn);
        have_next = true;
      }
    }

    SAR_LDS_SKEW();
#pragma unroll 1
    for (int ks = kh; ks < KS; ks += 3 * WK) {
#pragma unroll
      for (int half = 0; half < 3; ++half) {   // a ring of three register sets, two k-steps ahead
        const int kc = ks + half * WK;
        if (kc < KS) {
          if (kc + 2 * WK < KS) load_dout(kc + 2 * WK, raw[(half + 2) % 3]);
          const int nv = nlive - (16 * kc + 8 * hi);   // live elements of this lane's fragment
          unsigned bw[2][NTB][4];
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            float dv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              dv[j] = __uint_as_float(raw[half][mb][j >> 2][j & 3]);
              dv[j] = j < nv ? dv[j] : 0.f;
            }
            if (do_bias) {
#pragma unroll
              for (int kk = 0; kk < 3; ++kk) {
                const float4 c0v = *reinterpret_cast<const float4*>(&csl[kk * KS * 16 + 16 * kc + 8 * hi]);
                const float4 c1v = *reinterpret_cast<const float4*>(&csl[kk * KS * 16 + 16 * kc + 8 * hi + 4]);
                float sacc = bsum[kk][mb];
                sacc = fmaf(dv[0], c0v.x, sacc), sacc = fmaf(dv[1], c0v.y, sacc), sacc = fmaf(dv[2], c0v.z, sacc), sacc = fmaf(dv[3], c0v.w, sacc);
                sacc = fmaf(dv[4], c1v.x, sacc), sacc = fmaf(dv[5], c1v.y, sacc), sacc = fmaf(dv[6], c1v.z, sacc), sacc = fmaf(dv[7], c1v.w, sacc);
                bsum[kk][mb] = sacc;
              }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              unsigned w[NTB] = {};
              if (GW_ON(2)) split2<AR, false>(dv[2 * p] * sb, dv[2 * p + 1] * sb, w);
#pragma unroll
              for (int t = 0; t < NTB; ++t) bw[mb][t][p] = w[t];
            }
          }
          const unsigned a_ks = a_base + kc * 32;
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            u32x4 aq[NT];
#pragma unroll
            for (int tm = 0; tm < ntl<AR>(); ++tm) aq[tm] = *(lds_u128)(uintptr_t)(a_ks + (kk * NT + tm) * (CB * RS * 2));
            if constexpr (ntl<AR>() < NT) aq[NT - 1] = third_image(aq[0]);
            if (GW_ON(3))
#pragma unroll
            for (int p = 0; p < NPROD; ++p) {
              const int i = ar_pi(AR, p), j = ar_pj(AR, p);
#pragma unroll
              for (int mb = 0; mb < 2; ++mb) {
                const u32x4 bq = u32x4{bw[mb][j][0], bw[mb][j][1], bw[mb][j][2], bw[mb][j][3]};
                if (ar_f16(AR))
                  acc[kk][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&aq[i]),
                                                                      *reinterpret_cast<const f16x8*>(&bq), acc[kk][mb], 0, 0, 0);
                else
                  acc[kk][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&aq[i]),
                                                                       *reinterpret_cast<const bf16x8*>(&bq), acc[kk][mb], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }

  float* slab = d.slab + (int64_t)(sg * WK + kh) * (d.wsize + d.bsize);
  const float unscale = (nonfin ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + eb)));
#pragma unroll
  for (int kk = 0; kk < 3; ++kk)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int m = m0 + 32 * mb + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + mfma_row(r, hi);
        if (c < d.Kc && m < d.M && GW_ON(4)) slab[(int64_t)kk * d.w_stride_tap + (int64_t)c * d.w_stride_c + m] = acc[kk][mb][r] * unscale;
      }
      if (do_bias) {
        const float t = bsum[kk][mb] + __shfl_xor(bsum[kk][mb], 32);
        if (hi == 0 && m < d.M) slab[d.wsize + kk * d.M + m] = t;
      }
    }
}

// ---- The 9-tap temporal weight gradient at stride 1, second design ("ring"): frames at a pitch of 32 positions in LDS.
// tools/wsplit_timeline.py on conv_wgrad_split_kernel: a k-step takes 2 800 cycles for 864 of matrix work and ~250 instructions --
// V = 25 makes a tap shift 50 bytes, so every src fragment is read as five dwords and funnel-shifted (60 ds_read2 + 48 shifts + 37
// address adds per k-step): the vector issue port, not latency.  Here a frame occupies 32 positions of the LDS row (joints 0-24,
// then 7 zeros): a tap is a shift by whole frames = a multiple of 64 bytes, a src fragment ONE ds_read_b128, the 16 k of an MFMA =
// half a padded frame (22 % of the matrix work multiplies zeros -- bought back several times by the instruction count).  The row is a
// RING of 12 frames: a tile is 4 dout frames, its nine taps reach frames F - pad .. F - pad + 11, and the next tile (F + 4) replaces
// only the 4 oldest frames -- the stager stages 100 new positions per 100 dout positions (the flat window staged 392 per 192); a new
// sequence (or the first tile of a workgroup) primes all 12.  Slot of shifted frame g' = g + pad: g' mod 12; tap t of dout frame f reads
// slot (f + t) mod 12, whatever the padding.  The new frames of tile i + 1 are REQUESTED before the k-steps of tile i (16 registers).
constexpr int RING_F = 12, RING_FP = 32, RING_TF = 4, RING_RS = RING_F * RING_FP + 8;   // row: 392 elements = 49 units of 16 B (odd: conflict-free b128 reads)
// S2 (stride 2, T_src = 2 T_out): dout frame f reads src frames 2 f + t - pad -- shifted frame 2 f + t, slot (2 f + t) mod 12; a tile is
// TWO dout frames (4 k-steps: its taps reach 11 src frames), the next tile again replaces the 4 oldest src frames.
template <int AR, int WK, int S2 = 0>
__global__ __launch_bounds__(256, 2) void conv_wgrad_ring_kernel(const WgradKS k) {
  constexpr int NT = ar_nta(AR), NTB = ar_ntb(AR), NPROD = ar_nprod(AR), V = VJ, RS = RING_RS, TFR = S2 ? 2 : RING_TF, KS = 2 * TFR, FS = S2 ? 2 : 1;
  constexpr int WMM = 4 / WK, MBLK = 32 * WMM;
  __shared__ __attribute__((aligned(16))) unsigned short Hs[NT * CB * RS];
  __shared__ float2 bnp[CB];
  const sar_wgrad_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wmm = wave % WMM, kh = wave / WMM;

  int sg, by, bz;
  {
    const int nyz = k.gy * k.gz, ngrp = d.nsplit / WK, nwork = ngrp * nyz;
    const int per = (nwork + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int w = xcd * per + slot;
    if (slot >= per || w >= nwork) return;
    sg = w / nyz;
    const int yz = w - sg * nyz;
    bz = yz / k.gy;
    by = yz - bz * k.gy;
  }
  const int m0 = by * MBLK + wmm * 32, c0 = bz * CB;
  const int ngrp = d.nsplit / WK;

  int ea = 0, eb = 0;
  bool nonfin = false;   // an operand bound holds Inf / NaN bits: every output of the launch is NaN (split_scale.h)
  if (ar_f16(AR)) {
    ea = scale_exp(*k.src_bound);
    eb = scale_exp(*k.dout_bound);
    nonfin = bound_nonfinite(*k.src_bound) || bound_nonfinite(*k.dout_bound);
  }
  const float sa = __builtin_ldexpf(1.f, ea), sb = __builtin_ldexpf(1.f, eb);
  {
    float2 p = make_float2(sa, 0.f);
    if (tid < CB && d.pro_scale && c0 + tid < d.Kc) p = make_float2(d.pro_scale[c0 + tid] * sa, d.pro_shift[c0 + tid] * sa);
    if (tid < CB) bnp[tid] = p;
  }
  for (int i = tid; i < NT * CB * (RS / 2); i += 256) reinterpret_cast<unsigned*>(Hs)[i] = 0u;   // pad slots / row tails stay zero

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = d.bsize > 0 && bz == 0;

  const int tps = (k.ntiles + ngrp - 1) / ngrp;
  const int tile_lo = sg * tps;
  const int tile_hi = (tile_lo + tps < k.ntiles) ? tile_lo + tps : k.ntiles;
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  const int T = d.T_out, Ts = d.T_src;   // stride 1: T_src == T_out; S2: T_src = 2 T_out
  const int seq = T * V, seq_s = Ts * V;

  typedef const u32x4 __attribute__((address_space(3))) * lds_u128;
  const unsigned a_base = (unsigned)(uintptr_t)Hs + (unsigned)((l31 * RS + 8 * hi) * 2);
  const bool mrow_ok = (m0 + l31) < d.M;
  constexpr unsigned REJECT = 0xf0000000u;
  const int64_t dbytes = (int64_t)d.M * d.ld_dout * 4;
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.dout, 0, (unsigned)(dbytes < (int64_t)REJECT ? dbytes : (int64_t)REJECT), 0x00020000);
  const unsigned drow = (unsigned)(((int64_t)(mrow_ok ? m0 + l31 : 0) * d.ld_dout) * 4);

  // ---- the stager: one group = 4 shifted frames gp .. gp + 3 (128 row elements: a lane owns two adjacent ones), 8 channel rows per wave
  const int s_fr = lane >> 4, s_j = 2 * (lane & 15);          // frame of the group, joint (even) of this lane's pair
  auto issue_group = [&](int b, int gp, float (&x)[8][2]) {
    const int fr = gp + s_fr - d.pad;                          // real src frame
    const bool fok = (unsigned)fr < (unsigned)Ts;
    const float* src_b = d.src + (int64_t)b * seq_s;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = c0 + wave + 4 * q;
      const int cg = c < d.Kc ? c : 0;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, seq_s * 4, 0x00020000);
      const unsigned o0 = (fok && s_j < V) ? (unsigned)((fr * V + s_j) * 4) : REJECT, o1 = (fok && s_j + 1 < V) ? (unsigned)((fr * V + s_j + 1) * 4) : REJECT;
      x[q][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o0, 0, 0));   // rejected -> 0
      x[q][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o1, 0, 0));
    }
  };
  auto store_group = [&](int gp, const float (&x)[8][2]) {
    const int fr = gp + s_fr - d.pad;
    const bool fok = (unsigned)fr < (unsigned)Ts;
    const int slot = (gp + s_fr) % RING_F;
    const int col = slot * RING_FP + s_j;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = wave + 4 * q;
      const bool rok = c0 + row < d.Kc;
      const float2 ps = bnp[row];
      const float v0 = (rok && fok && s_j < V) ? fmaxf(fmaf(x[q][0], ps.x, ps.y), relu_lo) : 0.f;   // TF-SAME padding / pad joints stay exactly 0
      const float v1 = (rok && fok && s_j + 1 < V) ? fmaxf(fmaf(x[q][1], ps.x, ps.y), relu_lo) : 0.f;
      unsigned w[NT];
      split2<AR, true>(v0, v1, w);
#pragma unroll
      for (int t = 0; t < ntl<AR>(); ++t) *reinterpret_cast<unsigned*>(&Hs[(t * CB + row) * RS + col]) = w[t];
    }
  };

  float xn[8][2];          // the new frames of the NEXT tile, in flight during this tile's k-steps
  bool have_next = false;
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int b = tile / k.TPS;
    const int F = (tile - b * k.TPS) * TFR;      // first dout frame; FS F = its shifted src frame
    const bool prime = tile == tile_lo || F == 0;
    __syncthreads();   // closing: every wave has read its last fragment of the previous tile (first tile: the zero fill, bnp)
    auto load_dout = [&](int kc, u32x4 (&raw)[2]) {   // k-step kc: dout frame F + kc / 2, joints 16 (kc & 1) + 8 hi ..
      const int f = F + (kc >> 1), j0 = 16 * (kc & 1) + 8 * hi;
      const unsigned vo = (mrow_ok && f < T && j0 < V) ? drow + (unsigned)(((int64_t)b * seq + f * V + j0) * 4) : REJECT;
      raw[0] = __builtin_amdgcn_raw_buffer_load_b128(rdo, vo, 0, 0);
      raw[1] = __builtin_amdgcn_raw_buffer_load_b128(rdo, vo, 16, 0);
    };
    u32x4 raw[2][2];
    load_dout(kh, raw[0]);
    if (prime) {
#pragma unroll 1
      for (int g = 0; g < 3; ++g) {
        float x[8][2];
        issue_group(b, FS * F + 4 * g, x);
        store_group(FS * F + 4 * g, x);
      }
    } else {
      if (!have_next) issue_group(b, FS * F + 8, xn);
      store_group(FS * F + 8, xn);
    }
    have_next = false;
    __syncthreads();   // opening: the ring holds shifted frames FS F .. FS F + 11
    if (tile + 1 < tile_hi) {   // the next tile's new frames (if it continues this sequence)
      const int b1 = (tile + 1) / k.TPS, F1 = (tile + 1 - b1 * k.TPS) * TFR;
      if (F1 != 0) {
        issue_group(b1, FS * F1 + 8, xn);
        have_next = true;
      }
    }
    SAR_LDS_SKEW();
#pragma unroll 1
    for (int ks = kh; ks < KS; ks += 2 * WK) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int kc = ks + half * WK;
        if (kc < KS) {
          if (kc + WK < KS) load_dout(kc + WK, raw[half ^ 1]);
          const int f = F + (kc >> 1), j0 = 16 * (kc & 1) + 8 * hi;
          float dv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) dv[j] = __uint_as_float(raw[half][j >> 2][j & 3]);
          if ((kc & 1) && hi) {   // joints 24 .. 31: only joint 24 exists (the load ran into the next frame)
#pragma unroll
            for (int j = 1; j < 8; ++j) dv[j] = 0.f;
          }
          (void)j0;
          if (do_bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += dv[j];
          }
          unsigned bw[NTB][4];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            unsigned w[NTB];
            split2<AR, false>(dv[2 * p] * sb, dv[2 * p + 1] * sb, w);
#pragma unroll
            for (int t = 0; t < NTB; ++t) bw[t][p] = w[t];
          }
          // tap t reads slot (f + t) mod 12 of the ring, half frame kc & 1
          const int s0 = (FS * f) % RING_F;
          const unsigned a_ks = a_base + (unsigned)((kc & 1) * 32);
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            int st = s0 + t;
            st = st >= RING_F ? st - RING_F : st;
            const unsigned a_t = a_ks + (unsigned)(st * (RING_FP * 2));
            u32x4 aq[NT];
#pragma unroll
            for (int tm = 0; tm < ntl<AR>(); ++tm) aq[tm] = *(lds_u128)(uintptr_t)(a_t + tm * (CB * RS * 2));
            if constexpr (ntl<AR>() < NT) aq[NT - 1] = third_image(aq[0]);
#pragma unroll
            for (int p = 0; p < NPROD; ++p) {
              const int i = ar_pi(AR, p), j = ar_pj(AR, p);
              const u32x4 bq = u32x4{bw[j][0], bw[j][1], bw[j][2], bw[j][3]};
              if (ar_f16(AR))
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&aq[i]),
                                                                *reinterpret_cast<const f16x8*>(&bq), acc[t], 0, 0, 0);
              else
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&aq[i]),
                                                                 *reinterpret_cast<const bf16x8*>(&bq), acc[t], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  float* slab = d.slab + (int64_t)(sg * WK + kh) * (d.wsize + d.bsize);
  const float unscale = (nonfin ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + eb)));
  const int m = m0 + l31;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + mfma_row(r, hi);
      if (c < d.Kc && m < d.M) slab[(int64_t)t * d.w_stride_tap + (int64_t)c * d.w_stride_c + m] = acc[t][r] * unscale;
    }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32);
    if (hi == 0 && m < d.M) slab[d.wsize + m] = bsum;
  }
}

// ---- The 1-tap TEMPORAL weight gradient (the strided 1x1 residual convolution, models/stgcn.py:47-54) in the split arithmetics -- round 6:
// the last fp32 GEMM launch of the split engine's step besides the 3-channel layer (tools/leftover_bench.py: 176 / 271 us alone at
// 64 -> 128 / 128 -> 256 channels for 82 us of HBM time; 15.7 GFLOP are 100 us of fp32 matrix time).
//   dW[c][m] = sum_n src[c, s(n)] dout[m, n],  dbias[m] = sum_n dout[m, n];   n = (b, t, v) flat,  s(n) = n + (stride - 1) V floor(n / V)
// (T_src == stride T_out: the source frame of flat output frame f is frame stride f of the flat source).  Without taps a tile is just
// KT1 consecutive flat positions: workgroup = 64 src channels x 128 dout channels, the src tile through LDS ([term][64 rows][KT1] 2-byte
// elements, converted by the stager: a lane owns two adjacent positions of a row, their sources are adjacent unless a frame ends between
// them), wave w owns dout channels 32 w .. 32 w + 31 of the block: its dout fragments come straight from global memory, two k-steps
// ahead, and are summed for the bias gradient and converted in registers (conv_wgrad_split_kernel).  Two barriers per tile.
constexpr int KT1 = 128, KS1 = KT1 / 16, RS1 = KT1 + 8, CB1 = 64;   // row stride 272 bytes: 16-byte fragment reads of 32 rows hit distinct banks
template <int AR>
__global__ __launch_bounds__(256, 3) void wgrad_tap1_split_kernel(const WgradKS k) {
  constexpr int NT = ar_nta(AR), NTL = ntl<AR>(), NTB = ar_ntb(AR), NPROD = ar_nprod(AR), V = VJ;
  __shared__ __attribute__((aligned(16))) unsigned short Hs[NTL * CB1 * RS1];
  const sar_wgrad_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int sg, by, bz;
  {
    const int nyz = k.gy * k.gz, nwork = d.nsplit * nyz;
    const int per = (nwork + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int w = xcd * per + slot;
    if (slot >= per || w >= nwork) return;
    sg = w / nyz;
    const int yz = w - sg * nyz;
    bz = yz / k.gy;
    by = yz - bz * k.gy;
  }
  const int m0 = by * 128 + wave * 32, c0 = bz * CB1;
  int ea = 0, eb = 0;
  bool nonfin = false;
  if (ar_f16(AR)) {
    ea = scale_exp(*k.src_bound);
    eb = scale_exp(*k.dout_bound);
    nonfin = bound_nonfinite(*k.src_bound) || bound_nonfinite(*k.dout_bound);
  }
  const float sa = __builtin_ldexpf(1.f, ea), sb = __builtin_ldexpf(1.f, eb);
  const int64_t npos = (int64_t)d.B * d.T_out * V;          // flat output positions
  const int ntl_ = k.ntiles;                                 // ceil(npos / KT1)
  const int tps = (ntl_ + d.nsplit - 1) / d.nsplit;
  const int tile_lo = sg * tps;
  const int tile_hi = (tile_lo + tps < ntl_) ? tile_lo + tps : ntl_;
  const int smul = (d.stride - 1) * V;
  const unsigned src_bytes = (unsigned)((int64_t)d.B * d.T_src * V * 4);   // one row of the flat source (< 2^32: checked by the host)

  f32x16 acc[2];
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ms][r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = d.bsize > 0 && bz == 0;   // uniform
  constexpr unsigned REJECT = 0xf0000000u;
  const int64_t dbytes = (int64_t)d.M * d.ld_dout * 4;
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc((void*)d.dout, 0, (unsigned)(dbytes < (int64_t)REJECT ? dbytes : (int64_t)REJECT), 0x00020000);
  const bool mrow_ok = (m0 + l31) < d.M;
  const unsigned drow = (unsigned)(((int64_t)(mrow_ok ? m0 + l31 : 0) * d.ld_dout) * 4);
  // stager: thread -> (position pair 2 (tid & 63), rows (tid >> 6) + 4 q, q < 16)
  const int sp = 2 * (tid & 63), sr0 = tid >> 6;
  typedef const u32x4 __attribute__((address_space(3))) * lds_u128;
  const unsigned a_base = (unsigned)(uintptr_t)Hs + (unsigned)((l31 * RS1 + 8 * hi) * 2);

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int64_t n0 = (int64_t)tile * KT1;
    const int nlive = (int)((npos - n0 < KT1) ? npos - n0 : KT1);
    // the dout fragments of the first two k-steps, requested in front of the stager
    auto load_dout = [&](int ks, u32x4 (&raw)[2]) {
      const int pos = 16 * ks + 8 * hi;
      const unsigned vo = (mrow_ok && pos < nlive) ? drow + (unsigned)((n0 + pos) * 4) : REJECT;
      raw[0] = __builtin_amdgcn_raw_buffer_load_b128(rdo, vo, 0, 0);
      raw[1] = __builtin_amdgcn_raw_buffer_load_b128(rdo, vo, 16, 0);
    };
    u32x4 raw[3][2];
    load_dout(0, raw[0]);
    load_dout(1, raw[1]);
    {   // src tile: 64 rows x KT1 positions, two adjacent positions per lane
      const int64_t na = n0 + sp, nb = na + 1;
      const int64_t sa_ = na + (int64_t)smul * (na / V), sb_ = nb + (int64_t)smul * (nb / V);
      const bool la = sp < nlive, lb = sp + 1 < nlive;
      float x[16][2];
      const unsigned oa = la ? (unsigned)(sa_ * 4) : 0x7fffffffu, ob = lb ? (unsigned)(sb_ * 4) : 0x7fffffffu;   // dead positions: rejected -> 0
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int c = c0 + sr0 + 4 * q;   // wave-uniform: the row is a scalar descriptor (buffer loads: no FLAT loads beside the LDS traffic)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(d.src + (int64_t)(c < d.Kc ? c : 0) * d.ld_src), 0,
                                                                            c < d.Kc ? src_bytes : 0u, 0x00020000);
        x[q][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, oa, 0, 0));
        x[q][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, ob, 0, 0));
      }
      __syncthreads();   // closing: every wave has read its last fragment of the previous tile
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        unsigned w[NT];
        split2<AR, true>(ar_f16(AR) ? x[q][0] * sa : x[q][0], ar_f16(AR) ? x[q][1] * sa : x[q][1], w);
#pragma unroll
        for (int t = 0; t < NTL; ++t) *reinterpret_cast<unsigned*>(&Hs[(t * CB1 + sr0 + 4 * q) * RS1 + sp]) = w[t];
      }
    }
    __syncthreads();     // opening: the images are complete
    SAR_LDS_SKEW();
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      if (ks + 2 < KS1) load_dout(ks + 2, raw[(ks + 2) % 3]);
      const int nv = nlive - (16 * ks + 8 * hi);
      float dv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        dv[j] = __uint_as_float(raw[ks % 3][j >> 2][j & 3]);
        dv[j] = j < nv ? dv[j] : 0.f;
      }
      if (do_bias) bsum += ((dv[0] + dv[1]) + (dv[2] + dv[3])) + ((dv[4] + dv[5]) + (dv[6] + dv[7]));
      unsigned bw[NTB][4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        unsigned w[NTB];
        split2<AR, false>(dv[2 * p] * sb, dv[2 * p + 1] * sb, w);
#pragma unroll
        for (int t = 0; t < NTB; ++t) bw[t][p] = w[t];
      }
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        u32x4 aq[NT];
#pragma unroll
        for (int t = 0; t < NTL; ++t) aq[t] = *(lds_u128)(uintptr_t)(a_base + (unsigned)(((t * CB1 + ms * 32) * RS1 + 16 * ks) * 2));
        if constexpr (NTL < NT) aq[NT - 1] = third_image(aq[0]);
#pragma unroll
        for (int p = 0; p < NPROD; ++p) {
          const int i = ar_pi(AR, p), j = ar_pj(AR, p);
          const u32x4 bq = u32x4{bw[j][0], bw[j][1], bw[j][2], bw[j][3]};
          if (ar_f16(AR))
            acc[ms] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&aq[i]), *reinterpret_cast<const f16x8*>(&bq), acc[ms], 0, 0, 0);
          else
            acc[ms] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&aq[i]), *reinterpret_cast<const bf16x8*>(&bq), acc[ms], 0, 0, 0);
        }
      }
    }
  }

  float* slab = d.slab + (int64_t)sg * (d.wsize + d.bsize);
  const float unscale = (nonfin ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + eb)));
  const int m = m0 + l31;
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + ms * 32 + mfma_row(r, hi);
      if (c < d.Kc && m < d.M) slab[(int64_t)c * d.w_stride_c + m] = acc[ms][r] * unscale;
    }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32);
    if (hi == 0 && m < d.M) slab[d.wsize + m] = bsum;
  }
}

// SAR_WGRAD_RING=0: the first design (flat window, conv_wgrad_split_kernel) for the stride-1 temporal weight gradient (A/B runs)
bool wgrad_ring_on() {
  static const bool on = [] { const char* e = getenv("SAR_WGRAD_RING"); return !e || atoi(e) != 0; }();
  return on;
}

// which kernel: WK (1 or 2), or 0 = not built (the caller keeps sar_conv_wgrad_f32)
int wgrad_split_wk(const sar_wgrad_desc& d, int arith) {
  if (arith != AR_B6 && arith != AR_H3A) return 0;
  if (d.mode == SAR_CONV_GRAPH) {   // 3 slices, no folded prologue, <= 4-entry gather lists; the 3-channel first layer keeps its streaming kernel
    if (d.taps != 3 || d.V != VJ || d.T_src != d.T_out || d.pro_scale || d.Kc < 16 || d.Kc > 256 || !d.g_idx || !d.g_wt) return 0;
    for (int i = 0; i < 3; ++i)
      if (d.nz[i] < 1 || d.nz[i] > 4) return 0;
    return d.M > 128 ? 1 : (d.M > 64 ? 2 : 4);
  }
  if (d.mode == SAR_CONV_TEMPORAL && d.taps == 1) {   // wgrad_tap1_split_kernel (wk 1): no prologue, pad 0, whole strides
    if (d.V != VJ || d.pad != 0 || d.pro_scale || d.stride < 1 || d.T_src != d.stride * d.T_out || d.Kc < 16) return 0;
    if ((int64_t)d.B * d.T_src * d.V >= (1ll << 30)) return 0;
    return 1;
  }
  if (d.mode != SAR_CONV_TEMPORAL || d.taps != TAPS || d.V != VJ) return 0;
  if (!((d.stride == 1 && d.T_src == d.T_out) || (d.stride == 2 && d.pad == 3 && d.T_src == 2 * d.T_out))) return 0;
  if (d.Kc < 8 || d.Kc > 256 || d.pad < 0 || d.pad > 8) return 0;
  return d.M > 64 ? 1 : 2;
}

template <int AR, int NZ0, int NZ1, int NZ2, bool S0ID>
void launch_graph_wgrad_split(const WgradKS& k, int wk, dim3 grid, hipStream_t st) {
  if (wk == 1) hipLaunchKernelGGL((graph_wgrad_split_kernel<AR, 1, NZ0, NZ1, NZ2, S0ID>), grid, dim3(256), 0, st, k);
  else if (wk == 2) hipLaunchKernelGGL((graph_wgrad_split_kernel<AR, 2, NZ0, NZ1, NZ2, S0ID>), grid, dim3(256), 0, st, k);
  else hipLaunchKernelGGL((graph_wgrad_split_kernel<AR, 4, NZ0, NZ1, NZ2, S0ID>), grid, dim3(256), 0, st, k);
}

template <int AR>
int launch_wgrad_split(const sar_wgrad_desc& d, int wk, const unsigned* sb, const unsigned* db, hipStream_t st) {
  WgradKS k;
  k.d = d;
  k.src_bound = sb;
  k.dout_bound = db;
  k.ablate = 0;
#ifdef SAR_GW_ABLATE
  {
    const char* e = getenv("SAR_GW_ABLATE_BITS");
    k.ablate = e ? atoi(e) : 0;
  }
#endif
  if (d.mode == SAR_CONV_GRAPH) {
    k.TPS = (d.T_out + GFT - 1) / GFT;
    k.ntiles = d.B * k.TPS;
    k.gy = (d.M + 256 / wk - 1) / (256 / wk);
    k.gz = (d.Kc + CB - 1) / CB;
    const int nwork = (d.nsplit / wk) * k.gy * k.gz;
    const dim3 grid(((nwork + 7) / 8) * 8);
    const bool s0id = (d.g_flags & SAR_GRAPH_SLICE0_IDENTITY) != 0 && d.nz[0] == 1;   // the raw tile through LDS (one trip per tile)
    if (d.nz[0] == 1 && d.nz[1] == 1) s0id ? launch_graph_wgrad_split<AR, 1, 1, 4, true>(k, wk, grid, st) : launch_graph_wgrad_split<AR, 1, 1, 4, false>(k, wk, grid, st);
    else if (d.nz[0] == 1 && d.nz[2] == 1) s0id ? launch_graph_wgrad_split<AR, 1, 4, 1, true>(k, wk, grid, st) : launch_graph_wgrad_split<AR, 1, 4, 1, false>(k, wk, grid, st);
    else launch_graph_wgrad_split<AR, 4, 4, 4, false>(k, wk, grid, st);
    return 0;
  }
  if (d.taps == 1) {
    const int64_t npos = (int64_t)d.B * d.T_out * d.V;
    k.TPS = 0;
    k.ntiles = (int)((npos + KT1 - 1) / KT1);
    k.gy = (d.M + 127) / 128;
    k.gz = (d.Kc + CB1 - 1) / CB1;
    const int nwork1 = d.nsplit * k.gy * k.gz;
    hipLaunchKernelGGL((wgrad_tap1_split_kernel<AR>), dim3(((nwork1 + 7) / 8) * 8), dim3(256), 0, st, k);
    return 0;
  }
  const int seq = d.T_out * d.V;
  const int kt = d.stride == 2 ? Cfg<AR, 1>::KT : Cfg<AR, 0>::KT;
  k.TPS = (seq + kt - 1) / kt;
  k.ntiles = d.B * k.TPS;
  k.gy = (d.M + 128 / wk - 1) / (128 / wk);
  k.gz = (d.Kc + CB - 1) / CB;
  const int nwork = (d.nsplit / wk) * k.gy * k.gz;
  const dim3 grid(((nwork + 7) / 8) * 8), block(256);
  if (wgrad_ring_on()) {   // frames at pitch 32 in a ring of 12 (conv_wgrad_ring_kernel): tiles of 4 dout frames (stride 2: 2)
    const int tf = d.stride == 2 ? 2 : RING_TF;
    k.TPS = (d.T_out + tf - 1) / tf;
    k.ntiles = d.B * k.TPS;
    if (d.stride == 2) {
      if (wk == 1) hipLaunchKernelGGL((conv_wgrad_ring_kernel<AR, 1, 1>), grid, block, 0, st, k);
      else hipLaunchKernelGGL((conv_wgrad_ring_kernel<AR, 2, 1>), grid, block, 0, st, k);
    } else if (wk == 1) hipLaunchKernelGGL((conv_wgrad_ring_kernel<AR, 1, 0>), grid, block, 0, st, k);
    else hipLaunchKernelGGL((conv_wgrad_ring_kernel<AR, 2, 0>), grid, block, 0, st, k);
    return 0;
  }
  if (d.stride == 2) {
    if (wk == 1) hipLaunchKernelGGL((conv_wgrad_split_kernel<AR, 1, 1>), grid, block, 0, st, k);
    else hipLaunchKernelGGL((conv_wgrad_split_kernel<AR, 2, 1>), grid, block, 0, st, k);
  } else if (wk == 1) hipLaunchKernelGGL((conv_wgrad_split_kernel<AR, 1, 0>), grid, block, 0, st, k);
  else hipLaunchKernelGGL((conv_wgrad_split_kernel<AR, 2, 0>), grid, block, 0, st, k);
  return 0;
}

}  // namespace

#ifdef SAR_SPLIT_TL
extern "C" int sar_debug_wsplit_timeline(unsigned* out, int nwg, int reset) {   // out: [nwg][16] (host memory)
  if (nwg > WSPLIT_TL_WG) nwg = WSPLIT_TL_WG;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wsplit_tl), (size_t)nwg * 16 * sizeof(unsigned)) != hipSuccess) return -1;
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wsplit_tl)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned) * 16 * WSPLIT_TL_WG) != hipSuccess) return -1;
  }
  return nwg;
}
#endif

extern "C" int sar_conv_wgrad_split_blocks(const sar_wgrad_desc* d, int arith, int* wk_out, int* tile_positions) {
  if (!d) return SAR_E_ARG;
  const int wk = wgrad_split_wk(*d, arith);
  if (!wk) return SAR_E_UNSUP;
  if (wk_out) *wk_out = wk;
  if (d->mode == SAR_CONV_GRAPH) {
    if (tile_positions) *tile_positions = GKP;
    return ((d->M + 256 / wk - 1) / (256 / wk)) * ((d->Kc + CB - 1) / CB);
  }
  if (d->taps == 1) {
    if (tile_positions) *tile_positions = KT1;
    return ((d->M + 127) / 128) * ((d->Kc + CB1 - 1) / CB1);
  }
  if (tile_positions) *tile_positions = wgrad_ring_on() ? (d->stride == 2 ? 2 : RING_TF) * VJ : (d->stride == 2 ? Cfg<AR_H3A, 1>::KT : Cfg<AR_H3A, 0>::KT);
  return ((d->M + 128 / wk - 1) / (128 / wk)) * ((d->Kc + CB - 1) / CB);
}

extern "C" int sar_conv_wgrad_split(const sar_wgrad_desc* d, int arith, const uint32_t* src_bound, const uint32_t* dout_bound,
                                    sar_stream_t s) {
  SAR_REQUIRE(d != nullptr, "sar_conv_wgrad_split: null descriptor");
  const int wk = wgrad_split_wk(*d, arith);
  if (!wk) {
    sar_set_error("sar_conv_wgrad_split: built for the 9-tap temporal convolution at V = 25, stride 1 (or 2 with pad 3, even T), 8 <= Kc <= 256 and the graph "
                  "convolution at V = 25, 16 <= Kc <= 256 without a folded prologue, and the 1-tap temporal convolution at V = 25, pad 0, T_src = stride T_out, Kc >= 16, no prologue, "
                  "in the arithmetics bf16x6 / f16x3a (mode %d, taps %d, V %d, stride %d, Kc %d, arith %d): use sar_conv_wgrad_f32",
                  d->mode, d->taps, d->V, d->stride, d->Kc, arith);
    return SAR_E_UNSUP;
  }
  SAR_REQUIRE(d->B > 0 && d->T_out > 0 && d->M > 0, "sar_conv_wgrad_split: bad sizes");
  SAR_REQUIRE(d->src && d->dout && d->slab, "sar_conv_wgrad_split: null src/dout/slab");
  SAR_REQUIRE(d->nsplit >= wk && d->nsplit % wk == 0 && d->nsplit <= 65535,
              "sar_conv_wgrad_split: nsplit %d must be a positive multiple of %d", d->nsplit, wk);
  SAR_REQUIRE(d->ld_src >= (int64_t)d->B * d->T_src * d->V && d->ld_dout >= (int64_t)d->B * d->T_out * d->V,
              "sar_conv_wgrad_split: leading dimension smaller than B*T*V");
  SAR_REQUIRE((int64_t)d->T_src * d->V < (1 << 28), "sar_conv_wgrad_split: sequence row too long");
  SAR_REQUIRE((int64_t)d->M * d->ld_dout * 4 < 0xf0000000ll, "sar_conv_wgrad_split: dout larger than 3.75 GiB");
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv_wgrad_split: pro_scale/pro_shift mismatch");
  SAR_REQUIRE(d->wsize > 0 && (d->bsize == 0 || d->bsize == (d->mode == SAR_CONV_GRAPH ? 3 : 1) * (int64_t)d->M),
              "sar_conv_wgrad_split: bad slab sizes");
  SAR_REQUIRE(d->mode != SAR_CONV_GRAPH || d->bsize == 0 || d->g_colsum, "sar_conv_wgrad_split: the graph bias gradient needs g_colsum");
  SAR_REQUIRE(arith != AR_H3A || (src_bound && dout_bound), "sar_conv_wgrad_split: the fp16 arithmetic needs the operand bounds");
  if (arith == AR_H3A) launch_wgrad_split<AR_H3A>(*d, wk, src_bound, dout_bound, as_stream(s));
  else launch_wgrad_split<AR_B6>(*d, wk, src_bound, dout_bound, as_stream(s));
  SAR_LAUNCH_CHECK("sar_conv_wgrad_split");
  return 0;
}

#else   // ---------------------------------------------------------------------------- part 1 / 2: the resnet's 3x3 / stride-1 convolutions
}  // namespace
// dW[kh][kw][c][m] = sum_{b,h,w} pro(src)[c, (b, h + kh - 1, w + kw - 1)] * dout[m, (b, h, w)]   (models/resnet18.py:5-14 backward).
// Part 1 holds the f16x3a instantiations and the entry points, part 2 the bf16x6 ones (two compiler jobs).
int sar_c2d_wsplit_b6(const WgradKS& k, int wk, int W, dim3 grid, hipStream_t st);

namespace {
template <int AR, int W2>
void launch_c2d_w(const WgradKS& k, int wk, dim3 grid, hipStream_t st) {
  if (wk == 1) hipLaunchKernelGGL((conv_wgrad_split_kernel<AR, 1, 0, W2>), grid, dim3(256), 0, st, k);
  else hipLaunchKernelGGL((conv_wgrad_split_kernel<AR, 2, 0, W2>), grid, dim3(256), 0, st, k);
}
template <int AR>
int launch_c2d(const WgradKS& k, int wk, int W, dim3 grid, hipStream_t st) {
  switch (W) {
    case 8: launch_c2d_w<AR, 8>(k, wk, grid, st); return 0;
    case 16: launch_c2d_w<AR, 16>(k, wk, grid, st); return 0;
    case 32: launch_c2d_w<AR, 32>(k, wk, grid, st); return 0;
    case 64: launch_c2d_w<AR, 64>(k, wk, grid, st); return 0;
  }
  return SAR_E_UNSUP;
}
}  // namespace
#if SAR_WSPLIT_PART == 2
int sar_c2d_wsplit_b6(const WgradKS& k, int wk, int W, dim3 grid, hipStream_t st) { return launch_c2d<AR_B6>(k, wk, W, grid, st); }
#else
namespace {
// WK (1 or 2) of a descriptor, or 0 = not built (the caller keeps sar_conv2d_wgrad_f32)
int c2d_wsplit_wk(const sar_conv2d_desc& d, int arith) {
  if (arith != AR_B6 && arith != AR_H3A) return 0;
  if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.H_src != d.H_out || d.W_src != d.W_out) return 0;
  const int W = d.W_out;
  if (W != 8 && W != 16 && W != 32 && W != 64) return 0;
  if (d.Kc < 8 || d.M < 8 || d.H_out < 1 || (int64_t)d.B * d.H_out * W >= (1 << 22)) return 0;
  return d.M > 64 ? 1 : 2;
}
}  // namespace

extern "C" int sar_conv2d_wgrad_split_blocks(const sar_conv2d_desc* d, int arith, int* wk_out, int* tile_positions) {
  if (!d) return SAR_E_ARG;
  const int wk = c2d_wsplit_wk(*d, arith);
  if (!wk) return SAR_E_UNSUP;
  if (wk_out) *wk_out = wk;
  if (tile_positions) *tile_positions = Cfg<AR_H3A, 0, 8>::KT;
  return ((d->M + 128 / wk - 1) / (128 / wk)) * ((d->Kc + CB - 1) / CB);
}

extern "C" int sar_conv2d_wgrad_split(const sar_conv2d_desc* d, int arith, const uint32_t* src_bound, const uint32_t* dout_bound,
                                      sar_stream_t s) {
  SAR_REQUIRE(d != nullptr, "sar_conv2d_wgrad_split: null descriptor");
  const int wk = c2d_wsplit_wk(*d, arith);
  if (!wk) {
    sar_set_error("sar_conv2d_wgrad_split: built for 3x3 / stride 1 / pad 1 on images of width 8 / 16 / 32 / 64, Kc >= 8, M >= 8, in the "
                  "arithmetics bf16x6 / f16x3a (%dx%d, stride %d, pad %d, %dx%d, Kc %d, M %d, arith %d): use sar_conv2d_wgrad_f32",
                  d->KH, d->KW, d->stride, d->pad, d->H_out, d->W_out, d->Kc, d->M, arith);
    return SAR_E_UNSUP;
  }
  const int64_t npos = (int64_t)d->B * d->H_out * d->W_out;
  SAR_REQUIRE(d->B > 0 && d->src && d->dout && d->slab, "sar_conv2d_wgrad_split: null src/dout/slab or B <= 0");
  SAR_REQUIRE(d->nsplit >= wk && d->nsplit % wk == 0 && d->nsplit <= 65535,
              "sar_conv2d_wgrad_split: nsplit %d must be a positive multiple of %d", d->nsplit, wk);
  SAR_REQUIRE(d->ld_src >= npos && d->ld_dout >= npos, "sar_conv2d_wgrad_split: leading dimension smaller than B*H*W");
  SAR_REQUIRE((int64_t)d->M * d->ld_dout * 4 < 0xf0000000ll, "sar_conv2d_wgrad_split: dout larger than 3.75 GiB");
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv2d_wgrad_split: pro_scale/pro_shift mismatch");
  SAR_REQUIRE(arith != AR_H3A || (src_bound && dout_bound), "sar_conv2d_wgrad_split: the fp16 arithmetic needs the operand bounds");
  WgradKS k;
  k.ablate = 0;
  k.d = sar_wgrad_desc{};
  k.d.mode = SAR_CONV_TEMPORAL;
  k.d.B = 1, k.d.V = VJ, k.d.T_src = 1, k.d.T_out = 1;
  k.d.Kc = d->Kc, k.d.M = d->M, k.d.taps = TAPS, k.d.stride = 1, k.d.pad = 1, k.d.pro_relu = d->pro_relu;
  k.d.nsplit = d->nsplit;
  k.d.src = d->src, k.d.ld_src = d->ld_src, k.d.dout = d->dout, k.d.ld_dout = d->ld_dout;
  k.d.pro_scale = d->pro_scale, k.d.pro_shift = d->pro_shift;
  k.d.slab = d->slab;
  k.d.w_stride_tap = (int64_t)d->Kc * d->M, k.d.w_stride_c = d->M;
  k.d.wsize = (int64_t)TAPS * d->Kc * d->M, k.d.bsize = 0;
  k.src_bound = src_bound, k.dout_bound = dout_bound;
  k.H2 = d->H_out, k.seq2 = (int)npos, k.invH2 = 1.0f / (float)d->H_out;
  const int kt = Cfg<AR_H3A, 0, 8>::KT;
  k.TPS = (int)((npos + kt - 1) / kt);
  k.ntiles = k.TPS;
  k.gy = (d->M + 128 / wk - 1) / (128 / wk);
  k.gz = (d->Kc + CB - 1) / CB;
  const int nwork = (d->nsplit / wk) * k.gy * k.gz;
  const dim3 grid(((nwork + 7) / 8) * 8);
  const int rc = arith == AR_H3A ? launch_c2d<AR_H3A>(k, wk, d->W_out, grid, as_stream(s)) : sar_c2d_wsplit_b6(k, wk, d->W_out, grid, as_stream(s));
  if (rc) return rc;
  SAR_LAUNCH_CHECK("sar_conv2d_wgrad_split");
  return 0;
}
#endif
#endif

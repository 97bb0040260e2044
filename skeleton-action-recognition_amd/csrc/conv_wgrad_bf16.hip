// conv_wgrad_bf16.hip -- weight / bias gradient of the 9-tap temporal convolution (stride 1) with bf16 MFMA operands.
//
//   dW[tap][c][m] = sum_n bf16(pro(src))[c, n + (tap - pad) V] * bf16(dout)[m, n]     (tf.GradientTape of the Conv2D [9,1],
//   dbias[m]      = sum_n dout[m, n]  (fp32)                                           main_gnn.py:233, models/stgcn.py:29-36)
//
// Same slab contract as sar_conv_wgrad_f32 (include/sar_hip.h): split s writes slab[s][wsize + bsize], reduced by
// sar_slab_reduce_f32 in split order (deterministic, no atomics).  Both operands are rounded to bfloat16 when staged;
// products are exact, accumulation is fp32 (v_mfma_f32_32x32x16_bf16).  The bias gradient is summed from the fp32 values.
//
// Design (MI355X):
//  * The contraction runs over positions n, which is the contiguous axis of both operands: a lane's MFMA fragment is
//    8 consecutive positions of one row, i.e. 16 contiguous bytes of a row-major bf16 image in LDS.
//  * A temporal tap shifts the src window by tap*V positions.  V = 25 is odd, so odd taps start on an odd element:
//    the window is read as five aligned dwords and funnel-shifted by two bytes in registers (v_alignbyte_b32, which
//    issues beside the bf16 MFMA); even taps are four aligned dwords.  One LDS image serves all nine taps.
//  * A workgroup owns a 64 (c) x 64 (m) block of all nine taps (4 waves x 32x32 x 9 accumulators) and walks a
//    contiguous range of 8-frame tiles; the weight blocks that reduce the same tiles sit on one XCD (L2 reuse).
//  * At 16x the fp32 MFMA rate the kernel is bound by reading the two fp32 operands from HBM; two workgroups per CU
//    overlap one's staging with the other's MFMA phase.
#include "sar_common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WB_FT = 8;        // output frames per tile
constexpr int WB_TAPS = 9;
constexpr int WB_NPOS = 208;    // positions per tile, padded to 13 k-steps of 16 (FT*V <= 208)
constexpr int WB_KSTEPS = WB_NPOS / 16;
constexpr int WB_DYS = 216;     // dout row stride (elements): 108 dwords = 4*27 -> conflict-free ds_read_b128
constexpr int WB_SRCW = 416;    // staged src positions ((FT + 8) * V <= 416)
constexpr int WB_HS = 418;      // src row stride (elements): 209 dwords (odd)
constexpr int WB_BLK = 64;      // block edge along c and along m

struct WgradKB {
  sar_wgrad_desc d;
  int TPS;      // tiles per sequence
  int ntiles;   // B * TPS
  int gy, gz;   // blocks along m / c
};

template <int V>   // joints per frame, compile-time: every LDS window offset is an immediate and its parity is static
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const WgradKB k) {
  static_assert(WB_FT * V <= WB_NPOS && (WB_FT + 8) * V <= WB_SRCW, "tile too small for V");
  __shared__ __attribute__((aligned(16))) unsigned short Hs[WB_BLK * WB_HS];
  __shared__ __attribute__((aligned(16))) unsigned short Ds[WB_BLK * WB_DYS];
  const sar_wgrad_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  // workgroup id -> (split, m block, c block): the blocks of one split are adjacent slots of one XCD
  int split, by, bz;
  {
    const int nyz = k.gy * k.gz, nwork = d.nsplit * nyz;
    const int per = (nwork + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int w = xcd * per + slot;
    if (slot >= per || w >= nwork) return;
    split = w / nyz;
    const int yz = w - split * nyz;
    bz = yz / k.gy;
    by = yz - bz * k.gy;
  }
  const int m0 = by * WB_BLK, c0 = bz * WB_BLK;
  const int wc = wave >> 1, wmb = wave & 1;   // this wave's 32-row block along c and along m

  f32x16 acc[WB_TAPS];
#pragma unroll
  for (int t = 0; t < WB_TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  __shared__ float bacc[WB_BLK];   // fp32 row sums of dout (bias gradient); row r belongs to wave r % 4 only
  if (tid < WB_BLK) bacc[tid] = 0.f;

  // contiguous tile range of this split
  const int tps = (k.ntiles + d.nsplit - 1) / d.nsplit;
  const int tile_lo = split * tps;
  const int tile_hi = (tile_lo + tps < k.ntiles) ? tile_lo + tps : k.ntiles;

  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  const int seq_src = d.T_src * V, seq_out = d.T_out * V;
  // LDS read bases (bytes).  A = src rows (c), B = dout rows (m); k half hi -> +8 elements
  const unsigned a_base = (unsigned)(uintptr_t)Hs + (unsigned)(((wc * 32 + l31) * WB_HS + 8 * hi) * 2);
  const unsigned b_base = (unsigned)(uintptr_t)Ds + (unsigned)(((wmb * 32 + l31) * WB_DYS + 8 * hi) * 2);
  typedef const unsigned __attribute__((address_space(3))) * lds_u32;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef const u32x4 __attribute__((address_space(3))) * lds_u128;

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int b = tile / k.TPS;
    const int t0 = (tile - b * k.TPS) * WB_FT;
    const int t_lo = t0 - d.pad;
    // ---- stage src rows c0 .. c0+63: wave w takes rows w, w+4, ...; a lane converts pairs of positions
    {
      const float* src_b = d.src + (int64_t)b * seq_src;
      const int p_lo = t_lo * V;   // first staged position (may be negative)
      int vo[4][2];
      bool ok[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = 2 * lane + 128 * j;
        const int pa = p_lo + col;
        ok[j][0] = col < (WB_FT + 8) * V && (unsigned)pa < (unsigned)seq_src;
        ok[j][1] = col + 1 < (WB_FT + 8) * V && (unsigned)(pa + 1) < (unsigned)seq_src;
        vo[j][0] = pa * 4;   // the range check of the buffer load rejects negative / past-the-end offsets
        vo[j][1] = (pa + 1) * 4;
      }
#pragma unroll 1
      for (int rr = 0; rr < 16; rr += 2) {
        float x[2][4][2];
        float psc[2], psh[2];
        bool rok[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c = c0 + wave + 4 * (rr + q);
          rok[q] = c < d.Kc;
          const int cg = rok[q] ? c : 0;
          const __amdgpu_buffer_rsrc_t rs =
              __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, seq_src * 4, 0x00020000);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            x[q][j][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo[j][0], 0, 0));
            x[q][j][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo[j][1], 0, 0));
          }
          psc[q] = d.pro_scale ? d.pro_scale[cg] : 1.f;
          psh[q] = d.pro_scale ? d.pro_shift[cg] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int row = wave + 4 * (rr + q);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int col = 2 * lane + 128 * j;
            const float v0 = (ok[j][0] && rok[q]) ? fmaxf(fmaf(x[q][j][0], psc[q], psh[q]), relu_lo) : 0.f;
            const float v1 = (ok[j][1] && rok[q]) ? fmaxf(fmaf(x[q][j][1], psc[q], psh[q]), relu_lo) : 0.f;
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 p;
            p[0] = (__bf16)v0;
            p[1] = (__bf16)v1;
            if (col < WB_HS) *reinterpret_cast<unsigned*>(&Hs[row * WB_HS + col]) = *reinterpret_cast<unsigned*>(&p);
          }
        }
      }
    }
    // ---- stage dout rows m0 .. m0+63 (+ the fp32 row sums for the bias gradient)
    {
      const float* out_b = d.dout + (int64_t)b * seq_out;
      const int p_lo = t0 * V;
      int vo[2][2];
      bool ok[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = 2 * lane + 128 * j;
        ok[j][0] = col < WB_FT * V && p_lo + col < seq_out;
        ok[j][1] = col + 1 < WB_FT * V && p_lo + col + 1 < seq_out;
        vo[j][0] = (p_lo + col) * 4;
        vo[j][1] = (p_lo + col + 1) * 4;
      }
#pragma unroll 1
      for (int rr = 0; rr < 16; rr += 4) {
        float x[4][2][2];
        bool rok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + wave + 4 * (rr + q);
          rok[q] = m < d.M;
          const int mg = rok[q] ? m : 0;
          const __amdgpu_buffer_rsrc_t rs =
              __builtin_amdgcn_make_buffer_rsrc((void*)(out_b + (int64_t)mg * d.ld_dout), 0, seq_out * 4, 0x00020000);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            x[q][j][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo[j][0], 0, 0));
            x[q][j][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo[j][1], 0, 0));
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = wave + 4 * (rr + q);
          float rsum = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int col = 2 * lane + 128 * j;
            const float v0 = (ok[j][0] && rok[q]) ? x[q][j][0] : 0.f;
            const float v1 = (ok[j][1] && rok[q]) ? x[q][j][1] : 0.f;
            rsum += v0 + v1;
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 p;
            p[0] = (__bf16)v0;
            p[1] = (__bf16)v1;
            if (col < WB_DYS) *reinterpret_cast<unsigned*>(&Ds[row * WB_DYS + col]) = *reinterpret_cast<unsigned*>(&p);
          }
          if (d.bsize > 0 && bz == 0) {   // uniform
            rsum = wave_sum(rsum);
            if (lane == 0) bacc[row] += rsum;
          }
        }
      }
    }
    __syncthreads();
    // ---- MFMA phase: 13 k-steps x 9 taps
    for (int ks = 0; ks < WB_KSTEPS; ++ks) {
      const u32x4 bq = *(lds_u128)(uintptr_t)(b_base + ks * 32);
      const bf16x8 bv = *reinterpret_cast<const bf16x8*>(&bq);
      const unsigned a_ks = a_base + ks * 32;
#pragma unroll
      for (int t = 0; t < WB_TAPS; ++t) {
        // window start (elements, relative to the lane's base): t * V; V odd -> parity of t
        const int e = t * V;
        u32x4 aq;
        if ((e & 1) == 0) {
          lds_u32 p = (lds_u32)(uintptr_t)(a_ks + e * 2);
          aq = u32x4{p[0], p[1], p[2], p[3]};
        } else {
          lds_u32 p = (lds_u32)(uintptr_t)(a_ks + (e - 1) * 2);
          const unsigned w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4];
          aq = u32x4{__builtin_amdgcn_alignbyte(w1, w0, 2), __builtin_amdgcn_alignbyte(w2, w1, 2),
                     __builtin_amdgcn_alignbyte(w3, w2, 2), __builtin_amdgcn_alignbyte(w4, w3, 2)};
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&aq), bv, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- write this split's slab: rows c (registers), columns m (lanes: contiguous)
  float* slab = d.slab + (int64_t)split * (d.wsize + d.bsize);
  const int m = m0 + wmb * 32 + l31;
#pragma unroll
  for (int t = 0; t < WB_TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + wc * 32 + mfma_row(r, hi);
      if (c < d.Kc && m < d.M) slab[(int64_t)t * d.w_stride_tap + (int64_t)c * d.w_stride_c + m] = acc[t][r];
    }
  if (d.bsize > 0 && bz == 0 && tid < WB_BLK && m0 + tid < d.M) slab[d.wsize + m0 + tid] = bacc[tid];
}

// ---- stride 2 (the two down-sampling blocks, models/stgcn.py:117,120): out frame t reads src frame 2 t + tap - pad.
// The 24 src frames of a tile are de-interleaved by frame parity into two images (E: frames 2i, O: frames 2i + 1), in
// each of which a tap is again a shift by whole frames: tap -> (image, shift) is static for a given pad.  A workgroup
// owns 32 (c) x 64 (m); waves 0-1 multiply the taps that read E, waves 2-3 the taps that read O.
constexpr int S2_CB = 32;
constexpr int S2_HS = 314;   // image row stride (elements): 12 frames x 25 + window slack; 157 dwords (odd)

template <int V, int PAD>
__global__ __launch_bounds__(256, 2) void conv_wgrad_s2_bf16_kernel(const WgradKB k) {
  static_assert(WB_FT * V <= WB_NPOS && 12 * V + 12 <= S2_HS, "tile too small for V");
  __shared__ __attribute__((aligned(16))) unsigned short Hs[2 * S2_CB * S2_HS];
  __shared__ __attribute__((aligned(16))) unsigned short Ds[WB_BLK * WB_DYS];
  __shared__ float bacc[WB_BLK];
  const sar_wgrad_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int split, by, bz;
  {
    const int nyz = k.gy * k.gz, nwork = d.nsplit * nyz;
    const int per = (nwork + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int w = xcd * per + slot;
    if (slot >= per || w >= nwork) return;
    split = w / nyz;
    const int yz = w - split * nyz;
    bz = yz / k.gy;
    by = yz - bz * k.gy;
  }
  const int m0 = by * WB_BLK, c0 = bz * S2_CB;
  const int wmb = wave & 1, img = wave >> 1;   // img 0: taps with even (tap - PAD) read E; img 1: the others read O
  constexpr int NTE = (WB_TAPS - (PAD & 1) + 1) / 2;   // taps with tap == PAD (mod 2)
  constexpr int NTO = WB_TAPS - NTE;
  constexpr int NTMAX = NTE > NTO ? NTE : NTO;

  f32x16 acc[NTMAX];
#pragma unroll
  for (int t = 0; t < NTMAX; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  if (tid < WB_BLK) bacc[tid] = 0.f;

  const int tps = (k.ntiles + d.nsplit - 1) / d.nsplit;
  const int tile_lo = split * tps;
  const int tile_hi = (tile_lo + tps < k.ntiles) ? tile_lo + tps : k.ntiles;
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  const int seq_src = d.T_src * V, seq_out = d.T_out * V;
  const unsigned a_base = (unsigned)(uintptr_t)Hs + (unsigned)(((img * S2_CB + l31) * S2_HS + 8 * hi) * 2);
  const unsigned b_base = (unsigned)(uintptr_t)Ds + (unsigned)(((wmb * 32 + l31) * WB_DYS + 8 * hi) * 2);
  typedef const unsigned __attribute__((address_space(3))) * lds_u32;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef const u32x4 __attribute__((address_space(3))) * lds_u128;

  // src stager geometry: position p = lane + 64 j of the 24 x V contiguous positions -> (image, element)
  constexpr int SJ = (24 * V + 63) / 64;
  int sdst[SJ];
#pragma unroll
  for (int j = 0; j < SJ; ++j) {
    const int p = lane + 64 * j;
    const int f = p / V, v = p - f * V;
    sdst[j] = p < 24 * V ? ((f & 1) * S2_CB * S2_HS + (f >> 1) * V + v) : -1;
  }
  // the window slack behind the 12 staged frames of both images is read by the last k-steps: zero once
  for (int i = tid; i < 2 * S2_CB * (S2_HS - 12 * V); i += 256) {
    const int r = i / (S2_HS - 12 * V), cidx = 12 * V + i % (S2_HS - 12 * V);
    Hs[r * S2_HS + cidx] = 0;
  }

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int b = tile / k.TPS;
    const int t0 = (tile - b * k.TPS) * WB_FT;
    const int f_lo = 2 * (t0 - 2);   // first staged src frame (even); may be negative
    {
      const float* src_b = d.src + (int64_t)b * seq_src;
      const int p_lo = f_lo * V;
#pragma unroll 1
      for (int rr = 0; rr < 8; rr += 2) {
        float x[2][SJ];
        float psc[2], psh[2];
        bool rok[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c = c0 + wave + 4 * (rr + q);
          rok[q] = c < d.Kc;
          const int cg = rok[q] ? c : 0;
          const __amdgpu_buffer_rsrc_t rs =
              __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, seq_src * 4, 0x00020000);
#pragma unroll
          for (int j = 0; j < SJ; ++j)   // negative / past-the-end offsets are rejected by the range check -> 0
            x[q][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (p_lo + lane + 64 * j) * 4, 0, 0));
          psc[q] = d.pro_scale ? d.pro_scale[cg] : 1.f;
          psh[q] = d.pro_scale ? d.pro_shift[cg] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int row = wave + 4 * (rr + q);
#pragma unroll
          for (int j = 0; j < SJ; ++j) {
            const int pa = p_lo + lane + 64 * j;
            const bool ok = rok[q] && (unsigned)pa < (unsigned)seq_src;
            const __bf16 hv = (__bf16)(ok ? fmaxf(fmaf(x[q][j], psc[q], psh[q]), relu_lo) : 0.f);
            if (sdst[j] >= 0) Hs[row * S2_HS + sdst[j]] = *reinterpret_cast<const unsigned short*>(&hv);
          }
        }
      }
    }
    {
      const float* out_b = d.dout + (int64_t)b * seq_out;
      const int p_lo = t0 * V;
      int vo[2][2];
      bool ok[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = 2 * lane + 128 * j;
        ok[j][0] = col < WB_FT * V && p_lo + col < seq_out;
        ok[j][1] = col + 1 < WB_FT * V && p_lo + col + 1 < seq_out;
        vo[j][0] = (p_lo + col) * 4;
        vo[j][1] = (p_lo + col + 1) * 4;
      }
#pragma unroll 1
      for (int rr = 0; rr < 16; rr += 4) {
        float x[4][2][2];
        bool rok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + wave + 4 * (rr + q);
          rok[q] = m < d.M;
          const int mg = rok[q] ? m : 0;
          const __amdgpu_buffer_rsrc_t rs =
              __builtin_amdgcn_make_buffer_rsrc((void*)(out_b + (int64_t)mg * d.ld_dout), 0, seq_out * 4, 0x00020000);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            x[q][j][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo[j][0], 0, 0));
            x[q][j][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo[j][1], 0, 0));
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = wave + 4 * (rr + q);
          float rsum = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int col = 2 * lane + 128 * j;
            const float v0 = (ok[j][0] && rok[q]) ? x[q][j][0] : 0.f;
            const float v1 = (ok[j][1] && rok[q]) ? x[q][j][1] : 0.f;
            rsum += v0 + v1;
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 p;
            p[0] = (__bf16)v0;
            p[1] = (__bf16)v1;
            if (col < WB_DYS) *reinterpret_cast<unsigned*>(&Ds[row * WB_DYS + col]) = *reinterpret_cast<unsigned*>(&p);
          }
          if (d.bsize > 0 && bz == 0) {
            rsum = wave_sum(rsum);
            if (lane == 0) bacc[row] += rsum;
          }
        }
      }
    }
    __syncthreads();
    for (int ks = 0; ks < WB_KSTEPS; ++ks) {
      const u32x4 bq = *(lds_u128)(uintptr_t)(b_base + ks * 32);
      const bf16x8 bv = *reinterpret_cast<const bf16x8*>(&bq);
      const unsigned a_ks = a_base + ks * 32;
      auto tap_mma = [&](auto TP_, auto SLOT_) {
        constexpr int tp = decltype(TP_)::value, slot = decltype(SLOT_)::value;
        // src frame 2 t + tp - PAD: image parity (tp - PAD) & 1, image frame t + floor((tp - PAD) / 2); the images start
        // at image frame t0 - 2
        constexpr int diff = tp - PAD;
        constexpr int sh = (diff >= 0 ? diff / 2 : -((-diff + 1) / 2)) + 2;
        static_assert(sh >= 0 && sh <= 4, "tap outside the staged frames");
        constexpr int e = sh * V;
        u32x4 aq;
        if ((e & 1) == 0) {
          lds_u32 p = (lds_u32)(uintptr_t)(a_ks + e * 2);
          aq = u32x4{p[0], p[1], p[2], p[3]};
        } else {
          lds_u32 p = (lds_u32)(uintptr_t)(a_ks + (e - 1) * 2);
          const unsigned w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4];
          aq = u32x4{__builtin_amdgcn_alignbyte(w1, w0, 2), __builtin_amdgcn_alignbyte(w2, w1, 2),
                     __builtin_amdgcn_alignbyte(w3, w2, 2), __builtin_amdgcn_alignbyte(w4, w3, 2)};
        }
        acc[slot] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&aq), bv, acc[slot], 0, 0, 0);
      };
      constexpr int TE0 = PAD & 1;        // first tap that reads E
      constexpr int TO0 = 1 - (PAD & 1);  // first tap that reads O
      if (img == 0) {   // wave-uniform
        tap_mma(std::integral_constant<int, TE0>(), std::integral_constant<int, 0>());
        tap_mma(std::integral_constant<int, TE0 + 2>(), std::integral_constant<int, 1>());
        tap_mma(std::integral_constant<int, TE0 + 4>(), std::integral_constant<int, 2>());
        tap_mma(std::integral_constant<int, TE0 + 6>(), std::integral_constant<int, 3>());
        if constexpr (TE0 + 8 < WB_TAPS) tap_mma(std::integral_constant<int, TE0 + 8>(), std::integral_constant<int, 4>());
      } else {
        tap_mma(std::integral_constant<int, TO0>(), std::integral_constant<int, 0>());
        tap_mma(std::integral_constant<int, TO0 + 2>(), std::integral_constant<int, 1>());
        tap_mma(std::integral_constant<int, TO0 + 4>(), std::integral_constant<int, 2>());
        tap_mma(std::integral_constant<int, TO0 + 6>(), std::integral_constant<int, 3>());
        if constexpr (TO0 + 8 < WB_TAPS) tap_mma(std::integral_constant<int, TO0 + 8>(), std::integral_constant<int, 4>());
      }
    }
    __syncthreads();
  }

  float* slab = d.slab + (int64_t)split * (d.wsize + d.bsize);
  const int m = m0 + wmb * 32 + l31;
  const int tfirst = img == 0 ? (PAD & 1) : 1 - (PAD & 1);
#pragma unroll
  for (int sl = 0; sl < NTMAX; ++sl) {
    const int tp = tfirst + 2 * sl;
    if (tp < WB_TAPS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + mfma_row(r, hi);
        if (c < d.Kc && m < d.M) slab[(int64_t)tp * d.w_stride_tap + (int64_t)c * d.w_stride_c + m] = acc[sl][r];
      }
    }
  }
  if (d.bsize > 0 && bz == 0 && tid < WB_BLK && m0 + tid < d.M) slab[d.wsize + m0 + tid] = bacc[tid];
}

}  // namespace

extern "C" int sar_conv_wgrad_bf16(const sar_wgrad_desc* d, sar_stream_t s) {
  SAR_REQUIRE(d != nullptr, "sar_conv_wgrad_bf16: null descriptor");
  SAR_REQUIRE(d->mode == SAR_CONV_TEMPORAL && d->taps == 9 && (d->stride == 1 || (d->stride == 2 && d->pad == 3)),
              "sar_conv_wgrad_bf16: built for the 9-tap temporal convolution at stride 1, or stride 2 with pad 3 "
              "(others: sar_conv_wgrad_f32)");
  SAR_REQUIRE(d->B > 0 && d->V > 0 && d->T_src > 0 && d->T_out > 0 && d->Kc > 0 && d->M > 0, "sar_conv_wgrad_bf16: bad sizes");
  if (d->V != 25) {
    sar_set_error("sar_conv_wgrad_bf16: built for V = 25 joints (got %d); use sar_conv_wgrad_f32", d->V);
    return SAR_E_UNSUP;
  }
  SAR_REQUIRE(d->pad >= 0 && d->pad <= 8, "sar_conv_wgrad_bf16: bad pad");
  SAR_REQUIRE(d->src && d->dout && d->slab, "sar_conv_wgrad_bf16: null src/dout/slab");
  SAR_REQUIRE(d->nsplit >= 1 && d->nsplit <= 65535, "sar_conv_wgrad_bf16: nsplit %d out of range", d->nsplit);
  SAR_REQUIRE(d->ld_src >= (int64_t)d->B * d->T_src * d->V && d->ld_dout >= (int64_t)d->B * d->T_out * d->V,
              "sar_conv_wgrad_bf16: leading dimension smaller than B*T*V");
  SAR_REQUIRE((int64_t)d->T_src * d->V < (1 << 28) && (int64_t)d->T_out * d->V < (1 << 28), "sar_conv_wgrad_bf16: sequence row too long");
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv_wgrad_bf16: pro_scale/pro_shift mismatch");
  SAR_REQUIRE(d->wsize > 0 && (d->bsize == 0 || d->bsize == d->M), "sar_conv_wgrad_bf16: bad slab sizes");
  WgradKB k;
  k.d = *d;
  k.TPS = (d->T_out + WB_FT - 1) / WB_FT;
  k.ntiles = d->B * k.TPS;
  k.gy = (d->M + WB_BLK - 1) / WB_BLK;
  k.gz = (d->Kc + (d->stride == 2 ? S2_CB : WB_BLK) - 1) / (d->stride == 2 ? S2_CB : WB_BLK);
  const int nwork = d->nsplit * k.gy * k.gz;
  if (d->stride == 2)
    hipLaunchKernelGGL((conv_wgrad_s2_bf16_kernel<25, 3>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, as_stream(s), k);
  else
    hipLaunchKernelGGL(conv_wgrad_bf16_kernel<25>, dim3(((nwork + 7) / 8) * 8), dim3(256), 0, as_stream(s), k);
  SAR_LAUNCH_CHECK("sar_conv_wgrad_bf16");
  return 0;
}

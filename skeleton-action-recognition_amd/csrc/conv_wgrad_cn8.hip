// conv_wgrad_cn8.hip -- weight / bias gradients of the graph, temporal and residual convolutions on bf16 CN8 activations
// (cn8.h):
//
//   dW[tap][c][m] = sum_n OP_tap(pro(src))[c, n] * dout[m, n]      (tf.GradientTape of the convs, main_gnn.py:233;
//   dbias[m]      = sum_n dout[m, n] (x colsum(A_k)[w(n)] in GRAPH mode)   models/stgcn.py:29-36,47-54, models/gcn.py:199-209)
//
// Same slab contract as sar_conv_wgrad_f32 (include/sar_hip.h): split s writes slab[s][wsize + bsize] in fp32, reduced by
// sar_slab_reduce_f32 in split order -- deterministic, no atomics.  Products of the stored bf16 values are exact,
// accumulation is fp32 (v_mfma_f32_32x32x16_bf16); the bias sums are fp32 sums of the stored values.
//
// Design (MI355X):
//  * The contraction runs over columns n.  A lane's MFMA fragment is 8 consecutive columns of ONE channel -- the
//    transpose of the CN8 unit (8 channels of one column).  ds_read_b64_tr_b16 does that transpose in the LDS read
//    path: a 16-lane group fetches 4 columns x 16 channels (lane 4q+p supplies the address of column q / channels
//    4p..4p+3 = half a unit) and every lane receives 4 consecutive columns of its own channel.  So the LDS image is the
//    HBM image: staging is a 16-byte copy (through registers only where a folded BatchNorm + ReLU has to touch it).
//  * A temporal tap shifts the src window by tap*V whole units -- an immediate offset of the read, whatever the parity
//    of V = 25 (the fp32-storage kernel, conv_wgrad_bf16.hip, needs a funnel shift per odd tap).
//  * Plane stride = 4 (mod 16) units: the four planes a 32-lane half touches land on disjoint 16-bank groups --
//    conflict-free transposed reads.
//  * 9 taps: a workgroup owns 32 src channels x 64 out channels x all taps; the two wave pairs split the taps (5 + 4
//    accumulators of 32x32 per wave instead of 9: the prefetch registers of the next tile fit beside them).  Stride 2: the
//    staged src frames are de-interleaved by frame parity into two images (E: frames 2i, O: 2i+1), in each of which a tap is
//    again a whole-frame shift; waves 0-1 multiply the taps that read E, waves 2-3 those that read O.
//  * Every plane is staged by one wave with a per-(plane, sequence) buffer descriptor: TF-SAME padding, ragged sequence
//    ends and missing planes are the range check's zeros.  The loads of tile i+1 are in flight during the MFMA phase
//    of tile i (72 KB per workgroup): the kernel is bound by HBM, not by staging latency.
#include "cn8.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;

constexpr int WV = 25;   // joints per frame (compile-time: every window offset is an immediate)

struct WgradK8 {
  sar_wgrad_desc d;
  int TPS, ntiles, gy, gz;
  int Gs, Gd;   // CN8 planes of src / dout
};

// In-kernel phase stamps (diagnostic build -DSAR_CN8_STAMPS, tools/stamps8.sh; see conv_gemm_cn8.hip): wave 0 of every workgroup,
// shader-clock cycles in [0] store (wait for the tile's loads, folded BN + ReLU, LDS stores), [1] the barrier behind it, [2] load
// issue + MFMA phase, [3] the closing barrier, [4] prologue, [5] epilogue (slab stores); [6] workgroups, [7] lifetimes in 100 MHz
// ticks, [8] lifetimes in cycles, [9] tiles.
#ifdef SAR_CN8_STAMPS
constexpr int WSTAMP_WG = 8192;
__device__ unsigned g_wstamps8[WSTAMP_WG][10];
#define WSTAMP8(i)                                               \
  do {                                                           \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    st_acc[i] += t_ - st_last;                                   \
    st_last = t_;                                                \
  } while (0)
#else
#define WSTAMP8(i)
#endif

constexpr int pad_stride(int n) { return n + ((4 - n % 16) + 16) % 16; }   // smallest plane stride >= n that is 4 (mod 16)

// 8 consecutive columns (k = 8h .. 8h+7 of the k-step) of this lane's channel: two transposed reads, 4 columns each
__device__ __forceinline__ bf16x8 tr_frag(unsigned addr) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)addr);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(addr + 64));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return *reinterpret_cast<const bf16x8*>(&v);
}

// this lane's byte offset inside an image of plane stride PS (units) for the transposed reads of a 32-channel block
// that starts at plane `p0`: group G = lane>>4 -> (k half h = G>>1, channel half G&1); lane 4q+p of the group -> column q,
// channels 4p..4p+3 of the 16-channel half
__device__ __forceinline__ unsigned tr_lane_bytes(int lane, int PS, int p0) {
  const int G = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  return (unsigned)((((p0 + 2 * (G & 1) + (p >> 1)) * PS + 8 * (G >> 1) + q) * 16) + 8 * (p & 1));
}

__device__ __forceinline__ bool xcd_split(int nsplit, int gy, int gz, int& split, int& by, int& bz) {
  const int nyz = gy * gz, nwork = nsplit * nyz;
  const int per = (nwork + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int w = xcd * per + slot;
  if (slot >= per || w >= nwork) return false;
  split = w / nyz;
  const int yz = w - split * nyz;
  bz = yz / gy;
  by = yz - bz * gy;
  return true;
}

// ------------------------------------------------------------------------------------------------ temporal / residual
// WIDE (9 taps, stride 1): a workgroup owns 64 src x 64 out channels and every wave ALL nine taps of its 32 x 32 block (nine
// accumulators) instead of 32 src channels with the taps split 5 + 4 over the wave pairs: dout is re-read by half as many
// workgroups and no wave idles for a missing tap
template <int TAPS, int STRIDE, bool WIDE = false>
struct TCfg {
  static constexpr int FT = 7;                                   // output frames per tile
  static constexpr int NPOS = 176;                               // columns per tile padded to 11 k-steps of 16
  static constexpr int KSTEPS = NPOS / 16;
  static constexpr bool SPLIT = (STRIDE == 2 && TAPS == 9);      // parity images E / O
  static constexpr bool TSPLIT = (TAPS == 9) && !WIDE;           // taps split over the two wave pairs (32 src channels per workgroup)
  static constexpr int NIMG = SPLIT ? 2 : 1;
  static constexpr int CBLK = TSPLIT ? 32 : 64;                  // src channels per workgroup
  static constexpr int NSP = NIMG * CBLK / 8;                    // staged src image planes (8 or 4)
  static constexpr int IF = (STRIDE == 1) ? FT + TAPS - 1 : (SPLIT ? FT + 4 : FT);   // staged frames per image
  static constexpr int SU = IF * WV;                             // staged units per image plane
  static constexpr int MAXSH = (STRIDE == 1) ? TAPS - 1 : (SPLIT ? 4 : 0);            // largest window shift (frames)
  static constexpr int PS_S = pad_stride(NPOS + MAXSH * WV);     // src plane stride
  static constexpr int PS_D = pad_stride(NPOS);
  static constexpr int SJ = (SU + 63) / 64;                      // loads per src plane and lane
  static constexpr int DJ = (FT * WV + 63) / 64;
  static constexpr int NACC = TSPLIT ? 5 : TAPS;
  static constexpr int LDS_UNITS = NSP * PS_S + 8 * PS_D;
};

template <int TAPS, int STRIDE, bool WIDE = false>
__global__ __launch_bounds__(256, 2) void wgrad_cn8_kernel(const WgradK8 k) {
  using C = TCfg<TAPS, STRIDE, WIDE>;
  static_assert(!WIDE || (TAPS == 9 && STRIDE == 1), "the wide block is built for 9 taps at stride 1");
  constexpr int FT = C::FT, PS_S = C::PS_S, PS_D = C::PS_D, SJ = C::SJ, DJ = C::DJ, SU = C::SU, NACC = C::NACC;
  __shared__ uint4 lds[C::LDS_UNITS];
  __shared__ float bred[4][16];
  __shared__ __attribute__((aligned(16))) float pro_w[128];   // folded BN scale [0, 64) / shift [64, 128) of this workgroup's src channels
  constexpr int NSP = C::NSP, SQ = NSP / 4;   // src image planes; planes staged per wave (wave w: planes w, w + 4)
  uint4* Himg = lds;                 // src image planes (SPLIT: E planes 0-3, O planes 4-7)
  uint4* Dimg = lds + NSP * PS_S;    // 8 dout planes
  const sar_wgrad_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int split, by, bz;
  if (!xcd_split(d.nsplit, k.gy, k.gz, split, by, bz)) return;
#ifdef SAR_CN8_STAMPS
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_last = st_t0;
#endif
  const int m0 = by * 64, c0 = bz * C::CBLK;
  const int wmb = wave & 1;                 // this wave's 32-row block along m
  const int wc = C::TSPLIT ? 0 : wave >> 1;  // ... along c
  const int tg = C::TSPLIT ? wave >> 1 : 0;  // TSPLIT: this wave's tap group (stride 2: group 0 reads image E, group 1 image O)

  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum[2][8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[q][j] = 0.f;

  // zero the slack behind the staged units of every plane once (read by the window of the last k-steps; never restaged)
  for (int i = tid; i < NSP * (PS_S - SU); i += 256) Himg[(i / (PS_S - SU)) * PS_S + SU + i % (PS_S - SU)] = make_uint4(0u, 0u, 0u, 0u);
  for (int i = tid; i < 8 * (PS_D - FT * WV); i += 256)
    Dimg[(i / (PS_D - FT * WV)) * PS_D + FT * WV + i % (PS_D - FT * WV)] = make_uint4(0u, 0u, 0u, 0u);

  const int tps = (k.ntiles + d.nsplit - 1) / d.nsplit;
  const int tile_lo = split * tps;
  const int tile_hi = (tile_lo + tps < k.ntiles) ? tile_lo + tps : k.ntiles;
  const int seq_src = d.T_src * WV, seq_out = d.T_out * WV;
  const bool has_pro = d.pro_scale != nullptr;
  const bool pro_relu = d.pro_relu != 0;
  const bool do_bias = d.bsize > 0 && bz == 0;
  if (tid < 128) {
    const int c = c0 + (tid & 63);
    const float* pp = tid < 64 ? d.pro_scale : d.pro_shift;
    pro_w[tid] = (has_pro && c < d.Kc) ? pp[c] : 0.f;
  }
  __syncthreads();   // pro_w

  // ---- stager geometry.  Wave w stages src image planes w, w+4 and dout planes w, w+4; lane -> units lane + 64 j.
  // src unit i of an image plane = frame i / V of the image, joint i % V; its column offset inside the sequence is
  // linear in i at stride 1 and jumps 2 frames per image frame at stride 2: rel[j] holds the lane's part.
  int srel[SJ];
#pragma unroll
  for (int j = 0; j < SJ; ++j) {
    const int i = lane + 64 * j;
    const int f = i / WV;
    srel[j] = i < SU ? (STRIDE == 1 ? i : i + f * WV) : (1 << 27);   // beyond the staged width: rejected -> 0
  }
  uint4 sreg[SQ][SJ], dreg[2][DJ];
  unsigned sin[SQ];   // per plane: bit j set = unit j lies inside the sequence (prologue keeps the padding exactly 0)

  auto issue_loads = [&](int tile) {
    const int b = tile / k.TPS;
    const int t0 = (tile - b * k.TPS) * FT;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ip = wave + 4 * q;                                  // image plane
     if (q < SQ) {
      const int pim = C::SPLIT ? ip >> 2 : 0;                        // parity image of this plane
      const int g = c0 / 8 + (C::SPLIT ? (ip & 3) : ip);             // CN8 plane of src
      // first staged column of the image inside the sequence
      const int p0 = (STRIDE == 1) ? (t0 - d.pad) * WV : (C::SPLIT ? (2 * (t0 - 2) + pim) * WV : 2 * t0 * WV);
      const bool live = g < k.Gs;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)d.src + ((int64_t)(live ? g : 0) * d.ld_src + (int64_t)b * seq_src) * 16), 0,
          live ? (unsigned)seq_src * 16u : 0u, 0x00020000);
      sin[q] = 0;
#pragma unroll
      for (int j = 0; j < SJ; ++j) {
        const int col = p0 + srel[j];
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, col * 16, 0, 0);
        sreg[q][j] = make_uint4(v[0], v[1], v[2], v[3]);
        sin[q] |= ((unsigned)col < (unsigned)seq_src ? 1u : 0u) << j;
      }
     }
      const int gd = m0 / 8 + ip;
      const bool dlive = gd < k.Gd;
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)d.dout + ((int64_t)(dlive ? gd : 0) * d.ld_dout + (int64_t)b * seq_out) * 16), 0,
          dlive ? (unsigned)seq_out * 16u : 0u, 0x00020000);
#pragma unroll
      for (int j = 0; j < DJ; ++j) {
        const int i = lane + 64 * j;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rd, i < FT * WV ? (t0 * WV + i) * 16 : (1 << 30), 0, 0);
        dreg[q][j] = make_uint4(v[0], v[1], v[2], v[3]);
      }
    }
  };

  auto store_lds = [&]() {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ip = wave + 4 * q;
     if (q < SQ) {
      if (has_pro) {   // uniform: BatchNorm + ReLU of the producer folded into the operand (models/stgcn.py:27-28)
        // the plane's scale / shift from the workgroup's LDS copy (fetched here with cn8_params8 they were two dependent L2 round
        // trips per plane in EVERY tile's store phase: conv_gemm_cn8.hip)
        const int cr = 8 * (C::SPLIT ? (ip & 3) : ip);
        const float4* ps = reinterpret_cast<const float4*>(pro_w + cr);
        const float4* pt = reinterpret_cast<const float4*>(pro_w + 64 + cr);
        const float4 pa = ps[0], pb = ps[1], pc = pt[0], pe = pt[1];
        const float psc[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
        const float psh[8] = {pc.x, pc.y, pc.z, pc.w, pe.x, pe.y, pe.z, pe.w};
#pragma unroll
        for (int j = 0; j < SJ; ++j)   // padding / columns outside the sequence stay exactly 0 (keep mask)
          sreg[q][j] = cn8_bn_relu_unit(sreg[q][j], psc, psh, pro_relu, ((sin[q] >> j) & 1u) ? 0xffffffffu : 0u);
      }
#pragma unroll
      for (int j = 0; j < SJ; ++j)
        if ((j + 1) * 64 <= SU || lane + 64 * j < SU) Himg[ip * PS_S + lane + 64 * j] = sreg[q][j];
     }
#pragma unroll
      for (int j = 0; j < DJ; ++j) {
        if ((j + 1) * 64 <= FT * WV || lane + 64 * j < FT * WV) Dimg[ip * PS_D + lane + 64 * j] = dreg[q][j];
        if (do_bias) {   // uniform
          float f[8];
          cn8_unpack(dreg[q][j], f);   // lanes beyond the tile hold the range check's zeros
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[q][e] += f[e];
        }
      }
    }
  };

  // tap groups (TSPLIT).  stride 1: group 0 = taps 0-4, group 1 = taps 5-8, tap t reads the image shifted by t frames.
  // stride 2: out frame t, tap tp reads src frame 2 t + tp - pad = parity image (tp - pad) & 1 at image frame
  // t + floor((tp - pad) / 2); the images start at image frame t0 - 2.  pad 3 (even T): E <- taps 1,3,5,7 at shifts 1..4,
  // O <- taps 0,2,..,8 at shifts 0..4; pad 4 (odd T): E <- taps 0,2,..,8 at shifts 0..4, O <- taps 1,3,5,7 at shifts 0..3.
  int nt = C::TSPLIT ? 1 : TAPS, sh0 = 0, tap0 = 0;
  constexpr int tstep = C::SPLIT ? 2 : 1;
  if (C::TSPLIT && !C::SPLIT) {
    nt = tg == 0 ? 5 : 4;
    sh0 = tap0 = 5 * tg;
  } else if (C::SPLIT) {
    const int odd = d.pad & 1;             // pad 3 -> 1, pad 4 -> 0
    tap0 = (tg == 0) ? odd : 1 - odd;      // first tap of the group: E holds the taps congruent to pad (mod 2)
    nt = tap0 == 0 ? 5 : 4;
    sh0 = (tg == 0 && odd) ? 1 : 0;
  }
  const unsigned a_base = (unsigned)(uintptr_t)Himg + tr_lane_bytes(lane, PS_S, C::SPLIT ? 4 * tg : 4 * wc) + (unsigned)(sh0 * WV * 16);
  const unsigned b_base = (unsigned)(uintptr_t)Dimg + tr_lane_bytes(lane, PS_D, 4 * wmb);

  if (tile_lo < tile_hi) issue_loads(tile_lo);
  WSTAMP8(4);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    store_lds();
    WSTAMP8(0);
    __syncthreads();
    WSTAMP8(1);
    if (tile + 1 < tile_hi) issue_loads(tile + 1);   // in flight during the MFMA phase
    // k-steps, software-pipelined BY HAND over two fragment sets: the transposed reads of k-step ks + 1 are issued before
    // the MFMAs of k-step ks.  (As one un-pipelined loop body the compiler emitted read, wait, multiply three times per
    // k-step: three exposed LDS round trips for 160 cycles of matrix work.)
    constexpr int NFR = C::TSPLIT ? 5 : TAPS;
    bf16x8 fb[2], fa[2][NFR];
    auto frag_load = [&](int ks, bf16x8& bv, bf16x8 (&av)[NFR]) {
      bv = tr_frag(b_base + ks * 256);
      const unsigned a_ks = a_base + ks * 256;
      if constexpr (!C::TSPLIT) {
        // every tap of the kernel (1, or 9 in the wide block): tap t reads the image shifted by t frames.  One tap, stride 2
        // (pad 0): the E image as staged
#pragma unroll
        for (int t = 0; t < TAPS; ++t) av[t] = tr_frag(a_ks + t * WV * 16);
      } else {
        // this wave's tap group: accumulator s = tap tap0 + s * tstep, window shift (sh0 + s) frames
#pragma unroll
        for (int s = 0; s < 5; ++s) av[s] = tr_frag(a_ks + s * WV * 16);   // the 4-tap group reads (and ignores) a fifth window: inside the LDS image, and no branch
      }
    };
    auto frag_mma = [&](const bf16x8& bv, const bf16x8 (&av)[NFR]) {
      if constexpr (!C::TSPLIT) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[t], bv, acc[t], 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s], bv, acc[s], 0, 0, 0);
        if (nt == 5) acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[4], bv, acc[4], 0, 0, 0);   // wave-uniform
      }
    };
    frag_load(0, fb[0], fa[0]);
    int ks = 0;
#pragma unroll 1
    for (; ks + 2 < C::KSTEPS; ks += 2) {   // whole pairs with a successor: no branch between the reads and the MFMAs
      frag_load(ks + 1, fb[1], fa[1]);
      __builtin_amdgcn_sched_barrier(0);   // the reads of the next k-step stay AHEAD of this k-step's MFMAs
      frag_mma(fb[0], fa[0]);
      __builtin_amdgcn_sched_barrier(0);
      frag_load(ks + 2, fb[0], fa[0]);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(fb[1], fa[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (C::KSTEPS % 2 == 0) {
      frag_load(ks + 1, fb[1], fa[1]);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(fb[0], fa[0]);
      frag_mma(fb[1], fa[1]);
    } else {
      frag_mma(fb[0], fa[0]);
    }
    WSTAMP8(2);
    __syncthreads();
    WSTAMP8(3);
  }

  // ---- this split's slab: rows c (registers), columns m (lanes: contiguous)
  float* slab = d.slab + (int64_t)split * (d.wsize + d.bsize);
  const int m = m0 + wmb * 32 + l31;
#pragma unroll
  for (int s = 0; s < NACC; ++s) {
    const int tp = tap0 + s * tstep;
    if (s < nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wc * 32 + mfma_row(r, hi);
        if (c < d.Kc && m < d.M) slab[(int64_t)tp * d.w_stride_tap + (int64_t)c * d.w_stride_c + m] = acc[s][r];
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = wave_sum_dpp(bsum[q][e]);
        if (lane == 0) bred[wave][q * 8 + e] = t;
      }
    __syncthreads();
    if (tid < 64) {   // channel m0 + tid lives in dout plane tid / 8 = wave' + 4 q
      const int pl = tid >> 3, e = tid & 7;
      if (m0 + tid < d.M) slab[d.wsize + m0 + tid] = bred[pl & 3][(pl >> 2) * 8 + e];
    }
  }
#ifdef SAR_CN8_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  WSTAMP8(5);
  if (tid == 0 && blockIdx.x < WSTAMP_WG) {
    unsigned* row = g_wstamps8[blockIdx.x];
    for (int i2 = 0; i2 < 6; ++i2) row[i2] = (unsigned)st_acc[i2];
    row[6] = 1u;
    row[7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
    row[8] = (unsigned)(__builtin_amdgcn_s_memtime() - st_t0);
    row[9] = (unsigned)(tile_hi > tile_lo ? tile_hi - tile_lo : 0);
  }
#endif
}

// ------------------------------------------------------------------------------------------------ graph
//   dW[c][k F + m] = sum_n z_k[c, n] dout[m, n],  z_k = src . A_k (fp32 gather of the stored values, rounded once);
//   dbias[k][m] = sum_n dout[m, n] colsum(A_k)[w(n)].
// Tile = 5 whole frames (the gather never leaves a frame: no halo).  Per tile: raw units -> LDS, every thread builds
// the z_k units of its (plane quartet, column) -- an identity slice (ID0) is read straight from the raw image --, then
// 8 k-steps x 3 slices of MFMAs.
template <int NZ0, int NZ1, int NZ2, bool ID0>
__global__ __launch_bounds__(256, 2) void wgrad_graph_cn8_kernel(const WgradK8 k) {
  constexpr int FT = 5, NLIVE = FT * WV, NPOS = 128, KSTEPS = NPOS / 16;
  constexpr int PS = pad_stride(NPOS);   // 132
  constexpr int NZ[3] = {NZ0, NZ1, NZ2};
  constexpr int DENSE = ((NZ0 > 1) + (NZ1 > 1) + (NZ2 > 1) == 1) ? (NZ0 > 1 ? 0 : NZ1 > 1 ? 1 : 2) : -1;   // slice the matrix cores may gather
  constexpr int NZI = ID0 ? 2 : 3;       // built z images
  constexpr int XJ = (NLIVE + 63) / 64;
  __shared__ uint4 lds[(2 + NZI) * 8 * PS];
  __shared__ float bred[4][48];
  uint4* Ximg = lds;
  uint4* Dimg = lds + 8 * PS;
  uint4* Zimg = lds + 16 * PS;           // [slice - (ID0 ? 1 : 0)][8 planes]
  const sar_wgrad_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int split, by, bz;
  if (!xcd_split(d.nsplit, k.gy, k.gz, split, by, bz)) return;
#ifdef SAR_CN8_STAMPS   // graph kernel: [0] store_raw (waits for the loads), [1] barrier, [2] build_z + barrier, [3] k-steps + closing barrier, [4] prologue, [5] epilogue
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_last = st_t0;
#endif
  const int m0 = by * 64, c0 = bz * 64;
  const int wmb = wave & 1, wc = wave >> 1;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum[2][3][8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum[q][t][e] = 0.f;
  for (int i = tid; i < (2 + NZI) * 8 * (PS - NLIVE); i += 256)
    lds[(i / (PS - NLIVE)) * PS + NLIVE + i % (PS - NLIVE)] = make_uint4(0u, 0u, 0u, 0u);

  const int tps = (k.ntiles + d.nsplit - 1) / d.nsplit;
  const int tile_lo = split * tps;
  const int tile_hi = (tile_lo + tps < k.ntiles) ? tile_lo + tps : k.ntiles;
  const int seq = d.T_src * WV;
  const bool do_bias = d.bsize > 0 && bz == 0;

  // unit builder: thread -> (plane quartet ph, column pos); gather offsets / weights of its joint
  const int ph = tid >> 7, pos = tid & 127;
  const bool blive = pos < NLIVE;
  // matrix-core gather of the dense slice (conv_gemm_cn8.hip, conv_graph_cn8_kernel): gather weights exact in bfloat16
  const bool mg = DENSE >= 0 && (d.g_flags & SAR_GRAPH_WT_BF16_EXACT);   // uniform
  int go[3][4];
  float gwt[3][4];
  {
    const int pp = blive ? pos : 0;
    const int fo = pp / WV, v = pp - fo * WV;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < NZ[t] && !(mg && t == DENSE)) {
          go[t][j] = fo * WV + d.g_idx[(t * WV + v) * 4 + j];
          gwt[t][j] = d.g_wt[(t * WV + v) * 4 + j];
        }
  }
  // B fragments A_k[v = 8 G + j][w = (lane & 15) + 16 nb] and the lane's LDS addresses: wave w gathers channel group w
  // (planes 2 w, 2 w + 1) of every frame
  const int gG = lane >> 4, gi = lane & 15;
  bf16x8 bfr[2];
  unsigned tr_addr = 0;
  // A-operand keep mask of the matrix-core gather (element e of a lane's fragment = joint 8 G + e): joints >= V are the next
  // frame's columns; their B rows are zero but 0 x Inf = NaN, so they are cleared before the product (conv_gemm_cn8.hip)
  unsigned keep[4];
#pragma unroll
  for (int dd = 0; dd < 4; ++dd)
    keep[dd] = ((8 * gG + 2 * dd < WV) ? 0xffffu : 0u) | ((8 * gG + 2 * dd + 1 < WV) ? 0xffff0000u : 0u);
  int zst_unit = 0;
  if (mg) {
    constexpr int DS = DENSE >= 0 ? DENSE : 0;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int wv = gi + 16 * nb;
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
      if (wv < WV) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < NZ[DS]) {
            const int vi = d.g_idx[(DS * WV + wv) * 4 + e];
            const float wt = d.g_wt[(DS * WV + wv) * 4 + e];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += (vi == 8 * gG + j) ? wt : 0.f;
          }
      }
      const uint4 pk = cn8_pack(a);
      bfr[nb] = *reinterpret_cast<const bf16x8*>(&pk);
    }
    const int q = gi >> 2, p = gi & 3;
    tr_addr = (unsigned)(uintptr_t)Ximg + (unsigned)((((2 * wave + (p >> 1)) * PS + 8 * gG + q) * 16) + 8 * (p & 1));
    zst_unit = ((DS - (ID0 ? 1 : 0)) * 8 + 2 * wave + (gG >> 1)) * PS + gi;
  }
  // colsum(A_k)[w] of the dout units this lane stages (bias gradient)
  float cs[XJ][3];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int i = lane + 64 * j;
#pragma unroll
    for (int t = 0; t < 3; ++t) cs[j][t] = (i < NLIVE && d.g_colsum) ? d.g_colsum[t * WV + i % WV] : 0.f;
  }

  uint4 xreg[2][XJ], dreg[2][XJ];
  auto issue_loads = [&](int tile) {
    const int b = tile / k.TPS;
    const int t0 = (tile - b * k.TPS) * FT;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ip = wave + 4 * q;
      const int g = c0 / 8 + ip, gd = m0 / 8 + ip;
      const bool live = g < k.Gs, dlive = gd < k.Gd;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)d.src + ((int64_t)(live ? g : 0) * d.ld_src + (int64_t)b * seq) * 16), 0,
          live ? (unsigned)seq * 16u : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)d.dout + ((int64_t)(dlive ? gd : 0) * d.ld_dout + (int64_t)b * seq) * 16), 0,
          dlive ? (unsigned)seq * 16u : 0u, 0x00020000);
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        const int i = lane + 64 * j;
        const int vo = i < NLIVE ? (t0 * WV + i) * 16 : (1 << 30);
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0);
        const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rd, vo, 0, 0);
        xreg[q][j] = make_uint4(a[0], a[1], a[2], a[3]);
        dreg[q][j] = make_uint4(c[0], c[1], c[2], c[3]);
      }
    }
  };
  auto store_raw = [&]() {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ip = wave + 4 * q;
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        if ((j + 1) * 64 <= NLIVE || lane + 64 * j < NLIVE) {
          Ximg[ip * PS + lane + 64 * j] = xreg[q][j];
          Dimg[ip * PS + lane + 64 * j] = dreg[q][j];
        }
        if (do_bias) {
          float f[8];
          cn8_unpack(dreg[q][j], f);
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[q][t][e] = fmaf(f[e], cs[j][t], bsum[q][t][e]);
        }
      }
    }
  };
  auto build_z = [&]() {
    if (mg) {   // uniform: Z_f[16 channels x V] = X_f[16 x 32 joints] . A_k[32 x 32], two MFMAs per frame and channel group;
      // all transposed reads first, then the MFMAs, then the stores (conv_graph_cn8_kernel)
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      bf16x8 afr[FT];
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        const unsigned ra = tr_addr + (unsigned)(f * WV * 16);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)ra);
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(ra + 64));
        const s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        uint4 am = *reinterpret_cast<const uint4*>(&av);
        am.x &= keep[0], am.y &= keep[1], am.z &= keep[2], am.w &= keep[3];
        afr[f] = *reinterpret_cast<const bf16x8*>(&am);
      }
      f32x4 z[FT][2];
#pragma unroll
      for (int f = 0; f < FT; ++f)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          z[f][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[f], bfr[nb], zero, 0, 0, 0);
        }
#pragma unroll
      for (int f = 0; f < FT; ++f)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          if (gi + 16 * nb < WV)
            reinterpret_cast<uint2*>(Zimg + zst_unit + f * WV + 16 * nb)[gG & 1] =
                make_uint2(cn8_pack2(z[f][nb][0], z[f][nb][1]), cn8_pack2(z[f][nb][2], z[f][nb][3]));
    }
    if (!blive) return;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      const uint4* Xp = Ximg + (4 * ph + pl) * PS;
#pragma unroll
      for (int t = ID0 ? 1 : 0; t < 3; ++t) {
        if (mg && t == DENSE) continue;
        uint4 zu;
        if (NZ[t] == 1 && gwt[t][0] == 1.0f) {
          zu = Xp[go[t][0]];
        } else {
          float z[8], x[8];
          cn8_unpack(Xp[go[t][0]], x);
#pragma unroll
          for (int e = 0; e < 8; ++e) z[e] = gwt[t][0] * x[e];
#pragma unroll
          for (int j = 1; j < 4; ++j)
            if (j < NZ[t]) {
              cn8_unpack(Xp[go[t][j]], x);
#pragma unroll
              for (int e = 0; e < 8; ++e) z[e] = fmaf(gwt[t][j], x[e], z[e]);
            }
          zu = cn8_pack(z);
        }
        Zimg[((t - (ID0 ? 1 : 0)) * 8 + 4 * ph + pl) * PS + pos] = zu;
      }
    }
  };

  const unsigned lane_a = tr_lane_bytes(lane, PS, 4 * wc);
  const unsigned x_base = (unsigned)(uintptr_t)Ximg + lane_a;
  const unsigned z_base = (unsigned)(uintptr_t)Zimg + lane_a;
  const unsigned b_base = (unsigned)(uintptr_t)Dimg + tr_lane_bytes(lane, PS, 4 * wmb);

  if (tile_lo < tile_hi) issue_loads(tile_lo);
  WSTAMP8(4);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    store_raw();
    WSTAMP8(0);
    __syncthreads();
    WSTAMP8(1);
    if (tile + 1 < tile_hi) issue_loads(tile + 1);
    build_z();
    __syncthreads();
    WSTAMP8(2);
    // k-steps, software-pipelined by hand over two fragment sets (see wgrad_cn8_kernel): the eight transposed reads of
    // k-step ks + 1 are issued before the three MFMAs of k-step ks
    bf16x8 fb[2], fa[2][3];
    auto frag_load = [&](int ks, bf16x8& bv, bf16x8 (&av)[3]) {
      bv = tr_frag(b_base + ks * 256);
#pragma unroll
      for (int t = 0; t < 3; ++t)
        av[t] = tr_frag((ID0 && t == 0) ? x_base + ks * 256 : z_base + ((t - (ID0 ? 1 : 0)) * 8 * PS) * 16 + ks * 256);
    };
    auto frag_mma = [&](const bf16x8& bv, const bf16x8 (&av)[3]) {
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[t], bv, acc[t], 0, 0, 0);
    };
    frag_load(0, fb[0], fa[0]);
    int ks = 0;
#pragma unroll 1
    for (; ks + 2 < KSTEPS; ks += 2) {
      frag_load(ks + 1, fb[1], fa[1]);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(fb[0], fa[0]);
      __builtin_amdgcn_sched_barrier(0);
      frag_load(ks + 2, fb[0], fa[0]);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(fb[1], fa[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (KSTEPS % 2 == 0) {
      frag_load(ks + 1, fb[1], fa[1]);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(fb[0], fa[0]);
      frag_mma(fb[1], fa[1]);
    } else {
      frag_mma(fb[0], fa[0]);
    }
    __syncthreads();
    WSTAMP8(3);
  }

  float* slab = d.slab + (int64_t)split * (d.wsize + d.bsize);
  const int m = m0 + wmb * 32 + l31;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + wc * 32 + mfma_row(r, hi);
      if (c < d.Kc && m < d.M) slab[(int64_t)t * d.w_stride_tap + (int64_t)c * d.w_stride_c + m] = acc[t][r];
    }
  if (do_bias) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float s = wave_sum_dpp(bsum[q][t][e]);
          if (lane == 0) bred[wave][(q * 3 + t) * 8 + e] = s;
        }
    __syncthreads();
    if (tid < 192) {   // (slice t, channel m0 + mm): dout plane mm / 8 = wave' + 4 q
      const int t = tid >> 6, mm = tid & 63;
      const int pl = mm >> 3, e = mm & 7;
      if (m0 + mm < d.M) slab[d.wsize + (int64_t)t * d.M + m0 + mm] = bred[pl & 3][((pl >> 2) * 3 + t) * 8 + e];
    }
  }
#ifdef SAR_CN8_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  WSTAMP8(5);
  if (tid == 0 && blockIdx.x < WSTAMP_WG) {
    unsigned* row = g_wstamps8[blockIdx.x];
    for (int i2 = 0; i2 < 6; ++i2) row[i2] = (unsigned)st_acc[i2];
    row[6] = 1u;
    row[7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
    row[8] = (unsigned)(__builtin_amdgcn_s_memtime() - st_t0);
    row[9] = (unsigned)(tile_hi > tile_lo ? tile_hi - tile_lo : 0);
  }
#endif
}

// ------------------------------------------------------------------------------------------------ graph, read-gather
// The same weight / bias gradient without the z images (SAR_GRAPH_FEW_DENSE tables, conv_graph_cn8.hip): nearly every gather list
// of the NTU adjacency is {one joint, weight 1} or empty, so z_k is a PERMUTATION of the raw tile's columns -- and the transposed
// LDS read takes its column from the lane's address (lane 4q+p of a 16-lane group supplies column q): the A fragment of slice k
// is read straight from the raw image at the listed joints.  The few other lists are built per frame as virtual joints behind
// the raw columns (fp32 chain of the unit builder, rounded once).  In wgrad_graph_cn8_kernel the z builder was 3 700-4 200 of
// a tile's ~6 400 cycles (tools/stamps8.sh) and its images half of the LDS.  With the images gone there is room for TWO tiles:
// the units are staged by LDS-DMA into the buffer that is not being multiplied (no staging registers, no store phase, two
// barriers per tile).
//   X plane: [0, 125) raw units of the tile's 5 frames | [125, 128) zero | [128 + 4 f + r] virtual joint r of frame f
//   per lane: the unit of every (slice, k-step, half) it supplies, one byte each (4 registers per gathered slice)
constexpr int NVW = 4;   // virtual joints per frame this kernel has room for (NTU forward tables: 2)
template <bool ID0>
__global__ __launch_bounds__(256, 2) void wgrad_graph2_cn8_kernel(const WgradK8 k, const int nv_asserted) {
  constexpr int FT = 5, NLIVE = FT * WV, NPOS = 128, KSTEPS = NPOS / 16;
  constexpr int PSD = pad_stride(NPOS);              // 132
  constexpr int VBASE = NPOS, ZU = NLIVE;             // first virtual unit; a unit that is always zero
  constexpr int PSX = pad_stride(VBASE + FT * NVW);   // 148
  constexpr int XJ = (NLIVE + 63) / 64;
  constexpr int T0 = ID0 ? 1 : 0, NG = 3 - T0;        // gathered slices T0 .. 2
  static_assert(KSTEPS == 8 && PSX < 256, "one byte per (k-step, half)");
  constexpr int BUF = 8 * PSX + 8 * PSD;            // one tile: raw src planes (+ virtual joints) and dout planes
  __shared__ uint4 lds[2 * BUF];
  __shared__ float bred[4][48];
  __shared__ int vmap[3 * WV];
  __shared__ int vl_idx[NVW][4];
  __shared__ float vl_wt[NVW][4];
  __shared__ int nv_s;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const sar_wgrad_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int split, by, bz;
  if (!xcd_split(d.nsplit, k.gy, k.gz, split, by, bz)) return;
#ifdef SAR_CN8_STAMPS   // [0] store_raw (waits for the loads), [1] barrier, [2] virtual joints + barrier, [3] k-steps + closing barrier, [4] prologue, [5] epilogue
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_last = st_t0;
#endif
  const int m0 = by * 64, c0 = bz * 64;
  const int wmb = wave & 1, wc = wave >> 1;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum[2][3][8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum[q][t][e] = 0.f;
  // everything behind the raw columns starts as zero: [125, 128) and the dout tail stay so, the virtual joints are rewritten per tile
  for (int i = tid; i < 2 * 8 * (PSX - NLIVE); i += 256) {
    const int bf = i / (8 * (PSX - NLIVE)), r = i - bf * 8 * (PSX - NLIVE);
    lds[bf * BUF + (r / (PSX - NLIVE)) * PSX + NLIVE + r % (PSX - NLIVE)] = make_uint4(0u, 0u, 0u, 0u);
  }
  for (int i = tid; i < 2 * 8 * (PSD - NLIVE); i += 256) {
    const int bf = i / (8 * (PSD - NLIVE)), r = i - bf * 8 * (PSD - NLIVE);
    lds[bf * BUF + 8 * PSX + (r / (PSD - NLIVE)) * PSD + NLIVE + r % (PSD - NLIVE)] = make_uint4(0u, 0u, 0u, 0u);
  }

  const int tps = (k.ntiles + d.nsplit - 1) / d.nsplit;
  const int tile_lo = split * tps;
  const int tile_hi = (tile_lo + tps < k.ntiles) ? tile_lo + tps : k.ntiles;
  const int seq = d.T_src * WV;
  const bool do_bias = d.bsize > 0 && bz == 0;

  // colsum(A_k)[w] of the dout units this lane stages (bias gradient)
  float cs[XJ][3];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int i = lane + 64 * j;
#pragma unroll
    for (int t = 0; t < 3; ++t) cs[j][t] = (i < NLIVE && d.g_colsum) ? d.g_colsum[t * WV + i % WV] : 0.f;
  }
  // classify the 3 V gather lists (wave 0; conv_graph2_cn8_kernel): {one entry, weight 1} -> that raw joint, empty -> the zero unit,
  // anything else -> a virtual joint, ranked in (slice, joint) order
  if (wave == 0) {
    constexpr int NPASS = (3 * WV + 63) / 64;
    int ei[NPASS][4];
    float ew[NPASS][4];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int p = ps * 64 + lane;
      const int pc = p < 3 * WV ? p : 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) ei[ps][j] = d.g_idx[pc * 4 + j], ew[ps][j] = d.g_wt[pc * 4 + j];
    }
    const int nz0 = d.nz[0], nz1 = d.nz[1], nz2 = d.nz[2];
    int base = 0;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int p = ps * 64 + lane;
      const bool in = p < 3 * WV;
      const int tp = in ? p / WV : 0;
      const int nzl = tp == 0 ? nz0 : (tp == 1 ? nz1 : nz2);
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
      if (in) vmap[p] = virt ? ((rank < nv_asserted && rank < NVW) ? WV + rank : -1) : (cnt == 1 ? first : -1);
      if (virt && rank < nv_asserted && rank < NVW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) vl_idx[rank][j] = ei[ps][j], vl_wt[rank][j] = ew[ps][j];
      }
      base += __popcll(bal);
    }
    if (lane == 0) nv_s = base < nv_asserted ? (base < NVW ? base : NVW) : (nv_asserted < NVW ? nv_asserted : NVW);
  }
  __syncthreads();   // vmap, virtual lists, zero fill
  const int NV = nv_s;

  // the units this lane supplies to the transposed reads: position p = 16 ks + 8 (G >> 1) + q (+ 4 for the second read)
  const int G = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;
  unsigned goffs[NG][KSTEPS / 2];
#pragma unroll
  for (int t = T0; t < 3; ++t)
#pragma unroll
    for (int kp = 0; kp < KSTEPS / 2; ++kp) {
      unsigned w = 0;
#pragma unroll
      for (int b4 = 0; b4 < 4; ++b4) {
        const int ks = 2 * kp + (b4 >> 1), h2 = b4 & 1;
        const int p = 16 * ks + 8 * (G >> 1) + q4 + 4 * h2;
        int u = ZU;
        if (p < NLIVE) {
          const int f = p / WV, v = p - f * WV;
          const int vm = vmap[t * WV + v];
          u = vm < 0 ? ZU : (vm < WV ? f * WV + vm : VBASE + f * NVW + (vm - WV));
        }
        w |= (unsigned)u << (8 * b4);
      }
      goffs[t - T0][kp] = w;
    }
  // mini-builder: item = (plane, frame, virtual joint), one per thread (8 * 5 * NVW <= 256)
  int vb_dst = -1, vb_src[4] = {0, 0, 0, 0};
  float vb_wt[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid < 8 * FT * NV) {
    const int pl = tid / (FT * NV), rem = tid - pl * (FT * NV);
    const int f = rem / NV, r = rem - f * NV;
    vb_dst = pl * PSX + VBASE + f * NVW + r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vb_src[j] = pl * PSX + f * WV + vl_idx[r][j];
      vb_wt[j] = vl_wt[r][j];
    }
  }

  // Staging is LDS-DMA (buffer_load ... lds, conv_gemm_cn8_dma.hip): a CN8 unit needs no arithmetic between HBM and LDS, so the
  // tile's 16 planes go straight into the OTHER buffer while this one is multiplied -- no staging registers, no store phase.
  // Wave w requests planes w and w + 4 of src and dout, two 64-unit pieces each; lanes beyond the tile's 125 columns (and columns
  // beyond the sequence) are rejected by the range check and write zeros: units [125, 128) stay zero.
  int tb = 0, tt0 = 0;   // sequence / first frame of the tile about to be requested
  auto seek = [&](int tile) { tb = tile / k.TPS, tt0 = (tile - tb * k.TPS) * FT; };
  auto issue_dma = [&](int bufo) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ip = wave + 4 * q;
      const int g = c0 / 8 + ip, gd = m0 / 8 + ip;
      const bool live = g < k.Gs, dlive = gd < k.Gd;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)d.src + ((int64_t)(live ? g : 0) * d.ld_src + (int64_t)tb * seq) * 16), 0,
          live ? (unsigned)seq * 16u : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
          (void*)((const char*)d.dout + ((int64_t)(dlive ? gd : 0) * d.ld_dout + (int64_t)tb * seq) * 16), 0,
          dlive ? (unsigned)seq * 16u : 0u, 0x00020000);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = lane + 64 * j;
        const unsigned vo = i < NLIVE ? (unsigned)((tt0 * WV + i) * 16) : 0x7fffffffu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + bufo + ip * PSX + 64 * j), 16, vo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(lds + bufo + 8 * PSX + ip * PSD + 64 * j), 16, vo, 0, 0, 0);
      }
    }
    tt0 += FT;               // the next tile of this split
    if (tt0 >= d.T_out) tt0 = 0, ++tb;
  };
  auto bias_sums = [&](const uint4* Dimg) {   // this lane's dout units: planes wave, wave + 4, columns lane + 64 j
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ip = wave + 4 * q;
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        float f[8];
        cn8_unpack(Dimg[ip * PSD + lane + 64 * j], f);   // (columns 125 .. 127: zero units, cs = 0)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[q][t][e] = fmaf(f[e], cs[j][t], bsum[q][t][e]);
      }
    }
  };
  auto build_virtual = [&](uint4* Ximg) {
    if (vb_dst >= 0) {
      float z[8], x[8];
      cn8_unpack(Ximg[vb_src[0]], x);
#pragma unroll
      for (int c = 0; c < 8; ++c) z[c] = vb_wt[0] * x[c];
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        cn8_unpack(Ximg[vb_src[j]], x);
#pragma unroll
        for (int c = 0; c < 8; ++c) z[c] = vb_wt[j] != 0.f ? fmaf(vb_wt[j], x[c], z[c]) : z[c];
      }
      Ximg[vb_dst] = cn8_pack(z);
    }
  };

  // lane constants of the transposed reads (relative to a buffer): plane + channel-quad part (gathered reads add the unit), and the plain ones
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  const unsigned xg_rel = (unsigned)((((4 * wc + 2 * (G & 1) + (pp >> 1)) * PSX) * 16) + 8 * (pp & 1));
  const unsigned x_rel = tr_lane_bytes(lane, PSX, 4 * wc);
  const unsigned b_rel = (unsigned)(8 * PSX * 16) + tr_lane_bytes(lane, PSD, 4 * wmb);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  auto gfrag = [&](unsigned xg_base, unsigned w, int b4) {   // bytes b4, b4 + 1 of w: the units of the two reads of one k-step
    const unsigned u0 = (w >> (8 * b4)) & 0xffu, u1 = (w >> (8 * b4 + 8)) & 0xffu;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(xg_base + (u0 << 4)));
    const s16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(xg_base + (u1 << 4)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], h4[0], h4[1], h4[2], h4[3]};
    return *reinterpret_cast<const bf16x8*>(&v);
  };

  // Happens-before of the two buffers: the DMA of tile i + 1 into buffer (i + 1) & 1 is issued behind barrier A of tile i; that
  // buffer was last read (bias sums, mini-builder, k-steps) for tile i - 1, which every wave finished before it joined barrier A
  // of tile i.  The DMA of tile i is complete (vmcnt) in every wave before that wave joins barrier A of tile i.
  if (tile_lo < tile_hi) {
    seek(tile_lo);
    issue_dma((tile_lo & 1) * BUF);
  }
  WSTAMP8(4);
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int bufo = (tile & 1) * BUF;
    WSTAMP8(0);
    __syncthreads();         // A: this tile's units are in LDS (vmcnt(0) in front of the barrier)
    WSTAMP8(1);
    if (tile + 1 < tile_hi) issue_dma(((tile + 1) & 1) * BUF);
    if (do_bias) bias_sums(lds + bufo + 8 * PSX);
    if (NV > 0) {   // uniform
      build_virtual(lds + bufo);
      __syncthreads();       // B: virtual joints complete
    }
    WSTAMP8(2);
    // k-steps, fully unrolled (the gather units are register bytes), software-pipelined over two fragment sets
    SAR_LDS_SKEW();   // this wave reads buffer tile & 1 late: the others may only request tile + 2 into it behind barrier A of tile + 1
    const unsigned xg_base = lds0 + bufo * 16 + xg_rel, x_base = lds0 + bufo * 16 + x_rel, b_base = lds0 + bufo * 16 + b_rel;
    bf16x8 fb[2], fa[2][3];
    auto frag_load = [&](int ks, bf16x8& bv, bf16x8 (&av)[3]) {
      bv = tr_frag(b_base + ks * 256);
      if (ID0) av[0] = tr_frag(x_base + ks * 256);
#pragma unroll
      for (int t = T0; t < 3; ++t) av[t] = gfrag(xg_base, goffs[t - T0][ks >> 1], 2 * (ks & 1));
    };
    auto frag_mma = [&](const bf16x8& bv, const bf16x8 (&av)[3]) {
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[t], bv, acc[t], 0, 0, 0);
    };
    frag_load(0, fb[0], fa[0]);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + 1 < KSTEPS) frag_load(ks + 1, fb[(ks + 1) & 1], fa[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      frag_mma(fb[ks & 1], fa[ks & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    WSTAMP8(3);
  }

  float* slab = d.slab + (int64_t)split * (d.wsize + d.bsize);
  const int m = m0 + wmb * 32 + l31;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + wc * 32 + mfma_row(r, hi);
      if (c < d.Kc && m < d.M) slab[(int64_t)t * d.w_stride_tap + (int64_t)c * d.w_stride_c + m] = acc[t][r];
    }
  if (do_bias) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float s = wave_sum_dpp(bsum[q][t][e]);
          if (lane == 0) bred[wave][(q * 3 + t) * 8 + e] = s;
        }
    __syncthreads();
    if (tid < 192) {   // (slice t, channel m0 + mm): dout plane mm / 8 = wave' + 4 q
      const int t = tid >> 6, mm = tid & 63;
      const int pl = mm >> 3, e = mm & 7;
      if (m0 + mm < d.M) slab[d.wsize + (int64_t)t * d.M + m0 + mm] = bred[pl & 3][((pl >> 2) * 3 + t) * 8 + e];
    }
  }
#ifdef SAR_CN8_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  WSTAMP8(5);
  if (tid == 0 && blockIdx.x < WSTAMP_WG) {
    unsigned* row = g_wstamps8[blockIdx.x];
    for (int i2 = 0; i2 < 6; ++i2) row[i2] = (unsigned)st_acc[i2];
    row[6] = 1u;
    row[7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
    row[8] = (unsigned)(__builtin_amdgcn_s_memtime() - st_t0);
    row[9] = (unsigned)(tile_hi > tile_lo ? tile_hi - tile_lo : 0);
  }
#endif
}

template <int TAPS, int STRIDE, bool WIDE = false>
void launch_t(const WgradK8& k, hipStream_t st) {
  const int nwork = k.d.nsplit * k.gy * k.gz;
  hipLaunchKernelGGL((wgrad_cn8_kernel<TAPS, STRIDE, WIDE>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, st, k);
}

template <int NZ0, int NZ1, int NZ2>
void launch_g(const WgradK8& k, bool id0, hipStream_t st) {
  const int nwork = k.d.nsplit * k.gy * k.gz;
  if (id0) hipLaunchKernelGGL((wgrad_graph_cn8_kernel<NZ0, NZ1, NZ2, true>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, st, k);
  else hipLaunchKernelGGL((wgrad_graph_cn8_kernel<NZ0, NZ1, NZ2, false>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, st, k);
}

}  // namespace

#ifdef SAR_CN8_STAMPS
extern "C" int sar_debug_wgrad8_stamps(unsigned long long* out10, int reset) {
  static unsigned host[WSTAMP_WG][10];
  if (out10) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wstamps8), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 10; ++i) out10[i] = 0;
    for (int w = 0; w < WSTAMP_WG; ++w)
      for (int i = 0; i < 10; ++i) out10[i] += host[w][i];
  }
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wstamps8)) != hipSuccess || hipMemset(p, 0, sizeof(host)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

extern "C" int sar_conv_wgrad_cn8_tile_frames(int mode) { return mode == SAR_CONV_GRAPH ? 5 : 7; }

extern "C" int sar_conv_wgrad_cn8(const sar_wgrad_desc* d, int slice0_identity, sar_stream_t s) {
  SAR_REQUIRE(d != nullptr, "sar_conv_wgrad_cn8: null descriptor");
  SAR_REQUIRE(d->mode == SAR_CONV_TEMPORAL || d->mode == SAR_CONV_GRAPH, "sar_conv_wgrad_cn8: bad mode %d", d->mode);
  SAR_REQUIRE(d->B > 0 && d->T_src > 0 && d->T_out > 0 && d->Kc > 0 && d->M > 0, "sar_conv_wgrad_cn8: bad sizes");
  if (d->V != WV) {
    sar_set_error("sar_conv_wgrad_cn8: built for V = 25 joints (got %d)", d->V);
    return SAR_E_UNSUP;
  }
  SAR_REQUIRE(d->src && d->dout && d->slab && (((uintptr_t)d->src | (uintptr_t)d->dout) & 15) == 0,
              "sar_conv_wgrad_cn8: null / misaligned src, dout or slab");
  SAR_REQUIRE(d->nsplit >= 1 && d->nsplit <= 65535, "sar_conv_wgrad_cn8: nsplit %d out of range", d->nsplit);
  SAR_REQUIRE(d->ld_src >= (int64_t)d->B * d->T_src * d->V && d->ld_dout >= (int64_t)d->B * d->T_out * d->V,
              "sar_conv_wgrad_cn8: leading dimension smaller than B*T*V");
  SAR_REQUIRE((int64_t)d->T_src * d->V < (1 << 26) && (int64_t)d->T_out * d->V < (1 << 26), "sar_conv_wgrad_cn8: sequence too long");
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv_wgrad_cn8: pro_scale/pro_shift mismatch");
  SAR_REQUIRE(d->wsize > 0 && d->bsize >= 0, "sar_conv_wgrad_cn8: bad slab sizes");
  WgradK8 k;
  k.d = *d;
  k.Gs = (d->Kc + 7) / 8;
  k.Gd = (d->M + 7) / 8;
  k.gy = (d->M + 63) / 64;
  const int ft = sar_conv_wgrad_cn8_tile_frames(d->mode);
  k.TPS = (d->T_out + ft - 1) / ft;
  k.ntiles = d->B * k.TPS;
  if (d->mode == SAR_CONV_GRAPH) {
    SAR_REQUIRE(d->taps == 3 && d->T_src == d->T_out && d->g_idx && d->g_wt, "sar_conv_wgrad_cn8: graph mode needs 3 slices + tables");
    SAR_REQUIRE(d->bsize == 0 || (d->bsize == 3 * (int64_t)d->M && d->g_colsum), "sar_conv_wgrad_cn8: graph bias slab is [3][M]");
    SAR_REQUIRE(!d->pro_scale, "sar_conv_wgrad_cn8: no prologue in graph mode");
    for (int i = 0; i < 3; ++i)
      SAR_REQUIRE(d->nz[i] >= 1 && d->nz[i] <= 4, "sar_conv_wgrad_cn8: adjacency slice %d needs %d gather entries (max 4)", i, d->nz[i]);
    SAR_REQUIRE(!slice0_identity || d->nz[0] == 1, "sar_conv_wgrad_cn8: an identity slice has one gather entry");
    k.gz = (d->Kc + 63) / 64;
    // read-gather kernel: SAR_GRAPH_FEW_DENSE tables with at most NVW virtual joints (SAR_WGRAD_READ_GATHER=0: the z-image kernel)
    static const bool rg = [] { const char* e = getenv("SAR_WGRAD_READ_GATHER"); return !(e && e[0] == '0'); }();
    const int nvd = (d->g_flags >> SAR_GRAPH_FEW_DENSE_SHIFT) & 0xff;
    if (rg && (d->g_flags & SAR_GRAPH_FEW_DENSE) && nvd <= NVW) {
      const int nwork = k.d.nsplit * k.gy * k.gz;
      if (slice0_identity) hipLaunchKernelGGL((wgrad_graph2_cn8_kernel<true>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, as_stream(s), k, nvd);
      else hipLaunchKernelGGL((wgrad_graph2_cn8_kernel<false>), dim3(((nwork + 7) / 8) * 8), dim3(256), 0, as_stream(s), k, nvd);
    } else
    if (d->nz[0] == 1 && d->nz[1] == 1) launch_g<1, 1, 4>(k, slice0_identity != 0, as_stream(s));
    else if (d->nz[0] == 1 && d->nz[2] == 1) launch_g<1, 4, 1>(k, slice0_identity != 0, as_stream(s));
    else launch_g<4, 4, 4>(k, false, as_stream(s));
  } else {
    SAR_REQUIRE(d->bsize == 0 || d->bsize == d->M, "sar_conv_wgrad_cn8: temporal bias slab is [M]");
    const bool ok = (d->taps == 9 && d->stride == 1 && d->pad >= 0 && d->pad <= 8 && d->T_src == d->T_out) ||
                    (d->taps == 9 && d->stride == 2 && (d->pad == 3 || d->pad == 4)) || (d->taps == 1 && d->pad == 0 && (d->stride == 1 || d->stride == 2));
    if (!ok) {
      sar_set_error("sar_conv_wgrad_cn8: built for 9 taps (stride 1; stride 2 with the SAME pads 3 / 4) and 1 tap (stride 1 / 2, pad 0); got taps %d stride %d pad %d",
                    d->taps, d->stride, d->pad);
      return SAR_E_UNSUP;
    }
    static const bool wide = [] { const char* e = getenv("SAR_CN8_WGRAD_WIDE"); return e && e[0] == '1'; }();   // experiment switch
    if (d->taps == 9 && d->stride == 1 && wide) {
      k.gz = (d->Kc + 63) / 64;
      launch_t<9, 1, true>(k, as_stream(s));
    } else if (d->taps == 9 && d->stride == 1) {
      k.gz = (d->Kc + 31) / 32;
      launch_t<9, 1>(k, as_stream(s));
    } else if (d->taps == 9) {
      k.gz = (d->Kc + 31) / 32;
      launch_t<9, 2>(k, as_stream(s));
    } else if (d->stride == 1) {
      k.gz = (d->Kc + 63) / 64;
      launch_t<1, 1>(k, as_stream(s));
    } else {
      k.gz = (d->Kc + 63) / 64;
      launch_t<1, 2>(k, as_stream(s));
    }
  }
  SAR_LAUNCH_CHECK("sar_conv_wgrad_cn8");
  return 0;
}

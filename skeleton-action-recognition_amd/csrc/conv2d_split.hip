// conv2d_split.hip -- the 3x3 / stride-1 convolutions of the 1-channel ResNet-18 (models/resnet18.py:5-14,26-72: conv3x3 of the
// BasicBlocks) and their data gradients with fp32 results on the fp16 / bf16 matrix pipe: the "split" arithmetic of
// conv_gemm_split.hip (read its header and split_terms.h first) carried over to "image rows x image columns" (gfx950).
//
//   out[m, (b,h,w)] = sum_{kh,kw} sum_c W[kh][kw][c][m] * pro(src)[c, (b, h + kh - 1, w + kw - 1)]   (zero outside the image) ; epilogue
//
// The data gradient of such a convolution IS such a convolution of dout with the taps mirrored and (c, m) exchanged: the caller
// packs that view of the weight tensor (sar_pack_weights_split_batch items address elements by strides; a negative tap stride
// mirrors) and passes the transposed descriptor; one kernel serves both.
//
// Design (MI355X) = conv_gemm_split_kernel's, with conv2d.hip's padded image:
//  * tile 64 (m) x 256 columns = TH whole output rows of ONE image (64x64: 4 rows, 32x32: 8, 16x16: the image) or NI whole images
//    (8x8: 4); 4 waves side by side, wave tile 64 x 64.
//  * stage = 8 source channels; the staged window is the tile's rows + one halo row above / below, each row with one zero column left /
//    right, MATERIALISED in LDS ((TH + 2) x (W + 2) units per image, <= 512): a tap (kh, kw) is the uniform unit offset
//    kh (W + 2) + kw and the inner loop carries no bounds logic.  The 16 k of one MFMA are 8 channels x 2 taps (lanes 32-63 read the
//    next tap): nine taps = five k-steps, the last one half empty (zero weight slot / zero column).
//  * weights: term images [term][tap][G][M] written once per step by sar_pack_weights_split_batch, a stage's 27 pieces of 64 rows
//    by LDS-DMA; source: split in the stager behind the folded BatchNorm + ReLU (padding stays exactly 0).
//  * 50 KB of LDS (f16x3a): three workgroups per CU.  Epilogue: conv_epi_f32.h (NONE / STATS / MASK / ADD), partial sums
//    [M][4 ntiles][2].
#include "sar_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

#include "split_terms.h"

#include "conv_epi_f32.h"

constexpr int KC8 = 8;   // source channels per stage

struct C2S {
  sar_conv_desc d;   // the fields the shared epilogue and the stager read: M, Kc, src, out, aux*, pro_*, partials, epi
  const uint4* wp;   // term images [term][tap][G][M]
  int G;
  int B, H, W, TH, TPI, NI, Wq, IRW, RW, nparts, ntiles, ny;
  int ksplit;        // > 1 (DEEP kernels): the channel stages are divided among ksplit workgroups per output tile, each storing its raw fp32
                     // partial tile into slab[part][M][npix]; conv2d_split_reduce_kernel sums them in order and applies the epilogue
  float* slab;
  int64_t npix;
  int rb_major;      // workgroup order: 0 = the row blocks of a tile are neighbours (one XCD reads a source tile once), 1 = the tiles of a
                     // row block are (one XCD reads 1 / ny of the weight images: small feature maps with large weight tensors)
  const unsigned* src_bound;
  const unsigned* w_bound;
};

// NS: 32-column blocks per wave (2: tiles of 256 columns; 1: 128).  DEEP: both operand images double-buffered, ONE barrier per
// stage, the W DMA and the source loads of stage s + 1 in flight during the matrix phase of stage s -- for launches of at most a
// few workgroups per CU (small feature maps), where no other workgroup hides a stage's load latencies: 78-94 KB of LDS.
template <int AR, int NS, int DEEP>
__global__ __launch_bounds__(256, DEEP ? ((NS == 1 && AR == AR_H3A) ? 2 : 1) : (AR == AR_H3A ? 3 : 2)) void conv2d_split_kernel(const C2S k) {
  constexpr int NTA = ar_nta(AR), NTB = ar_ntb(AR), NPROD = ar_nprod(AR);
  constexpr bool SCALED = ar_f16(AR);
  constexpr int TAPS = 9, BM = 64, MS = 2, WN = 4;
  constexpr int CJ = NS, ZCOL = 256 * CJ, SCOLS = ZCOL + 1;   // staged elements 0 .. 256 CJ - 1, then the always-zero column
  constexpr int NBUF = DEEP ? 2 : 1;
  constexpr int WPIECES = NTA * TAPS, WPB = WPIECES * 64, ZSLOT = NBUF * WPB;   // weight pieces of 64 units per buffer, then the zero slot
  constexpr int NTA_LDS = nta_lds(AR);   // images of the weights that are moved to / read from LDS
  constexpr int WU = ZSLOT + 64, SUB = NTB * SCOLS, SU = NBUF * SUB;
  constexpr int PAREA_U = 4 * 16 * 65 / 4;   // the epilogue's transpose area aliases the image
  constexpr int IMG_U = (WU + SU) > PAREA_U ? (WU + SU) : PAREA_U;
  constexpr int PPW = (WPIECES + 3) / 4;
  constexpr int KCMAX = 512;
  __shared__ uint4 smem_u[IMG_U + BM + KCMAX / 2];   // image | per-row parameters (float4) | folded BN (scale, shift) per src channel
  uint4* Wl = smem_u;
  uint4* Sl = smem_u + WU;
  float* smem = reinterpret_cast<float*>(smem_u);
  float4* rowp = reinterpret_cast<float4*>(smem_u + IMG_U);
  float2* bnp = reinterpret_cast<float2*>(smem_u + IMG_U + BM);
  const sar_conv_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave;
  // workgroup -> (tile, row block), XCD-aware: the row blocks of a tile sit on one XCD (conv_gemm.hip)
  const int ny = k.ny, nwork = k.ntiles * ny * k.ksplit;
  int w = blockIdx.x;
  {
    const int per = (nwork + 7) / 8;
    const int xcd = w & 7, slot = w >> 3;
    w = xcd * per + slot;
    if (w >= nwork || slot >= per) return;
  }
  const int kpart = w % k.ksplit;
  w /= k.ksplit;
  const int tile = k.rb_major ? w % k.ntiles : w / ny;
  const int b = k.NI > 1 ? tile * k.NI : tile / k.TPI;
  const int h0 = k.NI > 1 ? 0 : (tile - b * k.TPI) * k.TH;
  const int m0 = (k.rb_major ? w / k.ntiles : w - tile * ny) * BM;
  const int H = k.H, W = k.W, Wq = k.Wq, opix = H * W;

  // ---- per-lane column geometry: column p of the tile = pixel (image im, row hl, column wo); its unit of tap (kh, kw) sits at
  // boff + kh Wq + kw of the staged window (origin: row h0 - 1, column -1 of image im)
  bool colok[NS];
  int64_t coln[NS];
  int boff[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    const int im = k.NI > 1 ? p / opix : 0;
    const int pp = p - im * opix;
    int hl = pp / W;
    int wo = pp - hl * W;
    colok[ns] = im < k.NI && (b + im) < k.B && hl < k.TH && (h0 + hl) < H;
    if (!colok[ns]) { hl = 0; wo = 0; }
    const int bi = b + (colok[ns] ? im : 0);
    coln[ns] = ((int64_t)bi * H + (h0 + hl)) * W + wo;
    boff[ns] = (colok[ns] ? im * k.IRW : 0) + hl * Wq + wo;   // a dead column reads pixel (0, 0) of the window: zeroed behind the loop
  }
  // k-step q multiplies taps 2 q (lanes 0-31) and 2 q + 1 (lanes 32-63); the 10th tap does not exist (zero slot / zero column)
  const int abase = hi * 64 + l31;

  int ea = 0, ew = 0;
  bool nonfin = false;   // an operand bound holds Inf / NaN bits: every output of the launch is NaN (split_scale.h)   // fp16 arithmetics: operand scale exponents (wave-uniform)
  if (SCALED) {
    ea = scale_exp(*k.src_bound);
    ew = scale_exp(*k.w_bound);
    nonfin = bound_nonfinite(*k.src_bound) || bound_nonfinite(*k.w_bound);
  }
  const float h3_sa = __builtin_ldexpf(1.f, ea);
  for (int c = tid; c < KCMAX; c += 256) {   // the folded prologue, with the source scale folded in (a power of two: exact)
    float2 p = make_float2(h3_sa, 0.f);
    if (d.pro_scale && c < d.Kc) p = make_float2(d.pro_scale[c] * h3_sa, d.pro_shift[c] * h3_sa);
    bnp[c] = p;
  }
  if (tid < NBUF * NTB) Sl[tid * SCOLS + ZCOL] = make_uint4(0u, 0u, 0u, 0u);
  if (tid >= 64 && tid < 128) Wl[ZSLOT + tid - 64] = make_uint4(0u, 0u, 0u, 0u);
  f32x16 acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

  const int img = opix;
  const float* src_b = d.src + (int64_t)b * img;

  // ---- source staging: a lane owns staged elements tid and tid + 256; offsets and masks once
  int svo[CJ];
  bool sok[CJ];
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int e = tid + 256 * j;
    const int im = e / k.IRW, ei = e - im * k.IRW;
    const int r = ei / Wq, q = ei - r * Wq;
    const int hs = h0 - 1 + r, ws = q - 1;
    sok[j] = e < k.RW && (b + im) < k.B && (unsigned)hs < (unsigned)H && (unsigned)ws < (unsigned)W;   // else zero padding
    svo[j] = sok[j] ? (im * img + hs * W + ws) * 4 : 0;
  }
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  float sreg[CJ][8];
  auto issue_s_loads = [&](int c0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = c0 + q;
      const int cg = c < d.Kc ? c : 0;   // wave-uniform
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, k.NI * img * 4, 0x00020000);
#pragma unroll
      for (int j = 0; j < CJ; ++j) sreg[j][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo[j], 0, 0));
    }
  };
  float psc[8], psh[8];
  auto load_bnp = [&](int c0) {   // ahead of the stage's W DMA (an LDS read behind an LDS-DMA makes the compiler wait for the DMA)
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      const float4 p2 = *reinterpret_cast<const float4*>(&bnp[c0 + 2 * q2]);
      psc[2 * q2] = p2.x, psh[2 * q2] = p2.y, psc[2 * q2 + 1] = p2.z, psh[2 * q2 + 1] = p2.w;
    }
  };
  auto store_piece = [&](int c0, int buf, auto J) {   // staged element tid + 256 j: folded BN + ReLU, split, NTB ds_write_b128
    constexpr int j = decltype(J)::value;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float val = fmaxf(fmaf(sreg[j][q], psc[q], psh[q]), relu_lo);
      v[q] = (sok[j] && c0 + q < d.Kc) ? val : 0.f;   // zero padding stays exactly 0 behind the folded BatchNorm
    }
    uint4 u[NTB];
    split8<AR, false>(v, u, 1.f);
#pragma unroll
    for (int t = 0; t < NTB; ++t) Sl[buf * SUB + t * SCOLS + tid + 256 * j] = u[t];
  };
  auto store_s = [&](int c0, int buf) {
    store_piece(c0, buf, std::integral_constant<int, 0>());
    if constexpr (CJ == 2) store_piece(c0, buf, std::integral_constant<int, 1>());
  };
  // ---- weight pieces by LDS-DMA: piece p = term * 9 + tap = 64 rows of one (term, tap) of channel group g
  const unsigned wbytes = (unsigned)((int64_t)NTA * TAPS * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  const unsigned wvo = (m0 + lane) < d.M ? (unsigned)((m0 + lane) * 16) : 0x80000000u;   // rows beyond M: rejected -> 0
  auto issue_w_dma = [&](int g, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + 4 * i;   // wave-uniform
      if (p < NTA_LDS * TAPS)   // (f16x3a: the third weight image is formed in registers: split_terms.h third_image)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(Wl + buf * WPB + p * 64), 16, wvo, (p * k.G + g) * d.M * 16, 0, 0);
    }
  };

  // this workgroup's channel stages (ksplit == 1: all of them)
  const int nst_all = (d.Kc + KC8 - 1) / KC8;
  const int s_lo = (int)((int64_t)kpart * nst_all / k.ksplit), nst = (int)((int64_t)(kpart + 1) * nst_all / k.ksplit);
  issue_s_loads(s_lo * KC8);
  __syncthreads();   // bnp, zero column / slot
  load_bnp(s_lo * KC8);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  issue_w_dma(s_lo, 0);

  // A k-step = fetch (NTA MS + NTB NS ds_read_b128) + NPROD MS NS MFMAs.  The fragments of k-step q + 1 are requested BEFORE the
  // MFMAs of k-step q issue (two register sets, the order pinned by sched_barrier): left to the compiler every group of four
  // reads was followed by s_waitcnt and its MFMAs in the same four registers -- with one or two waves per SIMD (small feature
  // maps) the LDS round trips were exposed: 6 000 cycles per stage for 1 920 of matrix work.
  uint4 fa[2][NTA][MS], fb[2][NTB][NS];
  auto fetch = [&](auto Q, int buf, uint4 (&a)[NTA][MS], uint4 (&bq)[NTB][NS]) {
    constexpr int q = decltype(Q)::value;
    constexpr bool last = q == 4;
    const bool dead = last && hi;   // this lane's half of the last k-step has no tap
#pragma unroll
    for (int t = 0; t < NTA_LDS; ++t)
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a[t][ms] = Wl[dead ? ZSLOT + l31 + ms * 32 : buf * WPB + t * (TAPS * 64) + abase + q * 128 + ms * 32];
    if constexpr (NTA_LDS < NTA) {
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a[NTA - 1][ms] = third_image(a[0][ms]);
    }
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      constexpr int t0 = 2 * q, t1 = 2 * q + 1;   // tap (kh, kw) = unit offset kh Wq + kw
      const int to = hi ? (t1 / 3) * Wq + (t1 % 3) : (t0 / 3) * Wq + (t0 % 3);
      const int bo = dead ? ZCOL : boff[ns] + to;
#pragma unroll
      for (int t = 0; t < NTB; ++t) bq[t][ns] = Sl[buf * SUB + t * SCOLS + bo];
    }
  };
  auto mma = [&](uint4 (&a)[NTA][MS], uint4 (&bq)[NTB][NS]) {
#pragma unroll
    for (int p = 0; p < NPROD; ++p) {
      const int i = ar_pi(AR, p), j = ar_pj(AR, p);
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          if (ar_f16(AR))
            acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<f16x8*>(&a[i][ms]),
                                                                 *reinterpret_cast<f16x8*>(&bq[j][ns]), acc[ms][ns], 0, 0, 0);
          else
            acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a[i][ms]),
                                                                  *reinterpret_cast<bf16x8*>(&bq[j][ns]), acc[ms][ns], 0, 0, 0);
        }
    }
  };
  auto ksteps = [&](int buf) {
    if constexpr (!DEEP) {   // three workgroups per CU hide the round trips; 168 registers do not hold a second fragment set
      fetch(std::integral_constant<int, 0>(), buf, fa[0], fb[0]);
      mma(fa[0], fb[0]);
      fetch(std::integral_constant<int, 1>(), buf, fa[0], fb[0]);
      mma(fa[0], fb[0]);
      fetch(std::integral_constant<int, 2>(), buf, fa[0], fb[0]);
      mma(fa[0], fb[0]);
      fetch(std::integral_constant<int, 3>(), buf, fa[0], fb[0]);
      mma(fa[0], fb[0]);
      fetch(std::integral_constant<int, 4>(), buf, fa[0], fb[0]);
      mma(fa[0], fb[0]);
      return;
    }
    fetch(std::integral_constant<int, 0>(), buf, fa[0], fb[0]);
    fetch(std::integral_constant<int, 1>(), buf, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    fetch(std::integral_constant<int, 2>(), buf, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    fetch(std::integral_constant<int, 3>(), buf, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    fetch(std::integral_constant<int, 4>(), buf, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa[0], fb[0]);
  };
  // DEEP: the matrix phase of stage s with the stager work of stage s + 1 INSIDE it -- with one or two waves per SIMD nothing else
  // fills the vector ALU / memory issue slots while the matrix pipe runs, and a wave issues in order: the split of a staged
  // element (~75 vector instructions) is placed between the MFMAs of a k-step (one MFMA, then six vector instructions), the
  // requests of stage s + 2 behind the last read of the staging registers.  (tools/c2s_probe.py: matrix phase, stager and W DMA
  // of a 512-channel 8 x 8 launch took 82 + 50 + 25 us one after the other.)
  auto mma_with = [&](uint4 (&a)[NTA][MS], uint4 (&bq)[NTB][NS], auto&& other) {
    other();
    mma(a, bq);
#pragma unroll
    for (int i = 0; i < NPROD * MS * NS; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    }
  };
  auto ksteps_deep = [&](int buf, int c1, bool more2) {
    fetch(std::integral_constant<int, 0>(), buf, fa[0], fb[0]);
    fetch(std::integral_constant<int, 1>(), buf, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    // (behind the last stage the pieces write zeros into the image nobody reads any more: no branch inside the interleaved region)
    mma_with(fa[0], fb[0], [&] { store_piece(c1, buf ^ 1, std::integral_constant<int, 0>()); });
    __builtin_amdgcn_sched_barrier(0);
    fetch(std::integral_constant<int, 2>(), buf, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mma_with(fa[1], fb[1], [&] { if constexpr (CJ == 2) store_piece(c1, buf ^ 1, std::integral_constant<int, 1>()); });
    __builtin_amdgcn_sched_barrier(0);
    fetch(std::integral_constant<int, 3>(), buf, fa[1], fb[1]);
    if (more2) issue_s_loads(c1 + KC8);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    fetch(std::integral_constant<int, 4>(), buf, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa[0], fb[0]);
  };

  if constexpr (!DEEP) {
    // Happens-before of the single image (conv_gemm_split.hip): store_s(s) and the W DMA of stage s write the image behind the CLOSING
    // barrier of stage s - 1; the OPENING barrier of stage s follows every wave's ds_writes and its vmcnt(0) (its DMA pieces landed).
    for (int s_ = s_lo; s_ < nst; ++s_) {
      store_s(s_ * KC8, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // opening
      if (s_ + 1 < nst) issue_s_loads((s_ + 1) * KC8);   // registers; in flight during the MFMA phase
      SAR_LDS_SKEW();
      ksteps(0);
      __syncthreads();   // closing: the image may be overwritten (next stage / the epilogue's transpose area)
      if (s_ + 1 < nst) {
        load_bnp((s_ + 1) * KC8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_w_dma(s_ + 1, 0);
      }
    }
  } else {
    // Two images.  Stage s multiplies image s & 1.  Behind the barrier of stage s every wave has finished its reads of stage s - 1
    // (image (s + 1) & 1) and written its part of image s & 1: the W DMA of stage s + 1 and store_s(s + 1) -- whose loads were
    // requested a stage ago -- fill image (s + 1) & 1 and the loads of stage s + 2 are requested, all of it INSIDE the matrix phase
    // of stage s (ksteps_deep).  Vector-memory order inside a stage: W DMA (s + 1), loads (s + 2) -- so at the top of stage s + 1 vmcnt(8 CJ)
    // (the loads may stay in flight) says this wave's DMA pieces have landed; the last stage has no loads behind its DMA.
    store_s(s_lo * KC8, 0);
    if (s_lo + 1 < nst) issue_s_loads((s_lo + 1) * KC8);
    for (int s_ = s_lo; s_ < nst; ++s_) {
      const int buf = (s_ - s_lo) & 1;
      if (s_ + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * CJ) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s_ + 1 < nst) {
        load_bnp((s_ + 1) * KC8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue_w_dma(s_ + 1, buf ^ 1);
      }
      SAR_LDS_SKEW();
      ksteps_deep(buf, (s_ + 1) * KC8, s_ + 2 < nst);
    }
    __syncthreads();   // the epilogue's transpose area aliases the images
  }

  {   // undo the operand scales; dead columns hold exact zeros (the epilogue's sums run over them)
    const float c0 = SCALED ? (nonfin ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + ew))) : 1.f;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ms][ns][r] = colok[ns] ? acc[ms][ns][r] * c0 : 0.f;
  }
  if (k.ksplit > 1) {   // raw partial tile -> slab[kpart]; the reduce kernel applies the epilogue
    sar_conv_desc ds = d;
    ds.out = k.slab + (int64_t)kpart * d.M * k.npix;
    ds.ld_out = k.npix;
    ds.epi = SAR_EPI_NONE;
    epilogue_b<MS, NS, WN, BM>(ds, k.nparts, tile, 0, wn, m0, colok, coln, acc, rowp, smem);
    return;
  }
  epilogue_b<MS, NS, WN, BM>(d, k.nparts, tile, 0, wn, m0, colok, coln, acc, rowp, smem);
}

// out[m, n] = epilogue(sum_p slab[p][m][n]) in slab order, partial sums [M][nparts][2] with nparts = chunks of 1024 columns
// (blockIdx.x = chunk, blockIdx.y = row): the second pass of a K-split launch.  One thread = four consecutive columns.
constexpr int RED_COLS = 1024;
__global__ __launch_bounds__(256) void conv2d_split_reduce_kernel(const sar_conv_desc d, const float* __restrict__ slab, int ksplit,
                                                                  int64_t npix, int nparts) {
  const int m = blockIdx.y, chunk = blockIdx.x;
  const int64_t col = (int64_t)chunk * RED_COLS + 4 * threadIdx.x;
  float s1 = 0.f, s2 = 0.f;
  if (col < npix) {
    const float* p = slab + (int64_t)m * npix + col;
    float4 v = *reinterpret_cast<const float4*>(p);
    for (int q = 1; q < ksplit; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)q * d.M * npix);
      v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
    }
    float e[4] = {v.x, v.y, v.z, v.w};
    if (d.epi == SAR_EPI_MASK || d.epi == SAR_EPI_ADD) {
      const float4 a4 = *reinterpret_cast<const float4*>(d.aux + (int64_t)m * d.ld_aux + col);
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
      if (d.epi == SAR_EPI_ADD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] += a[j];
      } else {
        const float sc = d.aux_scale[m], sh = d.aux_shift[m], mu = d.aux_mean ? d.aux_mean[m] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          e[j] = fmaf(a[j], sc, sh) > 0.f ? e[j] : 0.f;
          s1 += e[j];
          s2 = fmaf(e[j], a[j] - mu, s2);
        }
      }
    } else if (d.epi == SAR_EPI_STATS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1 += e[j];
        s2 = fmaf(e[j], e[j], s2);
      }
    }
    *reinterpret_cast<float4*>(d.out + (int64_t)m * d.ld_out + col) = make_float4(e[0], e[1], e[2], e[3]);
  }
  if (d.epi == SAR_EPI_STATS || d.epi == SAR_EPI_MASK) {   // fixed order: lanes by xor tree, then the four waves
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o), s2 += __shfl_xor(s2, o);
    __shared__ float w1[4], w2[4];
    if ((threadIdx.x & 63) == 0) w1[threadIdx.x >> 6] = s1, w2[threadIdx.x >> 6] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
      float* pp = d.partials + ((int64_t)m * nparts + chunk) * 2;
      pp[0] = (w1[0] + w1[1]) + (w1[2] + w1[3]);
      pp[1] = (w2[0] + w2[1]) + (w2[2] + w2[3]);
    }
  }
}

// ---- the 3x3 / stride-2 / pad-1 DATA GRADIENT (models/resnet18.py:5-14 with stride 2: conv1 of the first block of layers 2-4) in the
// split arithmetic.  Output pixel (2 i + py, 2 j + px) receives only the taps with kh = 1 + py (mod 2), kw = 1 + px (mod 2): four
// parity classes, each a stride-1 correlation of the COMPACT dout grid with 1 / 2 / 2 / 4 taps at offsets (dr, dc) in {0, 1}^2
// (kh = 0 reads dout row i + 1, kh = 2 row i, kh = 1 row i; columns alike).  One launch: blockIdx -> (class, tile, row block); a
// workgroup's k-steps = its class's taps, each over 16 channels (lanes 32-63: channels 8-15 of the stage) -- with one or two taps per
// 8-channel stage the stage overhead would dominate.  Tile = 128 compact pixels (4 waves x 32), window = the tile's rows + one row
// below, each row + one column to the right (zero beyond the image), <= 256 units; the weight image is the conv's data-gradient view
// [term][tap = kh 3 + kw][G][M] (no mirroring: the taps are addressed directly).  Epilogues: NONE / ADD (aux on the full grid, or --
// SAR_C2D_AUX_EVEN_PIXELS -- the compact gradient of the parallel 1x1 / stride-2 convolution, added by class (0, 0) only) / MASK
// (partial sums [M][4 classes x 4 ntiles][2]).
struct C2F {
  sar_conv_desc d;
  const uint4* wp;
  int G, B, Hc, Wc, TH, TPI, NI, Wq, IRW, RW, nparts, ntiles, ny, even_aux;
  const unsigned* src_bound;
  const unsigned* w_bound;
};

template <int AR>
__global__ __launch_bounds__(256, AR == AR_H3A ? 3 : 2) void conv2d_split_dgrad_s2_kernel(const C2F k) {
  constexpr int NTA = ar_nta(AR), NTB = ar_ntb(AR), NPROD = ar_nprod(AR);
  constexpr bool SCALED = ar_f16(AR);
  constexpr int BM = 64, MS = 2, NS = 1, WN = 4, KC16 = 16;
  constexpr int ZCOL = 256, SCOLS = ZCOL + 1;
  constexpr int WPIECES = NTA * 4 * 2, WU = WPIECES * 64, SU = NTB * 2 * SCOLS;   // weight pieces [term][tap slot][half][64 rows]; source [term][half][column]
  constexpr int PAREA_U = 4 * 16 * 65 / 4;
  constexpr int IMG_U = (WU + SU) > PAREA_U ? (WU + SU) : PAREA_U;
  constexpr int KCMAX = 512;
  __shared__ uint4 smem_u[IMG_U + BM + KCMAX / 2];
  uint4* Wl = smem_u;
  uint4* Sl = smem_u + WU;
  float* smem = reinterpret_cast<float*>(smem_u);
  float4* rowp = reinterpret_cast<float4*>(smem_u + IMG_U);
  float2* bnp = reinterpret_cast<float2*>(smem_u + IMG_U + BM);
  const sar_conv_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave;
  const int ny = k.ny, nwork = k.ntiles * ny * 4;
  int w = blockIdx.x;
  {
    const int per = (nwork + 7) / 8;
    const int xcd = w & 7, slot = w >> 3;
    w = xcd * per + slot;
    if (w >= nwork || slot >= per) return;
  }
  // the 4-tap class first: its workgroups take twice as long as the others'
  const int cls_i = w / (k.ntiles * ny);
  const int cls = 3 - cls_i, py = cls >> 1, px = cls & 1;
  w -= cls_i * (k.ntiles * ny);
  const int tile = w / ny;
  const int b = k.NI > 1 ? tile * k.NI : tile / k.TPI;
  const int h0 = k.NI > 1 ? 0 : (tile - b * k.TPI) * k.TH;
  const int m0 = (w - tile * ny) * BM;
  const int Hc = k.Hc, Wc = k.Wc, Wq = k.Wq, cpix = Hc * Wc;
  const int ntr = py + 1, ntc = px + 1, ntap = ntr * ntc;   // taps of this class

  bool colok[NS];
  int64_t coln[NS], colna[NS];
  int boff[NS];
  {
    const int p = wn * 32 + l31;
    const int im = k.NI > 1 ? p / cpix : 0;
    const int pp = p - im * cpix;
    int hl = pp / Wc;
    int wo = pp - hl * Wc;
    colok[0] = im < k.NI && (b + im) < k.B && hl < k.TH && (h0 + hl) < Hc;
    if (!colok[0]) { hl = 0; wo = 0; }
    const int bi = b + (colok[0] ? im : 0);
    coln[0] = ((int64_t)bi * (2 * Hc) + 2 * (h0 + hl) + py) * (2 * Wc) + 2 * wo + px;
    colna[0] = k.even_aux ? ((int64_t)bi * Hc + (h0 + hl)) * Wc + wo : coln[0];
    boff[0] = (colok[0] ? im * k.IRW : 0) + hl * Wq + wo;
  }

  int ea = 0, ew = 0;
  bool nonfin = false;   // an operand bound holds Inf / NaN bits: every output of the launch is NaN (split_scale.h)
  if (SCALED) {
    ea = scale_exp(*k.src_bound);
    ew = scale_exp(*k.w_bound);
    nonfin = bound_nonfinite(*k.src_bound) || bound_nonfinite(*k.w_bound);
  }
  const float h3_sa = __builtin_ldexpf(1.f, ea);
  for (int c = tid; c < KCMAX; c += 256) {
    float2 p = make_float2(h3_sa, 0.f);
    if (d.pro_scale && c < d.Kc) p = make_float2(d.pro_scale[c] * h3_sa, d.pro_shift[c] * h3_sa);
    bnp[c] = p;
  }
  if (tid < NTB * 2) Sl[tid * SCOLS + ZCOL] = make_uint4(0u, 0u, 0u, 0u);
  f32x16 acc[MS][NS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ms][0][r] = 0.f;

  const float* src_b = d.src + (int64_t)b * cpix;
  // staging: a lane owns staged element tid of the window (origin: row h0, column 0 of image im; one extra row / column of zeros)
  int svo;
  bool sok;
  {
    const int e = tid;
    const int im = e / k.IRW, ei = e - im * k.IRW;
    const int r = ei / Wq, q = ei - r * Wq;
    const int hs = h0 + r;
    sok = e < k.RW && (b + im) < k.B && hs < Hc && q < Wc;
    svo = sok ? (im * cpix + hs * Wc + q) * 4 : 0;
  }
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  float sreg[KC16], psc[KC16], psh[KC16];
  auto load_bnp = [&](int c0) {   // ahead of the stage's W DMA (an LDS read behind an LDS-DMA makes the compiler wait for the DMA)
#pragma unroll
    for (int q2 = 0; q2 < KC16 / 2; ++q2) {
      const float4 p2 = *reinterpret_cast<const float4*>(&bnp[c0 + 2 * q2]);
      psc[2 * q2] = p2.x, psh[2 * q2] = p2.y, psc[2 * q2 + 1] = p2.z, psh[2 * q2 + 1] = p2.w;
    }
  };
  auto issue_s_loads = [&](int c0) {
#pragma unroll
    for (int q = 0; q < KC16; ++q) {
      const int c = c0 + q;
      const int cg = c < d.Kc ? c : 0;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, k.NI * cpix * 4, 0x00020000);
      sreg[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo, 0, 0));
    }
  };
  auto store_s = [&](int c0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float val = fmaxf(fmaf(sreg[8 * h + q], psc[8 * h + q], psh[8 * h + q]), relu_lo);
        v[q] = (sok && c0 + 8 * h + q < d.Kc) ? val : 0.f;
      }
      uint4 u[NTB];
      split8<AR, false>(v, u, 1.f);
#pragma unroll
      for (int t = 0; t < NTB; ++t) Sl[(t * 2 + h) * SCOLS + tid] = u[t];
    }
  };
  // weight pieces: slot s (0 .. ntap - 1) = tap (kh, kw) = (ir == 0 ? (py ? 0 : 1) : 2, ic == 0 ? (px ? 0 : 1) : 2) with s = ir ntc + ic;
  // piece (term, slot, half) = 64 rows of channel group 2 g16 + half
  const unsigned wbytes = (unsigned)((int64_t)NTA * 9 * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  const unsigned wvo = (m0 + lane) < d.M ? (unsigned)((m0 + lane) * 16) : 0x80000000u;
  auto tap_of = [&](int s) {
    const int ir = s / ntc, ic = s - ir * ntc;
    const int kh = py ? (ir == 0 ? 0 : 2) : 1, kw = px ? (ic == 0 ? 0 : 2) : 1;
    return kh * 3 + kw;
  };
  auto issue_w_dma = [&](int g16) {
    const int np = nta_lds(AR) * ntap * 2;   // pieces of this class (wave-uniform; f16x3a: the third weight image is formed in registers)
    for (int p = wave; p < np; p += 4) {
      const int t = p / (ntap * 2), r = p - t * (ntap * 2), s = r >> 1, h = r & 1;
      const int g = 2 * g16 + h;
      const unsigned so = g < k.G ? (unsigned)(((t * 9 + tap_of(s)) * k.G + g) * d.M * 16) : 0u;
      const unsigned vo = g < k.G ? wvo : 0x80000000u;   // a channel group beyond Kc: zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(Wl + ((t * 4 + s) * 2 + h) * 64), 16, vo, so, 0, 0);
    }
  };

  issue_s_loads(0);
  __syncthreads();
  load_bnp(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  issue_w_dma(0);
  const int nst = (d.Kc + KC16 - 1) / KC16;
  for (int s_ = 0; s_ < nst; ++s_) {
    store_s(s_ * KC16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // opening
    if (s_ + 1 < nst) issue_s_loads((s_ + 1) * KC16);
    for (int s = 0; s < ntap; ++s) {   // tap slot s: kh = 0 reads the row BELOW (dr = 1), kh = 2 / 1 the row itself; columns alike
      const int ir = s / ntc, ic = s - ir * ntc;
      const int dr = (py && ir == 0) ? 1 : 0, dc = (px && ic == 0) ? 1 : 0;
      uint4 a[NTA][MS], bq[NTB];
#pragma unroll
      for (int t = 0; t < nta_lds(AR); ++t)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) a[t][ms] = Wl[((t * 4 + s) * 2 + hi) * 64 + l31 + ms * 32];
      if constexpr (nta_lds(AR) < NTA) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) a[NTA - 1][ms] = third_image(a[0][ms]);
      }
      const int bo = boff[0] + dr * Wq + dc;
#pragma unroll
      for (int t = 0; t < NTB; ++t) bq[t] = Sl[(t * 2 + hi) * SCOLS + bo];
#pragma unroll
      for (int p = 0; p < NPROD; ++p) {
        const int i = ar_pi(AR, p), j = ar_pj(AR, p);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
          if (ar_f16(AR))
            acc[ms][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<f16x8*>(&a[i][ms]), *reinterpret_cast<f16x8*>(&bq[j]), acc[ms][0], 0, 0, 0);
          else
            acc[ms][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a[i][ms]), *reinterpret_cast<bf16x8*>(&bq[j]), acc[ms][0], 0, 0, 0);
        }
      }
    }
    __syncthreads();   // closing
    if (s_ + 1 < nst) {
      load_bnp((s_ + 1) * KC16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue_w_dma(s_ + 1);
    }
  }
  {
    const float c0 = SCALED ? (nonfin ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + ew))) : 1.f;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][0][r] = colok[0] ? acc[ms][0][r] * c0 : 0.f;
  }
  sar_conv_desc de = d;
  if (k.even_aux && cls != 0) de.epi = SAR_EPI_NONE;   // the compact 1x1 gradient lands on the even pixels only
  epilogue_b<MS, NS, WN, BM>(de, k.nparts, cls * k.ntiles + tile, 0, wn, m0, colok, coln, acc, rowp, smem, colna);
}

bool ar_built(int ar) { return ar == AR_B6 || ar == AR_H3A; }

// the launches this unit is built for; fills the geometry for tiles of `tcols` (256 / 128) output pixels
bool geometry_2s(const sar_conv2d_desc& d, C2S& k, int tcols) {
  if (d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1) return false;
  if (d.H_src != d.H_out || d.W_src != d.W_out) return false;
  if (d.Kc < 8 || d.Kc > 512 || (d.M & 7) || d.flags) return false;
  const int H = d.H_out, W = d.W_out, opix = H * W;
  if (W > tcols || opix <= 0) return false;
  k.B = d.B, k.H = H, k.W = W, k.Wq = W + 2;
  if (opix <= tcols / 2) {
    k.NI = tcols / opix, k.TH = H, k.TPI = 1;
    k.IRW = (H + 2) * (W + 2), k.RW = k.NI * k.IRW;
    k.ntiles = (d.B + k.NI - 1) / k.NI;
  } else {
    k.NI = 1, k.TH = tcols / W < H ? tcols / W : H, k.TPI = (H + k.TH - 1) / k.TH;
    k.IRW = k.RW = (k.TH + 2) * (W + 2);
    k.ntiles = d.B * k.TPI;
  }
  if (k.RW > 2 * tcols) return false;
  k.ny = (d.M + 63) / 64;
  k.nparts = k.ntiles * 4;
  k.G = (d.Kc + 7) / 8;
  return true;
}

// Which kernel a descriptor takes: 0 = three workgroups per CU, tiles of 256 pixels (large launches: the other workgroups of the CU
// hide a stage's latencies); 1 = double-buffered, 256; 2 = double-buffered, 128 (small feature maps: twice the workgroups).
// SAR_C2S_VARIANT=0/1/2 pins it (experiments).  -1: not built.
int pick_variant(const sar_conv2d_desc& d, C2S& k, int64_t* want_slab = nullptr) {
  static const int pinned = [] { const char* e = getenv("SAR_C2S_VARIANT"); return e ? atoi(e) : -1; }();
  C2S k256, k128;
  const bool ok256 = geometry_2s(d, k256, 256), ok128 = geometry_2s(d, k128, 128);
  if (!ok256 && !ok128) return -1;
  int v;
  if (pinned >= 0 && pinned <= 2 && (pinned == 2 ? ok128 : ok256)) v = pinned;
  else if (!ok256) v = 2;
  else {
    // measured (tools/c2s_probe.py, bs = 32: variants 0 / 1 / 2 in us): 64 ch 64x64 (512 workgroups of 256 pixels) 44 / 56 / 57;
    // 128 ch 32x32 (256) 55 / 50 / 45; 256 ch 16x16 (128) 96 / 88 / 60; 512 ch 8x8 (64) 185 / 167 / 113 -- fp32 MFMA: 92 / 97 / 107 / 139
    const int nwork = k256.ntiles * k256.ny;
    v = nwork >= 2 * 256 ? 0 : (ok128 ? 2 : 1);
  }
  k = v == 2 ? k128 : k256;
  // K-split for launches that leave most of the chip idle (DEEP kernels only): workgroups up to ~2 per CU, at least 8 stages each;
  // needs the caller's workspace (d.slab: sar_conv2d_gemm_split_slab_bytes) and 16-byte rows.  SAR_C2S_KSPLIT pins it (experiments).
  k.ksplit = 1, k.slab = nullptr, k.npix = (int64_t)d.B * d.H_out * d.W_out;
  if (v != 0 && (k.npix & 3) == 0) {
    static const int pin_ks = [] { const char* e = getenv("SAR_C2S_KSPLIT"); return e ? atoi(e) : 0; }();
    const int nwork = k.ntiles * k.ny, nst = k.G;
    int ks = 1;
    while (ks < 8 && nwork * ks * 2 <= 512 && nst / (ks * 2) >= 8) ks *= 2;
    if (pin_ks > 0) ks = pin_ks;
    if (ks > 1 && ks <= nst && ks <= 8) k.ksplit = ks;   // planned; used when the caller provides the workspace (below)
  }
  if (want_slab) *want_slab = k.ksplit > 1 ? (int64_t)k.ksplit * d.M * k.npix * 4 : 0;
  if (k.ksplit > 1 && d.slab && (d.ld_out & 3) == 0 && (d.ld_aux & 3) == 0 && ((uintptr_t)d.out & 15) == 0 &&
      ((uintptr_t)d.aux & 15) == 0 && ((uintptr_t)d.slab & 15) == 0) {
    k.slab = d.slab;
    k.nparts = (int)((k.npix + RED_COLS - 1) / RED_COLS);
  } else {
    k.ksplit = 1;
  }
  {   // L2 footprint of one XCD (1 / 8 of the workgroups): all weight images + 1 / 8 of the source, or 1 / 8 of the weights + (all / part of) the source
    static const int pin_rb = [] { const char* e = getenv("SAR_C2S_RB_MAJOR"); return e ? atoi(e) : -1; }();
    const double wbytes = 3.0 * 9 * k.G * d.M * 16, sbytes = 4.0 * d.Kc * d.B * d.H_out * d.W_out;
    const double per_tile = wbytes + sbytes / 8, per_rb = wbytes / 8 + sbytes * (k.ny >= 8 ? 1.0 : k.ny / 8.0);
    k.rb_major = pin_rb >= 0 ? pin_rb : (per_rb < per_tile ? 1 : 0);
  }
  return v;
}

// the stride-2 data gradient launch (conv2d_split_dgrad_s2_kernel): compact dout grid Hc x Wc -> output 2 Hc x 2 Wc
bool geometry_2f(const sar_conv2d_desc& d, C2F& k) {
  if (!d.transposed || d.KH != 3 || d.KW != 3 || d.stride != 2 || d.pad != 1) return false;
  if (d.H_out != 2 * d.H_src || d.W_out != 2 * d.W_src) return false;
  if (d.Kc < 16 || d.Kc > 512 || (d.M & 7)) return false;
  if (d.flags & ~SAR_C2D_AUX_EVEN_PIXELS) return false;
  if ((d.flags & SAR_C2D_AUX_EVEN_PIXELS) && d.epi != SAR_EPI_ADD) return false;
  const int Hc = d.H_src, Wc = d.W_src, cpix = Hc * Wc;
  if (Wc > 128 || cpix <= 0) return false;
  k.B = d.B, k.Hc = Hc, k.Wc = Wc, k.Wq = Wc + 1;
  if (cpix <= 64) {
    k.NI = 128 / cpix, k.TH = Hc, k.TPI = 1;
    k.IRW = (Hc + 1) * (Wc + 1), k.RW = k.NI * k.IRW;
    k.ntiles = (d.B + k.NI - 1) / k.NI;
  } else {
    k.NI = 1, k.TH = 128 / Wc < Hc ? 128 / Wc : Hc, k.TPI = (Hc + k.TH - 1) / k.TH;
    k.IRW = k.RW = (k.TH + 1) * (Wc + 1);
    k.ntiles = d.B * k.TPI;
  }
  if (k.RW > 256) return false;
  k.ny = (d.M + 63) / 64;
  k.nparts = 4 * k.ntiles * 4;
  k.G = (d.Kc + 7) / 8;
  k.even_aux = (d.flags & SAR_C2D_AUX_EVEN_PIXELS) ? 1 : 0;
  return true;
}

void fill_desc(const sar_conv2d_desc& d, sar_conv_desc& o) {
  o = sar_conv_desc{};
  o.mode = SAR_CONV_TEMPORAL;
  o.B = d.B, o.Kc = d.Kc, o.M = d.M;
  o.pro_relu = d.pro_relu, o.epi = d.epi;
  o.src = d.src, o.ld_src = d.ld_src;
  o.out = d.out, o.ld_out = d.ld_out;
  o.pro_scale = d.pro_scale, o.pro_shift = d.pro_shift;
  o.aux = d.aux, o.ld_aux = d.ld_aux, o.aux_scale = d.aux_scale, o.aux_shift = d.aux_shift, o.aux_mean = d.aux_mean;
  o.partials = d.partials;
}

}  // namespace

extern "C" int64_t sar_conv2d_gemm_split_workspace_bytes(const sar_conv2d_desc* d, int arith) {
  if (!d || d->Kc <= 0 || d->M <= 0 || d->KH <= 0 || d->KW <= 0 || !ar_built(arith)) return SAR_E_ARG;
  return (int64_t)ar_nta(arith) * d->KH * d->KW * ((d->Kc + 7) / 8) * d->M * 16;
}

extern "C" int sar_conv2d_gemm_split_nparts(const sar_conv2d_desc* d) {
  if (!d || d->B <= 0 || d->M <= 0) return SAR_E_ARG;
  C2F kf;
  if (geometry_2f(*d, kf)) return kf.nparts;
  C2S k;
  if (pick_variant(*d, k) < 0) return SAR_E_UNSUP;
  return k.nparts;
}

extern "C" int sar_conv2d_gemm_split(const sar_conv2d_desc* d, int arith, const void* packed, const uint32_t* src_bound,
                                     const uint32_t* w_bound, sar_stream_t s) {
  SAR_REQUIRE(d != nullptr && packed != nullptr, "sar_conv2d_gemm_split: null descriptor / weight image");
  SAR_REQUIRE(((uintptr_t)packed & 15) == 0, "sar_conv2d_gemm_split: the weight image must be 16-byte aligned");
  SAR_REQUIRE(ar_built(arith), "sar_conv2d_gemm_split: built for bf16x6 / f16x3a (arith %d)", arith);
  SAR_REQUIRE(!ar_f16(arith) || (src_bound && w_bound), "sar_conv2d_gemm_split: fp16 arithmetics need the operand bounds");
  SAR_REQUIRE(d->B > 0 && d->Kc > 0 && d->M > 0 && d->H_src > 0 && d->W_src > 0, "sar_conv2d_gemm_split: bad sizes");
  C2F kf;
  if (geometry_2f(*d, kf)) {   // the 3x3 / stride-2 data gradient: four parity classes in one launch
    const int64_t nout = (int64_t)d->B * d->H_out * d->W_out, nsrc = (int64_t)d->B * d->H_src * d->W_src;
    SAR_REQUIRE(d->src && d->out && d->ld_src >= nsrc && d->ld_out >= nout, "sar_conv2d_gemm_split: null src/out or leading dimension too small");
    SAR_REQUIRE(d->ld_out < (1 << 22) && d->ld_aux < (1 << 22), "sar_conv2d_gemm_split: leading dimension too large (2^22 columns)");
    SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv2d_gemm_split: pro_scale/pro_shift mismatch");
    SAR_REQUIRE(d->epi >= SAR_EPI_NONE && d->epi <= SAR_EPI_ADD, "sar_conv2d_gemm_split: bad epilogue %d", d->epi);
    if (d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->partials && d->aux_scale && d->aux_shift, "sar_conv2d_gemm_split: partials / aux affine required");
    if (d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD)
      SAR_REQUIRE(d->aux && d->ld_aux >= (kf.even_aux ? nsrc : nout), "sar_conv2d_gemm_split: aux required");
    fill_desc(*d, kf.d);
    kf.wp = (const uint4*)packed;
    kf.src_bound = src_bound;
    kf.w_bound = w_bound;
    const dim3 grid(((kf.ntiles * kf.ny * 4 + 7) / 8) * 8), block(256);
    if (arith == AR_H3A) hipLaunchKernelGGL((conv2d_split_dgrad_s2_kernel<AR_H3A>), grid, block, 0, as_stream(s), kf);
    else hipLaunchKernelGGL((conv2d_split_dgrad_s2_kernel<AR_B6>), grid, block, 0, as_stream(s), kf);
    SAR_LAUNCH_CHECK("sar_conv2d_gemm_split");
    return 0;
  }
  C2S k;
  const int variant = pick_variant(*d, k);
  if (variant < 0) {
    sar_set_error("sar_conv2d_gemm_split: built for 3x3 / stride 1 / pad 1, 8 <= Kc <= 512, M %% 8 == 0, windows of <= 512 staged pixels "
                  "(%dx%d, stride %d, pad %d, Kc %d, M %d, %dx%d): use sar_conv2d_gemm_f32",
                  d->KH, d->KW, d->stride, d->pad, d->Kc, d->M, d->H_out, d->W_out);
    return SAR_E_UNSUP;
  }
  const int64_t npix = (int64_t)d->B * d->H_out * d->W_out;
  SAR_REQUIRE(d->src && d->out, "sar_conv2d_gemm_split: null src/out");
  SAR_REQUIRE(d->ld_src >= npix && d->ld_out >= npix, "sar_conv2d_gemm_split: leading dimension smaller than B*H*W");
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv2d_gemm_split: pro_scale/pro_shift mismatch");
  SAR_REQUIRE(d->ld_out < (1 << 22) && d->ld_aux < (1 << 22), "sar_conv2d_gemm_split: leading dimension too large (2^22 columns)");
  SAR_REQUIRE(sar_conv2d_gemm_split_workspace_bytes(d, arith) < (1ll << 31), "sar_conv2d_gemm_split: weight tensor too large");
  SAR_REQUIRE(d->epi >= SAR_EPI_NONE && d->epi <= SAR_EPI_ADD, "sar_conv2d_gemm_split: bad epilogue %d", d->epi);
  if (d->epi == SAR_EPI_STATS || d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->partials, "sar_conv2d_gemm_split: partials required");
  if (d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD) SAR_REQUIRE(d->aux && d->ld_aux >= npix, "sar_conv2d_gemm_split: aux required");
  if (d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->aux_scale && d->aux_shift, "sar_conv2d_gemm_split: aux affine required");
  fill_desc(*d, k.d);
  k.wp = (const uint4*)packed;
  k.src_bound = src_bound;
  k.w_bound = w_bound;
  const dim3 grid(((k.ntiles * k.ny * k.ksplit + 7) / 8) * 8), block(256);
  hipStream_t st = as_stream(s);
  if (arith == AR_H3A) {
    if (variant == 0) hipLaunchKernelGGL((conv2d_split_kernel<AR_H3A, 2, 0>), grid, block, 0, st, k);
    else if (variant == 1) hipLaunchKernelGGL((conv2d_split_kernel<AR_H3A, 2, 1>), grid, block, 0, st, k);
    else hipLaunchKernelGGL((conv2d_split_kernel<AR_H3A, 1, 1>), grid, block, 0, st, k);
  } else {
    if (variant == 0) hipLaunchKernelGGL((conv2d_split_kernel<AR_B6, 2, 0>), grid, block, 0, st, k);
    else if (variant == 1) hipLaunchKernelGGL((conv2d_split_kernel<AR_B6, 2, 1>), grid, block, 0, st, k);
    else hipLaunchKernelGGL((conv2d_split_kernel<AR_B6, 1, 1>), grid, block, 0, st, k);
  }
  if (k.ksplit > 1)
    hipLaunchKernelGGL(conv2d_split_reduce_kernel, dim3(k.nparts, d->M), dim3(256), 0, st, k.d, (const float*)k.slab, k.ksplit, k.npix, k.nparts);
  SAR_LAUNCH_CHECK("sar_conv2d_gemm_split");
  return 0;
}

extern "C" int64_t sar_conv2d_gemm_split_slab_bytes(const sar_conv2d_desc* d) {
  if (!d || d->B <= 0 || d->M <= 0 || d->H_out <= 0 || d->W_out <= 0) return SAR_E_ARG;
  C2F kf;
  if (geometry_2f(*d, kf)) return 0;
  C2S k;
  int64_t want = 0;
  if (pick_variant(*d, k, &want) < 0) return SAR_E_UNSUP;
  return want;
}

// conv_gemm_split.hip -- the 9-tap temporal convolution and its data gradients with fp32 results on the bf16 matrix pipe
// ("split" arithmetic, gfx950).
//
//   out[m, n] = sum_tap sum_c W[tap][c][m] * pro(src)[c, n + shift(tap)] (+ bias) ; epilogue          (models/stgcn.py:29-36)
//
// fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate on this chip.  Here every fp32 operand is written as a sum
// of bfloat16 TERMS,  a = a0 + a1 + a2  with  a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)  (round to nearest; the
// remainders are exact in fp32, 8 + 8 + 8 significant bits: |a - a0 - a1 - a2| <= 2^-27 |a|), and the product a b is the sum of
// the cross products a_i b_j with i + j <= 2 -- each EXACT in the fp32 accumulator of v_mfma_f32_32x32x16_bf16; the dropped
// products are <= 3 * 2^-27 |a b| -- issued smallest terms first into ONE fp32 accumulator ("x6").  Activations, weights,
// BatchNorm sums, bias and epilogue stay fp32: the kernel is a drop-in for conv_gemm_kernel<TEMPORAL, 9> (conv_gemm.hip) under
// the same parity tolerances.
//
// The product arithmetic is the fp16 form of the same idea, "f16x3s": a = h0 + h1 with h0 = fp16(s a), h1 = fp16(s a - h0) (11 + 11
// significant bits), products h0 g0 + h0 g1 + h1 g0 (the dropped h1 g1 <= 2^-24 |a b|) into one fp32 accumulator: HALF the matrix
// work of x6 and 4 instead of 6 bytes of LDS per operand element; measured error 0.6-0.9x the fp32 MFMA kernel's, 2.6-3.0x its
// speed (profiles/r05_split_probe_temporal_*.txt).  fp16 has 5 exponent bits, so each operand tensor is scaled by a power of two
// s = 2^e taken from an UPPER BOUND of its largest magnitude that the caller provides in device memory (no host sync):
// e = 14 - floor(log2(bound)) puts the bound in [2^14, 2^15).  Elements within 2^-18 of the bound keep 22 bits; smaller ones are
// represented to 2^-40 of the bound (fp16 subnormals) -- norm-wise below the fp32 accumulation error of any sum they enter.  The
// bound of a weight tensor is its amax (sar_pack_weights_split_batch), of a BatchNorm-ed source the Samuelson bound
// |gamma| sqrt(n - 1) + |beta| (sar_bn_bound_f32: no pass over the data) or scale * amax + shift (sar_affine_bound_f32), of a
// gradient tensor its amax (sar_amax_f32).  Values are clamped to +-65504 behind the scale: a stale bound saturates, it does not
// produce infinities.  The other arithmetics stay as measured data points: x1 (one bf16 term), x3, x9, and f16x3 with the second
// term scaled by 2^11 into its own accumulator (error 0.5x fp32, 128 accumulator registers).
//
// Design (MI355X):
//  * tile 64 (m) x 256 (columns = 10 frames x 25 joints), 4 waves side by side, wave tile 64 x 64 (2 x 2 MFMA blocks).
//  * stage = 8 source channels.  The 16 k of one MFMA are 8 channels x 2 TAPS: lanes 0-31 hold tap 2q, lanes 32-63 tap 2q + 1
//    (a tap is a shift of the column by V units, so the half's tap is part of the lane's LDS address).  Nine taps = five k-steps,
//    the last one half empty (its second half reads a zero weight slot): 10 % idle matrix work buys an operand image of 53 KB
//    -- three workgroups per CU -- instead of 99 KB with 16 channels per stage.
//  * LDS image: NT weight terms [term][tap][64 rows] + one zero slot, NT source terms [term][column] + one zero column; units of
//    16 bytes = 8 channels of one row / column, the k-innermost operand of the bf16 MFMA: every fragment is one ds_read_b128.
//  * weights are split ONCE per step into their term images by the pack kernel; a stage's weight pieces reach LDS by LDS-DMA
//    (buffer_load_dwordx4 ... lds: no registers, no VALU).  The source is split in the stager behind the folded BatchNorm +
//    ReLU: a lane owns a column, loads it from 8 channel rows (coalesced row segments), and writes NT ds_write_b128.
//  * single image; the W DMA of stage s + 1 is issued behind the closing barrier of stage s, the source loads of stage s + 1 are
//    in flight (registers) during the MFMA phase of stage s.  Happens-before chain: see the main loop.
#include "sar_common.h"
#include <type_traits>

namespace {

#include "split_terms.h"

#include "conv_epi_f32.h"

// Diagnostic build -DSAR_SPLIT_TL (tools/split_timeline.sh): wave 0 of every workgroup of the LAST launch writes one row -- start /
// end in 100 MHz ticks (s_memrealtime), HW_ID, XCC_ID and the shader-clock cycles it spent in each phase (summed over the stages) --
// DESIGN 3.9h's instrument for the split kernels.  No stamp executes in the product build.
#ifdef SAR_SPLIT_TL
constexpr int SPLIT_TL_WG = 16384;
__device__ unsigned g_split_tl[SPLIT_TL_WG][16];
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
    if (threadIdx.x == 0 && blockIdx.x < SPLIT_TL_WG) {                                     \
      unsigned* row = g_split_tl[blockIdx.x];                                               \
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

constexpr int KC8 = 8;    // source channels per stage
constexpr int VJ = 25;   // joints per frame: compile-time (tap shifts are immediates); other V stay on the fp32 kernel

// cell = max(cell, |x|) over a [C][n] matrix (row stride ld): bits of non-negative floats order like unsigned integers, and a
// maximum does not depend on the order of its operands: deterministic with atomics.  grid (chunks, C).
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int64_t n, int64_t ld, unsigned* __restrict__ cell) {
  const float* row = x + (int64_t)blockIdx.y * ld;
  unsigned m = 0;
  const bool vec = ((n | ld) & 3) == 0 && (((uintptr_t)x) & 15) == 0;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const uint4 v = reinterpret_cast<const uint4*>(row)[i];
      const unsigned a = (v.x & 0x7fffffffu) > (v.y & 0x7fffffffu) ? (v.x & 0x7fffffffu) : (v.y & 0x7fffffffu);
      const unsigned b = (v.z & 0x7fffffffu) > (v.w & 0x7fffffffu) ? (v.z & 0x7fffffffu) : (v.w & 0x7fffffffu);
      const unsigned c = a > b ? a : b;
      m = c > m ? c : m;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      const unsigned c = __float_as_uint(row[i]) & 0x7fffffffu;
      m = c > m ? c : m;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)m, o);
    m = t > m ? t : m;
  }
  __shared__ unsigned wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned a = wm[0] > wm[1] ? wm[0] : wm[1], b = wm[2] > wm[3] ? wm[2] : wm[3];
    a = a > b ? a : b;
    if (a > __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(cell, a);
  }
}

// mode 0 (Samuelson): cell = max_c |gamma_c| sqrt(count - 1) + |beta_c|  -- an upper bound of |gamma (x - mean) rstd + beta| for ANY
//   data whose mean / biased variance over `count` samples are the ones rstd was formed from (train-mode BatchNorm)
// mode 1 (affine): cell = max_c |scale_c| * bound(src) + max_c |shift_c|   (eval-mode BatchNorm, any folded affine)
// one workgroup; a 2^-10 margin covers the fp32 roundings of the folded evaluation
__global__ __launch_bounds__(256) void bound_kernel(const float* __restrict__ a, const float* __restrict__ b, int C, float root,
                                                    const unsigned* __restrict__ src_cell, int mode, unsigned* __restrict__ cell) {
  // maxima on the BITS of the (non-negative) values: unsigned order keeps a NaN / Inf in gamma, beta or the folded affine (fmaxf drops NaN)
  unsigned m0 = 0u, m1 = 0u;
  auto ubits = [](float v) { return __float_as_uint(v) & 0x7fffffffu; };
  auto umax = [](unsigned a, unsigned b) { return a > b ? a : b; };
  for (int c = threadIdx.x; c < C; c += 256) {
    const float av = fabsf(a[c]), bv = fabsf(b[c]);
    if (mode == 0) m0 = umax(m0, ubits(fmaf(av, root, bv)));
    else m0 = umax(m0, ubits(av)), m1 = umax(m1, ubits(bv));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m0 = umax(m0, (unsigned)__shfl_xor((int)m0, o)), m1 = umax(m1, (unsigned)__shfl_xor((int)m1, o));
  __shared__ unsigned w0[4], w1[4];
  if ((threadIdx.x & 63) == 0) w0[threadIdx.x >> 6] = m0, w1[threadIdx.x >> 6] = m1;
  __syncthreads();
  if (threadIdx.x == 0) {
    m0 = umax(umax(w0[0], w0[1]), umax(w0[2], w0[3]));
    m1 = umax(umax(w1[0], w1[1]), umax(w1[2], w1[3]));
    float v = mode == 0 ? __uint_as_float(m0) : fmaf(__uint_as_float(m0), __uint_as_float(*src_cell), __uint_as_float(m1));
    v *= 1.0009765625f;
    atomicMax(cell, __float_as_uint(v));
  }
}

// per-item amax of the weight tensors of a pack batch (blockIdx.y = item).  An item whose strides enumerate a DENSE block of
// taps Kc M floats (any permutation / mirroring of a packed tensor: the forward and the data-gradient view of the same weights) is
// scanned linearly -- the index arithmetic of the general path cost 244 us per step for the resnet's 22 M item elements.
__global__ __launch_bounds__(256) void pack_amax_kernel(const float* __restrict__ base, const sar_pack_item* __restrict__ items,
                                                        unsigned* __restrict__ item_amax) {
  const sar_pack_item it = items[blockIdx.y];
  const int64_t n = (int64_t)it.taps * it.Kc * it.M;
  const int64_t ast = it.st < 0 ? -it.st : it.st, asc = it.sc < 0 ? -it.sc : it.sc, asm_ = it.sm < 0 ? -it.sm : it.sm;
  const int64_t span = (it.taps - 1) * ast + (it.Kc - 1) * asc + (it.M - 1) * asm_ + 1;
  unsigned m = 0;
  if (span == n) {
    const int64_t lo = it.src_off + (it.st < 0 ? (it.taps - 1) * it.st : 0) + (it.sc < 0 ? (it.Kc - 1) * it.sc : 0) +
                       (it.sm < 0 ? (it.M - 1) * it.sm : 0);
    const float* p = base + lo;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < n; u += (int64_t)gridDim.x * 256) {
      const unsigned v = __float_as_uint(p[u]) & 0x7fffffffu;
      m = v > m ? v : m;
    }
  } else {
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < n; u += (int64_t)gridDim.x * 256) {
      const int mm = (int)(u % it.M);
      const int c = (int)((u / it.M) % it.Kc);
      const int tp = (int)(u / ((int64_t)it.M * it.Kc));
      const unsigned v = __float_as_uint(base[it.src_off + tp * it.st + c * it.sc + mm * it.sm]) & 0x7fffffffu;
      m = v > m ? v : m;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(item_amax + blockIdx.y, m);
}

// fp32 weights (element (tap, c, m) at src_off + tap*st + c*sc + m*sm) -> term images [term][tap][g][m], 16-byte units of 8
// channels, zero beyond Kc.  blockIdx.y = item (sar_pack_item; G = ceil(Kc / 8)).
template <int AR>
__global__ void pack_split_kernel(const float* __restrict__ base, const sar_pack_item* __restrict__ items,
                                  const unsigned* __restrict__ item_amax, uint4* __restrict__ out) {
  constexpr int NT = ar_nta(AR);
  const sar_pack_item it = items[blockIdx.y];
  const float wscale = ar_f16(AR) ? __builtin_ldexpf(1.f, scale_exp(item_amax[blockIdx.y])) : 1.f;
  const int64_t n = (int64_t)it.taps * it.G * it.M;
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const int m = (int)(u % it.M);
  const int g = (int)((u / it.M) % it.G);
  const int tp = (int)(u / ((int64_t)it.M * it.G));
  const float* W = base + it.src_off + tp * it.st + m * it.sm;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 8 * g + j;
    v[j] = c < it.Kc ? W[c * it.sc] : 0.f;
  }
  uint4 t[NT];
  split8<AR, true>(v, t, wscale);
#pragma unroll
  for (int i = 0; i < NT; ++i) out[it.dst_unit + i * n + u] = t[i];
}

struct ConvKS {
  sar_conv_desc d;
  const uint4* wp;   // term images [term][tap][G][M]
  int G;             // channel groups of 8
  int FT, TPS, RW, nparts, ntiles, ny;
  const unsigned* src_bound;   // fp16 arithmetics: bits of an upper bound of |pro(src)| / of |W| (device memory)
  const unsigned* w_bound;
  int ablate;        // -DSAR_G2_ABLATE builds only (tools/g2_ablate.sh): phases of conv_graph_split2_kernel switched off by SAR_G2_ABLATE_BITS
};
// Ablation build of the persistent graph kernel: bit 0 no raw DMA after the first two stages, 1 no regular conversion, 2 no virtual-joint
// conversion, 3 no MFMA, 4 no weight DMA after the first, 5 no epilogue.  Results are then WRONG; only the launch time is read.
#ifdef SAR_G2_ABLATE
#define G2_ON(bit) (!(k.ablate & (1 << (bit))))
#else
#define G2_ON(bit) true
#endif

// TR: 0 forward; 1 data gradient, stride 1; 3 data gradient, stride 2, parity-split column map (conv_gemm.hip).  WIDE: the
// staged window of a stride-2 forward tile (27 frames) instead of 18.
template <int TR, int AR, int WIDE>
__global__ __launch_bounds__(256, (WIDE || ar_two_acc(AR)) ? 2 : 3) void conv_gemm_split_kernel(const ConvKS k) {
  constexpr int NTA = ar_nta(AR), NTB = ar_ntb(AR), NPROD = ar_nprod(AR), NACC = ar_two_acc(AR) ? 2 : 1;
  constexpr bool SCALED = ar_f16(AR);   // the accumulators carry the operand scales: bias joins at the end
  constexpr int TAPS = 9, BM = 64, MS = 2, NS = 2, WN = 4, V = VJ;
  constexpr int PAR = (TR == 3);
  constexpr int ZCOL = WIDE ? 27 * V : 18 * V, SCOLS = ZCOL + 1;
  constexpr int CJ = (ZCOL + 255) / 256;
  constexpr int WPIECES = NTA * TAPS, ZSLOT = WPIECES * 64;   // weight pieces of 64 units, then the zero slot
  // f16x3a: the third image of the weights (w0 2^-11) is neither moved nor read -- every fragment of it is four v_pk_mul_f16 on the
  // fragment of w0 (an exact power-of-two scaling with the fp16 rounding the pack kernel applies): a third of the weight DMA pieces
  // and a fifth of the k-steps' LDS reads less (round 6; bit-identical results)
  constexpr int NTA_LDS = nta_lds(AR), WPIECES_DMA = NTA_LDS * TAPS;
  constexpr int WU = ZSLOT + 64, SU = NTB * SCOLS;
  constexpr int PAREA_U = 4 * 16 * 65 / 4;   // the epilogue's transpose area aliases the image
  constexpr int IMG_U = (WU + SU) > PAREA_U ? (WU + SU) : PAREA_U;
  constexpr int PPW = (WPIECES + 3) / 4;      // weight pieces per wave
  constexpr int KCMAX = 256;
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
  // workgroup -> (tile, row block), XCD-aware (conv_gemm.hip): the row blocks of a tile sit on one XCD
  const int ny = k.ny, nwork = k.ntiles * ny;
  int w = blockIdx.x;
  {
    const int per = (nwork + 7) / 8;
    const int xcd = w & 7, slot = w >> 3;
    w = xcd * per + slot;
    if (w >= nwork || slot >= per) return;
  }
  SPLIT_TL_BEGIN();
  const int tile = w / ny;
  const int b = tile / k.TPS;
  const int t0 = (tile - b * k.TPS) * k.FT;
  const int m0 = (w - tile * ny) * BM;

  // ---- per-lane column geometry.  Tap index i of this wave (TR 0 / 1: the tap; TR 3: the i-th tap of the wave's parity)
  // reads column  off0 + i * ostep;  k-step q multiplies tap indices 2q (lanes 0-31) and 2q + 1 (lanes 32-63).
  bool colok[NS];
  int64_t coln[NS];
  int boff[NS], bstep[NS];
  int t_lo;
  if (TR == 0) t_lo = t0 * d.stride - d.pad;
  else if (TR == 1) t_lo = t0 + d.pad - (TAPS - 1);
  else t_lo = floordiv(t0 + d.pad - (TAPS - 1), 2);
  constexpr int HALFC = 128;
  const int par = PAR ? (wn >= 2 ? 1 : 0) : 0;
  const int tp0 = PAR ? ((par + d.pad) & 1) : 0;
  const int ntap_w = PAR ? (TAPS - tp0 + 1) / 2 : TAPS;
  const int nq = (ntap_w + 1) >> 1;          // k-steps of this wave (wave-uniform)
  const bool last_half = (ntap_w & 1) != 0;  // the last k-step's second half is empty
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
    int off0, ostep;
    if (TR == 0) {
      off0 = fo * d.stride * V + v;
      ostep = V;
    } else if (TR == 1) {
      off0 = (fo + TAPS - 1) * V + v;
      ostep = -V;
    } else {
      off0 = (((t0 + fo + d.pad - tp0) >> 1) - t_lo) * V + v;
      ostep = -V;
    }
    boff[ns] = colok[ns] ? off0 + hi * ostep : ZCOL;   // a dead column multiplies the zero column in every k-step
    bstep[ns] = colok[ns] ? 2 * ostep : 0;
  }
  // weight slot of tap index i: TR 0 / 1 tap i; TR 3 tap tp0 + 2 i
  const int abase = (PAR ? (tp0 + 2 * hi) : hi) * 64 + l31;
  constexpr int ASTEP = (PAR ? 4 : 2) * 64;

  if (tid < BM) {
    const int row = m0 + tid;
    float4 bp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias && row < d.M) bp.x = d.bias[row];
    rowp[tid] = bp;
  }
  int ea = 0, ew = 0;
  bool nonfin = false;   // an operand bound holds Inf / NaN bits: every output of the launch is NaN (split_scale.h)   // fp16 arithmetics: operand scale exponents (wave-uniform)
  if (SCALED) {
    ea = scale_exp(*k.src_bound);
    ew = scale_exp(*k.w_bound);
    nonfin = bound_nonfinite(*k.src_bound) || bound_nonfinite(*k.w_bound);
  }
  const float h3_sa = __builtin_ldexpf(1.f, ea);
  if (tid < KCMAX) {   // the folded prologue, with the source scale folded in (a power of two: exact)
    float2 p = make_float2(h3_sa, 0.f);
    if (d.pro_scale && tid < d.Kc) p = make_float2(d.pro_scale[tid] * h3_sa, d.pro_shift[tid] * h3_sa);
    bnp[tid] = p;
  }
  if (tid < NTB) Sl[tid * SCOLS + ZCOL] = make_uint4(0u, 0u, 0u, 0u);
  if (tid >= 64 && tid < 128) Wl[ZSLOT + tid - 64] = make_uint4(0u, 0u, 0u, 0u);
  f32x16 acc[NACC][MS][NS];

  const int seq_len = d.T_src * V;
  const float* src_b = d.src + (int64_t)b * seq_len;

  // ---- source staging: per-lane offsets and masks once
  int svo[CJ];
  bool sok[CJ];
#pragma unroll
  for (int j = 0; j < CJ; ++j) {
    const int col = tid + 256 * j;
    const int rabs = t_lo * V + col;
    sok[j] = col < k.RW && (unsigned)rabs < (unsigned)seq_len;
    svo[j] = sok[j] ? rabs * 4 : 0;
  }
  const float relu_lo = d.pro_relu ? 0.f : -__builtin_inff();
  float sreg[CJ][8];
  auto issue_s_loads = [&](int c0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = c0 + q;
      const int cg = c < d.Kc ? c : 0;   // wave-uniform
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, seq_len * 4, 0x00020000);
#pragma unroll
      for (int j = 0; j < CJ; ++j) sreg[j][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo[j], 0, 0));
    }
  };
  float psc[8], psh[8];
  auto load_bnp = [&](int c0) {   // the folded BatchNorm of the stage's 8 channels (broadcast reads), AHEAD of the stage's W DMA:
#pragma unroll                    // an LDS read behind an LDS-DMA makes the compiler wait for the DMA
    for (int q2 = 0; q2 < 4; ++q2) {
      const float4 p2 = *reinterpret_cast<const float4*>(&bnp[c0 + 2 * q2]);
      psc[2 * q2] = p2.x, psh[2 * q2] = p2.y, psc[2 * q2 + 1] = p2.z, psh[2 * q2 + 1] = p2.w;
    }
  };
  auto store_s = [&](int c0) {
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float val = fmaxf(fmaf(sreg[j][q], psc[q], psh[q]), relu_lo);
        v[q] = (sok[j] && c0 + q < d.Kc) ? val : 0.f;   // TF-SAME padding stays exactly 0 behind the folded BatchNorm
      }
      uint4 u[NTB];
      split8<AR, false>(v, u, 1.f);
      if ((j + 1) * 256 <= ZCOL || tid + 256 * j < ZCOL) {
#pragma unroll
        for (int t = 0; t < NTB; ++t) Sl[t * SCOLS + tid + 256 * j] = u[t];
      }
    }
  };
  // ---- weight pieces by LDS-DMA: piece p = term * 9 + tap = 64 rows of one (term, tap) of channel group g
  const unsigned wbytes = (unsigned)((int64_t)NTA * TAPS * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  const unsigned wvo = (m0 + lane) < d.M ? (unsigned)((m0 + lane) * 16) : 0x80000000u;   // rows beyond M: rejected -> 0
  auto issue_w_dma = [&](int g) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + 4 * i;   // wave-uniform
      if (p < WPIECES_DMA)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(Wl + p * 64), 16, wvo, (p * k.G + g) * d.M * 16, 0, 0);
    }
  };

  issue_s_loads(0);
  __syncthreads();   // rowp, bnp, zero column / slot
  load_bnp(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  issue_w_dma(0);
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 bp = rowp[ms * 32 + mfma_row(r, hi)];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        if (!SCALED) acc[0][ms][ns][r] = colok[ns] ? bp.x : 0.f;
        else acc[0][ms][ns][r] = 0.f, acc[NACC - 1][ms][ns][r] = 0.f;
      }
    }

  auto kstep = [&](int q, bool last) {
    uint4 a[NTA][MS], bq[NTB][NS];
    int ao = abase + q * ASTEP;
    const bool dead = last && last_half && hi;   // this lane's half of the k-step has no tap
#pragma unroll
    for (int t = 0; t < NTA_LDS; ++t)
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a[t][ms] = Wl[dead ? ZSLOT + l31 + ms * 32 : t * (TAPS * 64) + ao + ms * 32];
    if constexpr (NTA_LDS < NTA) {   // f16x3a: the third weight image is w0 2^-11 -- formed here, exactly as the pack kernel rounds it
#pragma unroll
      for (int ms = 0; ms < MS; ++ms) a[NTA - 1][ms] = third_image(a[0][ms]);
    }
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const int bo = dead ? ZCOL : boff[ns] + q * bstep[ns];
#pragma unroll
      for (int t = 0; t < NTB; ++t) bq[t][ns] = Sl[t * SCOLS + bo];
    }
#pragma unroll
    for (int p = 0; p < NPROD; ++p) {
      const int i = ar_pi(AR, p), j = ar_pj(AR, p);
      const int ai = (NACC == 2 && i + j > 0) ? 1 : 0;   // h3: the cross terms carry 2^11 and have their own accumulator
#pragma unroll
      for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          if (ar_f16(AR))
            acc[ai][ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<f16x8*>(&a[i][ms]),
                                                                     *reinterpret_cast<f16x8*>(&bq[j][ns]), acc[ai][ms][ns], 0, 0, 0);
          else
            acc[ai][ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a[i][ms]),
                                                                      *reinterpret_cast<bf16x8*>(&bq[j][ns]), acc[ai][ms][ns], 0, 0, 0);
        }
    }
  };

  // Happens-before of the single image.  store_s(s) and the W DMA of stage s write the image behind the CLOSING barrier of
  // stage s - 1 (every wave has read its last fragment of stage s - 1); the OPENING barrier of stage s follows every wave's
  // ds_writes and its vmcnt(0) (its own DMA pieces have landed): behind it the whole image of stage s is in LDS.
  const int nst = (d.Kc + KC8 - 1) / KC8;
  SPLIT_TL(0);   // prologue
  for (int s_ = 0; s_ < nst; ++s_) {
    store_s(s_ * KC8);
    SPLIT_TL(1);   // wait for the stage's loads, folded BN + ReLU, split, LDS stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SPLIT_TL(2);   // wait for the W DMA
    __syncthreads();   // opening
    SPLIT_TL(3);
    if (s_ + 1 < nst) issue_s_loads((s_ + 1) * KC8);   // registers; in flight during the MFMA phase
    SPLIT_TL(8);   // load issue
    SAR_LDS_SKEW();
    if (PAR) {
#pragma unroll
      for (int q = 0; q < (TAPS + 1) / 4 + 1; ++q)
        if (q < nq) kstep(q, q == nq - 1);
    } else {
#pragma unroll
      for (int q = 0; q < (TAPS + 1) / 2; ++q) kstep(q, q == (TAPS + 1) / 2 - 1);
    }
    SPLIT_TL(4);   // load issue + MFMA phase
    __syncthreads();   // closing: the image may be overwritten (next stage / the epilogue's transpose area)
    if (s_ + 1 < nst) {
      load_bnp((s_ + 1) * KC8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue_w_dma(s_ + 1);
    }
    SPLIT_TL(5);   // closing barrier + DMA issue
  }

  if (SCALED) {   // fp16 terms: undo the operand scales, join the cross terms, add the bias
    const float c0 = (nonfin ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + ew))), c1 = NACC == 2 ? c0 / H3_LO : 0.f;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 bp = rowp[ms * 32 + mfma_row(r, hi)];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          const float v = fmaf(acc[NACC - 1][ms][ns][r], c1, acc[0][ms][ns][r] * c0) + bp.x;
          acc[0][ms][ns][r] = colok[ns] ? v : 0.f;
        }
      }
    __syncthreads();   // every wave has read its bias rows: the MASK epilogue rewrites rowp
  }
  epilogue_b<MS, NS, WN, BM>(d, k.nparts, tile, 0, wn, m0, colok, coln, acc[0], rowp, smem);
  SPLIT_TL(6);   // epilogue
  SPLIT_TL_END(w);
}

// ---- GraphConvTD (models/gcn.py:199-209) and its data gradient in the split arithmetics:
//   out[m, (t,w)] = sum_k sum_c W_k[c][m] z_k[c, (t,w)] + sum_k b_k[m] colsum(A_k)[w],   z_k[c, (t,w)] = sum_v x[c, (t,v)] A_k[v, w]
// The adjacency is applied when the operand is READ (the bf16 engine's conv_graph_cn8.hip, round 4): of the 3 V gather lists of the
// NTU graph most are {one entry of weight 1} -- the B fragment of slice k for column (t, w) is then the unit of the RAW tile at joint
// idx_k(w) -- or empty (the always-zero unit); the few others (2 forward, 8 transposed) become VIRTUAL joints behind the raw
// columns: z formed in fp32 (the fp32 kernel's fma chain) from <= 4 gathered columns, then split like any other column.  So only
// ONE image per term is staged, whatever the slice.  Stage = 16 src channels (lanes 0-31 multiply channels 0-7, lanes 32-63
// channels 8-15 of the stage), three k-steps = the three slices.  The caller asserts the list structure (SAR_GRAPH_FEW_DENSE with
// <= 16 such lists); no folded prologue (the engines have none in front of a graph convolution).
template <int AR>
__global__ __launch_bounds__(256, AR == AR_H3A ? 3 : 2) void conv_graph_split_kernel(const ConvKS k) {
  constexpr int NTA = ar_nta(AR), NTB = ar_ntb(AR), NPROD = ar_nprod(AR), NACC = ar_two_acc(AR) ? 2 : 1;
  constexpr bool SCALED = ar_f16(AR);
  constexpr int BM = 64, MS = 2, NS = 2, WN = 4, V = VJ, FTG = 10, NVMAX = 16, KC16 = 16;
  constexpr int VCOL0 = 256, ZCOL = VCOL0 + NVMAX * FTG, SC = ZCOL + 1;
  constexpr int WPIECES = NTA * 6, WU = WPIECES * 64, SU = NTB * 2 * SC;   // weight pieces [term][slice][half][64 rows]; source [term][half][column]
  constexpr int PAREA_U = 4 * 16 * 65 / 4;
  constexpr int IMG_U = (WU + SU) > PAREA_U ? (WU + SU) : PAREA_U;
  constexpr int PPW = (WPIECES + 3) / 4;
  __shared__ uint4 smem_u[IMG_U + BM];
  __shared__ int vmap[3 * V + 1];          // list (k, w) -> raw joint | V + virtual joint | -1 (empty)
  __shared__ int vl_idx[NVMAX * 4];        // the virtual joints' gather entries
  __shared__ float vl_wt[NVMAX * 4];
  __shared__ int nd_s[2];
  __shared__ float gs_s[2];                // largest sum of |weights| of a gather list (per classifying wave)
  uint4* Wl = smem_u;
  uint4* Sl = smem_u + WU;
  float* smem = reinterpret_cast<float*>(smem_u);
  float4* rowp = reinterpret_cast<float4*>(smem_u + IMG_U);
  const sar_conv_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave;
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
  const int nfr = (t0 + k.FT <= d.T_out) ? k.FT : d.T_out - t0;   // live frames of this tile
  const int ncols = nfr * V;
  SPLIT_TL_BEGIN();

  // ---- classify the 3 V gather lists (threads 0 .. 3 V - 1, two waves)
  int l_idx[4] = {0, 0, 0, 0};
  float l_wt[4] = {0.f, 0.f, 0.f, 0.f};
  int cnt = 0, first = 0;
  bool dense = false;
  if (tid < 3 * V) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      l_idx[j] = d.g_idx[tid * 4 + j];
      l_wt[j] = d.g_wt[tid * 4 + j];
    }
#pragma unroll
    for (int j = 3; j >= 0; --j)
      if (l_wt[j] != 0.f) {
        ++cnt;
        first = j;
      }
    dense = cnt > 1 || (cnt == 1 && l_wt[first] != 1.f);
  }
  const unsigned long long dmask = __ballot(dense);
  const int rank_w = __popcll(dmask & ((1ull << lane) - 1ull));
  float lsum = fabsf(l_wt[0]) + fabsf(l_wt[1]) + fabsf(l_wt[2]) + fabsf(l_wt[3]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) lsum = fmaxf(lsum, __shfl_xor(lsum, o));
  if (lane == 0 && wave < 2) {
    nd_s[wave] = __popcll(dmask);
    gs_s[wave] = lsum;
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
  if (tid < 2 * NTB) Sl[tid * SC + ZCOL] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const int nd_raw = nd_s[0] + nd_s[1];
  const int nd = nd_raw < NVMAX ? nd_raw : NVMAX;   // (the host refuses tables with more)
  if (tid < 3 * V) {
    const int rank = (wave == 1 ? nd_s[0] : 0) + rank_w;
    int code = -1;
    if (dense && rank < NVMAX) {
      code = V + rank;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vl_idx[rank * 4 + j] = l_idx[j];
        vl_wt[rank * 4 + j] = l_wt[j];
      }
    } else if (cnt == 1) code = l_idx[first];
    vmap[tid] = code;
  }
  __syncthreads();
  SPLIT_TL(0);   // table classification (two barriers)

  // ---- per-lane column geometry and the three slices' operand columns
  bool colok[NS];
  int64_t coln[NS], colna[NS];   // colna: the aux tensor's column (SAR_GRAPH_AUX_EVEN_FRAMES: even frames only, -1 on odd ones)
  const bool aux_even = (d.g_flags & SAR_GRAPH_AUX_EVEN_FRAMES) != 0;
  float gcs[3][NS];
  int baddr[3][NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    colok[ns] = p < ncols;
    const int pv = colok[ns] ? p : 0;
    coln[ns] = ((int64_t)b * d.T_out + t0) * V + pv;
    const int fo = pv / V, v = pv - fo * V;
    colna[ns] = aux_even ? (((t0 + fo) & 1) ? -1 : ((int64_t)b * ((d.T_out + 1) >> 1) + ((t0 + fo) >> 1)) * V + v) : coln[ns];
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) {
      gcs[tp][ns] = (d.g_colsum && colok[ns]) ? d.g_colsum[tp * V + v] : 0.f;
      const int code = vmap[tp * V + v];
      int col = ZCOL;
      if (colok[ns] && code >= 0) col = code < V ? fo * V + code : VCOL0 + fo * nd + (code - V);
      baddr[tp][ns] = col + hi * SC;
    }
  }

  int ea = 0, ew = 0;
  bool nonfin = false;   // an operand bound holds Inf / NaN bits: every output of the launch is NaN (split_scale.h)
  if (SCALED) {   // a virtual joint is a weighted SUM of <= 4 source values: the bound of what is staged is bound(src) x max sum |weights|
    const float gmax = fmaxf(fmaxf(gs_s[0], gs_s[1]), 1.f);
    ea = scale_exp(__float_as_uint(__uint_as_float(*k.src_bound) * gmax));
    ew = scale_exp(*k.w_bound);
    nonfin = bound_nonfinite(*k.src_bound) || bound_nonfinite(*k.w_bound);
  }
  const float sa = __builtin_ldexpf(1.f, ea);
  f32x16 acc[NACC][MS][NS];

  // ---- source staging.  Raw columns: thread = column, both channel halves.  Virtual joints: thread = (virtual column, half).
  const int seq_len = d.T_src * V;
  const float* src_t = d.src + ((int64_t)b * d.T_src + t0) * V;
  const unsigned tile_bytes = (unsigned)(seq_len - t0 * V) * 4;
  const int svo = tid < ncols ? tid * 4 : 0x7fffffff;   // rejected -> 0
  const int vt = tid >> 1, vh = tid & 1;
  const bool vact = vt < nfr * nd;
  const bool wave_virt = wave * 32 < nfr * nd;   // wave-uniform: this wave owns virtual joints (the others skip their 32 loads)
  int vvo[4];
  float vwt[4];
  {
    const int tf = nd > 0 ? vt / (nd > 0 ? nd : 1) : 0;
    const int l = vt - tf * nd;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vwt[j] = vact ? vl_wt[l * 4 + j] : 0.f;
      // byte offset from row c0 + q of the stage: the lane's channel half (8 rows further) + the gathered column; an unused entry
      // reads the tile's first float (weight 0)
      vvo[j] = (int)(((int64_t)vh * 8 * d.ld_src + ((vact && vwt[j] != 0.f) ? tf * V + vl_idx[l * 4 + j] : 0)) * 4);
    }
  }
  const unsigned vbytes = (unsigned)(8 * d.ld_src * 4) + tile_bytes;   // one descriptor per q covers both halves' rows
  float sreg[2][8], vreg[4][8];
  auto issue_s_loads = [&](int c0) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + 8 * h + q;
        const int cg = c < d.Kc ? c : 0;   // wave-uniform
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src_t + (int64_t)cg * d.ld_src), 0, tile_bytes, 0x00020000);
        sreg[h][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo, 0, 0));
      }
    // The virtual joints' gathered loads: BUFFER loads (the lane's channel half rides in its byte offset), unconditional per lane,
    // inside ONE wave-uniform branch.  (Plain pointer loads here were FLAT loads: they count in lgkmcnt too and complete out of
    // order, so the first LDS fragment read of the MFMA phase waited for all 32 of them -- 3 000 cycles per stage on the wave every
    // other wave waits for, more than the stage's MFMA phase; tools/split_timeline.sh.)
    if (wave_virt) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + q;   // Kc % 16 == 0: both halves' rows exist
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src_t + (int64_t)c * d.ld_src), 0, vbytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) vreg[j][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vvo[j], 0, 0));
      }
    }
  };
  auto store_s = [&](int c0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float x = SCALED ? sreg[h][q] * sa : sreg[h][q];
        v[q] = (tid < ncols && c0 + 8 * h + q < d.Kc) ? x : 0.f;
      }
      uint4 u[NTB];
      split8<AR, false>(v, u, 1.f);
#pragma unroll
      for (int t = 0; t < NTB; ++t) Sl[(t * 2 + h) * SC + tid] = u[t];
    }
    if (vact) {   // z = sum_j wt_j x_j: the fp32 kernel's chain (conv_gemm.hip), then the split
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float z = vwt[0] * vreg[0][q];
#pragma unroll
        for (int j = 1; j < 4; ++j) z = fmaf(vwt[j], vreg[j][q], z);
        z = SCALED ? z * sa : z;
        v[q] = (c0 + 8 * vh + q < d.Kc) ? z : 0.f;
      }
      uint4 u[NTB];
      split8<AR, false>(v, u, 1.f);
#pragma unroll
      for (int t = 0; t < NTB; ++t) Sl[(t * 2 + vh) * SC + VCOL0 + vt] = u[t];
    }
  };
  // weight pieces by LDS-DMA: piece p = (term * 3 + slice) * 2 + half = 64 rows of channel group g0 + half
  const unsigned wbytes = (unsigned)((int64_t)NTA * 3 * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  const unsigned wvo = (m0 + lane) < d.M ? (unsigned)((m0 + lane) * 16) : 0x80000000u;
  auto issue_w_dma = [&](int g0) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + 4 * i;   // wave-uniform
      if (p < WPIECES) {
        const int ts = p >> 1, h = p & 1;   // ts = term * 3 + slice
        const int g = g0 + h;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(Wl + p * 64), 16, g < k.G ? wvo : 0x80000000u, (ts * k.G + g) * d.M * 16, 0, 0);
      }
    }
  };

  issue_s_loads(0);
  issue_w_dma(0);
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 bp = rowp[ms * 32 + mfma_row(r, hi)];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        if (!SCALED) acc[0][ms][ns][r] = fmaf(bp.z, gcs[2][ns], fmaf(bp.y, gcs[1][ns], bp.x * gcs[0][ns]));
        else acc[0][ms][ns][r] = 0.f, acc[NACC - 1][ms][ns][r] = 0.f;
      }
    }

  const int abase = hi * 64 + l31;
  // Happens-before of the single image: as conv_gemm_split_kernel (store_s / DMA behind the closing barrier, the opening barrier
  // behind every wave's ds_writes and vmcnt(0))
  const int nst = (d.Kc + KC16 - 1) / KC16;
  SPLIT_TL(7);   // geometry, first requests, accumulator initialisation
  for (int s_ = 0; s_ < nst; ++s_) {
    store_s(s_ * KC16);
    SPLIT_TL(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SPLIT_TL(2);
    __syncthreads();   // opening
    SPLIT_TL(3);
    if (s_ + 1 < nst) issue_s_loads((s_ + 1) * KC16);
    SPLIT_TL(8);   // load issue
    SAR_LDS_SKEW();
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) {
      uint4 a[NTA][MS], bq[NTB][NS];
#pragma unroll
      for (int t = 0; t < NTA; ++t)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) a[t][ms] = Wl[(t * 3 + tp) * 128 + abase + ms * 32];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int t = 0; t < NTB; ++t) bq[t][ns] = Sl[t * 2 * SC + baddr[tp][ns]];
#pragma unroll
      for (int p = 0; p < NPROD; ++p) {
        const int i = ar_pi(AR, p), j = ar_pj(AR, p);
        const int ai = (NACC == 2 && i + j > 0) ? 1 : 0;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) {
            if (ar_f16(AR))
              acc[ai][ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<f16x8*>(&a[i][ms]),
                                                                       *reinterpret_cast<f16x8*>(&bq[j][ns]), acc[ai][ms][ns], 0, 0, 0);
            else
              acc[ai][ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&a[i][ms]),
                                                                        *reinterpret_cast<bf16x8*>(&bq[j][ns]), acc[ai][ms][ns], 0, 0, 0);
          }
      }
    }
    SPLIT_TL(4);
    __syncthreads();   // closing
    if (s_ + 1 < nst) issue_w_dma(2 * (s_ + 1));
    SPLIT_TL(5);
  }

  if (SCALED) {
    const float c0 = (nonfin ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + ew))), c1 = NACC == 2 ? c0 / H3_LO : 0.f;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 bp = rowp[ms * 32 + mfma_row(r, hi)];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          const float bias = fmaf(bp.z, gcs[2][ns], fmaf(bp.y, gcs[1][ns], bp.x * gcs[0][ns]));
          const float v = fmaf(acc[NACC - 1][ms][ns][r], c1, acc[0][ms][ns][r] * c0) + bias;
          acc[0][ms][ns][r] = colok[ns] ? v : 0.f;
        }
      }
    __syncthreads();   // every wave has read its bias rows: the gated epilogue rewrites rowp
  }
  epilogue_b<MS, NS, WN, BM>(d, k.nparts, tile, 0, wn, m0, colok, coln, acc[0], rowp, smem, colna);
  SPLIT_TL(6);
  SPLIT_TL_END(w);
}

// ---- The 1-tap TEMPORAL operator (the strided 1x1 residual convolution of models/stgcn.py:47-54 and the dense 1x1 products of its
// data gradient) in the split arithmetics -- round 6, VERDICT r05 next #1c ("fp32 leftovers").  These launches are memory-bound
// (<= 16 GFLOP for ~0.4-0.5 GB) but on the fp32 matrix pipe their matrix time alone (50 / 100 us at 64 -> 128 / 128 -> 256 channels)
// exceeded their HBM time (tools/leftover_bench.py: 163 / 241 us forward, 146 / 219 us dense data gradient).
//   out[m, (b, t, v)] = sum_c W[c][m] src[c, (b, t stride, v)] (+ bias[m]) ; epilogue NONE / STATS / MASK / ADD   (no folded prologue)
// The graph kernel's skeleton with ONE slice and no gather: tile 64 x 256 (10 output frames), 4 waves side by side, stage = 32
// source channels = two k-steps of 16 (lanes 0-31 channels 0-7, lanes 32-63 channels 8-15 of the k-step), weights by LDS-DMA
// ([term][group][64 rows] pieces), the source split by the stager (thread = output column; its source column is the same joint of
// frame t * stride: a stride-2 launch reads every other 100-byte frame), single image, two barriers per stage.
template <int AR>
__global__ __launch_bounds__(256, 3) void conv_tap1_split_kernel(const ConvKS k) {
  constexpr int NTA = ar_nta(AR), NTL = nta_lds(AR), NTB = ar_ntb(AR), NPROD = ar_nprod(AR);
  static_assert(!ar_two_acc(AR), "one accumulator set");
  constexpr bool SCALED = ar_f16(AR);
  constexpr int BM = 64, MS = 2, NS = 2, WN = 4, KCS = 32, NG = KCS / 8;
  constexpr int ZCOL = 256, SC = ZCOL + 1;
  constexpr int WPIECES = NTL * NG, WU = WPIECES * 64, SU = NTB * NG * SC;   // weight pieces [term][group][64 rows]; source [term][group][column]
  constexpr int PAREA_U = 4 * 16 * 65 / 4;
  constexpr int IMG_U = (WU + SU) > PAREA_U ? (WU + SU) : PAREA_U;
  constexpr int PPW = (WPIECES + 3) / 4;
  __shared__ uint4 smem_u[IMG_U + BM];
  uint4* Wl = smem_u;
  uint4* Sl = smem_u + WU;
  float* smem = reinterpret_cast<float*>(smem_u);
  float4* rowp = reinterpret_cast<float4*>(smem_u + IMG_U);
  const sar_conv_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave;
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
  const int V = d.V;
  const int nfr = (t0 + k.FT <= d.T_out) ? k.FT : d.T_out - t0;   // live frames of this tile
  const int ncols = nfr * V;

  // the bias row is requested here and stored behind the first operand requests
  float bias_v = 0.f;
  if (tid < BM && d.bias && m0 + tid < d.M) bias_v = d.bias[m0 + tid];

  bool colok[NS];
  int64_t coln[NS];
  int bcol[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    colok[ns] = p < ncols;
    coln[ns] = ((int64_t)b * d.T_out + t0) * V + (colok[ns] ? p : 0);
    bcol[ns] = colok[ns] ? p : ZCOL;
  }

  int ea = 0, ew = 0;
  bool nonfin = false;   // an operand bound holds Inf / NaN bits: every output of the launch is NaN (split_scale.h)
  if (SCALED) {
    ea = scale_exp(*k.src_bound);
    ew = scale_exp(*k.w_bound);
    nonfin = bound_nonfinite(*k.src_bound) || bound_nonfinite(*k.w_bound);
  }
  const float sa = __builtin_ldexpf(1.f, ea);
  f32x16 acc[MS][NS];

  // ---- source staging: thread = output column tid of the tile; its source column is joint v of frame (t0 + fo) * stride of
  // sequence b; the descriptor spans exactly the sequence row (a dead thread's offset is rejected: 0)
  const int seq_len = d.T_src * V;
  const float* src_b = d.src + (int64_t)b * seq_len;
  int svo;
  {
    const int fo = tid / V, v = tid - fo * V;
    svo = tid < ncols ? (((t0 + fo) * d.stride) * V + v) * 4 : 0x7fffffff;
  }
  float sreg[NG][8];
  auto issue_s_loads = [&](int c0) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = c0 + 8 * g + q;
        const int cg = c < d.Kc ? c : 0;   // wave-uniform (masked in store_s)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src_b + (int64_t)cg * d.ld_src), 0, (unsigned)seq_len * 4u, 0x00020000);
        sreg[g][q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, svo, 0, 0));
      }
  };
  auto store_s = [&](int c0) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float x = SCALED ? sreg[g][q] * sa : sreg[g][q];
        v[q] = (tid < ncols && c0 + 8 * g + q < d.Kc) ? x : 0.f;
      }
      uint4 u[NTB];
      split8<AR, false>(v, u, 1.f);
#pragma unroll
      for (int t = 0; t < NTB; ++t) Sl[(t * NG + g) * SC + tid] = u[t];
    }
  };
  // weight pieces by LDS-DMA: piece p = term * NG + group = 64 rows of channel group g0 + group of term image `term`
  const unsigned wbytes = (unsigned)((int64_t)NTA * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  const unsigned wvo = (m0 + lane) < d.M ? (unsigned)((m0 + lane) * 16) : 0x80000000u;
  auto issue_w_dma = [&](int g0) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + 4 * i;   // wave-uniform
      if (p < WPIECES) {
        const int t = p / NG, g = g0 + (p - t * NG);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(Wl + p * 64), 16, g < k.G ? wvo : 0x80000000u, (t * k.G + g) * d.M * 16, 0, 0);
      }
    }
  };

  issue_s_loads(0);
  issue_w_dma(0);
  asm volatile("" ::: "memory");
  if (tid < BM) rowp[tid] = make_float4(bias_v, 0.f, 0.f, 0.f);
  if (tid < NTB * NG) Sl[tid * SC + ZCOL] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int ms = 0; ms < MS; ++ms)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;

  // Happens-before of the single image: as conv_graph_split_kernel (store_s / DMA behind the closing barrier, the opening barrier
  // behind every wave's ds_writes and vmcnt(0)); rowp / the zero column are written in front of the first opening barrier
  const int nst = (d.Kc + KCS - 1) / KCS;
  for (int s_ = 0; s_ < nst; ++s_) {
    store_s(s_ * KCS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // opening
    if (s_ + 1 < nst) issue_s_loads((s_ + 1) * KCS);
    SAR_LDS_SKEW();
#pragma unroll
    for (int kk = 0; kk < NG / 2; ++kk) {
      uint4 a[NTA][MS], bq[NTB][NS];
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) a[t][ms] = Wl[(t * NG + 2 * kk + hi) * 64 + ms * 32 + l31];
      if constexpr (NTL < NTA)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) a[NTA - 1][ms] = third_image(a[0][ms]);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int t = 0; t < NTB; ++t) bq[t][ns] = Sl[(t * NG + 2 * kk + hi) * SC + bcol[ns]];
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
    }
    __syncthreads();   // closing
    if (s_ + 1 < nst) issue_w_dma(NG * (s_ + 1));
  }

  {
    const float c0 = SCALED ? (nonfin ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + ew))) : 1.f;
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 bp = rowp[ms * 32 + mfma_row(r, hi)];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          const float v = fmaf(acc[ms][ns][r], c0, bp.x);
          acc[ms][ns][r] = colok[ns] ? v : 0.f;
        }
      }
    __syncthreads();   // every wave has read its bias rows: the MASK epilogue rewrites rowp
  }
  epilogue_b<MS, NS, WN, BM>(d, k.nparts, tile, 0, wn, m0, colok, coln, acc, rowp, smem);
}

// ---- The same contraction as conv_graph_split_kernel -- BIT-IDENTICAL results: same term images, same products in the same order,
// same epilogue, same partial-sum layout -- rebuilt around what the per-workgroup timelines of round 5 showed
// (profiles/r05_f32split_timelines.txt: a 64-channel forward workgroup lived 45 000 cycles for 7 300 cycles of matrix work and the
// launch moved 2.2 TB/s): the source loads of a stage were in flight only during the previous stage's MFMA phase (a third of the
// time), every workgroup classified the same 75 gather lists, rebuilt its column geometry and re-read its bias rows for ONE tile
// of 250 columns, and the virtual joints' 32 gathered global loads per lane sat on the critical wave.
//  * PERSISTENT workgroups: 2 per CU, each walks its XCD's tiles (tile = xcd + 8 (lane + i lanes): the row blocks of a tile and
//    consecutive tiles share an L2); gather-list classification, column geometry, bias rows once per workgroup.
//  * the RAW fp32 source of a stage (16 channel rows x <= 250 columns) reaches LDS by LDS-DMA (buffer_load_dwordx4 ... lds on
//    whole lanes + one buffer_load_dword ... lds for the row's last ncols % 4 columns: row starts are only 4-byte aligned) into a
//    ring of two buffers, issued TWO stages ahead -- across tile boundaries, so the next tile's first stages are in flight during
//    this tile's epilogue.  No source registers, no per-lane descriptor arithmetic: four DMA instructions per wave and stage.
//  * the stager reads the raw tile from LDS (a lane = a column, conflict-free), scales, splits into the fp16 terms and writes the
//    single term image as before; the virtual joints (weighted sums of <= 4 joints of the same frame) are formed from the SAME raw
//    tile -- their global gathers are gone.
// Happens-before per stage g (buffers: raw ring g & 1, one term image Sl, one weight image Wl):
//   [C(g-1)] -> issue W DMA(g) -> convert raw(g) -> Sl -> s_waitcnt vmcnt(0) (own raw(g+1) rows, own W(g) pieces) -> barrier B(g)
//   -> issue raw DMA(g+2) into ring slot g & 1 (every wave is past its reads of raw(g)) -> MFMA(g) -> barrier C(g).
//   raw(g) was issued behind B(g-2) and waited for by its issuing wave in front of B(g-1): visible to every wave behind C(g-1).
// The epilogue's transpose area aliases Sl only (W DMA of the next tile's first stage is in flight during the epilogue); a barrier
// E separates it from the next tile's first convert.
template <int AR, int EPI>
__global__ __launch_bounds__(256, 2) void conv_graph_split2_kernel(const ConvKS k) {
  constexpr int NTA = ar_nta(AR), NTB = ar_ntb(AR), NPROD = ar_nprod(AR);
  static_assert(AR == AR_H3A, "built for the engine's arithmetic (one accumulator, scaled fp16 terms)");
  constexpr int BM = 64, MS = 2, NS = 2, WN = 4, V = VJ, FTG = 10, NVMAX = 16, KC16 = 16;
  constexpr int VCOL0 = 256, ZCOL = VCOL0 + NVMAX * FTG, SC = ZCOL + 1;
  constexpr int WPIECES = NTA * 6, WU = WPIECES * 64, SU = NTB * 2 * SC;
  constexpr int PAREA_U = 4 * 16 * 65 / 4;
  static_assert(SU >= PAREA_U, "the epilogue's transpose area must fit the term image");
  constexpr int PPW = (WPIECES + 3) / 4;
  constexpr int RAWP = 256, RAW_U = KC16 * RAWP / 4;   // raw stage buffer: 16 rows of 256 floats
  __shared__ uint4 smem_u[SU + WU + 2 * RAW_U + 2 * BM];
  __shared__ int vmap[3 * V + 1];
  __shared__ int vl_idx[NVMAX * 4];
  __shared__ float vl_wt[NVMAX * 4];
  __shared__ int nd_s[2];
  __shared__ float gs_s[2];
  uint4* Sl = smem_u;
  uint4* Wl = smem_u + SU;
  float* rawl = reinterpret_cast<float*>(smem_u + SU + WU);
  float4* rowp_b = reinterpret_cast<float4*>(smem_u + SU + WU + 2 * RAW_U);        // bias rows (per workgroup)
  float4* rowp_e = rowp_b + BM;                                                    // the epilogue's per-row parameters
  float* smem = reinterpret_cast<float*>(smem_u);
  const sar_conv_desc& d = k.d;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wn = wave;
  // ---- persistent assignment (gridDim.x % (8 ny) == 0): XCD = blockIdx & 7; inside an XCD slot -> (row block, tile lane)
  const int ny = k.ny;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nlanes = (int)(gridDim.x >> 3) / ny;
  const int my = slot % ny, tlane = slot / ny;
  const int m0 = my * BM;
  const int tstep = 8 * nlanes;
  int tile = xcd + 8 * tlane;
  if (tile >= k.ntiles) return;
  SPLIT_TL_BEGIN();

  // ---- classify the 3 V gather lists (as conv_graph_split_kernel), once per workgroup
  int l_idx[4] = {0, 0, 0, 0};
  float l_wt[4] = {0.f, 0.f, 0.f, 0.f};
  int cnt = 0, first = 0;
  bool dense = false;
  if (tid < 3 * V) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      l_idx[j] = d.g_idx[tid * 4 + j];
      l_wt[j] = d.g_wt[tid * 4 + j];
    }
#pragma unroll
    for (int j = 3; j >= 0; --j)
      if (l_wt[j] != 0.f) {
        ++cnt;
        first = j;
      }
    dense = cnt > 1 || (cnt == 1 && l_wt[first] != 1.f);
  }
  const unsigned long long dmask = __ballot(dense);
  const int rank_w = __popcll(dmask & ((1ull << lane) - 1ull));
  float lsum = fabsf(l_wt[0]) + fabsf(l_wt[1]) + fabsf(l_wt[2]) + fabsf(l_wt[3]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) lsum = fmaxf(lsum, __shfl_xor(lsum, o));
  if (lane == 0 && wave < 2) {
    nd_s[wave] = __popcll(dmask);
    gs_s[wave] = lsum;
  }
  if (tid < BM) {
    const int row = m0 + tid;
    float4 bp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.bias && row < d.M) {
      bp.x = d.bias[row];
      bp.y = d.bias[d.M + row];
      bp.z = d.bias[2 * d.M + row];
    }
    rowp_b[tid] = bp;
  }
  if (tid < 2 * NTB) Sl[tid * SC + ZCOL] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  const int nd_raw = nd_s[0] + nd_s[1];
  const int nd = nd_raw < NVMAX ? nd_raw : NVMAX;
  if (tid < 3 * V) {
    const int rank = (wave == 1 ? nd_s[0] : 0) + rank_w;
    int code = -1;
    if (dense && rank < NVMAX) {
      code = V + rank;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vl_idx[rank * 4 + j] = l_idx[j];
        vl_wt[rank * 4 + j] = l_wt[j];
      }
    } else if (cnt == 1) code = l_idx[first];
    vmap[tid] = code;
  }
  __syncthreads();

  // ---- tile-invariant column geometry: the three slices' operand columns of a FULL tile, the bias term's column sums
  int pcol[NS];
  float gcs[3][NS];
  int baddr_full[3][NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    const int p = (wn * NS + ns) * 32 + l31;
    pcol[ns] = p;
    const bool in = p < FTG * V;
    const int pv = in ? p : 0;
    const int fo = pv / V, v = pv - fo * V;
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) {
      gcs[tp][ns] = (d.g_colsum && in) ? d.g_colsum[tp * V + v] : 0.f;
      const int code = vmap[tp * V + v];
      int col = ZCOL;
      if (in && code >= 0) col = code < V ? fo * V + code : VCOL0 + fo * nd + (code - V);
      baddr_full[tp][ns] = col + hi * SC;
    }
  }
  const float gmax = fmaxf(fmaxf(gs_s[0], gs_s[1]), 1.f);
  const int ea = scale_exp(__float_as_uint(__uint_as_float(*k.src_bound) * gmax));
  const int ew = scale_exp(*k.w_bound);
  const bool nonfin = bound_nonfinite(*k.src_bound) || bound_nonfinite(*k.w_bound);
  const float sa = __builtin_ldexpf(1.f, ea);
  const float c0 = nonfin ? __uint_as_float(0x7fc00000u) : __builtin_ldexpf(1.f, -(ea + ew));
  const bool aux_even = (d.g_flags & SAR_GRAPH_AUX_EVEN_FRAMES) != 0;

  // ---- the virtual joints of the stager: thread = (virtual column, channel half); gathers inside the raw tile
  const int vt = tid >> 1, vh = tid & 1;
  const int vtf = nd > 0 ? vt / (nd > 0 ? nd : 1) : 0;
  const bool vact_full = nd > 0 && vt < FTG * nd;
  int vcol[4];
  float vwt[4];
  {
    const int l = vt - vtf * nd;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vwt[j] = vact_full ? vl_wt[l * 4 + j] : 0.f;
      vcol[j] = (vact_full && vwt[j] != 0.f) ? vtf * V + vl_idx[l * 4 + j] : 0;   // an unused entry reads the tile's first column (weight 0)
    }
  }

  // ---- DMA issue.  Raw rows: wave w moves rows w, w + 4, w + 8, w + 12 of a stage.  The prefetch cursor (pf_*) runs two stages
  // ahead of the stage being multiplied, across tile boundaries.
  const int seq_len = d.T_src * V;
  const int nst = d.Kc / KC16;
  auto tile_geo = [&](int t, const float*& src_t, int& ncols_t) {
    const int b = t / k.TPS;
    const int t0 = (t - b * k.TPS) * k.FT;
    const int nfr = (t0 + k.FT <= d.T_out) ? k.FT : d.T_out - t0;
    ncols_t = nfr * V;
    src_t = d.src + ((int64_t)b * d.T_src + t0) * V;
  };
  int pf_tile = tile, pf_s = 0, pf_ncols;
  const float* pf_src;
  tile_geo(pf_tile, pf_src, pf_ncols);
  auto issue_raw = [&](int buf) {   // raw(pf_tile, pf_s) -> ring slot buf; advances the cursor
    if (pf_tile < k.ntiles) {
      const int nfull = pf_ncols >> 2, rem = pf_ncols & 3;
      const unsigned rbytes = (unsigned)pf_ncols * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave + 4 * i;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(pf_src + (int64_t)(pf_s * KC16 + row) * d.ld_src), 0, rbytes, 0x00020000);
        float* dst = rawl + buf * (KC16 * RAWP) + row * RAWP;
        if (lane < nfull) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, lane * 16, 0, 0, 0);
        if (lane < rem) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + 4 * nfull), 4, (4 * nfull + lane) * 4, 0, 0, 0);
      }
      if (++pf_s == nst) {
        pf_s = 0;
        pf_tile += tstep;
        if (pf_tile < k.ntiles) tile_geo(pf_tile, pf_src, pf_ncols);
      }
    }
  };
  const unsigned wbytes = (unsigned)((int64_t)NTA * 3 * k.G * d.M * 16);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)k.wp, 0, wbytes, 0x00020000);
  const unsigned wvo = (m0 + lane) < d.M ? (unsigned)((m0 + lane) * 16) : 0x80000000u;
  auto issue_w_dma = [&](int g0) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + 4 * i;   // wave-uniform
      if (p < WPIECES) {
        const int ts = p >> 1, h = p & 1;   // ts = term * 3 + slice
        const int g = g0 + h;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(Wl + p * 64), 16, g < k.G ? wvo : 0x80000000u, (ts * k.G + g) * d.M * 16, 0, 0);
      }
    }
  };

  issue_raw(0);
  issue_raw(1);
  issue_w_dma(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue's only exposed round trip
  __syncthreads();
  SPLIT_TL(0);   // classification, geometry, first requests and their round trip
  int ring = 0;   // ring slot of the stage about to be converted
  const int abase = hi * 64 + l31;

  for (; tile < k.ntiles; tile += tstep) {
    const int b = tile / k.TPS;
    const int t0 = (tile - b * k.TPS) * k.FT;
    const int nfr = (t0 + k.FT <= d.T_out) ? k.FT : d.T_out - t0;
    const int ncols = nfr * V;
    bool colok[NS];
    int64_t coln[NS], colna[NS];
    int baddr[3][NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      colok[ns] = pcol[ns] < ncols;
      const int pv = colok[ns] ? pcol[ns] : 0;
      coln[ns] = ((int64_t)b * d.T_out + t0) * V + pv;
      const int fo = pv / V;
      colna[ns] = aux_even ? (((t0 + fo) & 1) ? -1 : ((int64_t)b * ((d.T_out + 1) >> 1) + ((t0 + fo) >> 1)) * V + (pv - fo * V)) : coln[ns];
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) baddr[tp][ns] = colok[ns] ? baddr_full[tp][ns] : ZCOL + hi * SC;
    }
    const bool vact = vact_full && vtf < nfr;
    f32x16 acc[MS][NS];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.f;
    SPLIT_TL(7);   // per-tile geometry + accumulator initialisation

    for (int s_ = 0; s_ < nst; ++s_) {
      // ---- convert raw(ring) -> the term image
      {
        const float* raw = rawl + ring * (KC16 * RAWP);
        if (G2_ON(1))
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float x = raw[(8 * h + q) * RAWP + tid] * sa;
            v[q] = tid < ncols ? x : 0.f;
          }
          uint4 u[NTB];
          split8<AR, false>(v, u, 1.f);
#pragma unroll
          for (int t = 0; t < NTB; ++t) Sl[(t * 2 + h) * SC + tid] = u[t];
        }
        if (vact && G2_ON(2)) {   // z = sum_j wt_j x_j: the fp32 kernel's chain (conv_gemm.hip), then the split
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float* rr = raw + (8 * vh + q) * RAWP;
            float z = vwt[0] * rr[vcol[0]];
#pragma unroll
            for (int j = 1; j < 4; ++j) z = fmaf(vwt[j], rr[vcol[j]], z);
            v[q] = z * sa;
          }
          uint4 u[NTB];
          split8<AR, false>(v, u, 1.f);
#pragma unroll
          for (int t = 0; t < NTB; ++t) Sl[(t * 2 + vh) * SC + VCOL0 + vt] = u[t];
        }
      }
      SPLIT_TL(1);   // convert
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own rows of the NEXT raw stage (issued a stage ago) and own pieces of this stage's weights
      SPLIT_TL(2);   // DMA wait
      __syncthreads();   // B: the term image and the weight pieces are visible; every wave is done with raw(ring)
      SPLIT_TL(3);
      if (G2_ON(0)) issue_raw(ring);   // two stages ahead
      SPLIT_TL(8);   // raw DMA issue
      SAR_LDS_SKEW();
      if (G2_ON(3))
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        uint4 a[NTA][MS], bq[NTB][NS];
#pragma unroll
        for (int t = 0; t < NTA; ++t)
#pragma unroll
          for (int ms = 0; ms < MS; ++ms) a[t][ms] = Wl[(t * 3 + tp) * 128 + abase + ms * 32];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
          for (int t = 0; t < NTB; ++t) bq[t][ns] = Sl[t * 2 * SC + baddr[tp][ns]];
#pragma unroll
        for (int p = 0; p < NPROD; ++p) {
          const int i = ar_pi(AR, p), j = ar_pj(AR, p);
#pragma unroll
          for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int ns = 0; ns < NS; ++ns)
              acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<f16x8*>(&a[i][ms]),
                                                                   *reinterpret_cast<f16x8*>(&bq[j][ns]), acc[ms][ns], 0, 0, 0);
        }
      }
      SPLIT_TL(4);   // k-steps
      __syncthreads();   // C: every wave is done with the term image and the weight pieces
      ring ^= 1;
      // the next stage's weights: this tile's next channel groups, or the next tile's first
      if (G2_ON(4)) {
        if (s_ + 1 < nst) issue_w_dma(2 * (s_ + 1));
        else if (tile + tstep < k.ntiles) issue_w_dma(0);
      }
      SPLIT_TL(5);   // closing barrier + W DMA issue
    }

    // ---- undo the operand scales, add the bias term, epilogue
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 bp = rowp_b[ms * 32 + mfma_row(r, hi)];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          const float bias = fmaf(bp.z, gcs[2][ns], fmaf(bp.y, gcs[1][ns], bp.x * gcs[0][ns]));
          const float v = acc[ms][ns][r] * c0 + bias;
          acc[ms][ns][r] = colok[ns] ? v : 0.f;
        }
      }
    if (G2_ON(5)) epilogue_b<MS, NS, WN, BM, EPI, true>(d, k.nparts, tile, 0, wn, m0, colok, coln, acc, rowp_e, smem, colna);
    SPLIT_TL(6);   // epilogue
    __syncthreads();   // E: the transpose area (inside Sl) is free again
    SPLIT_TL(9);
    if (tid < 2 * NTB) Sl[tid * SC + ZCOL] = make_uint4(0u, 0u, 0u, 0u);   // (the transpose area covers the zero column of term image 0 only when SC < PAREA: rewrite, cheap)
  }
  SPLIT_TL_END(blockIdx.x);
}

int geometry_s(const sar_conv_desc& d, int tr, ConvKS& k) {
  const int TAPS = 9;
  k.FT = 10;                                 // 256 / 25, parity split: 2 * (128 / 25)
  if (tr == 3) {
    const int t_even = d.T_out + (d.T_out & 1);
    if (k.FT > t_even) k.FT = t_even;
  } else if (k.FT > d.T_out) k.FT = d.T_out;
  k.TPS = (d.T_out + k.FT - 1) / k.FT;
  int nf;
  if (tr == 4 || tr == 5) nf = k.FT;
  else if (tr == 0) nf = (k.FT - 1) * d.stride + TAPS;
  else if (tr == 1) nf = k.FT + TAPS - 1;
  else nf = (k.FT - 1 + TAPS - 1) / 2 + 2;
  k.RW = nf * d.V;
  k.nparts = d.B * k.TPS * 4;
  k.ntiles = d.B * k.TPS;
  k.ny = (d.M + 63) / 64;
  k.G = (d.Kc + 7) / 8;
  return 0;
}

// which kernel a descriptor takes: 0 / 1 / 3 (TR), 4 = graph, or -1 = not built (the caller keeps sar_conv_gemm_f32)
int split_tr(const sar_conv_desc& d) {
  if (d.mode == SAR_CONV_GRAPH) {
    const int ndense = (d.g_flags >> SAR_GRAPH_FEW_DENSE_SHIFT) & 0xff;
    if (d.taps != 3 || d.V != VJ || d.T_src != d.T_out || d.pro_scale) return -1;
    if (!(d.g_flags & SAR_GRAPH_FEW_DENSE) || ndense > 16) return -1;
    if (d.Kc < 16 || d.Kc > 256 || (d.Kc & 15) || (d.M & 7)) return -1;
    return 4;
  }
  if (d.mode == SAR_CONV_TEMPORAL && d.taps == 1) {   // conv_tap1_split_kernel (5): forward form only (a dense data gradient is a forward launch with W^T)
    if (d.V != VJ || d.transposed || d.pad != 0 || d.pro_scale || (d.stride != 1 && d.stride != 2)) return -1;
    if (d.Kc < 16 || (d.M & 7) || d.T_src < (d.T_out - 1) * d.stride + 1) return -1;
    return 5;
  }
  if (d.mode != SAR_CONV_TEMPORAL || d.taps != 9 || d.V != VJ) return -1;
  if (d.Kc < 8 || d.Kc > 256 || (d.M & 7)) return -1;
  if (!d.transposed) return (d.stride == 1 || d.stride == 2) ? 0 : -1;
  if (d.stride == 1) return 1;
  if (d.stride == 2) return 3;
  return -1;
}

// process-wide constants read once (no mutable state behind them): the experiment switch of the graph kernel and the CU count
// which launches take conv_graph_split2_kernel: 0 none, 1 all, 2 (default) the data gradients (the forward launch is the one with the
// BatchNorm STATS epilogue; the descriptor's `transposed` is a TEMPORAL field) -- measured alone at the ten layer
// shapes (profiles/r06_graph_split2_kernel_bench.txt): data gradients 3.19 -> 3.00 ms per step, forward 2.53 -> 2.83
int graph_split_v2() {
  static const int v = [] {
    const char* e = getenv("SAR_GRAPH_SPLIT2");
    return !e ? 2 : (e[0] == '0' ? 0 : (e[0] == '1' ? 1 : 2));
  }();
  return v;
}
int device_cus() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
  }();
  return n;
}

template <int AR>
int launch_split(const sar_conv_desc& d, int tr, const uint4* wp, const unsigned* src_bound, const unsigned* w_bound, hipStream_t st) {
  ConvKS k;
  k.d = d;
  k.wp = wp;
  k.src_bound = src_bound;
  k.w_bound = w_bound;
  k.ablate = 0;
#ifdef SAR_G2_ABLATE
  {
    const char* e = getenv("SAR_G2_ABLATE_BITS");
    k.ablate = e ? atoi(e) : 0;
    k.d.reserved0 = k.ablate >> 6;      // bits 6 / 7: conv_epi_f32.h drops the output stores / the partial sums
  }
#endif
  geometry_s(d, tr, k);
  const dim3 grid(((k.ntiles * k.ny + 7) / 8) * 8), block(256);
  if (tr == 4) {
    if constexpr (AR == AR_H3A) {
      if ((graph_split_v2() == 1 || (graph_split_v2() == 2 && d.epi != SAR_EPI_STATS)) && !(d.g_flags & SAR_GRAPH_ONE_TILE_WG)) {   // persistent workgroups, two per CU (SAR_GRAPH_SPLIT2=0 / SAR_GRAPH_ONE_TILE_WG: the one-tile-per-workgroup kernel of round 5)
        const int per = 8 * k.ny;
        const int g2 = (2 * device_cus()) / per * per;
        const dim3 grid2(g2 > 0 ? g2 : per);
        switch (d.epi) {   // one instantiation per epilogue (conv_epi_f32.h: EPIF)
          case SAR_EPI_STATS: hipLaunchKernelGGL((conv_graph_split2_kernel<AR, SAR_EPI_STATS>), grid2, block, 0, st, k); break;
          case SAR_EPI_MASK: hipLaunchKernelGGL((conv_graph_split2_kernel<AR, SAR_EPI_MASK>), grid2, block, 0, st, k); break;
          case SAR_EPI_ADD: hipLaunchKernelGGL((conv_graph_split2_kernel<AR, SAR_EPI_ADD>), grid2, block, 0, st, k); break;
          case SAR_EPI_ADD_GATE: hipLaunchKernelGGL((conv_graph_split2_kernel<AR, SAR_EPI_ADD_GATE>), grid2, block, 0, st, k); break;
          default: hipLaunchKernelGGL((conv_graph_split2_kernel<AR, SAR_EPI_NONE>), grid2, block, 0, st, k); break;
        }
        return 0;
      }
    }
    if constexpr (AR == AR_B6 || AR == AR_H3A) hipLaunchKernelGGL((conv_graph_split_kernel<AR>), grid, block, 0, st, k);
    else return SAR_E_UNSUP;
  } else if (tr == 5) {
    if constexpr (AR == AR_B6 || AR == AR_H3A) hipLaunchKernelGGL((conv_tap1_split_kernel<AR>), grid, block, 0, st, k);
    else return SAR_E_UNSUP;
  } else if (tr == 0 && d.stride == 1) hipLaunchKernelGGL((conv_gemm_split_kernel<0, AR, 0>), grid, block, 0, st, k);
  else if (tr == 0) hipLaunchKernelGGL((conv_gemm_split_kernel<0, AR, 1>), grid, block, 0, st, k);
  else if (tr == 1) hipLaunchKernelGGL((conv_gemm_split_kernel<1, AR, 0>), grid, block, 0, st, k);
  else hipLaunchKernelGGL((conv_gemm_split_kernel<3, AR, 0>), grid, block, 0, st, k);
  return 0;
}

bool ar_known(int ar) { return ar == AR_B1 || ar == AR_B3 || ar == AR_B6 || ar == AR_B9 || ar == AR_H3 || ar == AR_H3S || ar == AR_H3A; }

}  // namespace

#ifdef SAR_SPLIT_TL
extern "C" int sar_debug_split_timeline(unsigned* out, int nwg, int reset) {   // out: [nwg][16] (host memory)
  if (nwg > SPLIT_TL_WG) nwg = SPLIT_TL_WG;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_split_tl), (size_t)nwg * 16 * sizeof(unsigned)) != hipSuccess) return -1;
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_split_tl)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned) * 16 * SPLIT_TL_WG) != hipSuccess) return -1;
  }
  return nwg;
}
#endif

extern "C" int64_t sar_conv_gemm_split_workspace_bytes(const sar_conv_desc* d, int arith) {
  if (!d || d->Kc <= 0 || d->M <= 0 || d->taps <= 0 || !ar_known(arith)) return SAR_E_ARG;
  return (int64_t)ar_nta(arith) * d->taps * ((d->Kc + 7) / 8) * d->M * 16;
}

extern "C" int sar_conv_gemm_split_nparts(const sar_conv_desc* d) {
  if (!d || d->V <= 0 || d->T_out <= 0 || d->B <= 0 || d->M <= 0) return SAR_E_ARG;
  const int tr = split_tr(*d);
  if (tr < 0) return SAR_E_UNSUP;
  ConvKS k;
  geometry_s(*d, tr, k);
  return k.nparts;
}

extern "C" int sar_amax_f32(const float* x, int C, int64_t n, int64_t ld, uint32_t* cell, sar_stream_t s) {
  SAR_REQUIRE(x && cell && C > 0 && C <= 65535 && n > 0 && ld >= n, "sar_amax_f32: bad arguments");
  int64_t chunks = (n + 8191) / 8192;
  if (chunks > 256) chunks = 256;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)chunks, C), dim3(256), 0, as_stream(s), x, n, ld, cell);
  SAR_LAUNCH_CHECK("sar_amax_f32");
  return 0;
}

extern "C" int sar_bn_bound_f32(const float* gamma, const float* beta, int C, double count, uint32_t* cell, sar_stream_t s) {
  SAR_REQUIRE(gamma && beta && cell && C > 0 && count >= 2, "sar_bn_bound_f32: bad arguments");
  hipLaunchKernelGGL(bound_kernel, dim3(1), dim3(256), 0, as_stream(s), gamma, beta, C, (float)sqrt(count - 1.0) * 1.000001f,
                     (const unsigned*)nullptr, 0, cell);
  SAR_LAUNCH_CHECK("sar_bn_bound_f32");
  return 0;
}

extern "C" int sar_affine_bound_f32(const float* scale, const float* shift, int C, const uint32_t* src_cell, uint32_t* cell,
                                    sar_stream_t s) {
  SAR_REQUIRE(scale && shift && src_cell && cell && C > 0, "sar_affine_bound_f32: bad arguments");
  hipLaunchKernelGGL(bound_kernel, dim3(1), dim3(256), 0, as_stream(s), scale, shift, C, 0.f, src_cell, 1, cell);
  SAR_LAUNCH_CHECK("sar_affine_bound_f32");
  return 0;
}

extern "C" int sar_pack_weights_split_batch(const float* base, const sar_pack_item* items, int nitems, int64_t max_units,
                                            int arith, void* out, uint32_t* item_amax, sar_stream_t s) {
  SAR_REQUIRE(base && items && out && nitems > 0 && max_units > 0, "sar_pack_weights_split_batch: bad arguments");
  SAR_REQUIRE(((uintptr_t)out & 15) == 0, "sar_pack_weights_split_batch: out must be 16-byte aligned");
  SAR_REQUIRE(nitems <= 65535 && (max_units + 255) / 256 < (1ll << 31), "sar_pack_weights_split_batch: too many items / units");
  SAR_REQUIRE(ar_known(arith), "sar_pack_weights_split_batch: unknown arithmetic %d", arith);
  if (ar_f16(arith)) {   // the scale of every item from its amax, on the device
    SAR_REQUIRE(item_amax != nullptr, "sar_pack_weights_split_batch: fp16 arithmetics need item_amax[nitems]");
    hipError_t e = hipMemsetAsync(item_amax, 0, sizeof(uint32_t) * nitems, as_stream(s));
    if (e != hipSuccess) { sar_set_error("sar_pack_weights_split_batch: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(pack_amax_kernel, dim3(64, nitems), dim3(256), 0, as_stream(s), base, items, item_amax);
  }
  const dim3 grid((unsigned)((max_units + 255) / 256), nitems), block(256);
  switch (arith) {
    case AR_B1: hipLaunchKernelGGL(pack_split_kernel<AR_B1>, grid, block, 0, as_stream(s), base, items, item_amax, (uint4*)out); break;
    case AR_B3: hipLaunchKernelGGL(pack_split_kernel<AR_B3>, grid, block, 0, as_stream(s), base, items, item_amax, (uint4*)out); break;
    case AR_B6: hipLaunchKernelGGL(pack_split_kernel<AR_B6>, grid, block, 0, as_stream(s), base, items, item_amax, (uint4*)out); break;
    case AR_B9: hipLaunchKernelGGL(pack_split_kernel<AR_B9>, grid, block, 0, as_stream(s), base, items, item_amax, (uint4*)out); break;
    case AR_H3S: hipLaunchKernelGGL(pack_split_kernel<AR_H3S>, grid, block, 0, as_stream(s), base, items, item_amax, (uint4*)out); break;
    case AR_H3A: hipLaunchKernelGGL(pack_split_kernel<AR_H3A>, grid, block, 0, as_stream(s), base, items, item_amax, (uint4*)out); break;
    default: hipLaunchKernelGGL(pack_split_kernel<AR_H3>, grid, block, 0, as_stream(s), base, items, item_amax, (uint4*)out); break;
  }
  SAR_LAUNCH_CHECK("sar_pack_weights_split_batch");
  return 0;
}

extern "C" int sar_conv_gemm_split(const sar_conv_desc* d, int arith, const void* packed, const uint32_t* src_bound,
                                   const uint32_t* w_bound, sar_stream_t s) {
  SAR_REQUIRE(d != nullptr && packed != nullptr, "sar_conv_gemm_split: null descriptor / weight image");
  SAR_REQUIRE(((uintptr_t)packed & 15) == 0, "sar_conv_gemm_split: the weight image must be 16-byte aligned");
  SAR_REQUIRE(ar_known(arith), "sar_conv_gemm_split: unknown arithmetic %d", arith);
  SAR_REQUIRE(!ar_f16(arith) || (src_bound && w_bound), "sar_conv_gemm_split: fp16 arithmetics need the operand bounds");
  SAR_REQUIRE(d->B > 0 && d->V > 0 && d->T_src > 0 && d->T_out > 0 && d->Kc > 0 && d->M > 0, "sar_conv_gemm_split: bad sizes");
  const int tr = split_tr(*d);
  if (tr < 0) {
    sar_set_error("sar_conv_gemm_split: built for the 9-tap temporal convolution at V = 25, stride 1 / 2, 8 <= Kc <= 256, M %% 8 == 0, and "
                  "for the graph convolution at V = 25, taps 3, T_src == T_out, 16 <= Kc <= 256, Kc %% 16 == 0, M %% 8 == 0, no prologue, "
                  "SAR_GRAPH_FEW_DENSE tables with <= 16 non-trivial lists, and for the 1-tap temporal operator at V = 25, forward form, stride 1 / 2, "
                  "pad 0, Kc >= 16, no prologue (mode %d, taps %d, V %d, stride %d, Kc %d, M %d): use sar_conv_gemm_f32",
                  d->mode, d->taps, d->V, d->stride, d->Kc, d->M);
    return SAR_E_UNSUP;
  }
  if (tr == 4) {
    SAR_REQUIRE(d->g_idx && d->g_wt, "sar_conv_gemm_split: graph gather tables required");
    SAR_REQUIRE(!d->bias || d->g_colsum, "sar_conv_gemm_split: graph bias needs g_colsum");
    SAR_REQUIRE(arith == AR_B6 || arith == AR_H3A, "sar_conv_gemm_split: the graph kernel is built for bf16x6 / f16x3a");
  } else {
    SAR_REQUIRE(d->stride >= 1 && d->pad >= 0 && d->pad <= 8, "sar_conv_gemm_split: bad stride/pad");
    if (tr == 5) SAR_REQUIRE(arith == AR_B6 || arith == AR_H3A, "sar_conv_gemm_split: the 1-tap kernel is built for bf16x6 / f16x3a");
  }
  SAR_REQUIRE(d->src && d->out, "sar_conv_gemm_split: null src/out");
  SAR_REQUIRE(d->ld_src >= (int64_t)d->B * d->T_src * d->V && d->ld_out >= (int64_t)d->B * d->T_out * d->V,
              "sar_conv_gemm_split: leading dimension smaller than B*T*V");
  SAR_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "sar_conv_gemm_split: pro_scale/pro_shift mismatch");
  SAR_REQUIRE((int64_t)d->T_src * d->V < (1 << 28), "sar_conv_gemm_split: sequence row too long");
  SAR_REQUIRE(d->ld_out < (1 << 22) && d->ld_aux < (1 << 22), "sar_conv_gemm_split: leading dimension too large (2^22 columns)");
  SAR_REQUIRE(sar_conv_gemm_split_workspace_bytes(d, arith) < (1ll << 31), "sar_conv_gemm_split: weight tensor too large");
  SAR_REQUIRE(d->epi >= SAR_EPI_NONE && d->epi <= SAR_EPI_ADD_GATE, "sar_conv_gemm_split: bad epilogue %d", d->epi);
  if (d->epi == SAR_EPI_STATS || d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD_GATE)
    SAR_REQUIRE(d->partials, "sar_conv_gemm_split: partials required");
  if (d->epi == SAR_EPI_MASK || d->epi == SAR_EPI_ADD || d->epi == SAR_EPI_ADD_GATE)
    SAR_REQUIRE(d->aux && d->ld_aux >= (int64_t)d->B * (aux_even_frames(*d) ? (d->T_out + 1) / 2 : d->T_out) * d->V, "sar_conv_gemm_split: aux required");
  SAR_REQUIRE(!aux_even_frames(*d) || (d->mode == SAR_CONV_GRAPH && (d->epi == SAR_EPI_ADD || d->epi == SAR_EPI_ADD_GATE)),
              "sar_conv_gemm_split: SAR_GRAPH_AUX_EVEN_FRAMES goes with the GRAPH operator and the ADD / ADD_GATE epilogues");
  if (d->epi == SAR_EPI_ADD_GATE)
    SAR_REQUIRE(tr == 4 && d->aux2 && d->aux_mask && (d->ld_aux2 & 3) == 0 && d->ld_aux2 >= (int64_t)d->B * d->T_out * d->V &&
                d->ld_aux2 < (1 << 22), "sar_conv_gemm_split: SAR_EPI_ADD_GATE is the graph data gradient's, with aux2 / aux_mask, ld_aux2 %% 4 == 0");
  if (d->epi == SAR_EPI_MASK) SAR_REQUIRE(d->aux_scale && d->aux_shift, "sar_conv_gemm_split: aux affine required");
  const uint4* wp = (const uint4*)packed;
  hipStream_t st = as_stream(s);
  int rc = 0;
  switch (arith) {
    case AR_B1: rc = launch_split<AR_B1>(*d, tr, wp, src_bound, w_bound, st); break;
    case AR_B3: rc = launch_split<AR_B3>(*d, tr, wp, src_bound, w_bound, st); break;
    case AR_B6: rc = launch_split<AR_B6>(*d, tr, wp, src_bound, w_bound, st); break;
    case AR_B9: rc = launch_split<AR_B9>(*d, tr, wp, src_bound, w_bound, st); break;
    case AR_H3S: rc = launch_split<AR_H3S>(*d, tr, wp, src_bound, w_bound, st); break;
    case AR_H3A: rc = launch_split<AR_H3A>(*d, tr, wp, src_bound, w_bound, st); break;
    default: rc = launch_split<AR_H3>(*d, tr, wp, src_bound, w_bound, st); break;
  }
  if (rc) return rc;
  SAR_LAUNCH_CHECK("sar_conv_gemm_split");
  return 0;
}

// conv_cn8_common.h -- shared by the CN8 convolution translation units (conv_gemm_cn8.hip: temporal / residual GEMMs and the
// gathering graph kernel; conv_graph_cn8.hip: the graph kernel that applies the adjacency when the operand is READ): kernel
// argument block, tile geometry, the epilogue (mask / add, BatchNorm partial sums, bf16 half-unit stores) and the XCD-aware
// work mapping.  Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "cn8.h"
#include <stdlib.h>
#include <type_traits>

// conv_graph_cn8.hip: GraphConvTD with the adjacency applied when the operand is read (library-internal, not part of the ABI)
constexpr int SAR_GRAPH2_NOT_APPLICABLE = -1000;
__attribute__((visibility("hidden"))) int sar_graph2_cn8_dispatch(const sar_conv_desc& d, const void* wp, hipStream_t st, int* np);

// conv_gemm_cn8_dma.hip: the 9-tap data gradients with LDS-DMA operand staging (experiment switch SAR_CN8_DMA=1)
constexpr int SAR_CN8_DMA_NOT_APPLICABLE = -1001;
__attribute__((visibility("hidden"))) int sar_cn8_dma_dispatch(int tr, const sar_conv_desc& d, const void* wp, hipStream_t st, int* np);

namespace {

constexpr int KC16 = 16;   // src channels per main-loop stage = one MFMA k-step = two CN8 planes

#ifndef SAR_ABLATE8
#define SAR_ABLATE8 0   // diagnostic builds only (tools/ablate8.sh): 1 no MFMA, 2 global loads of stage 0 only, 4 no epilogue, 8 LDS stores of stage 0 only
#endif

// In-kernel phase stamps of conv_gemm_cn8_kernel (diagnostic build -DSAR_CN8_STAMPS, tools/stamps8.sh): wave 0 of every
// workgroup adds the shader-clock cycles it spent in [0] store_lds, [1] the wait at the barrier behind it, [2] load issue +
// MFMA phase, [3] the wait at the closing barrier, [4] the prologue, [5] the epilogue; [6] = workgroups, [7] = their
// lifetimes in 100 MHz ticks (s_memrealtime), [8] = the same in shader-clock cycles, [9] = the graph kernel's loads + MFMA phase.
#ifdef SAR_CN8_STAMPS
constexpr int STAMP_WG = 8192;
__device__ unsigned g_stamps8[STAMP_WG][10];   // one row per workgroup (plain stores: same-address atomics would stall the L2 channel)
#define STAMP8_(i)                                               \
  do {                                                           \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    st_acc[i] += t_ - st_last;                                   \
    st_last = t_;                                                \
  } while (0)
#if SAR_CN8_STAMPS == 2   // the graph kernel's prologue in pieces ([0] geometry + colsum, [1] gather tables, [2] B fragments + zero fill, [3] loads + barrier + bias)
#define STAMP8(i)
#define STAMP8P(i) STAMP8_(i)
#else
#define STAMP8(i) STAMP8_(i)
#define STAMP8P(i)
#endif
#else
#define STAMP8(i)
#define STAMP8P(i)
#endif

struct ConvK8 {
  sar_conv_desc d;
  const uint4* wp;   // packed weights [taps][G][M] units of 8 bf16 (sar_pack_weights_bf16_batch)
  int G;             // channel groups of 8 in the packed weights (even)
  int Gs, Go;        // CN8 planes of src / out (= aux)
  int FT, TPS, NF, RW, nparts, ntiles, ny;
  int stagger, stagger_shift;   // experiment (SAR_CN8_STAGGER): first-round workgroups start (id >> shift) % 3 * stagger x 1024 cycles late
};

template <int TAPS, int MS, int NS, int WM, int WN>
struct TileCfg8 {
  static constexpr int BM = 32 * MS * WM;
  static constexpr int TN = 32 * NS * WN;
  static constexpr int RWMAX = (TN == 128 ? 448 : 704);   // staged columns: (FT - 1) * stride + taps frames of V joints
  static constexpr int SCOLS = RWMAX + 8;                 // + the always-zero column
  static constexpr int WUNITS = TAPS * 2 * BM;            // [tap][h][m]
  static constexpr int SUNITS = 2 * SCOLS;                // [h][col]
  static constexpr int CJ = (RWMAX + 255) / 256;          // S units per lane and plane
  static constexpr int WIT = (WUNITS + 255) / 256;        // W units per lane
  static constexpr int UNITS = WUNITS + SUNITS;
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
constexpr int pad_stride8(int n) { return n + ((4 - n % 16) + 16) % 16; }   // smallest plane stride >= n that is 4 (mod 16)

// ---- epilogue: mask / add, BatchNorm partial sums from the fp32 accumulators, bf16 half-unit stores.  Every wave is
// past its last MFMA phase and the closing barrier: the transpose area aliases the operand image.
// PRE: the caller has already (a) loaded the aux half units into axr[((ms * 2 + rb) * 2 + q2) * NS + ns] (issued before
// its last MFMA phase, so their HBM latency is hidden) and (b) staged the MASK parameters (scale, shift, mean) in rowp.
struct Epi8Desc {
  __amdgpu_buffer_rsrc_t ro, ra;
  int so_out, so_aux, g_w, rows_w;
};

// SAR_EPI_ADD_GATE: the second reduction operand (CN8, e.g. the tail's u) and the gate bytes, both [planes][ld_aux2], from plane g_w
struct Gate8Desc {
  __amdgpu_buffer_rsrc_t ru, rm;
  int so_u, so_m;
};
__device__ __forceinline__ Gate8Desc gate8_desc(const ConvK8& k, int g_w) {
  const sar_conv_desc& d = k.d;
  Gate8Desc g;
  const int64_t nu = (int64_t)(k.Go - g_w) * d.ld_aux2;
  const int64_t bu = nu * 16, bm = nu;
  g.ru = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)d.aux2 + (int64_t)g_w * d.ld_aux2 * 16), 0,
                                           (unsigned)(bu <= 0 ? 0 : (bu > 0x7fffffffll ? 0x7fffffffll : bu)), 0x00020000);
  g.rm = __builtin_amdgcn_make_buffer_rsrc((void*)(d.aux_mask + (int64_t)g_w * d.ld_aux2), 0,
                                           (unsigned)(bm <= 0 ? 0 : (bm > 0x7fffffffll ? 0x7fffffffll : bm)), 0x00020000);
  g.so_u = (int)(d.ld_aux2 * 16), g.so_m = (int)d.ld_aux2;
  return g;
}

template <int MS>
__device__ __forceinline__ Epi8Desc epi8_desc(const ConvK8& k, int wm, int m0, bool has_aux) {
  const sar_conv_desc& d = k.d;
  Epi8Desc e;
  e.rows_w = m0 + wm * MS * 32;          // first output row of this wave (multiple of 32)
  e.g_w = e.rows_w >> 3;                 // its first CN8 plane
  auto plane_bytes = [&](int64_t ld) {   // bytes from plane g_w to the end of the tensor (clamped to 2 GiB)
    const int64_t n = (int64_t)(k.Go - e.g_w) * ld * 16;
    return (unsigned)(n <= 0 ? 0 : (n > 0x7fffffffll ? 0x7fffffffll : n));
  };
  e.ro = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)d.out + (int64_t)e.g_w * d.ld_out * 16), 0, plane_bytes(d.ld_out),
                                           0x00020000);
  e.ra = __builtin_amdgcn_make_buffer_rsrc((void*)(has_aux ? (char*)d.aux + (int64_t)e.g_w * d.ld_aux * 16 : (char*)d.out), 0,
                                           has_aux ? plane_bytes(d.ld_aux) : 0u, 0x00020000);
  e.so_out = (int)(d.ld_out * 16), e.so_aux = (int)(d.ld_aux * 16);   // one plane
  return e;
}

// The operands one half block (two CN8 planes) of the epilogue reads from memory: aux half units (AUX) and, for
// SAR_EPI_ADD_GATE (GATE), the second reduction operand + the gate byte.  [q2][ns] = plane 2 hb + q2, column block ns.
template <int NS>
struct Epi8Half {
  u32x2 a[2][NS], u[2][NS];
  unsigned g[2][NS];
};
template <int NS, bool AUX, bool GATE>
__device__ __forceinline__ void epi8_half_loads(const Epi8Desc& e8, const Gate8Desc& g8, const unsigned (&vo)[NS], int hb,
                                                Epi8Half<NS>& h) {
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const int pl = 2 * hb + q2;
      if constexpr (AUX) h.a[q2][ns] = __builtin_amdgcn_raw_buffer_load_b64(e8.ra, vo[ns], pl * e8.so_aux, 0);
      if constexpr (GATE) {
        h.u[q2][ns] = __builtin_amdgcn_raw_buffer_load_b64(g8.ru, vo[ns], pl * g8.so_u, 0);
        h.g[q2][ns] = (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(g8.rm, vo[ns] >> 4, pl * g8.so_m, 0);   // byte = column (vo = 16 col + 8 hi)
      }
    }
}

// PRE: every aux half unit was requested by the kernel (axr).  pre0 (only without PRE): half block 0's operands were requested
// by the kernel before its last MFMA phase (epi8_half_loads with the same descriptors) -- the epilogue's only exposed round trip.
template <int MS, int NS, int WN, int BM, bool PRE = false>
__device__ __forceinline__ void epilogue8(const ConvK8& k, int tile, int wm, int wn, int m0, const unsigned (&vo)[NS],
                                          f32x16 (&acc)[MS][NS], float4* rowp, float* smem,
                                          const u32x2* axr = nullptr, const Epi8Half<NS>* pre0 = nullptr) {
  const sar_conv_desc& d = k.d;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int part = tile * WN + wn;
  auto run = [&](auto EPI_) {
    constexpr int EPI = decltype(EPI_)::value;
    constexpr bool gate = EPI == SAR_EPI_ADD_GATE;
    constexpr bool stats = EPI == SAR_EPI_STATS || EPI == SAR_EPI_MASK || gate;
    constexpr bool has_aux = EPI == SAR_EPI_MASK || EPI == SAR_EPI_ADD || gate;
    if (EPI == SAR_EPI_MASK && !PRE) {
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
    const Epi8Desc e8 = epi8_desc<MS>(k, wm, m0, has_aux);
    const int rows_w = e8.rows_w, g_w = e8.g_w;
    const __amdgpu_buffer_rsrc_t ro = e8.ro, ra = e8.ra;
    // vo[ns]: byte offset of this lane's half unit inside a plane (0x80000000 = off-tile column: rejected)
    const int so_out = e8.so_out, so_aux = e8.so_aux;
    Gate8Desc g8;
    if constexpr (gate) g8 = gate8_desc(k, g_w);
    float* P = smem + wave * (16 * 65);
    // Operands the epilogue reads from memory -- the aux half units (unless the kernel requested them all beforehand: PRE) and, for
    // SAR_EPI_ADD_GATE, the second reduction operand + the gate byte -- are requested ONE half block (two planes) ahead of their
    // use and consumed as raw bf16 pairs: loaded where they were used, every half block paid its own memory round trip, four per
    // 64-row wave tile (tools/g2_timeline.sh: 22 000 of a 64-channel data-gradient workgroup's 49 000 cycles)
    constexpr bool piped = (has_aux && !PRE) || gate;
    constexpr int HB = 2 * MS;
    // without the gate a half block is ~40 vector instructions -- nothing to hide the next one's round trip behind: then ALL half
    // blocks are requested up front (aux only: 8 NS registers per wave tile row block)
    constexpr bool all_ahead = piped && !gate;
    Epi8Half<NS> hbuf[all_ahead ? HB : 2];
    auto bf_elem = [](const u32x2& w, int i) {   // element i (0..3) of a half unit, as cn8_unpack4
      const unsigned dw = w[i >> 1];
      return __uint_as_float((i & 1) ? (dw & 0xffff0000u) : (dw << 16));
    };
    if constexpr (piped) {
      if (pre0) hbuf[0] = *pre0;   // uniform
      else epi8_half_loads<NS, has_aux && !PRE, gate>(e8, g8, vo, 0, hbuf[0]);
      if constexpr (all_ahead) {
#pragma unroll
        for (int h = 1; h < HB; ++h) epi8_half_loads<NS, true, false>(e8, g8, vo, h, hbuf[h]);
      }
    }
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        // registers 8 rb .. 8 rb + 7 = two groups of 4 consecutive channels: planes 4 ms + 2 rb + {0, 1}
        const int hb = 2 * ms + rb;
        constexpr int HM = all_ahead ? 0xff : 1;   // buffer of half block hb: hbuf[hb & HM]
        if constexpr (piped && !all_ahead)
          if (hb + 1 < HB) epi8_half_loads<NS, has_aux && !PRE, gate>(e8, g8, vo, hb + 1, hbuf[(hb + 1) & 1]);
        auto ax = [&](int ns, int r8) {
          if constexpr (PRE) return bf_elem(axr[(hb * 2 + (r8 >> 2)) * NS + ns], r8 & 3);
          else return bf_elem(hbuf[hb & HM].a[r8 >> 2][ns], r8 & 3);
        };
        auto ux = [&](int ns, int r8) { return bf_elem(hbuf[hb & HM].u[r8 >> 2][ns], r8 & 3); };
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = rb * 8 + r8;
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
              const float a = ax(ns, r8);
              val = (fmaf(a, ap.x, ap.y) > 0.f) ? val : 0.f;
              s1 += val;
              s2 = fmaf(val, a - ap.z, s2);
            } else if (EPI == SAR_EPI_ADD) {
              val += ax(ns, r8);
            } else if (gate) {   // channel 4 hi + (r8 & 3) of the unit in plane (r8 >> 2) of this half block
              val += ax(ns, r8);
              val = ((hbuf[hb & HM].g[r8 >> 2][ns] >> (4 * hi + (r8 & 3))) & 1u) ? val : 0.f;
              // the sums run over the value as STORED (rounded to bfloat16): they replace a pass that read the stored tensor, and
              // the apply pass of the block below centres exactly these values
              const float vr = __uint_as_float(cn8_pack2(val, 0.f) << 16);
              s1 += vr;
              s2 = fmaf(vr, ux(ns, r8) - ap.z, s2);
            }
            acc[ms][ns][r] = val;
          }
          if (stats) {
            P[(2 * r8) * 65 + lane] = s1;
            P[(2 * r8 + 1) * 65 + lane] = s2;
          }
        }
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int gq = 4 * ms + 2 * rb + q2;
          if (g_w + gq < k.Go) {   // wave-uniform
#pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
              const int r0 = rb * 8 + 4 * q2;
              u32x2 o;
              o[0] = cn8_pack2(acc[ms][ns][r0], acc[ms][ns][r0 + 1]);
              o[1] = cn8_pack2(acc[ms][ns][r0 + 2], acc[ms][ns][r0 + 3]);
              __builtin_amdgcn_raw_buffer_store_b64(o, ro, vo[ns], gq * so_out, 0);
            }
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
  switch (d.epi) {
    case SAR_EPI_STATS: run(std::integral_constant<int, SAR_EPI_STATS>()); break;
    case SAR_EPI_MASK: run(std::integral_constant<int, SAR_EPI_MASK>()); break;
    case SAR_EPI_ADD: run(std::integral_constant<int, SAR_EPI_ADD>()); break;
    case SAR_EPI_ADD_GATE: run(std::integral_constant<int, SAR_EPI_ADD_GATE>()); break;
    default: run(std::integral_constant<int, SAR_EPI_NONE>()); break;
  }
}

// workgroup id -> work item, XCD-aware (conv_gemm.hip): consecutive ids go to consecutive XCDs; each XCD walks a
// contiguous range of tiles so that the temporal halo and the row blocks of a tile share one L2
__device__ __forceinline__ int xcd_work(int nwork) {
  const int per = (nwork + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int w = xcd * per + slot;
  return (w >= nwork || slot >= per) ? -1 : w;
}

template <int WN>
int tile_geometry8(const sar_conv_desc& d, int NSv, bool parity, ConvK8& k, int rwmax_db = 0) {
  const int tile_n = 32 * NSv * WN;
  if (parity) {
    k.FT = 2 * ((tile_n / 2) / d.V);
    const int t_even = d.T_out + (d.T_out & 1);
    if (k.FT > t_even) k.FT = t_even;
  } else {
    k.FT = tile_n / d.V;
    if (k.FT > d.T_out) k.FT = d.T_out;
  }
  if (k.FT < 1) return -1;
  const int rwmax = rwmax_db ? rwmax_db : (d.mode == SAR_CONV_GRAPH) ? tile_n : (NSv * WN == 4 ? 448 : 704);
  for (;; k.FT -= parity ? 2 : 1) {      // as many frames as the staged window allows
    if (d.mode == SAR_CONV_GRAPH) k.NF = k.FT;
    else if (!d.transposed) k.NF = (k.FT - 1) * d.stride + d.taps;
    else k.NF = (k.FT - 1 + d.taps - 1) / d.stride + 2;
    k.RW = k.NF * d.V;
    if (k.RW <= rwmax || k.FT <= (parity ? 2 : 1)) break;
  }
  if (k.RW > rwmax) return -2;
  k.TPS = (d.T_out + k.FT - 1) / k.FT;
  k.nparts = d.B * k.TPS * WN;
  return 0;
}

void fill_common(const sar_conv_desc& d, const uint4* wp, ConvK8& k) {
  static const int stg = [] { const char* e = getenv("SAR_CN8_STAGGER"); return e ? atoi(e) : 0; }();
  static const int stg_sh = [] { const char* e = getenv("SAR_CN8_STAGGER_SHIFT"); return e ? atoi(e) : 8; }();
  k.stagger = stg, k.stagger_shift = stg_sh;
  k.d = d;
  k.wp = wp;
  k.G = 2 * ((d.Kc + 15) / 16);
  k.Gs = (d.Kc + 7) / 8;
  k.Go = (d.M + 7) / 8;
}

}  // namespace

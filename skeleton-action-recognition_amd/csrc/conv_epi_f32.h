// conv_epi_f32.h -- the fp32-storage epilogue shared by the bf16-operand kernels (conv_gemm_bf16.hip) and the split-arithmetic
// kernels (conv_gemm_split.hip): mask / add, buffer stores, BatchNorm partial sums through a wave-private LDS transpose
// (include/sar_hip.h: SAR_EPI_NONE / STATS / MASK / ADD; partials [M][nparts][2]).  Include inside the unit's anonymous namespace.
#pragma once
// (the including unit has <type_traits> and sar_common.h in scope)

// (-DSAR_G2_ABLATE builds, tools/g2_ablate.sh: desc.reserved0 bit 0 drops the output stores, bit 1 the partial sums)
#ifdef SAR_G2_ABLATE
#define EPI_ABLATE_STORES &&!(d.reserved0 & 1)
#define EPI_ABLATE_STATS &&!(d.reserved0 & 2)
#else
#define EPI_ABLATE_STORES
#define EPI_ABLATE_STATS
#endif
// ---- epilogue (as conv_gemm.hip): mask / add, store, BatchNorm partial sums.  Every wave is past its last MFMA phase
// and the closing barrier: the transpose area aliases the operand image.
// EPIF >= 0: only that epilogue is compiled in (a kernel instantiated per epilogue: the gated variant's 160 operand registers no
// longer set the register count of the launches that only store and sum)
// LEAN: half as many rounds' aux operands requested together (gate: one round = 48 registers instead of 96; mask: two rounds = 32
// instead of 64) -- for a kernel whose epilogue must fit beside a large persistent state (conv_graph_split3_kernel)
template <int MS, int NS, int WN, int BM, int EPIF = -1, bool LEAN = false>
__device__ __forceinline__ void epilogue_b(const sar_conv_desc& d, int nparts, int tile, int wm, int wn, int m0,
                                           const bool (&colok)[NS], const int64_t (&coln)[NS], f32x16 (&acc)[MS][NS],
                                           float4* rowp, float* smem, const int64_t* colna = nullptr,    // colna: the aux tensor's own column index (default: coln)
                                           int team_tid0 = 0, bool rowp_ready = false) {
  // team_tid0 / rowp_ready: a kernel whose epilogue runs on a SUBSET of its waves (conv_graph_split3_kernel: the four consumer waves,
  // threads team_tid0 .. team_tid0 + 255) passes the team's first thread -- the transpose areas and the row-parameter fill are indexed
  // by the thread's position in the team -- and fills rowp itself once per workgroup (rowp_ready: no fill, no barrier in here)
  const int tid = threadIdx.x - team_tid0;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int part = tile * WN + wn;
  const bool stats = d.epi == SAR_EPI_STATS || d.epi == SAR_EPI_MASK || d.epi == SAR_EPI_ADD_GATE;
  auto fast_epilogue = [&](auto EPI_) {
    constexpr int EPI = decltype(EPI_)::value;
    constexpr bool gate = EPI == SAR_EPI_ADD_GATE;   // out = gate(acc + aux), sums of the gated values (include/sar_hip.h; conv_gemm.hip)
    constexpr bool stats = EPI == SAR_EPI_STATS || EPI == SAR_EPI_MASK || gate;
    constexpr bool has_aux = EPI == SAR_EPI_MASK || EPI == SAR_EPI_ADD || gate;
    if ((EPI == SAR_EPI_MASK || gate) && !rowp_ready) {
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
    const int rows_w = m0 + wm * MS * 32;
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
    for (int ns = 0; ns < NS; ++ns) {
      vo_out[ns] = colok[ns] ? (unsigned)((coln[ns] + 4 * hi * d.ld_out) * 4) : 0x80000000u;
      // (colna < 0: this column has no aux element -- SAR_GRAPH_AUX_EVEN_FRAMES on an odd frame: the rejected load returns 0)
      vo_aux[ns] = (colok[ns] && (!colna || colna[ns] >= 0)) ? (unsigned)(((colna ? colna[ns] : coln[ns]) + 4 * hi * d.ld_aux) * 4) : 0x80000000u;
    }
    const int so_out = (int)(d.ld_out * 4), so_aux = (int)(d.ld_aux * 4);
    // SAR_EPI_ADD_GATE: aux2 [M][ld_aux2] fp32 and its gate bytes [M][ld_aux2 / 4] (bit j of byte i = column 4 i + j)
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
    float* P = smem + wave * (16 * 65);
    // The aux operands (ReLU-mask source / skip gradient; gate: + the second sum's tensor and the gate byte) of ALL rounds -- a
    // round = 8 accumulator registers = half a 32-row block -- are requested before the first one is processed (gate: two rounds at a
    // time): one exposed memory round trip per workgroup (gate: two) instead of one per round (tools/split_timeline.sh: the MASK epilogue took 27 000 cycles of a 90 000-cycle
    // 64-channel workgroup against 10 000 for STATS; the gated one 31 000).
    constexpr int NRB = (gate ? 2 : 2 * MS) / (LEAN ? 2 : 1);   // rounds requested together (gate: three operands per value -- 96 registers per pair of rounds)
    float axb[NRB][NS][8], uxb[gate ? NRB : 1][NS][8];
    unsigned gmb[gate ? NRB : 1][NS][8];
    auto load_round = [&](int ms, int rb, float (&ax)[NS][8], float (&ux)[NS][8], unsigned (&gm)[NS][8]) {
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          const int r = rb * 8 + r8;
          ax[ns][r8] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
              ra, vo_aux[ns], (ms * 32 + (r & 3) + 8 * (r >> 2)) * so_aux, 0));
          if (gate) {
            ux[ns][r8] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                ru, vo_u[ns], (ms * 32 + (r & 3) + 8 * (r >> 2)) * so_u, 0));
            gm[ns][r8] = (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(
                rm, vo_m[ns], (ms * 32 + (r & 3) + 8 * (r >> 2)) * so_m, 0);
          }
        }
    };
#pragma unroll
    for (int rd = 0; rd < 2 * MS; ++rd) {
      {
        const int ms = rd >> 1, rb = rd & 1;
        if (has_aux && rd % NRB == 0) {   // the whole batch of rounds is requested before its first round is processed
#pragma unroll
          for (int q = 0; q < NRB; ++q)
            if (rd + q < 2 * MS) load_round((rd + q) >> 1, (rd + q) & 1, axb[q], uxb[gate ? q : 0], gmb[gate ? q : 0]);
        }
        float (&ax)[NS][8] = axb[rd % NRB];
        float (&ux)[NS][8] = uxb[gate ? rd % NRB : 0];
        unsigned (&gm)[NS][8] = gmb[gate ? rd % NRB : 0];
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = rb * 8 + r8;
          const bool grp_ok = rows_w + ms * 32 + 8 * (r >> 2) < d.M;
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
              val = (fmaf(ax[ns][r8], ap.x, ap.y) > 0.f) ? val : 0.f;
              s1 += val;
              s2 = fmaf(val, ax[ns][r8] - ap.z, s2);
            } else if (EPI == SAR_EPI_ADD) {
              val += ax[ns][r8];
            } else if (gate) {   // replaces, for the block below, bn_add_relu_bwd_reduce and the masked-gradient write of the apply pass
              val += ax[ns][r8];
              val = ((gm[ns][r8] >> cbit[ns]) & 1u) ? val : 0.f;
              s1 += val;
              s2 = fmaf(val, ux[ns][r8] - ap.z, s2);
            }
            if (grp_ok EPI_ABLATE_STORES)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), ro, vo_out[ns],
                                                    (ms * 32 + (r & 3) + 8 * (r >> 2)) * so_out, 0);
          }
          if (stats EPI_ABLATE_STATS) {
            P[(2 * r8) * 65 + lane] = s1;
            P[(2 * r8 + 1) * 65 + lane] = s2;
          }
        }
        if (stats EPI_ABLATE_STATS) {
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
          if (sub == 0 && row < d.M) d.partials[((int64_t)row * nparts + part) * 2 + (q & 1)] = t;
        }
      }
    }
  };
  (void)stats;
  if constexpr (EPIF >= 0) {
    fast_epilogue(std::integral_constant<int, EPIF>());
    return;
  }
  switch (d.epi) {   // M % 8 == 0 is a precondition of this kernel (checked by the host)
    case SAR_EPI_STATS: fast_epilogue(std::integral_constant<int, SAR_EPI_STATS>()); break;
    case SAR_EPI_MASK: fast_epilogue(std::integral_constant<int, SAR_EPI_MASK>()); break;
    case SAR_EPI_ADD: fast_epilogue(std::integral_constant<int, SAR_EPI_ADD>()); break;
    case SAR_EPI_ADD_GATE: fast_epilogue(std::integral_constant<int, SAR_EPI_ADD_GATE>()); break;
    default: fast_epilogue(std::integral_constant<int, SAR_EPI_NONE>()); break;
  }
}

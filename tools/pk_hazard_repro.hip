// pk_hazard_repro.hip -- minimal reproducer attempt for the packed-fp32 hazard of csrc/Makefile (conv_gemm_cn8.o is built with
// -fno-slp-vectorize because hipcc packed the MASK epilogue's fma(aux, scale, shift) of two columns into
//     v_pk_fma_f32 v[a:a+1], v[x:x+1], v[a:a+1], v[a:a+1] op_sel:[0,0,1] op_sel_hi:[1,0,1]
// -- destination pair = the (scale, shift) source pair, and the HIGH half reads the LOW register that the low half overwrites --
// after which a few ReLU-mask bits per launch flipped, in lanes 32-63, differently from run to run).
// This program executes exactly that instruction (inline asm, operands pinned) in four contexts and compares every lane with
// the two scalar FMAs it stands for:
//   0  alone, one wave per SIMD                  1  alone, 8 waves per SIMD
//   2  directly behind an MFMA chain             3  operands fresh from an LDS read (ds_read_b64 of the (scale, shift) pair)
//   4  the epilogue's consumer sequence in ONE asm block (v_pk_fma_f32 -> v_cmp_lt_f32 on the high result with nothing in
//      between -> v_cndmask) while the OTHER waves of the workgroup run a v_mfma_f32_32x32x16_bf16 loop on the same SIMDs
//   5  as 4 with the s_nop 0 the compiler places behind the dst == src form
//   6  as 4 but the other waves exit at once (no matrix work beside it)      7  as 4 with a destination pair that is not a source
// Build: hipcc -O2 --offload-arch=gfx950 tools/pk_hazard_repro.hip -o tools/bin/pk_hazard_repro ; run: tools/bin/pk_hazard_repro [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CTX>
__global__ void pk_kernel(const float2* __restrict__ xs, const float2* __restrict__ ps, unsigned long long* bad, int iters) {
  __shared__ float2 lds_p[1024];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long nbad = 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const float2 x = xs[(gid + it * 7919) & 0xfffff];
    float2 p = ps[(gid * 3 + it) & 0xfffff];
    if (CTX == 3) {
      lds_p[threadIdx.x] = p;
      __syncthreads();
      p = lds_p[threadIdx.x ^ 1];
      __syncthreads();
    }
    const float want_lo = __builtin_fmaf(x.x, p.x, p.y), want_hi = __builtin_fmaf(x.y, p.x, p.y);
    f32x2 xv = {x.x, x.y}, pv = {p.x, p.y};
    if (CTX == 2) {   // an MFMA chain right in front (the epilogue follows the last MFMA phase)
      const float a = x.x, b = x.y;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    }
    asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "+v"(pv) : "v"(xv));
    if (pv[0] != want_lo || pv[1] != want_hi) ++nbad;
  }
  if (CTX == 2 && acc[0] == 12345.678f) nbad += 1ull << 40;   // keep the MFMAs
  if (nbad) atomicAdd(bad, nbad);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NOP, int MFMA, int SAFE>
__global__ __launch_bounds__(512) void pk_mixed_kernel(const float2* __restrict__ xs, const float2* __restrict__ ps, unsigned long long* bad,
                                                       int iters, float* sink) {
  const int wave = threadIdx.x >> 6;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (wave & 1) {   // odd waves: matrix work (two waves per SIMD: one multiplies, one runs the packed sequence)
    if (!MFMA) return;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(0.001f * (gid + i)), b[i] = (__bf16)(0.002f * (gid - i));
    for (int it = 0; it < iters * 6; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.678f) sink[0] = acc[1];
    return;
  }
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const float2 x = xs[(gid + it * 7919) & 0xfffff];
    const float2 p = ps[(gid * 3 + it) & 0xfffff];
    const float want_lo = __builtin_fmaf(x.x, p.x, p.y), want_hi = __builtin_fmaf(x.y, p.x, p.y);
    f32x2 xv = {x.x, x.y}, pv = {p.x, p.y};
    float mhi, mlo, rlo, rhi;
    const float one = 1.0f;
    // fixed registers v[20:21] = the (scale, shift) pair = the destination, exactly as in the kernel's epilogue
    if (SAFE)        // the same arithmetic into a destination pair that is NOT a source
      asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\ts_nop 1\n\t"
                   "v_pk_fma_f32 v[22:23], %6, v[20:21], v[20:21] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
                   "v_cmp_lt_f32 vcc, 0, v23\n\tv_cndmask_b32 %0, 0, %7, vcc\n\tv_cmp_lt_f32 vcc, 0, v22\n\tv_cndmask_b32 %1, 0, %7, vcc\n\t"
                   "v_mov_b32 %2, v22\n\tv_mov_b32 %3, v23"
                   : "=&v"(mhi), "=&v"(mlo), "=&v"(rlo), "=&v"(rhi) : "v"(p.x), "v"(p.y), "v"(xv), "v"(one) : "vcc", "v20", "v21", "v22", "v23");
    else if (NOP)
      asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\ts_nop 1\n\t"
                   "v_pk_fma_f32 v[20:21], %6, v[20:21], v[20:21] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"
                   "v_cmp_lt_f32 vcc, 0, v21\n\tv_cndmask_b32 %0, 0, %7, vcc\n\tv_cmp_lt_f32 vcc, 0, v20\n\tv_cndmask_b32 %1, 0, %7, vcc\n\t"
                   "v_mov_b32 %2, v20\n\tv_mov_b32 %3, v21"
                   : "=&v"(mhi), "=&v"(mlo), "=&v"(rlo), "=&v"(rhi) : "v"(p.x), "v"(p.y), "v"(xv), "v"(one) : "vcc", "v20", "v21");
    else
      asm volatile("v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\ts_nop 1\n\t"
                   "v_pk_fma_f32 v[20:21], %6, v[20:21], v[20:21] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
                   "v_cmp_lt_f32 vcc, 0, v21\n\tv_cndmask_b32 %0, 0, %7, vcc\n\tv_cmp_lt_f32 vcc, 0, v20\n\tv_cndmask_b32 %1, 0, %7, vcc\n\t"
                   "v_mov_b32 %2, v20\n\tv_mov_b32 %3, v21"
                   : "=&v"(mhi), "=&v"(mlo), "=&v"(rlo), "=&v"(rhi) : "v"(p.x), "v"(p.y), "v"(xv), "v"(one) : "vcc", "v20", "v21");
    pv[0] = rlo, pv[1] = rhi;
    const float whi = want_hi > 0.f ? 1.f : 0.f, wlo = want_lo > 0.f ? 1.f : 0.f;
    if (mhi != whi || mlo != wlo || pv[0] != want_lo || pv[1] != want_hi) ++nbad;
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20;
  const int N = 1 << 20;
  std::vector<float2> hx(N), hp(N);
  srand(7);
  for (int i = 0; i < N; ++i) {
    hx[i] = make_float2((rand() % 20001 - 10000) * 1e-3f, (rand() % 20001 - 10000) * 1e-3f);
    hp[i] = make_float2((rand() % 4001 - 2000) * 1e-3f, (rand() % 4001 - 2000) * 1e-3f);
  }
  float2 *dx, *dp;
  unsigned long long* dbad;
  hipMalloc(&dx, N * sizeof(float2));
  hipMalloc(&dp, N * sizeof(float2));
  hipMalloc(&dbad, 8 * sizeof(unsigned long long));
  float* dsink;
  hipMalloc(&dsink, 64);
  hipMemcpy(dx, hx.data(), N * sizeof(float2), hipMemcpyHostToDevice);
  hipMemcpy(dp, hp.data(), N * sizeof(float2), hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 8 * sizeof(unsigned long long));
  const int iters = 2000;
  unsigned long long ops[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < rounds; ++r) {
    hipLaunchKernelGGL(pk_kernel<0>, dim3(1024), dim3(256), 0, 0, dx, dp, dbad + 0, iters);   // 4 waves per CU-ish: one per SIMD
    ops[0] += 1024ull * 256 * iters;
    hipLaunchKernelGGL(pk_kernel<1>, dim3(8192), dim3(256), 0, 0, dx, dp, dbad + 1, iters);
    ops[1] += 8192ull * 256 * iters;
    hipLaunchKernelGGL(pk_kernel<2>, dim3(4096), dim3(256), 0, 0, dx, dp, dbad + 2, iters);
    ops[2] += 4096ull * 256 * iters;
    hipLaunchKernelGGL(pk_kernel<3>, dim3(4096), dim3(256), 0, 0, dx, dp, dbad + 3, iters);
    ops[3] += 4096ull * 256 * iters;
    hipLaunchKernelGGL((pk_mixed_kernel<0, 1, 0>), dim3(2048), dim3(512), 0, 0, dx, dp, dbad + 4, iters, dsink);
    ops[4] += 2048ull * 256 * iters;
    hipLaunchKernelGGL((pk_mixed_kernel<1, 1, 0>), dim3(2048), dim3(512), 0, 0, dx, dp, dbad + 5, iters, dsink);
    ops[5] += 2048ull * 256 * iters;
    hipLaunchKernelGGL((pk_mixed_kernel<0, 0, 0>), dim3(2048), dim3(512), 0, 0, dx, dp, dbad + 6, iters, dsink);
    ops[6] += 2048ull * 256 * iters;
    hipLaunchKernelGGL((pk_mixed_kernel<0, 1, 1>), dim3(2048), dim3(512), 0, 0, dx, dp, dbad + 7, iters, dsink);
    ops[7] += 2048ull * 256 * iters;
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
  unsigned long long hb[8];
  hipMemcpy(hb, dbad, sizeof(hb), hipMemcpyDeviceToHost);
  const char* name[8] = {"alone, low occupancy", "alone, 8 waves per SIMD", "behind an MFMA chain", "operands fresh from LDS",
                         "dst == src, MFMA waves beside", "same with s_nop 0 behind it", "dst == src, NO MFMA beside", "dst != src, MFMA waves beside"};
  unsigned long long tot = 0;
  for (int c = 0; c < 8; ++c) {
    printf("context %d (%-30s): %llu wrong lanes in %.3e executions of the instruction\n", c, name[c], hb[c], (double)ops[c]);
    tot += hb[c];
  }
  printf(tot ? "REPRODUCED\n" : "not reproduced\n");
  return 0;
}

// pk_hazard_forms.hip -- which packed-fp32 forms go wrong beside which matrix instruction?  (follow-up of pk_hazard_repro.hip,
// which showed: v_pk_fma_f32 ... op_sel:[0,0,1] op_sel_hi:[1,0,1] is exact in 2.7e11 executions on its own and wrong in
// ~1.3e-4 of them while ANOTHER wave of the same SIMD issues v_mfma_f32_32x32x16_bf16 -- whether or not its destination
// overlaps a source).  Even waves run the packed instruction on fixed registers, odd waves a matrix loop.
// Build: hipcc -O2 --offload-arch=gfx950 tools/pk_hazard_forms.hip -o tools/bin/pk_hazard_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// FORM 0: fma, op_sel:[0,0,1] op_sel_hi:[1,0,1]   (lo = x.lo p.lo + p.hi ; hi = x.hi p.lo + p.hi)
// FORM 1: fma, no modifiers                        (lo = x.lo p.lo + q.lo ; hi = x.hi p.hi + q.hi)
// FORM 2: mul, no modifiers                        FORM 3: add, no modifiers
// FORM 4: fma, op_sel_hi:[1,0,1] only              (lo = x.lo p.lo + q.lo ; hi = x.hi p.lo + q.hi)
// FORM 5: FORM 0's modifiers on DISTINCT pairs     (lo = x.lo p.lo + q.hi ; hi = x.hi p.lo + q.hi)
// FORM 6: src1 == src2 pair, no modifiers          (lo = x.lo p.lo + p.lo ; hi = x.hi p.hi + p.hi)
// FORM 7: op_sel:[0,0,1] only, distinct pairs      (lo = x.lo p.lo + q.hi ; hi = x.hi p.hi + q.hi)
// FORM 8: v_pk_mul_f32 d, p, p op_sel:[0,1] op_sel_hi:[1,0]   (lo = p.lo p.hi ; hi = p.hi p.lo)   -- the form hipcc emits in
// FORM 9: v_pk_add_f32 d, p, p op_sel:[0,1] op_sel_hi:[1,0]   (lo = p.lo + p.hi ; hi = p.hi + p.lo)  conv2d / radar / elementwise_cn8
// MT 0: no matrix work beside; 1: v_mfma_f32_32x32x16_bf16; 2: v_mfma_f32_32x32x2_f32; 3: v_mfma_f32_16x16x32_bf16
template <int FORM, int MT>
__global__ __launch_bounds__(512) void forms_kernel(const float2* __restrict__ xs, const float2* __restrict__ ps, const float2* __restrict__ qs,
                                                    unsigned long long* bad, int iters, float* sink) {
  const int wave = threadIdx.x >> 6;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (wave & 1) {
    if (MT == 0) return;
    if (MT == 1) {
      f32x16 acc;
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) a[i] = (__bf16)(0.001f * (gid + i)), b[i] = (__bf16)(0.002f * (gid - i));
      for (int it = 0; it < iters * 6; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      if (acc[0] == 12345.678f) sink[0] = acc[1];
    } else if (MT == 2) {
      f32x16 acc;
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float a = 0.001f * gid, b = 0.002f * gid;
      for (int it = 0; it < iters * 3; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      if (acc[0] == 12345.678f) sink[0] = acc[1];
    } else {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) a[i] = (__bf16)(0.001f * (gid + i)), b[i] = (__bf16)(0.002f * (gid - i));
      for (int it = 0; it < iters * 12; ++it) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
      if (acc[0] == 12345.678f) sink[0] = acc[1];
    }
    return;
  }
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const float2 x = xs[(gid + it * 7919) & 0xfffff];
    const float2 p = ps[(gid * 3 + it) & 0xfffff];
    const float2 q = qs[(gid * 5 + it * 3) & 0xfffff];
    float wlo, whi;
    if (FORM == 0) wlo = __builtin_fmaf(x.x, p.x, p.y), whi = __builtin_fmaf(x.y, p.x, p.y);
    else if (FORM == 1) wlo = __builtin_fmaf(x.x, p.x, q.x), whi = __builtin_fmaf(x.y, p.y, q.y);
    else if (FORM == 2) wlo = x.x * p.x, whi = x.y * p.y;
    else if (FORM == 3) wlo = x.x + p.x, whi = x.y + p.y;
    else if (FORM == 4) wlo = __builtin_fmaf(x.x, p.x, q.x), whi = __builtin_fmaf(x.y, p.x, q.y);
    else if (FORM == 5) wlo = __builtin_fmaf(x.x, p.x, q.y), whi = __builtin_fmaf(x.y, p.x, q.y);
    else if (FORM == 6) wlo = __builtin_fmaf(x.x, p.x, p.x), whi = __builtin_fmaf(x.y, p.y, p.y);
    else if (FORM == 7) wlo = __builtin_fmaf(x.x, p.x, q.y), whi = __builtin_fmaf(x.y, p.y, q.y);
    else if (FORM == 8) wlo = p.x * p.y, whi = p.y * p.x;
    else wlo = p.x + p.y, whi = p.y + p.x;
    float rlo, rhi;
#define PRE "v_mov_b32 v20, %2\n\tv_mov_b32 v21, %3\n\tv_mov_b32 v24, %4\n\tv_mov_b32 v25, %5\n\tv_mov_b32 v26, %6\n\tv_mov_b32 v27, %7\n\ts_nop 4\n\t"
#define POST "\n\ts_nop 4\n\tv_mov_b32 %0, v22\n\tv_mov_b32 %1, v23"
#define OPS : "=&v"(rlo), "=&v"(rhi) : "v"(p.x), "v"(p.y), "v"(x.x), "v"(x.y), "v"(q.x), "v"(q.y) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27"
    if (FORM == 0) asm volatile(PRE "v_pk_fma_f32 v[22:23], v[24:25], v[20:21], v[20:21] op_sel:[0,0,1] op_sel_hi:[1,0,1]" POST OPS);
    else if (FORM == 1) asm volatile(PRE "v_pk_fma_f32 v[22:23], v[24:25], v[20:21], v[26:27]" POST OPS);
    else if (FORM == 2) asm volatile(PRE "v_pk_mul_f32 v[22:23], v[24:25], v[20:21]" POST OPS);
    else if (FORM == 3) asm volatile(PRE "v_pk_add_f32 v[22:23], v[24:25], v[20:21]" POST OPS);
    else if (FORM == 4) asm volatile(PRE "v_pk_fma_f32 v[22:23], v[24:25], v[20:21], v[26:27] op_sel_hi:[1,0,1]" POST OPS);
    else if (FORM == 5) asm volatile(PRE "v_pk_fma_f32 v[22:23], v[24:25], v[20:21], v[26:27] op_sel:[0,0,1] op_sel_hi:[1,0,1]" POST OPS);
    else if (FORM == 6) asm volatile(PRE "v_pk_fma_f32 v[22:23], v[24:25], v[20:21], v[20:21]" POST OPS);
    else if (FORM == 7) asm volatile(PRE "v_pk_fma_f32 v[22:23], v[24:25], v[20:21], v[26:27] op_sel:[0,0,1]" POST OPS);
    else if (FORM == 8) asm volatile(PRE "v_pk_mul_f32 v[22:23], v[20:21], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]" POST OPS);
    else asm volatile(PRE "v_pk_add_f32 v[22:23], v[20:21], v[20:21] op_sel:[0,1] op_sel_hi:[1,0]" POST OPS);
    if (rlo != wlo || rhi != whi) ++nbad;
  }
  if (nbad) atomicAdd(bad, nbad);
}

template <int FORM, int MT>
void run(const float2* dx, const float2* dp, const float2* dq, unsigned long long* dbad, float* dsink, int rounds, const char* fname, const char* mname) {
  hipMemset(dbad, 0, 8);
  const int iters = 2000;
  for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL((forms_kernel<FORM, MT>), dim3(2048), dim3(512), 0, 0, dx, dp, dq, dbad, iters, dsink);
  hipDeviceSynchronize();
  unsigned long long hb = 0;
  hipMemcpy(&hb, dbad, 8, hipMemcpyDeviceToHost);
  printf("%-44s beside %-26s: %9llu wrong of %.2e\n", fname, mname, hb, (double)rounds * 2048 * 256 * iters);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  const int N = 1 << 20;
  std::vector<float2> hx(N), hp(N), hq(N);
  srand(7);
  for (int i = 0; i < N; ++i) {
    hx[i] = make_float2((rand() % 20001 - 10000) * 1e-3f, (rand() % 20001 - 10000) * 1e-3f);
    hp[i] = make_float2((rand() % 4001 - 2000) * 1e-3f, (rand() % 4001 - 2000) * 1e-3f);
    hq[i] = make_float2((rand() % 4001 - 2000) * 1e-3f, (rand() % 4001 - 2000) * 1e-3f);
  }
  float2 *dx, *dp, *dq;
  unsigned long long* dbad;
  float* dsink;
  hipMalloc(&dx, N * sizeof(float2)); hipMalloc(&dp, N * sizeof(float2)); hipMalloc(&dq, N * sizeof(float2));
  hipMalloc(&dbad, 64); hipMalloc(&dsink, 64);
  hipMemcpy(dx, hx.data(), N * sizeof(float2), hipMemcpyHostToDevice);
  hipMemcpy(dp, hp.data(), N * sizeof(float2), hipMemcpyHostToDevice);
  hipMemcpy(dq, hq.data(), N * sizeof(float2), hipMemcpyHostToDevice);
  const char* fn[10] = {"v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,0,1]", "v_pk_fma_f32 (no modifiers)", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32 op_sel_hi:[1,0,1]",
                       "same modifiers as row 1, distinct pairs", "v_pk_fma_f32 src1 == src2, no modifiers", "v_pk_fma_f32 op_sel:[0,0,1], distinct pairs",
                        "v_pk_mul_f32 d,p,p op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 d,p,p op_sel:[0,1] op_sel_hi:[1,0]"};
  const char* mn[4] = {"nothing", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x32_bf16"};
#define RUN(F, M) run<F, M>(dx, dp, dq, dbad, dsink, rounds, fn[F], mn[M])
  RUN(0, 0); RUN(0, 1); RUN(0, 2); RUN(0, 3);
  RUN(1, 0); RUN(1, 1); RUN(1, 2); RUN(1, 3);
  RUN(2, 1); RUN(2, 2); RUN(3, 1); RUN(3, 2);
  RUN(4, 1); RUN(4, 2);
  RUN(5, 1); RUN(5, 3); RUN(6, 1); RUN(6, 3); RUN(7, 1); RUN(7, 3);
  RUN(8, 0); RUN(8, 1); RUN(8, 2); RUN(8, 3); RUN(9, 0); RUN(9, 1); RUN(9, 2); RUN(9, 3);
  return 0;
}

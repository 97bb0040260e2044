// mfma_fill.hip -- how much vector / LDS issue hides under one v_mfma_f32_32x32x2_f32 (64 cyc/SIMD) on gfx950?
// One wave per SIMD (256 blocks x 256 threads) and two; hand-placed fillers between the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a))
#define DSR(x) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(addr))
#define DSR2(x) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(x) : "v"(addr))

template <int NV, int ND>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  __shared__ float s[1024];
  s[threadIdx.x] = 1.f; s[threadIdx.x + 256] = 1.f;
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float d[4] = {0, 0, 0, 0};
  int addr = (threadIdx.x & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      MFMA(acc[j]);
#pragma unroll
      for (int q = 0; q < NV; ++q) VADD(f[q & 7]);
#pragma unroll
      for (int q = 0; q < ND; ++q) DSR(d[q & 3]);
    }
    if (ND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float r = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int q = 0; q < 16; ++q) r += acc[j][q];
  for (int q = 0; q < 8; ++q) r += f[q];
  for (int q = 0; q < 4; ++q) r += d[q];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NV, int ND>
void run(int blocks, int iters) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV, ND><<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<NV, ND><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double n_mfma = (double)iters * 4;                 // per wave
  double waves_per_simd = blocks / 256.0;
  double cyc = best * 1e-3 * 2.4e9 / (n_mfma * waves_per_simd);
  printf("v_add/mfma %2d  ds_read/mfma %d  waves/SIMD %.0f : %7.2f TFLOP/s  %6.1f cyc/MFMA (at 2.4 GHz)\n", NV, ND,
         waves_per_simd, (double)blocks * 4 * n_mfma * 4096.0 / best / 1e9, cyc);
  hipFree(out);
}

int main() {
  const int it = 20000;
  run<0, 0>(256, it); run<1, 0>(256, it); run<2, 0>(256, it); run<4, 0>(256, it); run<8, 0>(256, it); run<12, 0>(256, it); run<16, 0>(256, it);
  run<0, 1>(256, it); run<0, 2>(256, it); run<0, 4>(256, it); run<4, 1>(256, it); run<4, 2>(256, it);
  run<4, 0>(512, it); run<8, 0>(512, it); run<0, 1>(512, it); run<4, 1>(512, it); run<4, 2>(512, it);
  return 0;
}

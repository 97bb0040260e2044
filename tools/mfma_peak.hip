// mfma_peak.hip -- what fp32 MFMA rate does this box sustain?  (ceiling for every conv GEMM in this repo)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
// Variants: pure v_mfma_f32_32x32x2_f32 chains (4 independent accumulators per wave), the same with one
// ds_read_b32 per MFMA, at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, int iters) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = (float)i * 1e-6f;
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  int o = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float a0 = a, a1 = a + 1.f, b0 = b, b1 = b + 1.f;
      if (LDS) {
        a0 = s[(o + u * 64) & 4095]; a1 = s[(o + u * 64 + 1024) & 4095];
        b0 = s[(o + u * 64 + 2048) & 4095]; b1 = s[(o + u * 64 + 3072) & 4095];
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
    o += 7;
  }
  float r = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int q = 0; q < 16; ++q) r += acc[j][q];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// software-pipelined variant: the operands of group u+1 are read while the MFMAs of group u issue
template <int SCHED>
__global__ __launch_bounds__(256, 2) void mfma_pipe(float* out, int iters) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = (float)i * 1e-6f;
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  int o = threadIdx.x;
  float a0 = s[o & 4095], a1 = s[(o + 1024) & 4095], b0 = s[(o + 2048) & 4095], b1 = s[(o + 3072) & 4095];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int on = o + (u + 1) * 64;
      const float na0 = s[on & 4095], na1 = s[(on + 1024) & 4095], nb0 = s[(on + 2048) & 4095], nb1 = s[(on + 3072) & 4095];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
      if (SCHED) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        }
      }
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    o += 7;
  }
  float r = a0 + a1 + b0 + b1;
  for (int j = 0; j < 4; ++j)
    for (int q = 0; q < 16; ++q) r += acc[j][q];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename K>
void run_k(const char* name, K kern, int blocks, int iters) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flops = (double)blocks * 4 * iters * 32 * 4096.0;
  printf("%-34s blocks %5d  %8.3f ms  %7.2f TFLOP/s\n", name, blocks, best, flops / best / 1e9);
  hipFree(out);
}

template <int LDS>
void run(const char* name, int blocks, int iters) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<LDS><<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    mfma_loop<LDS><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flops = (double)blocks * 4 * iters * 32 * 4096.0;
  printf("%-34s blocks %5d  %8.3f ms  %7.2f TFLOP/s\n", name, blocks, best, flops / best / 1e9);
  hipFree(out);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s  CUs %d  clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int cu = p.multiProcessorCount;
  run<0>("pure mfma, 1 wave/SIMD", cu, 4000);
  run<0>("pure mfma, 2 waves/SIMD", cu * 2, 4000);
  run<0>("pure mfma, 2 waves/SIMD, 8 rounds", cu * 16, 1000);
  run<1>("mfma + 1 ds_read/mfma, 1 wave/SIMD", cu, 4000);
  run<1>("mfma + 1 ds_read/mfma, 2 waves/SIMD", cu * 2, 4000);
  run_k("pipelined ds_read, 1 wave/SIMD", mfma_pipe<0>, cu, 4000);
  run_k("pipelined ds_read, 2 waves/SIMD", mfma_pipe<0>, cu * 2, 4000);
  run_k("pipelined+sched_group, 1 wave/SIMD", mfma_pipe<1>, cu, 4000);
  run_k("pipelined+sched_group, 2 waves/SIMD", mfma_pipe<1>, cu * 2, 4000);
  run<0>("pure mfma, long (power steady state)", cu * 2, 40000);
  return 0;
}

#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// kernels with a controlled number of live accumulators (each f32x16 = 16 registers)
template <int NACC>
__global__ __launch_bounds__(256) void probe(const float* a, float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x = a[threadIdx.x], y = a[threadIdx.x + 256];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void report(const char* name) {
  hipFuncAttributes at; hipFuncGetAttributes(&at, (const void*)probe<NACC>);
  int n0 = -1, n32 = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n0, probe<NACC>, 256, 0);
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n32, probe<NACC>, 256, 32 * 1024);
  // measure: time a grid of 256*k blocks; time per block-slot shows residency
  float* a; float* out; hipMalloc(&a, 4096); hipMemset(a, 0, 4096); hipMalloc(&out, 256 * 16 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[5];
  int blocks[5] = {256, 512, 768, 1024, 2048};
  for (int j = 0; j < 5; ++j) {
    hipLaunchKernelGGL(probe<NACC>, dim3(blocks[j]), dim3(256), 0, 0, a, out, 2000);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(probe<NACC>, dim3(blocks[j]), dim3(256), 0, 0, a, out, 2000); hipEventRecord(e1);
    hipEventSynchronize(e1); hipEventElapsedTime(&ms[j], e0, e1);
  }
  printf("%s numRegs=%d  api(lds0)=%d api(lds32K)=%d  ms@256/512/768/1024/2048 blocks: %.2f %.2f %.2f %.2f %.2f\n", name, at.numRegs, n0, n32, ms[0], ms[1], ms[2], ms[3], ms[4]);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("regsPerBlock %d regsPerMultiprocessor %d sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu maxThreadsPerMP %d\n", p.regsPerBlock, p.regsPerMultiprocessor, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.maxThreadsPerMultiProcessor);
  report<2>("acc2 "); report<4>("acc4 "); report<6>("acc6 "); report<7>("acc7 "); report<8>("acc8 "); report<10>("acc10"); report<12>("acc12"); report<14>("acc14"); report<15>("acc15");
  return 0;
}

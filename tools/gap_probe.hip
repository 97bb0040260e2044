// gap_probe: what makes back-to-back DEPENDENT launches on one stream cost ~6 us on this box?  Each variant launches the same
// tiny amount of work N times in a row; (total time) / N is the per-launch cost.   hipcc --offload-arch=gfx950 -O3 tools/gap_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
struct Big { float v[96]; float* p; };                      // 392-byte kernarg
__global__ void k_plain(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_bigarg(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] += b.v[3]; }
__global__ __launch_bounds__(256) void k_lds(float* p) {
  __shared__ float s[12 * 1024];                             // 48 KB
  s[threadIdx.x * 48] = p[0];
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = s[48] + 1.f;
}
__global__ __launch_bounds__(256) void k_vgpr(float* p, int n) {
  float a[160];
#pragma unroll
  for (int i = 0; i < 160; ++i) a[i] = p[(i * 7 + threadIdx.x) & 1023];
  for (int j = 0; j < n; ++j)
#pragma unroll
    for (int i = 0; i < 160; ++i) a[i] = a[i] * 1.0001f + a[(i + 1) % 160];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 160; ++i) s += a[i];
  if (s == 123.456f) p[0] = s;
}
__global__ void k_write(float4* q, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) q[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_read(const float4* q, size_t n, float* p) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += q[i].x;
  if (s == 123.456f) p[0] = s;
}
template <class F> double run(const char* name, int N, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < N; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %7.2f us per launch\n", name, ms * 1e3 / N);
  return ms * 1e3 / N;
}
int main() {
  float* p; hipMalloc(&p, 1 << 20); hipMemset(p, 0, 1 << 20);
  const size_t nq = (size_t)128 << 20 >> 4;                   // 128 MB
  float4* q; hipMalloc(&q, nq * 16);
  Big b; for (int i = 0; i < 96; ++i) b.v[i] = 0.f; b.p = p;
  const int N = 400;
  run("plain, 1024 workgroups", N, [&] { hipLaunchKernelGGL(k_plain, dim3(1024), dim3(256), 0, 0, p); });
  run("392-byte kernarg", N, [&] { hipLaunchKernelGGL(k_bigarg, dim3(1024), dim3(256), 0, 0, b); });
  run("48 KB static LDS", N, [&] { hipLaunchKernelGGL(k_lds, dim3(1024), dim3(256), 0, 0, p); });
  run("160 VGPRs", N, [&] { hipLaunchKernelGGL(k_vgpr, dim3(1024), dim3(256), 0, 0, p, 0); });
  run("writes 128 MB", 100, [&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, q, nq); });
  run("reads 128 MB", 100, [&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, q, nq, p); });
  run("writes 128 MB then plain", 100, [&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, q, nq); hipLaunchKernelGGL(k_plain, dim3(1024), dim3(256), 0, 0, p); });
  run("reads 128 MB then plain", 100, [&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, q, nq, p); hipLaunchKernelGGL(k_plain, dim3(1024), dim3(256), 0, 0, p); });
  run("48 KB LDS then plain", N, [&] { hipLaunchKernelGGL(k_lds, dim3(1024), dim3(256), 0, 0, p); hipLaunchKernelGGL(k_plain, dim3(1024), dim3(256), 0, 0, p); });
  run("160 VGPRs then plain", N, [&] { hipLaunchKernelGGL(k_vgpr, dim3(1024), dim3(256), 0, 0, p, 0); hipLaunchKernelGGL(k_plain, dim3(1024), dim3(256), 0, 0, p); });
  return 0;
}

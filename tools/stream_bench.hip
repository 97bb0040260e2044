// stream_bench.hip -- which loop shape streams a [C][n] fp32 row kernel (out = k1[c]*a + k2[c]*b + k3[c]) fastest?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int TPB = 256;

// A: the shape used by csrc/elementwise.hip today: grid-stride loop, one float4 pair in flight per iteration
__global__ __launch_bounds__(TPB) void kA(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ k1,
                                          float* out, int64_t n) {
  const int c = blockIdx.y;
  const int64_t base = (int64_t)c * n;
  const float a1 = k1[c], a2 = k1[c] * 0.5f, a3 = 0.25f;
  for (int64_t i = ((int64_t)blockIdx.x * TPB + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * TPB * 4) {
    const float4 x = *reinterpret_cast<const float4*>(a + base + i), y = *reinterpret_cast<const float4*>(b + base + i);
    float4 o;
    o.x = fmaf(a1, x.x, fmaf(a2, y.x, a3)); o.y = fmaf(a1, x.y, fmaf(a2, y.y, a3));
    o.z = fmaf(a1, x.z, fmaf(a2, y.z, a3)); o.w = fmaf(a1, x.w, fmaf(a2, y.w, a3));
    *reinterpret_cast<float4*>(out + base + i) = o;
  }
}

// B: U chunks per thread, all loads issued before the first use; optional nontemporal stores
template <int U, int NT>
__global__ __launch_bounds__(TPB) void kB(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ k1,
                                          float* out, int64_t n) {
  const int c = blockIdx.y;
  const int64_t base = (int64_t)c * n;
  const float a1 = k1[c], a2 = k1[c] * 0.5f, a3 = 0.25f;
  const int64_t i0 = ((int64_t)blockIdx.x * U * TPB + threadIdx.x) * 4;
  float4 x[U], y[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = i0 + (int64_t)u * TPB * 4;
    if (i < n) {
      x[u] = *reinterpret_cast<const float4*>(a + base + i);
      y[u] = *reinterpret_cast<const float4*>(b + base + i);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = i0 + (int64_t)u * TPB * 4;
    if (i < n) {
      float4 o;
      o.x = fmaf(a1, x[u].x, fmaf(a2, y[u].x, a3)); o.y = fmaf(a1, x[u].y, fmaf(a2, y[u].y, a3));
      o.z = fmaf(a1, x[u].z, fmaf(a2, y[u].z, a3)); o.w = fmaf(a1, x[u].w, fmaf(a2, y[u].w, a3));
      if (NT) {
        __builtin_nontemporal_store(o.x, out + base + i); __builtin_nontemporal_store(o.y, out + base + i + 1);
        __builtin_nontemporal_store(o.z, out + base + i + 2); __builtin_nontemporal_store(o.w, out + base + i + 3);
      } else {
        *reinterpret_cast<float4*>(out + base + i) = o;
      }
    }
  }
}

template <typename F>
void run(const char* name, F launch, double bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-44s %8.3f ms  %6.2f TB/s\n", name, best, bytes / best / 1e9);
}

int main() {
  const int C = 64; const int64_t n = 960000;
  float *a, *b, *o, *k;
  hipMalloc(&a, C * n * 4); hipMalloc(&b, C * n * 4); hipMalloc(&o, C * n * 4); hipMalloc(&k, C * 4);
  hipMemset(a, 0, C * n * 4); hipMemset(b, 0, C * n * 4); hipMemset(k, 0, C * 4);
  const double bytes = 3.0 * C * n * 4;
  for (int per : {4, 8, 16}) {
    char nm[64]; snprintf(nm, 64, "A grid-stride loop, ~%d iterations/thread", per);
    int gx = (int)((n + (int64_t)TPB * 4 * per - 1) / ((int64_t)TPB * 4 * per));
    run(nm, [&] { kA<<<dim3(gx, C), TPB>>>(a, b, k, o, n); }, bytes);
  }
  auto gb = [&](int U) { return (int)((n + (int64_t)TPB * 4 * U - 1) / ((int64_t)TPB * 4 * U)); };
  run("B 1 chunk / thread", [&] { kB<1, 0><<<dim3(gb(1), C), TPB>>>(a, b, k, o, n); }, bytes);
  run("B 2 chunks / thread, loads first", [&] { kB<2, 0><<<dim3(gb(2), C), TPB>>>(a, b, k, o, n); }, bytes);
  run("B 4 chunks / thread, loads first", [&] { kB<4, 0><<<dim3(gb(4), C), TPB>>>(a, b, k, o, n); }, bytes);
  run("B 8 chunks / thread, loads first", [&] { kB<8, 0><<<dim3(gb(8), C), TPB>>>(a, b, k, o, n); }, bytes);
  run("B 4 chunks / thread, nontemporal stores", [&] { kB<4, 1><<<dim3(gb(4), C), TPB>>>(a, b, k, o, n); }, bytes);
  return 0;
}

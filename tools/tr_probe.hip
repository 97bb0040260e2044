// Probe of ds_read_b64_tr_b16 (cdna_hip_programming.md T10) as the CN8 weight-gradient kernels use it: per 16-lane group a
// block of 4 rows x 16 columns of 16-bit elements, lane 4q+p supplies the address of row q / columns 4p..4p+3, lane i
// receives column i of the 4 rows.  Build: hipcc --offload-arch=gfx950 tools/tr_probe.hip -o tools/bin/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[64 * 16];
  for (int i = threadIdx.x; i < 64 * 16; i += 64) img[i] = (unsigned short)i;   // element (row r, col c) = 16 r + c
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, q = i >> 2, p = i & 3, g = l >> 4;
  const unsigned short* a = img + (4 * g + q) * 16 + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  int* d;
  hipMalloc(&d, 256 * sizeof(int));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int expect = (4 * (l >> 4) + j) * 16 + (l & 15);   // row 4g + j, column i
      if (h[l * 4 + j] != expect) ++bad;
    }
  printf("tr_probe: %d mismatches; lane 0: %d %d %d %d, lane 17: %d %d %d %d\n", bad, h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71]);
  return bad != 0;
}

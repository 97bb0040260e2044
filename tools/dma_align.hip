// dma_align.hip -- what does buffer_load_dwordx4 ... lds accept on gfx950?  (design input of conv_graph_split2_kernel, round 6)
//   (1) a global address that is only 4-byte aligned (row starts of the CN layout are multiples of 100 bytes)
//   (2) EXEC-masked lanes: untouched LDS?  (3) a lane whose 16 bytes straddle num_records: per-dword zero fill or whole-lane?
//   hipcc --offload-arch=gfx950 -O3 tools/dma_align.hip -o tools/bin/dma_align && tools/bin/dma_align
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const float* p, float* out, int shift, int nrec_bytes, int nlanes, int size16) {
  __shared__ float s[512];
  for (int i = threadIdx.x; i < 512; i += 64) s[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(p + shift), 0, nrec_bytes, 0x00020000);
  if ((int)threadIdx.x < nlanes) {
    if (size16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)s, 16, threadIdx.x * 16, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)s, 4, threadIdx.x * 4, 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main() {
  float h[2048]; for (int i = 0; i < 2048; ++i) h[i] = i + 1;
  float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 512 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  float r[512];
  for (int size16 = 1; size16 >= 0; --size16)
    for (int shift = 0; shift < 4; ++shift) {
      k<<<1, 64>>>(d, o, shift, 1 << 20, 64, size16);
      hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
      int n = size16 ? 256 : 64, bad = 0;
      for (int i = 0; i < n; ++i) bad += r[i] != h[shift + i];
      printf("size %2d, global address shifted by %d floats: %d of %d values wrong (first: got %.0f want %.0f), beyond: %.0f\n", size16 ? 16 : 4, shift, bad, n, r[0], h[shift], r[n]);
    }
  k<<<1, 64>>>(d, o, 1, 1 << 20, 40, 1);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("EXEC mask: 40 of 64 lanes active: lds[159] = %.0f (want 161), lds[160] = %.0f (untouched = -1)\n", r[159], r[160]);
  k<<<1, 64>>>(d, o, 1, 1000, 64, 1);      // 250 floats in range: lane 62 holds floats 248..251, lane 63 252..255
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("num_records 1000 B: floats 246..255 = %.0f %.0f | %.0f %.0f %.0f %.0f | %.0f %.0f %.0f %.0f  (want 248 249 | 250 251 then per-dword zeros or whole-lane zeros)\n",
         r[246], r[247], r[248], r[249], r[250], r[251], r[252], r[253], r[254], r[255]);
  return 0;
}

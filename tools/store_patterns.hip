// store_patterns.hip -- how fast does the chip WRITE a [64 rows][N cols] fp32 matrix, as a function of the store pattern of the conv
// kernels' epilogue?  (round 6: the epilogue of the graph kernels costs 39 % of a 64-channel launch, tools/g2_ablate.sh)
//   A  the MFMA C layout as epilogue_b stores it: one buffer_store_dword per accumulator register -- lanes 0-31 = 32 consecutive
//      columns of row r (128 B), lanes 32-63 = row r + 4 -- tile of 250 columns (row segments start at multiples of 1000 B)
//   B  the same with tiles of 256 columns (segments 128-byte aligned)
//   C  dwordx4 per lane: a wave instruction writes 4 rows x 256 B contiguous (what an LDS-transposed epilogue would issue), 250-column tiles
//   D  as C with 256-column tiles
// Persistent grid of 512 workgroups x 256 threads walking tiles (tile = 64 rows x 250 / 256 columns), like conv_graph_split2_kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/store_patterns.hip -o tools/bin/store_patterns && tools/bin/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PAT, int TW>
__global__ __launch_bounds__(256) void k(float* out, int64_t ld, int ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t col0 = (int64_t)tile * TW;
    if (PAT == 0) {   // C layout: wave owns columns [wave * 64, +64) x 64 rows: 2 x 2 blocks of 32 x 32, 16 registers each
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int ns = 0; ns < 2; ++ns) {
            const int c = wave * 64 + ns * 32 + l31;
            const int row = ms * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (c < TW) out[row * ld + col0 + c] = (float)(r + c);
          }
    } else {          // transposed: wave owns 16 rows x all columns? keep the wave's 64-column strip: instruction = 4 rows x 64 columns
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int c = wave * 64 + (lane & 15) * 4;
        if (c + 3 < TW) *reinterpret_cast<float4*>(out + row * ld + col0 + c) = make_float4(c, c + 1.f, c + 2.f, c + 3.f);
        else if (c < TW) { out[row * ld + col0 + c] = c; if (c + 1 < TW) out[row * ld + col0 + c + 1] = c; }
      }
    }
  }
}

template <int PAT, int TW>
void run(const char* name, float* out, int64_t ld, int64_t n) {
  const int ntiles = (int)(n / TW);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<PAT, TW><<<512, 256>>>(out, ld, ntiles); hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0); k<PAT, TW><<<512, 256>>>(out, ld, ntiles); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%-64s %8.1f us  %6.0f GB/s\n", name, best * 1e3, 64.0 * ntiles * TW * 4 / best / 1e6);
}

int main() {
  const int64_t n = 960000, ld = 960000 + 256;
  float* out; hipMalloc(&out, 64 * ld * 4 + 4096);
  run<0, 250>("A dword, C layout, 250-column tiles (1000-byte starts)", out, ld, n);
  run<0, 256>("B dword, C layout, 256-column tiles (aligned)", out, ld, n);
  run<1, 250>("C dwordx4 rows, 250-column tiles", out + 2, ld, n);     // (+2 floats: 8-byte aligned like odd tiles of 250)
  run<1, 256>("D dwordx4 rows, 256-column tiles (aligned)", out, ld, n);
  return 0;
}

// buffer_range.hip -- which offsets does the gfx950 raw-buffer range check see?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* p, float* out, int nrec, unsigned vo, int so) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nrec, 0x00020000);
  out[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
  out[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo + 64, so, 0));
}
int main() {
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  struct { int nrec; unsigned vo; int so; const char* what; } c[] = {
    {400, 40, 0, "in range: expect 11"},
    {400, 400, 0, "voffset == num_records: expect 0"},
    {400, 40, 400, "soffset beyond num_records, voffset inside"},
    {400, 40, 2000, "soffset far beyond"},
    {400, 0xFFFFFFF0u, 0, "negative voffset (wraps)"},
    {400, 0xFFFFFFF0u, 64, "negative voffset + soffset 64 -> byte 48 if wrapped before the check"},
    {400, 396, 0, "last dword"},
    {400, 398, 0, "straddles the end"},
  };
  for (auto& t : c) {
    k<<<1, 1>>>(d, o, t.nrec, t.vo, t.so);
    float r[2]; hipMemcpy(r, o, 8, hipMemcpyDeviceToHost);
    printf("nrec %4d voffset %10u soffset %5d -> %6.0f (+64: %6.0f)   %s\n", t.nrec, t.vo, t.so, r[0], r[1], t.what);
  }
  return 0;
}

// radar.hip -- VirtualRadar forward (reference layers/virtual_radar.py:79-134) on gfx950.
//
// Kernel 1 (sar_vr_signal_f32, :93-123): one wave per 64 consecutive frames of one clip.  The
//   (3, 64, V*M) slab of joint coordinates is staged once in LDS with coalesced loads (odd row
//   stride -> conflict-free per-frame reads); each lane then walks the E edges x M bodies of its
//   frame entirely in registers: range, aspect angles, ellipsoid RCS, phase, complex sum.  The dozens
//   of full-tensor temporaries the reference materialises (each a HBM round trip) never exist.
//   Numerics: built with -ffp-contract=off, IEEE-correct sqrtf and '/' (NOT __fsqrt_rn, which lowers to the
//   1-ulp v_sqrt_f32); range and phase follow the oracle's operation order bit for bit (psi ~ 1e5 rad at lambda = 5e-4, so one ulp of range is 0.02 rad of phase); sin/cos
//   are the accurate ocml routines (Payne-Hanek reduction), never the fast-math approximations.
// Kernel 2 (sar_stft_logmag_f32, :124-133 + nnAudio 0.1.1 STFT): one workgroup per (clip, output
//   frame): reflect-padded, Hann-windowed complex frame and the n_fft-entry twiddle table live in
//   LDS; thread k accumulates bin k of the length-n_fft DFT, then log(|Z|+1e-6) is written to row
//   (k + n_fft/2) % n_fft (the reference's roll) -- and only for the frames the nearest-neighbour
//   F.interpolate of models/resnet.py:26 actually consumes when out_cols > 0.
#include "sar_common.h"

namespace {

constexpr int FRAMES = 64;

__global__ __launch_bounds__(FRAMES) void vr_signal_kernel(const float* __restrict__ x, int T, int V, int M,
                                                           const int* __restrict__ e_src, const int* __restrict__ e_dst,
                                                           int E, const float* __restrict__ loc_p,
                                                           const float* __restrict__ lam_p, float* __restrict__ z_re,
                                                           float* __restrict__ z_im) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int VM = V * M;
  const int RS = VM | 1;                     // odd row stride
  float* xs = smem;                          // [3][FRAMES][RS]
  int* es = (int*)(xs + 3 * FRAMES * RS);    // [E] src joints
  int* ed = es + E;                          // [E] dst joints
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * FRAMES;
  const int nt = min(FRAMES, T - t0);
  for (int c = 0; c < 3; ++c) {
    const float* g = x + (((int64_t)b * 3 + c) * T + t0) * VM;   // nt*VM contiguous floats
    for (int i = threadIdx.x; i < nt * VM; i += FRAMES) {
      const int tt = i / VM;
      xs[(c * FRAMES + tt) * RS + (i - tt * VM)] = g[i];
    }
  }
  for (int i = threadIdx.x; i < E; i += FRAMES) {
    es[i] = e_src[i];
    ed[i] = e_dst[i];
  }
  __syncthreads();
  const int tt = threadIdx.x;
  if (tt >= nt) return;
  const float lx = loc_p[0], ly = loc_p[1], lz = loc_p[2];
  const float lam = lam_p[0];
  const float* X0 = xs + (0 * FRAMES + tt) * RS;
  const float* X1 = xs + (1 * FRAMES + tt) * RS;
  const float* X2 = xs + (2 * FRAMES + tt) * RS;
  const float PI_F = 3.14159274101257324f;        // float32(np.pi)
  const float FOURPI_F = 12.5663706143591725f;    // rounds to float32(4*np.pi)
  float zr = 0.f, zi = 0.f;
  // c[m] = (mean_e |S - D|)^2, layers/virtual_radar.py:110-113 (mean over the edge axis)
  float cm[4];
  for (int m = 0; m < M; ++m) {
    float acc = 0.f;
    for (int e = 0; e < E; ++e) {
      const int js = es[e] * M + m, jd = ed[e] * M + m;
      const float dx = X0[js] - X0[jd], dy = X1[js] - X1[jd], dz = X2[js] - X2[jd];
      acc = acc + sqrtf((dx * dx + dy * dy) + dz * dz);
    }
    const float c = acc / (float)E;
    cm[m] = c * c;
  }
  for (int e = 0; e < E; ++e) {
    for (int m = 0; m < M; ++m) {
      const int js = es[e] * M + m, jd = ed[e] * M + m;
      const float sx = X0[js], sy = X1[js], sz = X2[js];
      const float dx = X0[jd], dy = X1[jd], dz = X2[jd];
      const float rx = fabsf(sx - lx), ry = fabsf(sy - ly), rz = fabsf(sz - lz);
      const float rxy2 = rx * rx + ry * ry;
      const float dist = sqrtf(rxy2 + rz * rz);
      const float ax = lx - ((sx + dx) / (2.f)), ay = ly - ((sy + dy) / (2.f)), az = lz - ((sz + dz) / (2.f));
      const float bx = dx - sx, by = dy - sy, bz = dz - sz;
      const float dot = (ax * bx + ay * by) + az * bz;
      const float nA = sqrtf((ax * ax + ay * ay) + az * az);
      const float nB = sqrtf((bx * bx + by * by) + bz * bz);
      const float theta = acosf(((dot) / (nA * nB + 1e-6f)));
      const float phi = asinf((ly - sy) / (sqrtf(rxy2) + 1e-6f));
      const float st = sinf(theta), ct = cosf(theta), sp = sinf(phi), cp = cosf(phi);
      const float c = cm[m];
      const float den = ((st * st) * (cp * cp) + (st * st) * (sp * sp)) + c * (ct * ct);
      const float amp = sqrtf(((PI_F * c) / (den * den)));
      const float psi = ((FOURPI_F * dist) / (lam));
      zr = zr + amp * cosf(psi);
      zi = zi + amp * sinf(psi);
    }
  }
  z_re[(int64_t)b * T + t0 + tt] = zr;
  z_im[(int64_t)b * T + t0 + tt] = zi;
}

__global__ __launch_bounds__(256) void stft_logmag_kernel(const float* __restrict__ z_re, const float* __restrict__ z_im,
                                                          int T, int n_fft, int hop, const float* __restrict__ window,
                                                          int F, int ncols, int select, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* wz = (float2*)smem;            // [n_fft] windowed complex frame
  float2* tw = wz + n_fft;               // [n_fft] (cos, sin)(2 pi m / n_fft)
  const int b = blockIdx.y, j = blockIdx.x;
  int f = j;
  if (select) {   // F.interpolate nearest: src = min(floor(j * fl32(F/ncols)), F-1)
    const float scale = (float)F / (float)ncols;
    f = min((int)floorf((float)j * scale), F - 1);
  }
  const int half = n_fft / 2;
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    int i = f * hop + n - half;          // ReflectionPad1d(n_fft/2)
    if (i < 0) i = -i;
    if (i >= T) i = 2 * (T - 1) - i;
    const float w = window[n];
    wz[n] = make_float2(w * z_re[(int64_t)b * T + i], w * z_im[(int64_t)b * T + i]);
    float s, c;
    sincospif((float)(2 * n) / (float)n_fft, &s, &c);
    tw[n] = make_float2(c, s);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_fft; k += blockDim.x) {
    float re = 0.f, im = 0.f;
    int idx = 0;
    for (int n = 0; n < n_fft; ++n) {
      const float2 v = wz[n];
      const float2 t = tw[idx];
      // (a + jb)(cos - j sin)
      re = fmaf(v.x, t.x, fmaf(v.y, t.y, re));
      im = fmaf(v.y, t.x, fmaf(-v.x, t.y, im));
      idx += k;
      if (idx >= n_fft) idx -= n_fft;
    }
    const float mag = sqrtf(re * re + im * im);
    const int row = (k + half) % n_fft;
    // log in double, rounded once: silent frames give exactly float32(log(1e-6)) like the reference's CPU logf
    out[((int64_t)b * n_fft + row) * ncols + j] = (float)log((double)(mag + 1e-6f));
  }
}

}  // namespace

extern "C" int sar_vr_signal_f32(const float* x, int B, int T, int V, int M, const int32_t* e_src, const int32_t* e_dst,
                                 int E, const float* loc, const float* wavelength, float* z_re, float* z_im,
                                 sar_stream_t s) {
  SAR_REQUIRE(x && e_src && e_dst && loc && wavelength && z_re && z_im, "sar_vr_signal: null pointer");
  SAR_REQUIRE(B > 0 && T > 0 && V > 0 && M > 0 && M <= 4 && E > 0, "sar_vr_signal: bad sizes (M <= 4)");
  const size_t lds = sizeof(float) * 3 * FRAMES * ((V * M) | 1) + sizeof(int) * 2 * E;
  SAR_REQUIRE(lds <= 64 * 1024, "sar_vr_signal: V*M = %d too large for the LDS slab", V * M);
  dim3 grid((T + FRAMES - 1) / FRAMES, B);
  hipLaunchKernelGGL(vr_signal_kernel, grid, dim3(FRAMES), lds, as_stream(s), x, T, V, M, e_src, e_dst, E, loc,
                     wavelength, z_re, z_im);
  SAR_LAUNCH_CHECK("sar_vr_signal_f32");
  return 0;
}

extern "C" int sar_stft_logmag_f32(const float* z_re, const float* z_im, int B, int T, int n_fft, int hop,
                                   const float* window, int out_cols, float* out, sar_stream_t s) {
  SAR_REQUIRE(z_re && z_im && window && out, "sar_stft_logmag: null pointer");
  SAR_REQUIRE(B > 0 && n_fft >= 2 && n_fft <= 2048 && (n_fft % 2) == 0 && hop > 0, "sar_stft_logmag: bad sizes");
  SAR_REQUIRE(T > n_fft / 2, "sar_stft_logmag: reflect padding needs T > n_fft/2 (T=%d, n_fft=%d)", T, n_fft);
  const int F = T / hop + 1;
  const int ncols = out_cols > 0 ? out_cols : F;
  dim3 grid(ncols, B);
  hipLaunchKernelGGL(stft_logmag_kernel, grid, dim3(256), sizeof(float2) * 2 * n_fft, as_stream(s), z_re, z_im, T, n_fft,
                     hop, window, F, ncols, out_cols > 0 ? 1 : 0, out);
  SAR_LAUNCH_CHECK("sar_stft_logmag_f32");
  return 0;
}

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
#include <stdlib.h>

namespace {

constexpr int FRAMES = 64;

// Slab fill shared by the forward and backward signal kernels.  SPLINE == 0: frames t0.. of the clip itself.
// SPLINE == 1: frame j of the UP-SAMPLED clip (utils.py:134-140: cubic interpolation of the Gaussian-smoothed clip
// at np.linspace(0, 1, P*T)) evaluated on the fly from the per-interval cubic pieces of sar_upsample_prepare_f64 --
// the (B,3,P*T,V,M) tensor (45 MB per clip at P = 250) is never materialised.  Evaluation in float64, rounded once
// to float32 like the reference's `.type(torch.FloatTensor)`.
template <int SPLINE>
__device__ __forceinline__ void fill_slab(float* xs, int RS, const float* __restrict__ x, const double* __restrict__ coef,
                                          int b, int T, int Tup, int VM, int t0, int nt) {
  if (!SPLINE) {
    for (int c = 0; c < 3; ++c) {
      const float* g = x + (((int64_t)b * 3 + c) * T + t0) * VM;   // nt*VM contiguous floats
      for (int i = threadIdx.x; i < nt * VM; i += FRAMES) {
        const int tt = i / VM;
        xs[(c * FRAMES + tt) * RS + (i - tt * VM)] = g[i];
      }
    }
  } else {
    const int tt = threadIdx.x;
    if (tt < nt) {
      const int j = t0 + tt;
      // x_new = j / (Tup - 1) on knots i / (T - 1): interval i = floor(x_new (T-1)), local dx in knot units / (T-1)
      const double xn = (double)j / (double)(Tup - 1);
      int i = (int)floor(xn * (double)(T - 1));
      if (i > T - 2) i = T - 2;
      const double dx = xn - (double)i / (double)(T - 1);
      const double* cf = coef + ((int64_t)b * (T - 1) + i) * (3 * VM * 4);
      for (int c = 0; c < 3; ++c)
        for (int vm = 0; vm < VM; ++vm) {
          const double* q = cf + (c * VM + vm) * 4;
          const double v = q[0] + dx * (q[1] + dx * (q[2] + dx * q[3]));
          xs[(c * FRAMES + tt) * RS + vm] = (float)v;
        }
    }
  }
}

template <int SPLINE>
__global__ __launch_bounds__(FRAMES) void vr_signal_kernel(const float* __restrict__ x, const double* __restrict__ coef,
                                                           int T, int Tup, int V, int M,
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
  const int To = SPLINE ? Tup : T;           // frames of the signal
  const int nt = min(FRAMES, To - t0);
  fill_slab<SPLINE>(xs, RS, x, coef, b, T, Tup, VM, t0, nt);
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
  z_re[(int64_t)b * To + t0 + tt] = zr;
  z_im[(int64_t)b * To + t0 + tt] = zi;
}

// ---- the up-sampled signal (sar_vr_signal_upsampled_f32): 75 000 frames per clip instead of 300, i.e. where this kernel is 30 %
// of the Path B step (profiles/r02_pathB_pad250_*).  Same frame-per-lane scheme, three changes (round 4):
//  * ONE BODY AT A TIME through the LDS slab: 19 KB instead of 39 KB per wave, two waves per SIMD instead of one (the
//    kernel is a chain of dependent vector instructions: a second wave hides half of every latency);
//  * the aspect angles never leave the algebra: sin^2(theta) cos^2(phi) + sin^2(theta) sin^2(phi) = sin^2(theta) = 1 - q^2 with
//    q = <A,B> / (|A||B| + 1e-6) = cos(theta), so  den = (1 - q)(1 + q) + c q^2  -- no acos / asin / 4 x sin, cos per term
//    (phi drops out of layers/virtual_radar.py:114-116 altogether); q itself keeps the oracle's roundings.  Measured against the
//    oracle's literal evaluation in tests/test_gpu_radar.py;
//  * RANGE AND PHASE keep the oracle's float32 operation order bit for bit (IEEE sqrt and division: one ulp of range is
//    0.02 rad of phase at lambda = 5e-4), but cos / sin of the ~1e5 rad phase share ONE argument reduction done in float64
//    (psi is exact in float64; n = rint(psi 2/pi), r = psi - n pi/2 with a two-term pi/2: |error| < 1e-11 rad) followed by
//    the Cephes minimax polynomials on [-pi/4, pi/4] -- instead of two Payne-Hanek reductions in ocml's sinf / cosf.
__device__ __forceinline__ void sincos_phase(float psi, float& sn, float& cs) {
  const double x = (double)psi;
  const double n = __builtin_rint(x * 0.63661977236758134308);
  double r = __builtin_fma(-n, 1.57079632679489655800e+00, x);
  r = __builtin_fma(-n, 6.12323399573676603587e-17, r);
  const float rf = (float)r;
  const int q = (int)n;
  const float z = rf * rf;
  const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, rf, rf);
  const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                        fmaf(-0.5f, z, 1.0f));
  const bool swap = q & 1;
  const float a = swap ? cp : sp, b = swap ? sp : cp;
  sn = (q & 2) ? -a : a;
  cs = ((q + 1) & 2) ? -b : b;
}

template <int SPLINE>
__global__ __launch_bounds__(FRAMES) void vr_signal_fast_kernel(const float* __restrict__ x, const double* __restrict__ coef,
                                                                int T, int Tup, int V, int M,
                                                                const int* __restrict__ e_src, const int* __restrict__ e_dst,
                                                                int E, const float* __restrict__ loc_p,
                                                                const float* __restrict__ lam_p, float* __restrict__ z_re,
                                                                float* __restrict__ z_im) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int VM = V * M;
  const int RS = V | 1;                      // odd row stride
  float* xs = smem;                          // [3][FRAMES][RS]: ONE body
  int* es = (int*)(xs + 3 * FRAMES * RS);    // [E] src joints
  int* ed = es + E;                          // [E] dst joints
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * FRAMES;
  const int To = SPLINE ? Tup : T;           // frames of the signal
  const int nt = min(FRAMES, To - t0);
  const int tt = threadIdx.x;
  for (int i = tt; i < E; i += FRAMES) {
    es[i] = e_src[i];
    ed[i] = e_dst[i];
  }
  const float lx = loc_p[0], ly = loc_p[1], lz = loc_p[2];
  const float lam = lam_p[0];
  const float PI_F = 3.14159274101257324f;        // float32(np.pi)
  const float FOURPI_F = 12.5663706143591725f;    // rounds to float32(4*np.pi)
  // spline piece of this lane's frame (fill_slab<1>)
  double sdx = 0.0;
  const double* cf = nullptr;
  if (SPLINE && tt < nt) {
    const int j = t0 + tt;
    const double xn = (double)j / (double)(Tup - 1);
    int i = (int)floor(xn * (double)(T - 1));
    if (i > T - 2) i = T - 2;
    sdx = xn - (double)i / (double)(T - 1);
    cf = coef + ((int64_t)b * (T - 1) + i) * (3 * VM * 4);
  }
  float zr = 0.f, zi = 0.f;
  for (int m = 0; m < M; ++m) {
    __syncthreads();   // the previous body's slab has been consumed (and es / ed are visible)
    if (!SPLINE) {
      for (int c = 0; c < 3; ++c) {
        const float* g = x + (((int64_t)b * 3 + c) * T + t0) * VM;   // nt*VM contiguous floats
        for (int i = tt; i < nt * V; i += FRAMES) {
          const int f = i / V, v = i - f * V;
          xs[(c * FRAMES + f) * RS + v] = g[f * VM + v * M + m];
        }
      }
    } else if (tt < nt) {
      for (int c = 0; c < 3; ++c)
        for (int v = 0; v < V; ++v) {
          const double* q = cf + (c * VM + v * M + m) * 4;
          const double val = q[0] + sdx * (q[1] + sdx * (q[2] + sdx * q[3]));
          xs[(c * FRAMES + tt) * RS + v] = (float)val;
        }
    }
    __syncthreads();
    if (tt < nt) {
      const float* X0 = xs + (0 * FRAMES + tt) * RS;
      const float* X1 = xs + (1 * FRAMES + tt) * RS;
      const float* X2 = xs + (2 * FRAMES + tt) * RS;
      // c = (mean_e |S - D|)^2, layers/virtual_radar.py:110-113 (mean over the edge axis); amplitude only: 1-ulp square root
      float acc = 0.f;
      for (int e = 0; e < E; ++e) {
        const int js = es[e], jd = ed[e];
        const float dx = X0[js] - X0[jd], dy = X1[js] - X1[jd], dz = X2[js] - X2[jd];
        acc = acc + __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz);
      }
      float c = acc / (float)E;
      c = c * c;
      const float spc = __builtin_amdgcn_sqrtf(PI_F * c);
      for (int e = 0; e < E; ++e) {
        const int js = es[e], jd = ed[e];
        const float sx = X0[js], sy = X1[js], sz = X2[js];
        const float dx = X0[jd], dy = X1[jd], dz = X2[jd];
        // range and phase: the oracle's operation order, IEEE sqrt and division
        const float rx = fabsf(sx - lx), ry = fabsf(sy - ly), rz = fabsf(sz - lz);
        const float rxy2 = rx * rx + ry * ry;
        const float dist = sqrtf(rxy2 + rz * rz);
        const float psi = ((FOURPI_F * dist) / (lam));
        // amplitude: q = cos(theta), den = sin^2(theta) + c cos^2(theta)
        const float ax = lx - ((sx + dx) * 0.5f), ay = ly - ((sy + dy) * 0.5f), az = lz - ((sz + dz) * 0.5f);
        const float bx = dx - sx, by = dy - sy, bz = dz - sz;
        const float dot = (ax * bx + ay * by) + az * bz;
        const float nA2 = (ax * ax + ay * ay) + az * az;
        const float nB2 = (bx * bx + by * by) + bz * bz;
        // q with the oracle's roundings (IEEE sqrt and division): near |q| = 1 the denominator is a cancellation, and agreement
        // with the reference there means the SAME q, not merely an accurate one (an rcp-based q: 6e-5 of the signal's scale)
        const float q = dot / (sqrtf(nA2) * sqrtf(nB2) + 1e-6f);
        const float den = fmaf(c * q, q, (1.f - q) * (1.f + q));
        const float amp = spc * __builtin_amdgcn_rcpf(fabsf(den));
        float sn, cs;
        sincos_phase(psi, sn, cs);
        zr = zr + amp * cs;
        zi = zi + amp * sn;
      }
    }
  }
  if (tt < nt) {
    z_re[(int64_t)b * To + t0 + tt] = zr;
    z_im[(int64_t)b * To + t0 + tt] = zi;
  }
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

// ------------------------------------------------------------------------------------------------
// Backward (training radar_location / wavelength, main_spectrogram.py:133-136 + torch autograd of
// layers/virtual_radar.py:93-133).  Only 4 scalars are trainable, so the signal stage is differentiated in
// FORWARD mode: each (clip, frame) thread re-evaluates the geometry with the tangents w.r.t. (loc_x, loc_y,
// loc_z, lambda) and contracts d z / d p with the upstream cotangent of z; the STFT stage is the adjoint of the
// forward kernel (recomputes Z, no saved spectrum).  All reductions have a fixed order (no atomics).

// Stage 1, one workgroup per (clip, STFT frame f): dS = sum of dout over the output columns that consumed f,
// dZ = dS * Z / (|Z| (|Z| + 1e-6)), adjoint DFT, times the window -> G[b][f][n] (gradient of the reflect-padded
// sample f*hop + n - n_fft/2).
__global__ __launch_bounds__(256) void stft_logmag_bwd_frames_kernel(const float* __restrict__ z_re,
                                                                     const float* __restrict__ z_im, int T, int n_fft,
                                                                     int hop, const float* __restrict__ window, int F,
                                                                     int ncols, int select,
                                                                     const float* __restrict__ dout,
                                                                     float2* __restrict__ G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* wz = (float2*)smem;            // [n_fft] windowed complex frame
  float2* tw = wz + n_fft;               // [n_fft] (cos, sin)(2 pi m / n_fft)
  float2* dZ = tw + n_fft;               // [n_fft]
  const int b = blockIdx.y, f = blockIdx.x;
  // output columns that read frame f: j with min(floor(j * fl32(F/ncols)), F-1) == f (select) or j == f
  int j0 = f, j1 = f + 1;
  if (select) {
    const float scale = (float)F / (float)ncols;
    j0 = ncols;
    j1 = 0;
    for (int j = 0; j < ncols; ++j) {   // uniform scalar loop; ncols is small
      const int src = min((int)floorf((float)j * scale), F - 1);
      if (src == f) {
        j0 = min(j0, j);
        j1 = max(j1, j + 1);
      }
    }
  }
  float2* Gf = G + ((int64_t)b * F + f) * n_fft;
  if (j0 >= j1) {   // frame not consumed
    for (int n = threadIdx.x; n < n_fft; n += blockDim.x) Gf[n] = make_float2(0.f, 0.f);
    return;
  }
  const int half = n_fft / 2;
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    int i = f * hop + n - half;
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
      re = fmaf(v.x, t.x, fmaf(v.y, t.y, re));
      im = fmaf(v.y, t.x, fmaf(-v.x, t.y, im));
      idx += k;
      if (idx >= n_fft) idx -= n_fft;
    }
    const float mag = sqrtf(re * re + im * im);
    const int row = (k + half) % n_fft;
    float ds = 0.f;
    for (int j = j0; j < j1; ++j) ds += dout[((int64_t)b * n_fft + row) * ncols + j];
    // d log(|Z| + eps) / d(re, im) = (re, im) / (|Z| (|Z| + eps)); |Z| = 0 takes the zero subgradient
    const float g = mag > 0.f ? ds / (mag * (mag + 1e-6f)) : 0.f;
    dZ[k] = make_float2(g * re, g * im);
  }
  __syncthreads();
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    // re_k = sum_n vx cos + vy sin ; im_k = sum_n vy cos - vx sin  (theta = 2 pi k n / n_fft)
    float gx = 0.f, gy = 0.f;
    int idx = 0;
    for (int k = 0; k < n_fft; ++k) {
      const float2 dz = dZ[k];
      const float2 t = tw[idx];
      gx = fmaf(dz.x, t.x, fmaf(-dz.y, t.y, gx));
      gy = fmaf(dz.x, t.y, fmaf(dz.y, t.x, gy));
      idx += n;
      if (idx >= n_fft) idx -= n_fft;
    }
    const float w = window[n];
    Gf[n] = make_float2(w * gx, w * gy);
  }
}

// Stage 2, one thread per (clip, sample t): gather the frame gradients that touch sample t -- directly or through
// the reflect padding at either end -- in a fixed order.
__global__ __launch_bounds__(256) void stft_logmag_bwd_gather_kernel(const float2* __restrict__ G, int B, int T,
                                                                     int n_fft, int hop, int F,
                                                                     float* __restrict__ dz_re,
                                                                     float* __restrict__ dz_im) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * T) return;
  const int b = (int)(gid / T), t = (int)(gid - (int64_t)b * T);
  const int half = n_fft / 2;
  float sx = 0.f, sy = 0.f;
  // a padded index i maps to sample t if i == t, or i == -t (i < 0), or i == 2(T-1) - t (i >= T)
  const int cand[3] = {t, -t, 2 * (T - 1) - t};
  for (int c = 0; c < 3; ++c) {
    const int i = cand[c];
    if (c == 1 && !(i < 0)) continue;
    if (c == 2 && !(i >= T)) continue;
    // frames with 0 <= i + half - f*hop < n_fft
    int f_lo = (i + half - n_fft + hop) / hop;   // ceil((i + half - n_fft + 1) / hop) for the non-negative case
    if (i + half - n_fft + 1 <= 0) f_lo = 0;
    int f_hi = (i + half >= 0) ? (i + half) / hop : -1;
    if (f_hi > F - 1) f_hi = F - 1;
    for (int f = f_lo; f <= f_hi; ++f) {
      const int n = i + half - f * hop;
      if (n < 0 || n >= n_fft) continue;
      const float2 g = G[((int64_t)b * F + f) * n_fft + n];
      sx += g.x;
      sy += g.y;
    }
  }
  dz_re[gid] = sx;
  dz_im[gid] = sy;
}

// Signal stage: partials[block][4] = sum over the block's frames of Re(conj(dz) . d z / d p), p = (loc_x, loc_y,
// loc_z, lambda).  Same slab staging as the forward kernel.
template <int SPLINE>
__global__ __launch_bounds__(FRAMES) void vr_signal_bwd_kernel(const float* __restrict__ x, const double* __restrict__ coef,
                                                               int T, int Tup, int V, int M,
                                                               const int* __restrict__ e_src,
                                                               const int* __restrict__ e_dst, int E,
                                                               const float* __restrict__ loc_p,
                                                               const float* __restrict__ lam_p,
                                                               const float* __restrict__ dz_re,
                                                               const float* __restrict__ dz_im,
                                                               float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int VM = V * M;
  const int RS = VM | 1;
  float* xs = smem;
  int* es = (int*)(xs + 3 * FRAMES * RS);
  int* ed = es + E;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * FRAMES;
  const int To = SPLINE ? Tup : T;
  const int nt = min(FRAMES, To - t0);
  fill_slab<SPLINE>(xs, RS, x, coef, b, T, Tup, VM, t0, nt);
  for (int i = threadIdx.x; i < E; i += FRAMES) {
    es[i] = e_src[i];
    ed[i] = e_dst[i];
  }
  __syncthreads();
  const int tt = threadIdx.x;
  float gp[4] = {0.f, 0.f, 0.f, 0.f};
  if (tt < nt) {
    const float lx = loc_p[0], ly = loc_p[1], lz = loc_p[2];
    const float lam = lam_p[0];
    const float* X0 = xs + (0 * FRAMES + tt) * RS;
    const float* X1 = xs + (1 * FRAMES + tt) * RS;
    const float* X2 = xs + (2 * FRAMES + tt) * RS;
    const float PI_F = 3.14159274101257324f;
    const float FOURPI_F = 12.5663706143591725f;
    const float ur = dz_re[(int64_t)b * To + t0 + tt], ui = dz_im[(int64_t)b * To + t0 + tt];
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
        // value path: same expressions as the forward kernel
        const float ex = sx - lx, ey = sy - ly, ez = sz - lz;
        const float rx = fabsf(ex), ry = fabsf(ey), rz = fabsf(ez);
        const float rxy2 = rx * rx + ry * ry;
        const float rxy = sqrtf(rxy2);
        const float dist = sqrtf(rxy2 + rz * rz);
        const float ax = lx - ((sx + dx) / 2.f), ay = ly - ((sy + dy) / 2.f), az = lz - ((sz + dz) / 2.f);
        const float bx = dx - sx, by = dy - sy, bz = dz - sz;
        const float dot = (ax * bx + ay * by) + az * bz;
        const float nA = sqrtf((ax * ax + ay * ay) + az * az);
        const float nB = sqrtf((bx * bx + by * by) + bz * bz);
        const float den_u = nA * nB + 1e-6f;
        const float u = dot / den_u;
        const float theta = acosf(u);
        const float den_q = rxy + 1e-6f;
        const float q = (ly - sy) / den_q;
        const float phi = asinf(q);
        const float st = sinf(theta), ct = cosf(theta), sp = sinf(phi), cp = cosf(phi);
        const float c = cm[m];
        const float den = ((st * st) * (cp * cp) + (st * st) * (sp * sp)) + c * (ct * ct);
        const float amp = sqrtf((PI_F * c) / (den * den));
        const float psi = (FOURPI_F * dist) / lam;
        const float cps = cosf(psi), sps = sinf(psi);
        // tangents w.r.t. loc (3 components); d|e|/dl = -e/|e| component-wise (e = s - l)
        // Degenerate points (a joint at the radar position, an absent all-zero body, |u| or |q| = 1) take the zero
        // subgradient.  The reference's autograd returns NaN there (0/0 in the norm backward, sqrt'(0) * 0 for an
        // absent body's zero RCS), i.e. NaN radar_location gradients on every NTU clip with a missing second body.
        const float idist = dist > 0.f ? 1.f / dist : 0.f, irxy = rxy > 0.f ? 1.f / rxy : 0.f, inA = nA > 0.f ? 1.f / nA : 0.f;
        const float ddist[3] = {-ex * idist, -ey * idist, -ez * idist};
        const float drxy[3] = {-ex * irxy, -ey * irxy, 0.f};
        const float ddot[3] = {bx, by, bz};
        const float dnA[3] = {ax * inA, ay * inA, az * inA};
        const float su2 = 1.f - u * u, sq2 = 1.f - q * q;
        const float inv_su = su2 > 0.f ? -1.f / sqrtf(su2) : 0.f;       // d acos
        const float inv_sq = sq2 > 0.f ? 1.f / sqrtf(sq2) : 0.f;        // d asin
        for (int p = 0; p < 3; ++p) {
          const float du = (ddot[p] * den_u - dot * nB * dnA[p]) / (den_u * den_u);
          const float dth = inv_su * du;
          const float dq = ((p == 1 ? 1.f : 0.f) * den_q - (ly - sy) * drxy[p]) / (den_q * den_q);
          const float dph = inv_sq * dq;
          const float dden = 2.f * st * ct * dth * (cp * cp + sp * sp) + (st * st) * (2.f * sp * cp - 2.f * cp * sp) * dph -
                             2.f * c * ct * st * dth;
          const float damp = -amp * dden / den;
          const float dpsi = FOURPI_F * ddist[p] / lam;
          const float dzr = damp * cps - amp * sps * dpsi;
          const float dzi = damp * sps + amp * cps * dpsi;
          gp[p] += ur * dzr + ui * dzi;
        }
        {   // wavelength: only the phase depends on it
          const float dpsi = -psi / lam;
          gp[3] += ur * (-amp * sps * dpsi) + ui * (amp * cps * dpsi);
        }
      }
    }
  }
  // block reduction (one wave), fixed order
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float t = wave_sum(gp[p]);
    if (threadIdx.x == 0) partials[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + p] = t;
  }
}

// Up-sampling preparation (utils.py:134-140, Dataset.pad_frames): per series (clip, coordinate, joint, body) along T
//   1. scipy.ndimage.gaussian_filter1d(sigma, mode='reflect', truncate=4): symmetric FIR of radius int(4 sigma + .5),
//      accumulated in float64 in scipy's order (centre, then the farthest pair inwards), stored as float32 (the
//      filter keeps the input dtype);
//   2. scipy.interpolate.interp1d(np.linspace(0,1,T), y, 'cubic') = the C2 cubic spline with not-a-knot end
//      conditions on uniform knots, in float64: second derivatives by the Thomas algorithm
//      (m[1] = r[1]/6, m[T-2] = r[T-2]/6 from the not-a-knot conditions, m[0] = 2 m[1] - m[2], ...), then the
//      per-interval cubic pieces  a + b dx + c dx^2 + d dx^3  (dx in units of x, knot spacing h = 1/(T-1)).
// One thread per series (150 per clip); scratch [T][nseries] so that neighbouring threads touch neighbouring words.
// Step 1 as its own launch (round 4): one thread per (frame, series) -- the 2 radius + 1 taps of one output are independent of
// every other output, and inside the per-series kernel they were 7 500 serial iterations in front of the spline solve
// (0.6 of its 0.89 ms at bs = 32).  Same summation order, same float64 accumulation, same float32 result.
__global__ __launch_bounds__(256) void upsample_smooth_kernel(const float* __restrict__ x, int B, int T, int VM,
                                                              const double* __restrict__ w, int radius, float* __restrict__ sm) {
  const int nser = B * 3 * VM;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)nser * T) return;
  const int sidx = (int)(gid % nser), t = (int)(gid / nser);
  const int vm = sidx % VM, c = (sidx / VM) % 3, b = sidx / (3 * VM);
  const float* xs = x + ((int64_t)(b * 3 + c) * T) * VM + vm;        // element t at xs[t * VM]
  auto refl = [&](int q) {                                          // scipy 'reflect': (d c b a | a b c d | d c b a)
    while (q < 0 || q >= T) {
      if (q < 0) q = -q - 1;
      if (q >= T) q = 2 * T - 1 - q;
    }
    return q;
  };
  double tmp = (double)xs[(int64_t)t * VM] * w[0];
  for (int k = radius; k >= 1; --k)
    tmp += ((double)xs[(int64_t)refl(t - k) * VM] + (double)xs[(int64_t)refl(t + k) * VM]) * w[k];
  sm[(int64_t)t * nser + sidx] = (float)tmp;
}

// The spline solve with its per-series state in LDS (round 4; T <= PREP_TMAX).  upsample_prepare_kernel below keeps the second
// derivatives and the sweep coefficients in global scratch: every step of its 300-step recurrences is a dependent global-memory
// round trip, and 75 waves cannot hide it (0.31 ms at bs = 32 for 2e6 vector instructions: 1 % of the issue capacity).  Here the
// right-hand sides r(i) are computed first (independent loads), the Thomas sweeps run on LDS ([T][64] float64, lanes = series:
// conflict-free), and the sweep's coefficient sequence -- which does not depend on the data -- is kept once per workgroup.  Same
// formulas in the same order: bit-identical pieces.
constexpr int PREP_TMAX = 304;
__global__ __launch_bounds__(64) void upsample_prepare_lds_kernel(int B, int T, int VM, const float* __restrict__ sm,
                                                                  double* __restrict__ coef) {
  __shared__ double m2s[PREP_TMAX * 64];
  __shared__ double cps[PREP_TMAX];
  const int nser = B * 3 * VM;
  const int lane = threadIdx.x;
  const int sidx = blockIdx.x * 64 + lane;
  const bool live = sidx < nser;
  const int sx = live ? sidx : nser - 1;     // idle lanes shadow the last series (no divergent barriers, nothing stored)
  const int vm = sx % VM, c = (sx / VM) % 3, b = sx / (3 * VM);
  const int n = T - 1;                       // intervals
  const double h = 1.0 / (double)n, ih2 = 6.0 / (h * h);
  auto y = [&](int t) { return (double)sm[(int64_t)t * nser + sx]; };
  auto M2 = [&](int i) -> double& { return m2s[i * 64 + lane]; };
  // r(i) = 6 / h^2 (y(i-1) - 2 y(i) + y(i+1)) for i = 1 .. n-1: independent loads, three values in flight
  {
    double ym = y(0), y0 = y(1);
    for (int i = 1; i <= n - 1; ++i) {
      const double yp = y(i + 1);
      M2(i) = ih2 * (ym - 2.0 * y0 + yp);
      ym = y0;
      y0 = yp;
    }
  }
  const double r1 = M2(1), rn1 = M2(n - 1);
  M2(1) = r1 / 6.0;
  M2(n - 1) = rn1 / 6.0;
  if (n - 2 >= 2) {   // interior unknowns m[2..n-2]: m[i-1] + 4 m[i] + m[i+1] = r[i]
    double cprev = 0.0, dprev = 0.0;
    const double m1 = M2(1), mn1 = M2(n - 1);
    for (int i = 2; i <= n - 2; ++i) {
      double rhs = M2(i);
      if (i == 2) rhs -= m1;
      if (i == n - 2) rhs -= mn1;
      const double lower = (i == 2) ? 0.0 : 1.0;
      const double denom = 4.0 - lower * cprev;
      cprev = ((i == n - 2) ? 0.0 : 1.0) / denom;
      dprev = (rhs - lower * dprev) / denom;
      if (lane == 0) cps[i] = cprev;     // the same sequence in every lane
      M2(i) = dprev;
    }
    __syncthreads();                     // cps (one wave: the barrier is a wait for the LDS stores)
    for (int i = n - 3; i >= 2; --i) M2(i) = M2(i) - cps[i] * M2(i + 1);
  }
  M2(0) = 2.0 * M2(1) - M2(2);
  M2(n) = 2.0 * M2(n - 1) - M2(n - 2);
  if (!live) return;
  double yi = y(0);
  for (int i = 0; i < n; ++i) {
    const double mi = M2(i), mj = M2(i + 1), yj = y(i + 1);
    double* q = coef + ((int64_t)b * n + i) * (3 * VM * 4) + (c * VM + vm) * 4;
    q[0] = yi;
    q[1] = (yj - yi) / h - h * (2.0 * mi + mj) / 6.0;
    q[2] = mi / 2.0;
    q[3] = (mj - mi) / (6.0 * h);
    yi = yj;
  }
}

__global__ __launch_bounds__(64) void upsample_prepare_kernel(int B, int T, int VM, const float* __restrict__ sm,
                                                              double* __restrict__ m2, double* __restrict__ cp,
                                                              double* __restrict__ coef) {
  const int nser = B * 3 * VM;
  const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= nser) return;
  const int vm = sidx % VM, c = (sidx / VM) % 3, b = sidx / (3 * VM);
  const int n = T - 1;                       // intervals
  const double h = 1.0 / (double)n, ih2 = 6.0 / (h * h);
  auto y = [&](int t) { return (double)sm[(int64_t)t * nser + sidx]; };
  auto r = [&](int i) { return ih2 * (y(i - 1) - 2.0 * y(i) + y(i + 1)); };
  auto M2 = [&](int i) -> double& { return m2[(int64_t)i * nser + sidx]; };
  auto CP = [&](int i) -> double& { return cp[(int64_t)i * nser + sidx]; };
  M2(1) = r(1) / 6.0;
  M2(n - 1) = r(n - 1) / 6.0;
  if (n - 2 >= 2) {   // interior unknowns m[2..n-2]: m[i-1] + 4 m[i] + m[i+1] = r[i]
    double cprev = 0.0, dprev = 0.0;
    for (int i = 2; i <= n - 2; ++i) {
      double rhs = r(i);
      if (i == 2) rhs -= M2(1);
      if (i == n - 2) rhs -= M2(n - 1);
      const double lower = (i == 2) ? 0.0 : 1.0;
      const double denom = 4.0 - lower * cprev;
      cprev = ((i == n - 2) ? 0.0 : 1.0) / denom;
      dprev = (rhs - lower * dprev) / denom;
      CP(i) = cprev;
      M2(i) = dprev;
    }
    for (int i = n - 3; i >= 2; --i) M2(i) = M2(i) - CP(i) * M2(i + 1);
  }
  M2(0) = 2.0 * M2(1) - M2(2);
  M2(n) = 2.0 * M2(n - 1) - M2(n - 2);
  for (int i = 0; i < n; ++i) {
    const double mi = M2(i), mj = M2(i + 1), yi = y(i), yj = y(i + 1);
    double* q = coef + ((int64_t)b * n + i) * (3 * VM * 4) + (c * VM + vm) * 4;
    q[0] = yi;
    q[1] = (yj - yi) / h - h * (2.0 * mi + mj) / 6.0;
    q[2] = mi / 2.0;
    q[3] = (mj - mi) / (6.0 * h);
  }
}


// ------------------------------------------------------------------------------------------------
// Trainable Fourier kernels (layers/virtual_radar.py:71-76 train_stft_kernel -> nnAudio STFT(trainable=True)): the DFT
// matrices wcos[k][n], wsin[k][n] (window folded in) are Parameters, so the transform is a plain matrix product with
// whatever they currently hold and the backward pass also produces d wcos / d wsin:
//   Z_re[k] = sum_n fr[n] wcos[k][n] + fi[n] wsin[k][n]        Z_im[k] = sum_n fi[n] wcos[k][n] - fr[n] wsin[k][n]
// (fr, fi = the reflect-padded frame of z_re, z_im).  Forward: KF output columns per workgroup, thread k walks n with
// the TRANSPOSED kernels [n][k] (coalesced).  Backward: per (clip, frame) dZ (kept in HBM for the kernel gradient) and the
// frame cotangent G; the kernel gradient is a (k x n) outer-product accumulation over all frames, split over workgroups
// into slabs that are summed in a fixed order.
constexpr int KF = 4;      // output columns per workgroup in the forward kernel
constexpr int KWB = 16;    // rows k per workgroup in the kernel-gradient reduction

__device__ __forceinline__ int reflect_index(int i, int T) {
  if (i < 0) i = -i;
  if (i >= T) i = 2 * (T - 1) - i;
  return i;
}

__device__ __forceinline__ int nearest_src(int j, int F, int ncols, int select) {
  if (!select) return j;
  const float scale = (float)F / (float)ncols;
  return min((int)floorf((float)j * scale), F - 1);
}

__global__ __launch_bounds__(256) void stft_kernels_fwd_kernel(const float* __restrict__ z_re, const float* __restrict__ z_im,
                                                               int T, int n_fft, int hop, const float* __restrict__ wcosT,
                                                               const float* __restrict__ wsinT, int F, int ncols, int select,
                                                               float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* fr = (float2*)smem;            // [KF][n_fft] (re, im) of the reflect-padded frames
  const int b = blockIdx.y, j0 = blockIdx.x * KF;
  const int half = n_fft / 2;
  for (int q = 0; q < KF; ++q) {
    const int j = min(j0 + q, ncols - 1);
    const int f = nearest_src(j, F, ncols, select);
    for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
      const int i = reflect_index(f * hop + n - half, T);
      fr[q * n_fft + n] = make_float2(z_re[(int64_t)b * T + i], z_im[(int64_t)b * T + i]);
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_fft; k += blockDim.x) {
    float re[KF], im[KF];
#pragma unroll
    for (int q = 0; q < KF; ++q) re[q] = im[q] = 0.f;
    for (int n = 0; n < n_fft; ++n) {
      const float c = wcosT[(int64_t)n * n_fft + k], s = wsinT[(int64_t)n * n_fft + k];
#pragma unroll
      for (int q = 0; q < KF; ++q) {
        const float2 v = fr[q * n_fft + n];
        re[q] = fmaf(v.x, c, fmaf(v.y, s, re[q]));
        im[q] = fmaf(v.y, c, fmaf(-v.x, s, im[q]));
      }
    }
    const int row = (k + half) % n_fft;
#pragma unroll
    for (int q = 0; q < KF; ++q)
      if (j0 + q < ncols) {
        const float mag = sqrtf(re[q] * re[q] + im[q] * im[q]);
        out[((int64_t)b * n_fft + row) * ncols + j0 + q] = (float)log((double)(mag + 1e-6f));
      }
  }
}

// one workgroup per (clip, frame f): dZ[b][f][k] (zero when no output column reads f) and, when G != NULL, the frame
// cotangent G[b][f][n]
__global__ __launch_bounds__(256) void stft_kernels_bwd_frames_kernel(const float* __restrict__ z_re, const float* __restrict__ z_im,
                                                                      int T, int n_fft, int hop, const float* __restrict__ wcos,
                                                                      const float* __restrict__ wsin, const float* __restrict__ wcosT,
                                                                      const float* __restrict__ wsinT, int F, int ncols, int select,
                                                                      const float* __restrict__ dout, float2* __restrict__ dZg,
                                                                      float2* __restrict__ G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* fr = (float2*)smem;            // [n_fft]
  float2* dZ = fr + n_fft;               // [n_fft]
  const int b = blockIdx.y, f = blockIdx.x;
  int j0 = f, j1 = f + 1;
  if (select) {
    j0 = ncols;
    j1 = 0;
    for (int j = 0; j < ncols; ++j)
      if (nearest_src(j, F, ncols, 1) == f) {
        j0 = min(j0, j);
        j1 = max(j1, j + 1);
      }
  }
  float2* dZf = dZg + ((int64_t)b * F + f) * n_fft;
  float2* Gf = G ? G + ((int64_t)b * F + f) * n_fft : nullptr;
  if (j0 >= j1) {
    for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
      dZf[n] = make_float2(0.f, 0.f);
      if (Gf) Gf[n] = make_float2(0.f, 0.f);
    }
    return;
  }
  const int half = n_fft / 2;
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    const int i = reflect_index(f * hop + n - half, T);
    fr[n] = make_float2(z_re[(int64_t)b * T + i], z_im[(int64_t)b * T + i]);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_fft; k += blockDim.x) {
    float re = 0.f, im = 0.f;
    for (int n = 0; n < n_fft; ++n) {
      const float c = wcosT[(int64_t)n * n_fft + k], s = wsinT[(int64_t)n * n_fft + k];
      const float2 v = fr[n];
      re = fmaf(v.x, c, fmaf(v.y, s, re));
      im = fmaf(v.y, c, fmaf(-v.x, s, im));
    }
    const float mag = sqrtf(re * re + im * im);
    const int row = (k + half) % n_fft;
    float ds = 0.f;
    for (int j = j0; j < j1; ++j) ds += dout[((int64_t)b * n_fft + row) * ncols + j];
    const float g = mag > 0.f ? ds / (mag * (mag + 1e-6f)) : 0.f;
    const float2 d = make_float2(g * re, g * im);
    dZ[k] = d;
    dZf[k] = d;
  }
  if (!Gf) return;
  __syncthreads();
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    float gx = 0.f, gy = 0.f;
    for (int k = 0; k < n_fft; ++k) {
      const float c = wcos[(int64_t)k * n_fft + n], s = wsin[(int64_t)k * n_fft + n];
      const float2 d = dZ[k];
      gx = fmaf(d.x, c, fmaf(-d.y, s, gx));
      gy = fmaf(d.x, s, fmaf(d.y, c, gy));
    }
    Gf[n] = make_float2(gx, gy);
  }
}

// d wcos[k][n] = sum_{b,f} dZre fr[n] + dZim fi[n];  d wsin[k][n] = sum_{b,f} dZre fi[n] - dZim fr[n].
// Workgroup (k block of KWB rows, split): thread n owns column n of the block; slab[split][2][n_fft][n_fft].
__global__ __launch_bounds__(1024) void stft_kernels_wgrad_kernel(const float* __restrict__ z_re, const float* __restrict__ z_im,
                                                                  int B, int T, int n_fft, int hop, int F,
                                                                  const float2* __restrict__ dZg, int nsplit,
                                                                  float* __restrict__ slab) {
  __shared__ float2 dz[KWB];
  const int k0 = blockIdx.x * KWB, split = blockIdx.y, n = threadIdx.x;
  const int half = n_fft / 2;
  const int64_t total = (int64_t)B * F;
  const int64_t per = (total + nsplit - 1) / nsplit;
  const int64_t lo = split * per, hi = (lo + per < total) ? lo + per : total;
  float ac[KWB], as[KWB];
#pragma unroll
  for (int q = 0; q < KWB; ++q) ac[q] = as[q] = 0.f;
  for (int64_t bf = lo; bf < hi; ++bf) {
    const int b = (int)(bf / F), f = (int)(bf - (int64_t)b * F);
    __syncthreads();
    if (n < KWB) dz[n] = (k0 + n < n_fft) ? dZg[bf * n_fft + k0 + n] : make_float2(0.f, 0.f);
    __syncthreads();
    const int i = reflect_index(f * hop + n - half, T);
    const float xr = z_re[(int64_t)b * T + i], xi = z_im[(int64_t)b * T + i];
#pragma unroll
    for (int q = 0; q < KWB; ++q) {
      const float2 d = dz[q];
      ac[q] = fmaf(d.x, xr, fmaf(d.y, xi, ac[q]));
      as[q] = fmaf(d.x, xi, fmaf(-d.y, xr, as[q]));
    }
  }
  float* sc = slab + (int64_t)split * 2 * n_fft * n_fft;
  float* ss = sc + (int64_t)n_fft * n_fft;
#pragma unroll
  for (int q = 0; q < KWB; ++q)
    if (k0 + q < n_fft) {
      sc[(int64_t)(k0 + q) * n_fft + n] = ac[q];
      ss[(int64_t)(k0 + q) * n_fft + n] = as[q];
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
  const char* fe = getenv("SAR_VR_FAST_PLAIN");   // test switch (read per call): the up-sampled path's arithmetic on plain clips
  const bool fast = fe && fe[0] == '1';
  if (fast) {
    const size_t lds1 = sizeof(float) * 3 * FRAMES * (V | 1) + sizeof(int) * 2 * E;
    hipLaunchKernelGGL(vr_signal_fast_kernel<0>, grid, dim3(FRAMES), lds1, as_stream(s), x, nullptr, T, T, V, M, e_src, e_dst, E,
                       loc, wavelength, z_re, z_im);
  } else {
    hipLaunchKernelGGL(vr_signal_kernel<0>, grid, dim3(FRAMES), lds, as_stream(s), x, nullptr, T, T, V, M, e_src, e_dst, E,
                       loc, wavelength, z_re, z_im);
  }
  SAR_LAUNCH_CHECK("sar_vr_signal_f32");
  return 0;
}

extern "C" int64_t sar_upsample_workspace_bytes(int B, int T, int V, int M) {
  if (B <= 0 || T < 5 || V <= 0 || M <= 0) return -1;
  const int64_t nser = (int64_t)B * 3 * V * M;
  return nser * T * (4 + 8 + 8);
}

extern "C" int64_t sar_upsample_coef_doubles(int B, int T, int V, int M) {
  if (B <= 0 || T < 5 || V <= 0 || M <= 0) return -1;
  return (int64_t)B * (T - 1) * 3 * V * M * 4;
}

extern "C" int sar_upsample_prepare_f64(const float* x, int B, int T, int V, int M, const double* weights, int radius,
                                        void* workspace, double* coef, sar_stream_t s) {
  SAR_REQUIRE(x && weights && workspace && coef, "sar_upsample_prepare: null pointer");
  SAR_REQUIRE(B > 0 && T >= 5 && V > 0 && M > 0 && radius >= 0, "sar_upsample_prepare: bad sizes (T >= 5)");
  const int64_t nser = (int64_t)B * 3 * V * M;
  SAR_REQUIRE(nser < (1ll << 31), "sar_upsample_prepare: too many series");
  float* sm = (float*)workspace;
  double* m2 = (double*)((char*)workspace + (((nser * T * 4) + 7) / 8) * 8);
  double* cp = m2 + nser * T;
  hipLaunchKernelGGL(upsample_smooth_kernel, dim3((unsigned)((nser * T + 255) / 256)), dim3(256), 0, as_stream(s), x, B, T, V * M,
                     weights, radius, sm);
  const char* pe = getenv("SAR_UPSAMPLE_LDS");   // A/B switch (read per call): 0 = the global-scratch solve
  if (T <= PREP_TMAX && !(pe && pe[0] == '0'))
    hipLaunchKernelGGL(upsample_prepare_lds_kernel, dim3((unsigned)((nser + 63) / 64)), dim3(64), 0, as_stream(s), B, T, V * M, sm, coef);
  else
    hipLaunchKernelGGL(upsample_prepare_kernel, dim3((unsigned)((nser + 63) / 64)), dim3(64), 0, as_stream(s), B, T, V * M, sm, m2,
                       cp, coef);
  SAR_LAUNCH_CHECK("sar_upsample_prepare_f64");
  return 0;
}

extern "C" int sar_vr_signal_upsampled_f32(const double* coef, int B, int T, int P, int V, int M, const int32_t* e_src,
                                           const int32_t* e_dst, int E, const float* loc, const float* wavelength,
                                           float* z_re, float* z_im, sar_stream_t s) {
  SAR_REQUIRE(coef && e_src && e_dst && loc && wavelength && z_re && z_im, "sar_vr_signal_upsampled: null pointer");
  SAR_REQUIRE(B > 0 && T >= 5 && P >= 1 && V > 0 && M > 0 && M <= 4 && E > 0, "sar_vr_signal_upsampled: bad sizes");
  SAR_REQUIRE((int64_t)T * P < (1ll << 31), "sar_vr_signal_upsampled: T*P too large");
  const size_t lds = sizeof(float) * 3 * FRAMES * ((V * M) | 1) + sizeof(int) * 2 * E;
  SAR_REQUIRE(lds <= 64 * 1024, "sar_vr_signal_upsampled: V*M = %d too large for the LDS slab", V * M);
  const int Tup = T * P;
  dim3 grid((Tup + FRAMES - 1) / FRAMES, B);
  const char* fe = getenv("SAR_VR_FAST");   // A/B switch (read per call): 0 = the literal evaluation
  const bool fast = !(fe && fe[0] == '0');
  if (fast) {
    const size_t lds1 = sizeof(float) * 3 * FRAMES * (V | 1) + sizeof(int) * 2 * E;   // one body at a time
    hipLaunchKernelGGL(vr_signal_fast_kernel<1>, grid, dim3(FRAMES), lds1, as_stream(s), nullptr, coef, T, Tup, V, M, e_src, e_dst,
                       E, loc, wavelength, z_re, z_im);
  } else {
    hipLaunchKernelGGL(vr_signal_kernel<1>, grid, dim3(FRAMES), lds, as_stream(s), nullptr, coef, T, Tup, V, M, e_src, e_dst,
                       E, loc, wavelength, z_re, z_im);
  }
  SAR_LAUNCH_CHECK("sar_vr_signal_upsampled_f32");
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

extern "C" int64_t sar_stft_logmag_bwd_workspace_floats(int B, int T, int n_fft, int hop) {
  if (B <= 0 || T <= 0 || n_fft <= 0 || hop <= 0) return -1;
  return (int64_t)B * (T / hop + 1) * n_fft * 2;
}

extern "C" int sar_stft_logmag_bwd_f32(const float* z_re, const float* z_im, int B, int T, int n_fft, int hop,
                                       const float* window, int out_cols, const float* dout, float* workspace,
                                       float* dz_re, float* dz_im, sar_stream_t s) {
  SAR_REQUIRE(z_re && z_im && window && dout && workspace && dz_re && dz_im, "sar_stft_logmag_bwd: null pointer");
  SAR_REQUIRE(B > 0 && n_fft >= 2 && n_fft <= 2048 && (n_fft % 2) == 0 && hop > 0, "sar_stft_logmag_bwd: bad sizes");
  SAR_REQUIRE(T > n_fft / 2, "sar_stft_logmag_bwd: reflect padding needs T > n_fft/2 (T=%d, n_fft=%d)", T, n_fft);
  const int F = T / hop + 1;
  const int ncols = out_cols > 0 ? out_cols : F;
  hipLaunchKernelGGL(stft_logmag_bwd_frames_kernel, dim3(F, B), dim3(256), sizeof(float2) * 3 * n_fft, as_stream(s),
                     z_re, z_im, T, n_fft, hop, window, F, ncols, out_cols > 0 ? 1 : 0, dout, (float2*)workspace);
  SAR_LAUNCH_CHECK("sar_stft_logmag_bwd_f32 (frames)");
  const int64_t n = (int64_t)B * T;
  hipLaunchKernelGGL(stft_logmag_bwd_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(s),
                     (const float2*)workspace, B, T, n_fft, hop, F, dz_re, dz_im);
  SAR_LAUNCH_CHECK("sar_stft_logmag_bwd_f32 (gather)");
  return 0;
}

extern "C" int sar_stft_kernels_fwd_f32(const float* z_re, const float* z_im, int B, int T, int n_fft, int hop,
                                        const float* wcosT, const float* wsinT, int out_cols, float* out, sar_stream_t s) {
  SAR_REQUIRE(z_re && z_im && wcosT && wsinT && out, "sar_stft_kernels_fwd: null pointer");
  SAR_REQUIRE(B > 0 && n_fft >= 2 && n_fft <= 1024 && (n_fft % 2) == 0 && hop > 0, "sar_stft_kernels_fwd: bad sizes (n_fft <= 1024)");
  SAR_REQUIRE(T > n_fft / 2, "sar_stft_kernels_fwd: reflect padding needs T > n_fft/2 (T=%d, n_fft=%d)", T, n_fft);
  const int F = T / hop + 1;
  const int ncols = out_cols > 0 ? out_cols : F;
  hipLaunchKernelGGL(stft_kernels_fwd_kernel, dim3((ncols + KF - 1) / KF, B), dim3(256), sizeof(float2) * KF * n_fft,
                     as_stream(s), z_re, z_im, T, n_fft, hop, wcosT, wsinT, F, ncols, out_cols > 0 ? 1 : 0, out);
  SAR_LAUNCH_CHECK("sar_stft_kernels_fwd_f32");
  return 0;
}

extern "C" int64_t sar_stft_kernels_bwd_workspace_floats(int B, int T, int n_fft, int hop, int nsplit) {
  if (B <= 0 || T <= 0 || n_fft <= 0 || hop <= 0 || nsplit <= 0) return -1;
  return (int64_t)B * (T / hop + 1) * n_fft * 4 + (int64_t)nsplit * 2 * n_fft * n_fft;
}

extern "C" int sar_stft_kernels_bwd_f32(const float* z_re, const float* z_im, int B, int T, int n_fft, int hop,
                                        const float* wcos, const float* wsin, const float* wcosT, const float* wsinT,
                                        int out_cols, const float* dout, float* workspace, int nsplit, float* dw,
                                        float* dz_re, float* dz_im, sar_stream_t s) {
  SAR_REQUIRE(z_re && z_im && wcos && wsin && wcosT && wsinT && dout && workspace && dw, "sar_stft_kernels_bwd: null pointer");
  SAR_REQUIRE((dz_re == nullptr) == (dz_im == nullptr), "sar_stft_kernels_bwd: dz_re / dz_im go together");
  SAR_REQUIRE(B > 0 && n_fft >= KWB && n_fft <= 1024 && (n_fft % 2) == 0 && hop > 0 && nsplit > 0 && nsplit <= 65535,
              "sar_stft_kernels_bwd: bad sizes (16 <= n_fft <= 1024)");
  SAR_REQUIRE(T > n_fft / 2, "sar_stft_kernels_bwd: reflect padding needs T > n_fft/2 (T=%d, n_fft=%d)", T, n_fft);
  const int F = T / hop + 1;
  const int ncols = out_cols > 0 ? out_cols : F;
  float2* dZ = (float2*)workspace;
  float2* G = dZ + (int64_t)B * F * n_fft;
  float* slab = (float*)(G + (int64_t)B * F * n_fft);
  hipLaunchKernelGGL(stft_kernels_bwd_frames_kernel, dim3(F, B), dim3(256), sizeof(float2) * 2 * n_fft, as_stream(s), z_re, z_im,
                     T, n_fft, hop, wcos, wsin, wcosT, wsinT, F, ncols, out_cols > 0 ? 1 : 0, dout, dZ, dz_re ? G : nullptr);
  SAR_LAUNCH_CHECK("sar_stft_kernels_bwd_f32 (frames)");
  hipLaunchKernelGGL(stft_kernels_wgrad_kernel, dim3((n_fft + KWB - 1) / KWB, nsplit), dim3(n_fft), 0, as_stream(s), z_re, z_im, B,
                     T, n_fft, hop, F, (const float2*)dZ, nsplit, slab);
  SAR_LAUNCH_CHECK("sar_stft_kernels_bwd_f32 (kernel gradient)");
  const int64_t nw = (int64_t)2 * n_fft * n_fft;
  if (int rc = sar_slab_reduce_f32(slab, nsplit, nw, nw, dw, s)) return rc;
  if (dz_re) {
    const int64_t n = (int64_t)B * T;
    hipLaunchKernelGGL(stft_logmag_bwd_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(s),
                       (const float2*)G, B, T, n_fft, hop, F, dz_re, dz_im);
    SAR_LAUNCH_CHECK("sar_stft_kernels_bwd_f32 (gather)");
  }
  return 0;
}

extern "C" int sar_vr_signal_bwd_nparts(int B, int T) {
  if (B <= 0 || T <= 0) return SAR_E_ARG;
  return B * ((T + FRAMES - 1) / FRAMES);
}

extern "C" int sar_vr_signal_bwd_f32(const float* x, int B, int T, int V, int M, const int32_t* e_src,
                                     const int32_t* e_dst, int E, const float* loc, const float* wavelength,
                                     const float* dz_re, const float* dz_im, float* partials, sar_stream_t s) {
  SAR_REQUIRE(x && e_src && e_dst && loc && wavelength && dz_re && dz_im && partials, "sar_vr_signal_bwd: null pointer");
  SAR_REQUIRE(B > 0 && T > 0 && V > 0 && M > 0 && M <= 4 && E > 0, "sar_vr_signal_bwd: bad sizes (M <= 4)");
  const size_t lds = sizeof(float) * 3 * FRAMES * ((V * M) | 1) + sizeof(int) * 2 * E;
  SAR_REQUIRE(lds <= 64 * 1024, "sar_vr_signal_bwd: V*M = %d too large for the LDS slab", V * M);
  dim3 grid((T + FRAMES - 1) / FRAMES, B);
  hipLaunchKernelGGL(vr_signal_bwd_kernel<0>, grid, dim3(FRAMES), lds, as_stream(s), x, nullptr, T, T, V, M, e_src, e_dst,
                     E, loc, wavelength, dz_re, dz_im, partials);
  SAR_LAUNCH_CHECK("sar_vr_signal_bwd_f32");
  return 0;
}

extern "C" int sar_vr_signal_upsampled_bwd_f32(const double* coef, int B, int T, int P, int V, int M, const int32_t* e_src,
                                               const int32_t* e_dst, int E, const float* loc, const float* wavelength,
                                               const float* dz_re, const float* dz_im, float* partials, sar_stream_t s) {
  SAR_REQUIRE(coef && e_src && e_dst && loc && wavelength && dz_re && dz_im && partials, "sar_vr_signal_upsampled_bwd: null pointer");
  SAR_REQUIRE(B > 0 && T >= 5 && P >= 1 && V > 0 && M > 0 && M <= 4 && E > 0, "sar_vr_signal_upsampled_bwd: bad sizes");
  const size_t lds = sizeof(float) * 3 * FRAMES * ((V * M) | 1) + sizeof(int) * 2 * E;
  SAR_REQUIRE(lds <= 64 * 1024, "sar_vr_signal_upsampled_bwd: V*M = %d too large for the LDS slab", V * M);
  const int Tup = T * P;
  dim3 grid((Tup + FRAMES - 1) / FRAMES, B);
  hipLaunchKernelGGL(vr_signal_bwd_kernel<1>, grid, dim3(FRAMES), lds, as_stream(s), nullptr, coef, T, Tup, V, M, e_src,
                     e_dst, E, loc, wavelength, dz_re, dz_im, partials);
  SAR_LAUNCH_CHECK("sar_vr_signal_upsampled_bwd_f32");
  return 0;
}

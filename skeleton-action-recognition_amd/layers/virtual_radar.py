"""Drop-in for the reference's layers/virtual_radar.py: `VirtualRadar(...)(x)` with the same constructor
signature (layers/virtual_radar.py:36-45) and output (B, n_fft, T//hop_length + 1) = log|STFT| of the
simulated radar return, frequency axis rolled by n_fft/2 (layers/virtual_radar.py:79-134).

The arithmetic runs in two HIP kernels of libsar_hip.so (csrc/radar.hip): sar_vr_signal_f32 (geometry, RCS,
phase, sum over edges and bodies) and sar_stft_logmag_f32 (reflect pad, periodic-Hann windowed DFT, log
magnitude, roll).  The window is the one nnAudio 0.1.1's STFT builds
(scipy.signal.get_window('hann', n_fft, fftbins=True)); no nnAudio dependency.

`wavelength` and `radar_location` are torch Parameters with the reference's names.  When either requires a gradient
(train_wavelength / train_radar_location, or `requires_grad = True` set later as main_spectrogram.py:133-136 does)
the forward records an autograd node whose backward runs sar_stft_logmag_bwd_f32 (adjoint of the STFT / log-magnitude
stage) and sar_vr_signal_bwd_f32 (forward-mode tangents of the geometry w.r.t. the 4 scalars, contracted with the
signal cotangent).  The skeleton input gets no gradient (it is data).

train_stft_kernel=True (layers/virtual_radar.py:71-76 -> nnAudio STFT(trainable=True)) turns the two conv1d Fourier kernels
into the Parameters `stft.wcos` / `stft.wsin` of shape (n_fft, 1, n_fft) (window folded in, float32 of the float64
formula like nnAudio's create_fourier_kernels).  The transform is then the matrix product with whatever the kernels
currently hold (sar_stft_kernels_fwd_f32) and the backward pass also returns their gradients
(sar_stft_kernels_bwd_f32: outer-product accumulation over all frames, slab-reduced in a fixed order).
"""
import numpy as np
import torch

from sar_amd import _lib as L
from sar_amd import profiler
from sar_amd._lib import check, ptr, stream_ptr

# layers/virtual_radar.py:10-13: the skeleton's bones without the hand-tip / thumb / foot-tip stubs the reference
# drops, written as chains of joints
_CHAINS = [(0, 1, 20, 2, 3), (20, 4, 5, 6, 7, 21), (7, 22), (20, 8, 9, 10, 11, 23), (11, 24), (0, 16), (0, 12, 13, 14, 15),
           (16, 17, 18, 19)]
edges = [(c[i], c[i + 1]) for c in _CHAINS for i in range(len(c) - 1)]


def gaussian_weights(sigma, truncate=4.0):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius): one-sided weights w[0..radius], float64."""
    radius = int(truncate * float(sigma) + 0.5)
    k = np.arange(-radius, radius + 1, dtype=np.float64)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * k ** 2)
    phi = phi / phi.sum()
    return phi[radius:].copy(), radius


class _RadarFunction(torch.autograd.Function):
    """log|STFT(z(x; loc, lambda))| with gradients for radar_location and wavelength."""

    @staticmethod
    def forward(ctx, loc, wavelength, mod, x, out_cols, num_pad_frames, sigma):
        coef = mod.spline_pieces(x, sigma) if num_pad_frames else None
        zr, zi = mod.signal(x, num_pad_frames, coef)
        ctx.mod, ctx.out_cols, ctx.P, ctx.coef = mod, out_cols, num_pad_frames, coef
        ctx.save_for_backward(x, zr, zi)
        return mod._stft(zr, zi, out_cols)

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        x, zr, zi = ctx.saved_tensors
        lib = L.load()
        B, T = zr.shape
        _, _, _, V, M = x.shape
        dout = dout.contiguous().float()
        ws = torch.empty(lib.sar_stft_logmag_bwd_workspace_floats(B, T, mod.n_fft, mod.hop_length), dtype=torch.float32,
                         device=x.device)
        dzr, dzi = torch.empty_like(zr), torch.empty_like(zi)
        check(lib.sar_stft_logmag_bwd_f32(ptr(zr), ptr(zi), B, T, mod.n_fft, mod.hop_length, ptr(mod.window), ctx.out_cols,
                                          ptr(dout), ptr(ws), ptr(dzr), ptr(dzi), stream_ptr()), "sar_stft_logmag_bwd_f32")
        g = _signal_backward(mod, x, ctx.coef, ctx.P, dzr, dzi)
        return g[:3].clone(), g[3].reshape(mod.wavelength.shape), None, None, None, None, None


class _KernelRadarFunction(torch.autograd.Function):
    """The layer with nnAudio's trainable Fourier kernels: gradients for wcos / wsin and, when they train too, for
    radar_location / wavelength."""

    @staticmethod
    def forward(ctx, loc, wavelength, wcos, wsin, mod, x, out_cols, num_pad_frames, sigma):
        coef = mod.spline_pieces(x, sigma) if num_pad_frames else None
        zr, zi = mod.signal(x, num_pad_frames, coef)
        wT = mod.stft.transposed()
        ctx.mod, ctx.out_cols, ctx.P, ctx.coef, ctx.wT = mod, out_cols, num_pad_frames, coef, wT
        ctx.save_for_backward(x, zr, zi)
        return mod._stft(zr, zi, out_cols, wT)

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        x, zr, zi = ctx.saved_tensors
        lib = L.load()
        B, T = zr.shape
        _, _, _, V, M = x.shape
        n_fft, hop = mod.n_fft, mod.hop_length
        dout = dout.contiguous().float()
        need_radar = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        nsplit = max(1, min(B * (T // hop + 1), 512 // max(1, n_fft // 16)))
        ws = torch.empty(lib.sar_stft_kernels_bwd_workspace_floats(B, T, n_fft, hop, nsplit), dtype=torch.float32, device=x.device)
        dw = torch.empty((2, n_fft, 1, n_fft), dtype=torch.float32, device=x.device)
        dzr = torch.empty_like(zr) if need_radar else None
        dzi = torch.empty_like(zi) if need_radar else None
        wcosT, wsinT = ctx.wT
        check(lib.sar_stft_kernels_bwd_f32(ptr(zr), ptr(zi), B, T, n_fft, hop, ptr(mod.stft.wcos.data), ptr(mod.stft.wsin.data),
                                           ptr(wcosT), ptr(wsinT), ctx.out_cols, ptr(dout), ptr(ws), nsplit, ptr(dw), ptr(dzr),
                                           ptr(dzi), stream_ptr()), "sar_stft_kernels_bwd_f32")
        dloc = dlam = None
        if need_radar:
            g = _signal_backward(mod, x, ctx.coef, ctx.P, dzr, dzi)
            dloc, dlam = g[:3].clone(), g[3].reshape(mod.wavelength.shape)
        return dloc, dlam, dw[0], dw[1], None, None, None, None, None


def _signal_backward(mod, x, coef, P, dzr, dzi):
    """d loss / d (loc_x, loc_y, loc_z, wavelength) from the signal cotangent (sar_vr_signal*_bwd_f32)"""
    lib = L.load()
    B, _, _, V, M = x.shape
    T = dzr.shape[1]
    nparts = lib.sar_vr_signal_bwd_nparts(B, T)
    part = torch.empty((nparts, 4), dtype=torch.float32, device=x.device)
    if P:
        check(lib.sar_vr_signal_upsampled_bwd_f32(ptr(coef), B, x.shape[2], P, V, M, ptr(mod._src), ptr(mod._dst), len(mod.src),
                                                  ptr(mod.radar_location.data), ptr(mod.wavelength.data.reshape(1)), ptr(dzr),
                                                  ptr(dzi), ptr(part), stream_ptr()), "sar_vr_signal_upsampled_bwd_f32")
    else:
        check(lib.sar_vr_signal_bwd_f32(ptr(x), B, T, V, M, ptr(mod._src), ptr(mod._dst), len(mod.src),
                                        ptr(mod.radar_location.data), ptr(mod.wavelength.data.reshape(1)), ptr(dzr), ptr(dzi),
                                        ptr(part), stream_ptr()), "sar_vr_signal_bwd_f32")
    return part.double().sum(0).float()          # fixed-order reduction of the per-block partial sums


class _FourierKernels(torch.nn.Module):
    """nnAudio 0.1.1 STFT(trainable=True)'s parameters: wsin / wcos (n_fft, 1, n_fft), k-th row = window[n] * sin / cos
    (2 pi k n / n_fft) computed in float64 and stored as float32 (create_fourier_kernels, freq_scale='no', hann)."""

    def __init__(self, n_fft, device):
        super().__init__()
        n = np.arange(n_fft, dtype=np.float64)
        w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
        ang = 2.0 * np.pi * n[:, None] * n[None, :] / n_fft
        self.wsin = torch.nn.Parameter(torch.from_numpy((w * np.sin(ang)).astype(np.float32))[:, None, :].contiguous().to(device))
        self.wcos = torch.nn.Parameter(torch.from_numpy((w * np.cos(ang)).astype(np.float32))[:, None, :].contiguous().to(device))
        self.n_fft = n_fft

    def transposed(self):
        """[n][k] copies of the current kernels for the forward kernel's coalesced reads"""
        N = self.n_fft
        out = torch.empty((2, N, N), dtype=torch.float32, device=self.wcos.device)
        for i, w in enumerate((self.wcos, self.wsin)):
            check(L.load().sar_transpose_f32(ptr(w.data), ptr(out[i]), 1, N, N, stream_ptr()), "sar_transpose_f32")
        return out[0], out[1]


class VirtualRadar(torch.nn.Module):
    def __init__(self, edges=edges, wavelength=1e-3, radar_location=[0., 0., 0.], train_wavelength=False,
                 train_radar_location=False, train_stft_kernel=False, n_fft=256, hop_length=16, device='cuda:0'):
        super().__init__()
        L.load()
        assert not train_stft_kernel or 16 <= n_fft <= 1024, "trainable STFT kernels are built for 16 <= n_fft <= 1024"
        self.stft = _FourierKernels(n_fft, device) if train_stft_kernel else None
        # layers/virtual_radar.py:46-52
        self.wavelength = torch.nn.Parameter(torch.as_tensor(wavelength, dtype=torch.float32),
                                             requires_grad=bool(train_wavelength))
        self.radar_location = torch.nn.Parameter(torch.as_tensor(radar_location, dtype=torch.float32),
                                                 requires_grad=bool(train_radar_location))
        self.src, self.dst = map(list, zip(*edges))
        self.n_fft, self.hop_length = n_fft, hop_length
        n = np.arange(n_fft, dtype=np.float64)
        self.register_buffer("window", torch.from_numpy((0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)),
                             persistent=False)
        self.register_buffer("_src", torch.tensor(self.src, dtype=torch.int32), persistent=False)
        self.register_buffer("_dst", torch.tensor(self.dst, dtype=torch.int32), persistent=False)
        self.to(device)

    def spline_pieces(self, x, sigma=3):
        """Gaussian smoothing + not-a-knot cubic spline of every coordinate series along T (utils.py:134-140), as
        per-interval cubic pieces [B][T-1][3][V*M][4] float64 for the fused up-sampled signal kernel."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[1] == 3
        x = x.contiguous()
        B, _, T, V, M = x.shape
        lib = L.load()
        key = (float(sigma), x.device)      # the smoothing weights live on the device: no host copy per step (and none inside a graph capture)
        cache = self.__dict__.setdefault("_gauss_cache", {})
        if key not in cache:
            w, radius = gaussian_weights(sigma)
            cache[key] = (torch.from_numpy(w).to(x.device), radius)
        w_dev, radius = cache[key]
        ws = torch.empty(lib.sar_upsample_workspace_bytes(B, T, V, M), dtype=torch.uint8, device=x.device)
        coef = torch.empty(lib.sar_upsample_coef_doubles(B, T, V, M), dtype=torch.float64, device=x.device)
        # algorithmic float64 work: (2 radius + 1)-tap smoothing + ~8 T flops of the tridiagonal solve per series
        with profiler.region("radar_upsample_prepare", B * 3.0 * V * M * T * (2.0 * (2 * radius + 1) + 8.0)):
            check(lib.sar_upsample_prepare_f64(ptr(x), B, T, V, M, ptr(w_dev), radius, ptr(ws), ptr(coef), stream_ptr()),
                  "sar_upsample_prepare_f64")
        return coef

    def signal(self, x, num_pad_frames=0, coef=None, sigma=3):
        """Complex baseband signal z[b, t] (layers/virtual_radar.py:93-123) as (z_re, z_im); with num_pad_frames = P
        the signal of the clip up-sampled to P*T frames (utils.py:134-140), frames evaluated on the fly."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[1] == 3
        x = x.contiguous()
        B, _, T, V, M = x.shape
        Tz = T * num_pad_frames if num_pad_frames else T
        zr = torch.empty((B, Tz), dtype=torch.float32, device=x.device)
        zi = torch.empty_like(zr)
        if num_pad_frames:
            coef = coef if coef is not None else self.spline_pieces(x, sigma)
            # float64 work of the on-the-fly up-sampling: one cubic (3 DFMA) per coordinate and up-sampled frame
            with profiler.region("radar_signal_upsampled", 6.0 * B * Tz * 3 * V * M):
                check(L.load().sar_vr_signal_upsampled_f32(ptr(coef), B, T, num_pad_frames, V, M, ptr(self._src), ptr(self._dst),
                                                           len(self.src), ptr(self.radar_location.data),
                                                           ptr(self.wavelength.data.reshape(1)), ptr(zr), ptr(zi), stream_ptr()),
                      "sar_vr_signal_upsampled_f32")
        else:
            check(L.load().sar_vr_signal_f32(ptr(x), B, T, V, M, ptr(self._src), ptr(self._dst), len(self.src),
                                             ptr(self.radar_location.data), ptr(self.wavelength.data.reshape(1)), ptr(zr),
                                             ptr(zi), stream_ptr()), "sar_vr_signal_f32")
        return zr, zi

    def forward(self, x, out_cols=0, num_pad_frames=0, sigma=3):
        """out_cols > 0 produces only the frames a nearest-neighbour F.interpolate(..., out_cols) would read
        (models/resnet.py:26 fused as a column select) -> (B, n_fft, out_cols).
        num_pad_frames = P > 0: x is the RAW (B,3,T,V,M) clip and the result is what the reference computes from
        utils.Dataset.pad_frames(x) (Gaussian smoothing sigma + cubic interpolation to P*T frames, utils.py:134-140) --
        the up-sampled tensor is never built."""
        radar_grad = self.radar_location.requires_grad or self.wavelength.requires_grad
        if self.stft is not None:      # nnAudio's trainable kernels: always the matrix product with their current values
            if torch.is_grad_enabled() and (radar_grad or self.stft.wcos.requires_grad or self.stft.wsin.requires_grad):
                return _KernelRadarFunction.apply(self.radar_location, self.wavelength, self.stft.wcos, self.stft.wsin, self,
                                                  x.contiguous(), out_cols, num_pad_frames, sigma)
            zr, zi = self.signal(x, num_pad_frames, None, sigma)
            return self._stft(zr, zi, out_cols, self.stft.transposed())
        if torch.is_grad_enabled() and radar_grad:
            return _RadarFunction.apply(self.radar_location, self.wavelength, self, x.contiguous(), out_cols, num_pad_frames,
                                        sigma)
        zr, zi = self.signal(x, num_pad_frames, None, sigma)
        return self._stft(zr, zi, out_cols)

    def _stft(self, zr, zi, out_cols=0, kernels_T=None):
        B, T = zr.shape
        F_ = T // self.hop_length + 1
        out = torch.empty((B, self.n_fft, out_cols if out_cols > 0 else F_), dtype=torch.float32, device=zr.device)
        if kernels_T is not None:
            check(L.load().sar_stft_kernels_fwd_f32(ptr(zr), ptr(zi), B, T, self.n_fft, self.hop_length, ptr(kernels_T[0]),
                                                    ptr(kernels_T[1]), out_cols, ptr(out), stream_ptr()), "sar_stft_kernels_fwd_f32")
            return out
        check(L.load().sar_stft_logmag_f32(ptr(zr), ptr(zi), B, T, self.n_fft, self.hop_length, ptr(self.window), out_cols,
                                           ptr(out), stream_ptr()), "sar_stft_logmag_f32")
        return out

"""Drop-in for the reference's layers/virtual_radar.py: `VirtualRadar(...)(x)` with the same constructor
signature (layers/virtual_radar.py:36-45) and output (B, n_fft, T//hop_length + 1) = log|STFT| of the
simulated radar return, frequency axis rolled by n_fft/2 (layers/virtual_radar.py:79-134).

The arithmetic runs in two HIP kernels of libsar_hip.so (csrc/radar.hip): sar_vr_signal_f32 (geometry, RCS,
phase, sum over edges and bodies) and sar_stft_logmag_f32 (reflect pad, periodic-Hann windowed DFT, log
magnitude, roll).  The window is the one nnAudio 0.1.1's STFT builds
(scipy.signal.get_window('hann', n_fft, fftbins=True)); no nnAudio dependency.

`wavelength` and `radar_location` are torch Parameters with the reference's names; training them
(train_wavelength / train_radar_location / train_stft_kernel) needs the backward kernels, which are not
built yet: requesting them raises NotImplementedError instead of silently freezing.
"""
import numpy as np
import torch

from sar_amd import _lib as L
from sar_amd._lib import check, ptr, stream_ptr

# layers/virtual_radar.py:10-13: the skeleton's bones without the hand-tip / thumb / foot-tip stubs the reference
# drops, written as chains of joints
_CHAINS = [(0, 1, 20, 2, 3), (20, 4, 5, 6, 7, 21), (7, 22), (20, 8, 9, 10, 11, 23), (11, 24), (0, 16), (0, 12, 13, 14, 15),
           (16, 17, 18, 19)]
edges = [(c[i], c[i + 1]) for c in _CHAINS for i in range(len(c) - 1)]


class VirtualRadar(torch.nn.Module):
    def __init__(self, edges=edges, wavelength=1e-3, radar_location=[0., 0., 0.], train_wavelength=False,
                 train_radar_location=False, train_stft_kernel=False, n_fft=256, hop_length=16, device='cuda:0'):
        super().__init__()
        if train_wavelength or train_radar_location or train_stft_kernel:
            raise NotImplementedError("trainable radar parameters need the VirtualRadar backward kernels (not built yet)")
        L.load()
        self.wavelength = torch.nn.Parameter(torch.as_tensor(wavelength, dtype=torch.float32), requires_grad=False)
        self.radar_location = torch.nn.Parameter(torch.as_tensor(radar_location, dtype=torch.float32),
                                                 requires_grad=False)
        self.src, self.dst = map(list, zip(*edges))
        self.n_fft, self.hop_length = n_fft, hop_length
        n = np.arange(n_fft, dtype=np.float64)
        self.register_buffer("window", torch.from_numpy((0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)),
                             persistent=False)
        self.register_buffer("_src", torch.tensor(self.src, dtype=torch.int32), persistent=False)
        self.register_buffer("_dst", torch.tensor(self.dst, dtype=torch.int32), persistent=False)
        self.to(device)

    def signal(self, x):
        """Complex baseband signal z[b, t] (layers/virtual_radar.py:93-123) as (z_re, z_im)."""
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[1] == 3
        x = x.contiguous()
        B, _, T, V, M = x.shape
        zr = torch.empty((B, T), dtype=torch.float32, device=x.device)
        zi = torch.empty_like(zr)
        check(L.load().sar_vr_signal_f32(ptr(x), B, T, V, M, ptr(self._src), ptr(self._dst), len(self.src),
                                         ptr(self.radar_location.data), ptr(self.wavelength.data.reshape(1)), ptr(zr),
                                         ptr(zi), stream_ptr()), "sar_vr_signal_f32")
        return zr, zi

    def forward(self, x, out_cols=0):
        """out_cols > 0 produces only the frames a nearest-neighbour F.interpolate(..., out_cols) would read
        (models/resnet.py:26 fused as a column select) -> (B, n_fft, out_cols)."""
        zr, zi = self.signal(x)
        B, T = zr.shape
        F_ = T // self.hop_length + 1
        out = torch.empty((B, self.n_fft, out_cols if out_cols > 0 else F_), dtype=torch.float32, device=x.device)
        check(L.load().sar_stft_logmag_f32(ptr(zr), ptr(zi), B, T, self.n_fft, self.hop_length, ptr(self.window), out_cols,
                                           ptr(out), stream_ptr()), "sar_stft_logmag_f32")
        return out

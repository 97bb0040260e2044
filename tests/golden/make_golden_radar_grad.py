"""Generates tests/golden/radar_reference_grads.npz: gradients of a seeded scalar loss w.r.t. `radar_location` and
`wavelength`, computed by torch autograd THROUGH THE REFERENCE's VirtualRadar.forward code (layers/virtual_radar.py,
imported from /root/reference, train_wavelength = train_radar_location = True) with the nnAudio-0.1.1 STFT
restatement of make_golden_radar.py injected.  Build container only.  Loss = sum(out * weights), weights =
numpy default_rng(7).standard_normal(out.shape) (float32).
Inputs: two SYNTHETIC clips with both bodies present (0.12 * standard_normal, clipped to [-1.1, 0.75], seed 11; stored
in the fixture).  Real NTU clips cannot be used for radar_location: wherever a body is absent or a frame is zero padded
the reference computes c = 0 -> rcs = 0 -> sqrt'(0) = inf times a zero inner derivative = NaN, so its radar_location
gradient is NaN on every such clip (recorded below for clip 0 / 2 of the bundled examples as `ntu_dloc_is_nan`).
The float64 columns repeat the computation with the module converted to double (the float32 result at small
lambda is dominated by the rounding of a 1e4..1e5 rad phase)."""
import os
import sys
import types

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)


class STFT(torch.nn.Module):
    """nnAudio 0.1.1 Spectrogram.STFT semantics (freq_scale='no', hann, center, reflect, Complex)."""

    def __init__(self, n_fft=2048, freq_bins=None, hop_length=512, window='hann', freq_scale='no', center=True,
                 pad_mode='reflect', trainable=False, output_format='Magnitude', device='cpu', **kw):
        super().__init__()
        assert freq_bins == n_fft and output_format == 'Complex'
        self.n_fft, self.stride = n_fft, hop_length
        n = np.arange(n_fft, dtype=np.float64)
        w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
        k = n[:, None]
        self.register_buffer("wsin", torch.tensor((w * np.sin(2 * np.pi * k * n / n_fft))[:, None, :], dtype=torch.float))
        self.register_buffer("wcos", torch.tensor((w * np.cos(2 * np.pi * k * n / n_fft))[:, None, :], dtype=torch.float))

    def forward(self, x):
        x = torch.nn.ReflectionPad1d(self.n_fft // 2)(x[:, None, :])
        spec_imag = torch.nn.functional.conv1d(x, self.wsin, stride=self.stride)
        spec_real = torch.nn.functional.conv1d(x, self.wcos, stride=self.stride)
        return torch.stack((spec_real, -spec_imag), -1)


mod = types.ModuleType("nnAudio"); sub = types.ModuleType("nnAudio.Spectrogram"); sub.STFT = STFT
mod.Spectrogram = sub
sys.modules["nnAudio"] = mod; sys.modules["nnAudio.Spectrogram"] = sub
sys.path.insert(0, "/root/reference")
from layers.virtual_radar import VirtualRadar  # noqa: E402  (reference code, executed not copied)

x_ntu = torch.from_numpy(np.load(os.path.join(here, "ntu_clips_0_2.npy")))
x = torch.from_numpy(np.clip(0.12 * np.random.default_rng(11).standard_normal((2, 3, 300, 25, 2)), -1.1, 0.75).astype(np.float32))
out = {"x": x.numpy()}
vr = VirtualRadar(wavelength=0.1, radar_location=[0.5, -1.0, 2.0], train_wavelength=True, train_radar_location=True, device='cpu')
vr(x_ntu).sum().backward()
out["ntu_dloc_is_nan"] = np.isnan(vr.radar_location.grad.numpy())
print("reference radar_location gradient on the bundled NTU clips:", vr.radar_location.grad.numpy())
for lam, loc in [(1e-1, [0.5, -1.0, 2.0]), (1e-2, [0.3, 0.2, -1.5]), (5e-4, [0., 0., 0.])]:
    key = "lam%g" % lam
    for dt, tag in [(torch.float32, "f32"), (torch.float64, "f64")]:
        vr = VirtualRadar(wavelength=lam, radar_location=loc, train_wavelength=True, train_radar_location=True, device='cpu')
        if dt == torch.float64:
            vr = vr.double()
        y = vr(x.to(dt))
        w = torch.from_numpy(np.random.default_rng(7).standard_normal(tuple(y.shape)).astype(np.float32)).to(dt)
        (y * w).sum().backward()
        out[key + "_dloc_" + tag] = vr.radar_location.grad.numpy().astype(np.float64)
        out[key + "_dlam_" + tag] = vr.wavelength.grad.numpy().astype(np.float64)
        print(key, tag, "dloc", out[key + "_dloc_" + tag], "dlam", out[key + "_dlam_" + tag])
np.savez_compressed(os.path.join(here, "radar_reference_grads.npz"), **out)

"""Generates tests/golden/radar_*.npz by running the REFERENCE's VirtualRadar.forward code
(layers/virtual_radar.py, imported from /root/reference) with a restatement of nnAudio-0.1.1's STFT
injected as `nnAudio.Spectrogram.STFT` (nnAudio itself is not installable here).  Build container only.
Inputs are clips 0 and 2 of the reference's bundled data/NTU_preprocessed_skeleton_examples.npy.
"""
import os
import sys
import types

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))


class STFT(torch.nn.Module):
    """nnAudio 0.1.1 Spectrogram.STFT semantics (freq_scale='no', hann, center, reflect, Complex)."""

    def __init__(self, n_fft=2048, freq_bins=None, hop_length=512, window='hann', freq_scale='no', center=True,
                 pad_mode='reflect', trainable=False, output_format='Magnitude', device='cpu', **kw):
        super().__init__()
        assert freq_bins == n_fft and output_format == 'Complex'
        self.n_fft, self.stride = n_fft, hop_length
        s = np.arange(0, n_fft, 1.)
        n = np.arange(n_fft, dtype=np.float64)
        w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)          # scipy get_window('hann', fftbins=True)
        wsin = np.empty((n_fft, 1, n_fft)); wcos = np.empty((n_fft, 1, n_fft))
        for k in range(n_fft):
            wsin[k, 0, :] = w * np.sin(2 * np.pi * k * s / n_fft)
            wcos[k, 0, :] = w * np.cos(2 * np.pi * k * s / n_fft)
        self.wsin = torch.tensor(wsin, dtype=torch.float)
        self.wcos = torch.tensor(wcos, dtype=torch.float)

    def forward(self, x):
        x = x[:, None, :]
        x = torch.nn.ReflectionPad1d(self.n_fft // 2)(x)
        spec_imag = torch.nn.functional.conv1d(x, self.wsin, stride=self.stride)
        spec_real = torch.nn.functional.conv1d(x, self.wcos, stride=self.stride)
        return torch.stack((spec_real, -spec_imag), -1)


mod = types.ModuleType("nnAudio"); sub = types.ModuleType("nnAudio.Spectrogram"); sub.STFT = STFT
mod.Spectrogram = sub
sys.modules["nnAudio"] = mod; sys.modules["nnAudio.Spectrogram"] = sub
sys.path.insert(0, "/root/reference")
from layers.virtual_radar import VirtualRadar  # noqa: E402  (reference code, executed not copied)

data = np.load("/root/reference/data/NTU_preprocessed_skeleton_examples.npy")
x = np.ascontiguousarray(data[[0, 2]]).astype(np.float32)             # (2,3,300,25,2)
np.save(os.path.join(here, "ntu_clips_0_2.npy"), x)
out = {}
for lam, loc in [(5e-4, [0., 0., 0.]), (1e-3, [0., 0., 0.]), (1e-1, [0., 0., 0.]), (1e-1, [0.5, -1.0, 2.0])]:
    vr = VirtualRadar(wavelength=lam, radar_location=loc, device='cpu')
    with torch.no_grad():
        y = vr(torch.from_numpy(x)).numpy()
    key = "lam%g_loc%g" % (lam, loc[2])
    out[key] = y.astype(np.float32)
    print(key, y.shape, y.min(), y.max())
np.savez_compressed(os.path.join(here, "radar_reference_outputs.npz"), **out)

# notebook known answers (virtual_radar_example.ipynb cell 4: NTU clip, upsampled): shape only needs T

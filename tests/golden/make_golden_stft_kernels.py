"""Generates tests/golden/stft_kernel_reference_grads.npz: outputs and gradients of the REFERENCE's VirtualRadar.forward code
(layers/virtual_radar.py, imported from /root/reference) with train_stft_kernel=True, i.e. with nnAudio's STFT(trainable=True)
whose conv1d Fourier kernels `wsin` / `wcos` are Parameters.  nnAudio is not installable here; the stand-in below restates
nnAudio 0.1.1's Spectrogram.STFT (freq_scale='no', hann, center, reflect, Complex) including its `trainable` switch
(`self.wsin = torch.nn.Parameter(self.wsin)`).  Build container only.

Loss = sum(out * weights), weights = default_rng(7).standard_normal(out.shape) (float32).  Inputs: the two synthetic clips of
radar_reference_grads.npz (both bodies present).  Two configurations: (n_fft, hop) = (64, 8) with the complete kernel
gradients, and the layer's defaults (256, 16) with the rows K256 of the kernel gradients (fixture size).  Every run is done
in float32 (as the reference runs) and with the module / input converted to float64 (`*_f64`); besides the analytic kernels
a PERTURBED kernel pair (what training produces) is used so that nothing relies on the DFT structure."""
import os
import sys
import types

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
K256 = [0, 1, 2, 64, 127, 128, 129, 200, 255]


class STFT(torch.nn.Module):
    def __init__(self, n_fft=2048, freq_bins=None, hop_length=512, window='hann', freq_scale='no', center=True,
                 pad_mode='reflect', trainable=False, output_format='Magnitude', device='cpu', **kw):
        super().__init__()
        assert freq_bins == n_fft and output_format == 'Complex'
        self.n_fft, self.stride = n_fft, hop_length
        n = np.arange(n_fft, dtype=np.float64)
        w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
        k = n[:, None]
        self.wsin = torch.tensor((w * np.sin(2 * np.pi * k * n / n_fft))[:, None, :], dtype=torch.float)
        self.wcos = torch.tensor((w * np.cos(2 * np.pi * k * n / n_fft))[:, None, :], dtype=torch.float)
        if trainable:
            self.wsin = torch.nn.Parameter(self.wsin)
            self.wcos = torch.nn.Parameter(self.wcos)

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

x = torch.from_numpy(np.load(os.path.join(here, "radar_reference_grads.npz"))["x"])
out = {"K256": np.array(K256)}
for n_fft, hop in [(64, 8), (256, 16)]:
    rng = np.random.default_rng(100 + n_fft)
    pert = [0.05 * rng.standard_normal((n_fft, 1, n_fft)).astype(np.float32) for _ in range(2)]
    if n_fft == 64:
        out["perturbation_seed64"] = np.array([100 + n_fft])
    for variant in ("analytic", "perturbed"):
        for dt, tag in [(torch.float32, "f32"), (torch.float64, "f64")]:
            vr = VirtualRadar(wavelength=0.1, radar_location=[0.5, -1.0, 2.0], train_wavelength=True, train_radar_location=True,
                              train_stft_kernel=True, n_fft=n_fft, hop_length=hop, device='cpu')
            names = [k for k, _ in vr.named_parameters()]
            assert "stft.wsin" in names and "stft.wcos" in names, names
            if variant == "perturbed":
                with torch.no_grad():
                    vr.stft.wsin += torch.from_numpy(pert[0])
                    vr.stft.wcos += torch.from_numpy(pert[1])
            if dt == torch.float64:
                vr = vr.double()
            y = vr(x.to(dt))
            w = torch.from_numpy(np.random.default_rng(7).standard_normal(tuple(y.shape)).astype(np.float32)).to(dt)
            (y * w).sum().backward()
            key = "n%d_%s_%s_" % (n_fft, variant, tag)
            rows = slice(None) if n_fft == 64 else K256
            out[key + "out"] = y.detach().numpy() if n_fft == 64 else y.detach().numpy()[:, K256]
            out[key + "dwsin"] = vr.stft.wsin.grad.numpy()[rows, 0].astype(np.float64)
            out[key + "dwcos"] = vr.stft.wcos.grad.numpy()[rows, 0].astype(np.float64)
            out[key + "dloc"] = vr.radar_location.grad.numpy().astype(np.float64)
            out[key + "dlam"] = vr.wavelength.grad.numpy().astype(np.float64)
            print(key, tuple(y.shape), "|dwsin| max %.3e |dwcos| max %.3e dloc %s dlam %s" % (
                np.abs(out[key + "dwsin"]).max(), np.abs(out[key + "dwcos"]).max(), out[key + "dloc"], out[key + "dlam"]))
np.savez_compressed(os.path.join(here, "stft_kernel_reference_grads.npz"), **out)
print("bytes:", os.path.getsize(os.path.join(here, "stft_kernel_reference_grads.npz")))

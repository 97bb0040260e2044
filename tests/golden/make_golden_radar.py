"""Generates tests/golden/radar_*.npz by running the REFERENCE's VirtualRadar.forward code
(layers/virtual_radar.py, imported from /root/reference) with a restatement of nnAudio-0.1.1's STFT
injected as `nnAudio.Spectrogram.STFT` (nnAudio itself is not installable here).  Build container only.  Every configuration is run twice: as the reference
runs it (float32) and with the module and the input converted to float64 (`*_f64` keys).
Inputs are clips 0 and 2 of the reference's bundled data/NTU_preprocessed_skeleton_examples.npy.
"""
import os
import sys
import types

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))


sys.path.insert(0, here)
from make_golden_radar_grad import STFT  # noqa: E402,F401  (the nnAudio-0.1.1 STFT restatement; also injects the nnAudio stub)

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
    # the same reference code in float64 on the same float32 inputs / parameters: the yardstick for "how far is a float32
    # evaluation allowed to be" at radar wavelengths, where the phase 4*pi*d/lambda ~ 1e4..1e5 rad loses 3-4 digits
    vr64 = VirtualRadar(wavelength=lam, radar_location=loc, device='cpu').double()
    with torch.no_grad():
        y64 = vr64(torch.from_numpy(x).double()).numpy()
    out[key + "_f64"] = y64.astype(np.float64)
    m32, m64 = np.exp(y.astype(np.float64)) - 1e-6, np.exp(y64) - 1e-6
    print(key, y.shape, y.min(), y.max(), "float32-vs-float64 |Z| distance / peak: %.3e" % (np.abs(m32 - m64).max() / m64.max()))
np.savez_compressed(os.path.join(here, "radar_reference_outputs.npz"), **out)

# notebook known answers (virtual_radar_example.ipynb cell 4: NTU clip, upsampled): shape only needs T

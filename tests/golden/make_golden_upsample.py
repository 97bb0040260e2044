"""Generates tests/golden/upsample_reference.npz with the REFERENCE's utils.Dataset.pad_frames (utils.py:134-140, called
unbound on a small stand-in object -- the method only uses self.T, self.sigma, self.num_pad_frames) on a seeded synthetic
clip, plus the reference VirtualRadar spectrogram of the up-sampled clip (nnAudio STFT restatement injected as in
make_golden_radar.py).  Build container only.  Stored: the raw clip, samples of the up-sampled tensor at a strided set of
output frames (the full tensor is 3 x 75000 x 25 x 2), and the 256 spectrogram columns models/resnet.py:26 would select."""
import os
import sys
import types

import numpy as np
import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
from make_golden_radar_grad import STFT  # noqa: E402  (also injects the nnAudio stub and /root/reference on sys.path)
import utils as ref_utils  # noqa: E402  (reference code, executed not copied)
from layers.virtual_radar import VirtualRadar  # noqa: E402

x = np.clip(0.12 * np.random.default_rng(21).standard_normal((3, 300, 25, 2)), -1.1, 0.75).astype(np.float32)
x[:, 260:] = 0                                   # zero-padded tail like real NTU clips
obj = types.SimpleNamespace(T=300, sigma=3, num_pad_frames=250)
up = ref_utils.Dataset.pad_frames(obj, x)          # float64 (3, 75000, 25, 2)
up32 = torch.from_numpy(up).type(torch.FloatTensor).numpy()
idx = np.unique(np.concatenate([np.arange(0, 75000, 997), np.arange(0, 300), np.arange(74700, 75000), [37499, 37500]]))
out = {"x": x, "frame_idx": idx, "up_frames": up32[:, idx]}
for lam in (1e-1, 5e-4):
    vr = VirtualRadar(wavelength=lam, radar_location=[0., 0., 0.], device='cpu')
    with torch.no_grad():
        spec = vr(torch.from_numpy(up32)[None]).numpy()[0]            # (256, 4688)
    F = spec.shape[1]
    cols = np.minimum(np.floor(np.arange(256, dtype=np.float32) * (np.float32(F) / np.float32(256))).astype(np.int64), F - 1)
    out["spec_lam%g" % lam] = spec[:, cols].astype(np.float32)
    # float64 evaluation of the same reference code on the same float32 up-sampled input (yardstick, see make_golden_radar.py)
    vr64 = VirtualRadar(wavelength=lam, radar_location=[0., 0., 0.], device='cpu').double()
    with torch.no_grad():
        spec64 = vr64(torch.from_numpy(up32)[None].double()).numpy()[0]
    out["spec_lam%g_f64" % lam] = spec64[:, cols].astype(np.float64)
    # ... and end to end in float64 (the up-sampled coordinates NOT rounded to float32): against this one the reference's
    # own float32 pipeline error includes its float32 cast of the up-sampled clip (utils.py:130)
    with torch.no_grad():
        spec64e = vr64(torch.from_numpy(up)[None]).numpy()[0]
    out["spec_lam%g_f64_e2e" % lam] = spec64e[:, cols].astype(np.float64)
    print("   end-to-end float64: reference float32 pipeline distance / peak %.3e"
          % (np.abs(np.exp(spec[:, cols].astype(np.float64)) - np.exp(spec64e[:, cols])).max() / np.exp(spec64e[:, cols]).max()))
    m32, m64 = np.exp(spec[:, cols].astype(np.float64)) - 1e-6, np.exp(spec64[:, cols]) - 1e-6
    print(lam, spec.shape, spec.min(), spec.max(), "float32-vs-float64 |Z| distance / peak: %.3e" % (np.abs(m32 - m64).max() / m64.max()))
np.savez_compressed(os.path.join(here, "upsample_reference.npz"), **out)

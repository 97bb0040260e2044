import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")): sys.path.insert(0, p)
import numpy as np, torch
from oracle import radar as R
from layers.virtual_radar import VirtualRadar
dev = torch.device("cuda:0")
clips = np.load(os.path.join(ROOT, "tests/golden/ntu_clips_0_2.npy"))
for lam, loc in [(1e-1, (0., 0., 0.)), (5e-4, (0., 0., 0.))]:
    vr = VirtualRadar(wavelength=lam, radar_location=list(loc), device=dev)
    zr, zi = vr.signal(torch.from_numpy(clips).to(dev)); torch.cuda.synchronize()
    zr, zi = zr.cpu().numpy(), zi.cpu().numpy()
    rr, ri = R.radar_signal(clips, wavelength=lam, radar_location=loc)
    scale = max(np.abs(rr).max(), np.abs(ri).max())
    e = np.abs(zr - rr) / scale
    print("lam", lam, "scale", scale, "max err re", e.max(), "at", np.unravel_index(e.argmax(), e.shape), "im", (np.abs(zi - ri) / scale).max())
    b, t = np.unravel_index(e.argmax(), e.shape)
    print("  kernel", zr[b, t], zi[b, t], "oracle", rr[b, t], ri[b, t])
    print("  first frames kernel", zr[0, :4], "oracle", rr[0, :4])
    print("  nonzero frames oracle", (np.abs(rr) > 0).sum(), "kernel", (np.abs(zr) > 0).sum())
print("log(1e-6f) on gpu:", torch.log(torch.tensor([1e-6], device=dev)).item(), "numpy:", float(np.log(np.float32(1e-6))))

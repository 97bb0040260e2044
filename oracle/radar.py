"""Oracle: VirtualRadar forward (reference layers/virtual_radar.py:79-134) and the
nnAudio-0.1.1 STFT it calls, restated in numpy float32 with an EXPLICIT operation order.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Why an explicit order: the phase psi = 4*pi*d/lambda reaches ~1e5 rad at the reference's
lambda = 5e-4 (models/resnet.py:20), so one ulp of the range d moves psi by ~0.02 rad.  The
reference's own result therefore depends on the summation order inside torch.norm (which differs
between its CPU and GPU kernels).  The oracle fixes the order

    d = sqrt((rx*rx + ry*ry) + rz*rz)      (each op one IEEE float32 rounding, no FMA)
    psi = (fl32(4*pi) * d) / lambda        (layers/virtual_radar.py:119; np.pi is a Python float,
                                            so torch multiplies by the float32 scalar 12.566371)

and the HIP kernel follows the same order bit for bit.

Third-party dependency: nnAudio (requirements.txt:2, unpinned; the notebook's pip log shows 0.1.1 and
the `device=` kwarg at layers/virtual_radar.py:76 exists only in < 0.2.0).  Its STFT (absent from
/root/reference) is restated from its published algorithm: freq_scale='no', window='hann' periodic
(scipy.signal.get_window('hann', n_fft, fftbins=True)), center=True with ReflectionPad1d(n_fft//2),
two conv1d with kernels wcos[k,n] = w[n]cos(2 pi k n/n_fft), wsin[k,n] = w[n]sin(2 pi k n/n_fft)
(float64 then cast to float32), stride hop, output_format='Complex' -> stack(real, -imag).

Parity status: PINNED against outputs of the reference's own forward() code run in the build
container with that STFT restatement injected as `nnAudio.Spectrogram.STFT`
(tests/golden/make_golden_radar.py -> tests/golden/radar_*.npz): tight at lambda = 0.1 where the
phase is well conditioned, statistically at lambda = 5e-4 / 1e-3; plus the notebook's printed shapes
and minimum (virtual_radar_example.ipynb cells 2-7).  The STFT restatement itself is cross-checked
against numpy.fft.
"""
import numpy as np

# layers/virtual_radar.py:10-13
EDGES = [(0, 1), (1, 20), (20, 2), (2, 3), (20, 4), (4, 5), (5, 6), (6, 7), (7, 21), (7, 22), (20, 8), (8, 9),
         (9, 10), (10, 11), (11, 23), (11, 24), (0, 16), (0, 12), (12, 13), (13, 14), (14, 15), (16, 17), (17, 18),
         (18, 19)]

f32 = np.float32


def hann_periodic(n_fft):
    """scipy.signal.get_window('hann', n_fft, fftbins=True) in float64."""
    n = np.arange(n_fft, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)


def radar_signal(x, edges=EDGES, wavelength=1e-3, radar_location=(0., 0., 0.)):
    """layers/virtual_radar.py:93-123.  x: (B,3,T,V,M) float32 -> z_re, z_im (B,T) float32."""
    x = np.asarray(x, dtype=f32)
    src, dst = map(list, zip(*edges))
    loc = np.asarray(radar_location, dtype=f32)
    lam = f32(wavelength)
    S = x[:, :, :, src]                     # (B,3,T,E,M)
    D = x[:, :, :, dst]
    L = loc[:, None, None, None]
    rev = np.abs(S - L)                     # :96-98
    rx, ry, rz = rev[:, 0], rev[:, 1], rev[:, 2]
    dist = np.sqrt((rx * rx + ry * ry) + rz * rz)                       # :99   (B,T,E,M)
    Av = L - ((S + D) / f32(2))             # :101-102
    Bv = D - S                              # :103
    dot = (Av[:, 0] * Bv[:, 0] + Av[:, 1] * Bv[:, 1]) + Av[:, 2] * Bv[:, 2]
    nA = np.sqrt((Av[:, 0] * Av[:, 0] + Av[:, 1] * Av[:, 1]) + Av[:, 2] * Av[:, 2])
    nB = np.sqrt((Bv[:, 0] * Bv[:, 0] + Bv[:, 1] * Bv[:, 1]) + Bv[:, 2] * Bv[:, 2])
    theta = np.arccos(dot / (nA * nB + f32(1e-6)))                      # :104-105
    phi = np.arcsin((loc[1] - S[:, 1]) / (np.sqrt(rx * rx + ry * ry) + f32(1e-6)))   # :106-108
    SD = S - D
    ce = np.sqrt((SD[:, 0] * SD[:, 0] + SD[:, 1] * SD[:, 1]) + SD[:, 2] * SD[:, 2])   # (B,T,E,M)
    acc = np.zeros(ce[:, :, 0].shape, dtype=f32)
    for e in range(ce.shape[2]):            # mean over the EDGE axis, keepdim (:110-112), sequential order
        acc = acc + ce[:, :, e]
    c = (acc / f32(ce.shape[2]))[:, :, None, :]
    c = c * c                               # :113
    st, ct = np.sin(theta), np.cos(theta)
    sp, cp = np.sin(phi), np.cos(phi)
    den = ((st * st) * (cp * cp) + (st * st) * (sp * sp)) + c * (ct * ct)   # :114-116
    rcs = (f32(np.pi) * c) / (den * den)
    amp = np.sqrt(rcs)                      # :118
    psi = (f32(4 * np.pi) * dist) / lam     # :119
    re = amp * np.cos(psi)                  # :121-122
    im = amp * np.sin(psi)
    B, T = re.shape[0], re.shape[1]
    zr = np.zeros((B, T), dtype=f32)
    zi = np.zeros((B, T), dtype=f32)
    for e in range(re.shape[2]):            # sum over dims [2,3] (:123), e-major then m, sequential
        for m in range(re.shape[3]):
            zr = zr + re[:, :, e, m]
            zi = zi + im[:, :, e, m]
    return zr.astype(f32), zi.astype(f32)


def stft_kernels(n_fft):
    """nnAudio 0.1.1 create_fourier_kernels(freq_scale='no', window='hann'): float64 -> float32."""
    n = np.arange(n_fft, dtype=np.float64)
    k = np.arange(n_fft, dtype=np.float64)[:, None]
    w = hann_periodic(n_fft)
    wcos = (w * np.cos(2 * np.pi * k * n / n_fft)).astype(f32)
    wsin = (w * np.sin(2 * np.pi * k * n / n_fft)).astype(f32)
    return wcos, wsin


def stft_complex(u, n_fft=256, hop=16):
    """nnAudio 0.1.1 STFT.forward for one real signal u (B,T), output_format='Complex':
    returns (real, -imag), each (B, n_fft, F) with F = T//hop + 1."""
    u = np.asarray(u, dtype=f32)
    B, T = u.shape
    pad = n_fft // 2
    assert T > pad, "ReflectionPad1d needs T > n_fft/2"
    up = np.pad(u, ((0, 0), (pad, pad)), mode="reflect")
    F_ = (up.shape[1] - n_fft) // hop + 1
    frames = np.stack([up[:, f * hop:f * hop + n_fft] for f in range(F_)], axis=1)   # (B,F,n_fft)
    wcos, wsin = stft_kernels(n_fft)
    real = np.einsum("bfn,kn->bkf", frames, wcos).astype(f32)
    imag = np.einsum("bfn,kn->bkf", frames, wsin).astype(f32)
    return real, -imag


def log_spectrogram(zr, zi, n_fft=256, hop=16):
    """layers/virtual_radar.py:124-133: complex STFT of z = zr + j zi, log magnitude, fftshift."""
    a_re, a_im = stft_complex(zr, n_fft, hop)
    b_re, b_im = stft_complex(zi, n_fft, hop)
    Z_re = a_re - b_im                      # :126-129
    Z_im = a_im + b_re
    mag = np.sqrt(Z_re * Z_re + Z_im * Z_im)
    out = np.log(mag + f32(1e-6))
    return np.roll(out, n_fft // 2, axis=1).astype(f32), mag.astype(f32)


def virtual_radar(x, edges=EDGES, wavelength=1e-3, radar_location=(0., 0., 0.), n_fft=256, hop=16):
    zr, zi = radar_signal(x, edges, wavelength, radar_location)
    return log_spectrogram(zr, zi, n_fft, hop)[0]


def nearest_columns(F_, out_cols):
    """F.interpolate(x, out_cols) nearest along the frame axis (models/resnet.py:26):
    source column of output column j = min(floor(j * fl32(F/out_cols)), F-1)."""
    scale = f32(F_) / f32(out_cols)
    j = np.arange(out_cols, dtype=f32)
    return np.minimum(np.floor(j * scale).astype(np.int64), F_ - 1)


def spectrogram_torch(x, loc, lam, edges=EDGES, n_fft=256, hop=16, out_cols=0, dtype=None, wcos=None, wsin=None):
    """Differentiable restatement of the whole layer (layers/virtual_radar.py:93-133) in torch, for the gradients of
    radar_location / wavelength (autograd of the reference when train_* are set, main_spectrogram.py:133-136).
    x (B,3,T,V,M) tensor; loc (3,) and lam () tensors (requires_grad as wanted); float64 unless dtype is given.
    out_cols > 0 applies the nearest column select of models/resnet.py:26.
    wcos / wsin: (n_fft, n_fft) [k, n] tensors replacing the analytic Fourier kernels -- nnAudio's STFT(trainable=True)
    holds them as Parameters of shape (n_fft, 1, n_fft) (layers/virtual_radar.py:71-76 train_stft_kernel) and convolves the
    reflect-padded signal with them; pass leaves with requires_grad to get their gradients."""
    import torch
    dtype = dtype or torch.float64
    x = torch.as_tensor(x).to(dtype)
    src, dst = map(list, zip(*edges))
    S, D = x[:, :, :, src], x[:, :, :, dst]                         # (B,3,T,E,M)
    L = loc.to(dtype)[:, None, None, None]
    rev = (S - L).abs()
    rx, ry, rz = rev[:, 0], rev[:, 1], rev[:, 2]
    dist = torch.sqrt(rx * rx + ry * ry + rz * rz)
    Av = L - (S + D) / 2
    Bv = D - S
    dot = (Av * Bv).sum(1)
    nA, nB = torch.sqrt((Av * Av).sum(1)), torch.sqrt((Bv * Bv).sum(1))
    theta = torch.acos(dot / (nA * nB + 1e-6))
    phi = torch.asin((loc.to(dtype)[1] - S[:, 1]) / (torch.sqrt(rx * rx + ry * ry) + 1e-6))
    c = torch.sqrt(((S - D) ** 2).sum(1)).mean(dim=2, keepdim=True) ** 2
    st, ct, sp, cp = torch.sin(theta), torch.cos(theta), torch.sin(phi), torch.cos(phi)
    den = st * st * cp * cp + st * st * sp * sp + c * ct * ct
    amp = torch.sqrt(np.pi * c / (den * den))
    psi = 4 * np.pi * dist / lam.to(dtype)
    zr, zi = (amp * torch.cos(psi)).sum(dim=[2, 3]), (amp * torch.sin(psi)).sum(dim=[2, 3])       # (B,T)
    pad = n_fft // 2
    w = torch.from_numpy(hann_periodic(n_fft)).to(dtype)
    n = torch.arange(n_fft, dtype=dtype)
    ang = 2 * np.pi * n[:, None] * n[None, :] / n_fft                                          # [k, n]
    wcos = w * torch.cos(ang) if wcos is None else wcos.to(dtype).reshape(n_fft, n_fft)
    wsin = w * torch.sin(ang) if wsin is None else wsin.to(dtype).reshape(n_fft, n_fft)

    def frames(u):
        up = torch.nn.functional.pad(u[:, None, :], (pad, pad), mode="reflect")[:, 0]
        return up.unfold(1, n_fft, hop)                                                         # (B,F,n_fft)
    fr, fi = frames(zr), frames(zi)
    # Z[k,f] = sum_n (zr + j zi)[n] w[n] e^{-j 2 pi k n / N}
    Z_re = torch.einsum("bfn,kn->bkf", fr, wcos) + torch.einsum("bfn,kn->bkf", fi, wsin)
    Z_im = torch.einsum("bfn,kn->bkf", fi, wcos) - torch.einsum("bfn,kn->bkf", fr, wsin)
    out = torch.log(torch.sqrt(Z_re * Z_re + Z_im * Z_im) + 1e-6)
    out = torch.roll(out, n_fft // 2, dims=1)
    if out_cols > 0:
        out = out[:, :, torch.from_numpy(nearest_columns(out.shape[2], out_cols).astype(np.int64))]
    return out


def radar_param_grads(x, weights, loc, lam, **kw):
    """d/d(loc, lam) of sum(spectrogram * weights) in float64 -> (dloc (3,), dlam ()) numpy."""
    import torch
    loc_t = torch.tensor(np.asarray(loc, dtype=np.float64), requires_grad=True)
    lam_t = torch.tensor(float(lam), dtype=torch.float64, requires_grad=True)
    out = spectrogram_torch(x, loc_t, lam_t, **kw)
    (out * torch.as_tensor(weights).to(out.dtype)).sum().backward()
    return loc_t.grad.numpy(), lam_t.grad.numpy()


def pad_frames(data, num_pad_frames=250, sigma=3):
    """utils.py:134-140 (Dataset.pad_frames) with the reference's own scipy calls: Gaussian smoothing along T then cubic
    interpolation to num_pad_frames*T frames.  data (..., T, V, M) float32 with T at axis -3 -> float32 (the
    reference's `.type(torch.FloatTensor)`)."""
    from scipy.interpolate import interp1d
    from scipy.ndimage import gaussian_filter1d
    data = np.asarray(data)
    T = data.shape[-3]
    f = interp1d(np.linspace(0, 1, T), gaussian_filter1d(data, sigma, axis=-3), 'cubic', axis=-3)
    return f(np.linspace(0, 1, num_pad_frames * T)).astype(f32)

"""CPU: the VirtualRadar oracle (oracle/radar.py) is pinned against
 (1) outputs of the reference's own layers/virtual_radar.py forward code (tests/golden/make_golden_radar.py),
 (2) numpy.fft for the nnAudio-0.1.1 STFT restatement,
 (3) the notebook's printed known answers (virtual_radar_example.ipynb cells 2-7): output shapes
     (256, T//16+1) and the exact minimum -13.815511 = log(1e-6) on all-zero frames."""
import os

import numpy as np
import pytest

from oracle import radar as R


@pytest.fixture(scope="module")
def clips(golden_dir):
    return np.load(os.path.join(golden_dir, "ntu_clips_0_2.npy"))


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "radar_reference_outputs.npz"))


def _mag(logspec):
    return np.exp(np.roll(logspec, -128, axis=1)) - 1e-6


@pytest.mark.parametrize("lam,loc,key,tol", [(1e-1, (0., 0., 0.), "lam0.1_loc0", 1e-4),
                                             (1e-1, (0.5, -1.0, 2.0), "lam0.1_loc2", 1e-4)])
def test_matches_reference_where_phase_is_well_conditioned(clips, gold, lam, loc, key, tol):
    """|Z| within 1e-4 (relative to the spectrogram peak) of the reference's own forward()."""
    zr, zi = R.radar_signal(clips, wavelength=lam, radar_location=loc)
    out, mag = R.log_spectrogram(zr, zi)
    ref = gold[key]
    assert out.shape == ref.shape == (2, 256, 19)
    refmag = _mag(ref)
    assert np.abs(mag - refmag).max() / refmag.max() < tol
    assert np.array_equal(out[ref == ref.min()], ref[ref == ref.min()])     # exact log(1e-6) frames


@pytest.mark.parametrize("lam,key", [(5e-4, "lam0.0005_loc0"), (1e-3, "lam0.001_loc0")])
def test_matches_reference_statistically_at_radar_wavelengths(clips, gold, lam, key):
    """At lambda <= 1e-3 the phase 4*pi*d/lambda ~ 1e4..1e5 rad amplifies one ulp of torch.norm's summation
    order into ~1e-2 rad, so agreement with the reference's CPU run is statistical, not elementwise."""
    out = R.virtual_radar(clips, wavelength=lam)
    ref = gold[key]
    refmag, mag = _mag(ref), _mag(out)
    assert np.abs(mag - refmag).max() / refmag.max() < 2e-3
    assert np.median(np.abs(mag - refmag) / (refmag + 1e-3 * refmag.max())) < 5e-4
    assert np.corrcoef(out.ravel(), ref.ravel())[0, 1] > 0.99999
    assert ref.min() == out.min() == np.float32(np.log(np.float32(1e-6)))


@pytest.mark.parametrize("lam", [5e-4, 1e-3, 1e-1])
def test_oracle_pinned_to_the_float64_reference_run(clips, gold, lam):
    """The oracle is pinned at every wavelength by the reference's own forward code evaluated in float64 (`*_f64` in the
    fixture, tests/golden/make_golden_radar.py): it must be no further from that truth than 1.5x the reference's own
    float32 run (measured 0.84x / 1.05x / 0.91x) -- i.e. the oracle is as good a float32 evaluation of
    layers/virtual_radar.py:93-133 as the reference itself, also where elementwise agreement between two float32
    evaluations is impossible."""
    out = R.virtual_radar(clips, wavelength=lam)
    m64 = np.exp(np.roll(gold["lam%g_loc0_f64" % lam], -128, axis=1)) - 1e-6
    d_ref = np.abs(_mag(gold["lam%g_loc0" % lam]).astype(np.float64) - m64).max() / m64.max()
    d_ora = np.abs(_mag(out).astype(np.float64) - m64).max() / m64.max()
    assert d_ora <= 1.5 * d_ref and d_ora < 5e-4, (d_ora, d_ref)


def test_stft_restatement_equals_windowed_fft():
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal((2, 300)).astype(np.float32), rng.standard_normal((2, 300)).astype(np.float32)
    _, mag = R.log_spectrogram(u, v)
    z = np.pad((u + 1j * v).astype(np.complex128), ((0, 0), (128, 128)), mode="reflect")
    w = R.hann_periodic(256)
    ref = np.stack([np.abs(np.fft.fft(z[:, f * 16:f * 16 + 256] * w, axis=1)) for f in range(19)], axis=2)
    assert np.abs(mag - ref).max() / ref.max() < 1e-6


@pytest.mark.parametrize("T,frames", [(3438 * 16, 3439), (5120 * 16, 5121), (300, 19)])
def test_notebook_output_shapes(T, frames):
    """virtual_radar_example.ipynb cells 2-4 print (256, 3439) / (256, 5121) / (256, 10313): F = T//hop + 1."""
    z = np.zeros((1, T), dtype=np.float32)
    out, _ = R.log_spectrogram(z, z)
    assert out.shape == (1, 256, frames)
    assert out.min() == out.max() == np.float32(-13.815511)        # the notebook's printed minimum


def test_nearest_column_select_matches_torch_interpolate():
    import torch
    for F_ in (19, 4688, 300, 257):
        x = torch.arange(F_, dtype=torch.float32).view(1, 1, 1, F_).expand(1, 1, 256, F_)
        y = torch.nn.functional.interpolate(x, 256)[0, 0, 0].numpy().astype(np.int64)
        assert np.array_equal(R.nearest_columns(F_, 256), y)


def test_zero_body_and_degenerate_geometry_are_finite(clips):
    x = clips.copy()
    x[:, :, :, :, 1] = 0                                          # empty second body
    zr, zi = R.radar_signal(x, wavelength=5e-4)
    assert np.isfinite(zr).all() and np.isfinite(zi).all()


@pytest.mark.parametrize("lam,loc", [(1e-1, [0.5, -1.0, 2.0]), (1e-2, [0.3, 0.2, -1.5]), (5e-4, [0., 0., 0.])])
def test_parameter_gradients_match_the_reference_autograd(golden_dir, lam, loc):
    """d sum(out*w) / d(radar_location, wavelength): the float64 torch restatement against autograd through the
    reference's own layers/virtual_radar.py run in float64 (tests/golden/make_golden_radar_grad.py)."""
    g = np.load(os.path.join(golden_dir, "radar_reference_grads.npz"))
    x = g["x"]
    w = np.random.default_rng(7).standard_normal((2, 256, 19)).astype(np.float32)
    # the reference's parameters are float32 tensors (converted to double for the float64 run)
    dloc, dlam = R.radar_param_grads(x, w, np.asarray(loc, dtype=np.float32).astype(np.float64), float(np.float32(lam)))
    key = "lam%g" % lam
    # 2e-5: the reference's double run keeps its float32-rounded DFT kernels and constants (measured 3e-7 .. 6e-6)
    assert np.abs(dloc - g[key + "_dloc_f64"]).max() <= 2e-5 * np.abs(g[key + "_dloc_f64"]).max()
    assert abs(dlam - g[key + "_dlam_f64"]) <= 2e-5 * abs(g[key + "_dlam_f64"])
    assert g["ntu_dloc_is_nan"].all()     # the reference's own gradient is NaN on clips with an absent body / zero padding


@pytest.mark.parametrize("n_fft,hop", [(64, 8), (256, 16)])
@pytest.mark.parametrize("variant", ["analytic", "perturbed"])
def test_trainable_stft_kernel_gradients_match_the_reference_autograd(golden_dir, n_fft, hop, variant):
    """train_stft_kernel=True (layers/virtual_radar.py:71-76): output and d sum(out*w) / d(wsin, wcos, radar_location,
    wavelength) of the float64 restatement with the Fourier kernels as leaves, against autograd through the reference's
    own layer run in float64 with nnAudio's trainable kernels (tests/golden/make_golden_stft_kernels.py) -- at the analytic
    kernels and at a perturbed pair."""
    import torch
    g = np.load(os.path.join(golden_dir, "stft_kernel_reference_grads.npz"))
    x = np.load(os.path.join(golden_dir, "radar_reference_grads.npz"))["x"]
    wcos32, wsin32 = R.stft_kernels(n_fft)
    if variant == "perturbed":
        rng = np.random.default_rng(100 + n_fft)
        wsin32 = wsin32 + 0.05 * rng.standard_normal((n_fft, 1, n_fft)).astype(np.float32)[:, 0]
        wcos32 = wcos32 + 0.05 * rng.standard_normal((n_fft, 1, n_fft)).astype(np.float32)[:, 0]
    wcos = torch.tensor(wcos32.astype(np.float64), requires_grad=True)
    wsin = torch.tensor(wsin32.astype(np.float64), requires_grad=True)
    loc = torch.tensor(np.asarray([0.5, -1.0, 2.0], dtype=np.float32).astype(np.float64), requires_grad=True)
    lam = torch.tensor(float(np.float32(0.1)), dtype=torch.float64, requires_grad=True)
    out = R.spectrogram_torch(x, loc, lam, n_fft=n_fft, hop=hop, wcos=wcos, wsin=wsin)
    w = torch.from_numpy(np.random.default_rng(7).standard_normal(tuple(out.shape)).astype(np.float32)).double()
    (out * w).sum().backward()
    key = "n%d_%s_f64_" % (n_fft, variant)
    rows = slice(None) if n_fft == 64 else g["K256"]
    ref_out = g[key + "out"]
    got_out = out.detach().numpy() if n_fft == 64 else out.detach().numpy()[:, g["K256"]]
    assert np.abs(got_out - ref_out).max() <= 1e-6 * np.abs(ref_out).max()
    for name, got in (("dwsin", wsin.grad.numpy()[rows]), ("dwcos", wcos.grad.numpy()[rows]), ("dloc", loc.grad.numpy()),
                      ("dlam", lam.grad.numpy())):
        ref = g[key + name]
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), name


def test_pad_frames_restatement_equals_the_reference(golden_dir):
    """utils.py:134-140 (Gaussian smoothing + cubic up-sampling x250): the oracle's scipy calls against the reference's
    own Dataset.pad_frames output at ~1 000 of the 75 000 frames (tests/golden/make_golden_upsample.py)."""
    g = np.load(os.path.join(golden_dir, "upsample_reference.npz"))
    up = R.pad_frames(g["x"], 250, 3)
    assert up.shape == (3, 75000, 25, 2) and up.dtype == np.float32
    assert np.array_equal(up[:, g["frame_idx"]], g["up_frames"])

"""Box calibration: what the MI355X this process landed on sustains (csrc/box_probe.hip through the C ABI).

`measure(dev)` runs, in about two seconds, a dense loop of each matrix instruction the conv kernels use -- fp32
(v_mfma_f32_32x32x2_f32), fp16 and bf16 (v_mfma_f32_32x32x16_*) -- with the shader clock the chip held under each, and a
float4 copy of 1 GiB.  bench.py reports every leg's roofline fraction against the guide's peaks (`frac`) AND against these
(`frac_of_box`): boxes of the pool differ by ~10 %, and without the second figure a slow box reads as a regression
(VERDICT r05 weak #2).  Measurement only: nothing in the training path calls this module."""
import torch

from . import _lib

KINDS = {"f32": 0, "f16": 1, "bf16": 2}
# iterations per launch: ~120-130 ms at the guide's peaks (157.3 TF fp32, 2.5 PF fp16 / bf16) on 512 workgroups -- long enough for the
# power management to settle on the clock it HOLDS under the load (a 15 ms burst runs at 2.27-2.39 GHz; the split kernels' own
# timelines see 1.6-1.9 GHz inside a training step)
ITERS = {"f32": 72000, "f16": 144000, "bf16": 144000}


def _timed(fn, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]     # ms; the launches run on torch's current stream, which is what the events see


def mfma(kind, dev, blocks=512, reps=4, iters=None):
    """(sustained TFLOP/s, held clock in GHz) of one matrix instruction: `reps` back-to-back launches after one untimed launch,
    the MEDIAN launch; the clock is the median over workgroups of s_memtime cycles / s_memrealtime ticks x 100 MHz in the last."""
    lib = _lib.load()
    iters = iters or ITERS[kind]
    sink = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    clocks = torch.zeros(blocks, 2, dtype=torch.int32, device=dev)
    k = KINDS[kind]

    def launch():
        _lib.check(lib.sar_box_mfma(k, blocks, iters, sink.data_ptr(), clocks.data_ptr(), _lib.stream_ptr()), "sar_box_mfma")

    launch()
    ms = sorted(_timed(launch, reps))[reps // 2]
    c = clocks.cpu().to(torch.float64)
    ghz = float((c[:, 0] / c[:, 1].clamp(min=1)).median()) * 0.1
    return lib.sar_box_mfma_flops(k, blocks, iters) / (ms * 1e-3) / 1e12, ghz


def copy(dev, nbytes=1 << 30, reps=5):
    """float4 copy of `nbytes` (read + write = 2 x nbytes of HBM traffic) in GB/s, the median of `reps` launches"""
    lib = _lib.load()
    n = nbytes // 4
    src = torch.ones(n, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)

    def launch():
        _lib.check(lib.sar_box_copy_f32(src.data_ptr(), dst.data_ptr(), n, _lib.stream_ptr()), "sar_box_copy_f32")

    launch()
    ms = sorted(_timed(launch, reps))[reps // 2]
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def measure(dev, quick=False):
    """the `box` object of the bench line"""
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = {}
    for kind in ("f32", "f16", "bf16"):
        tf, ghz = mfma(kind, dev, reps=2 if quick else 3, iters=ITERS[kind] // (64 if quick else 1))
        out["%s_mfma_tflops" % kind] = round(tf, 1)
        out["%s_mfma_clock_ghz" % kind] = round(ghz, 3)
    out["copy_gbps"] = round(copy(dev, (1 << 26) if quick else (1 << 30), reps=3 if quick else 5), 0)
    torch.cuda.synchronize()
    out["probe_s"] = round(time.perf_counter() - t0, 2)
    return out

"""Per-kernel timing with HIP events on the launch stream (used by bench.py for the roofline object).

Kernels are launched on torch's current stream, so torch.cuda.Event brackets exactly the kernel.
Disabled (zero overhead) unless a KernelTimer is installed with `install()`.
"""
import contextlib

import torch

_active = None


class KernelTimer:
    def __init__(self):
        self.records = []   # (tag, flops, bytes, start, end)

    @contextlib.contextmanager
    def region(self, tag, flops=0.0, nbytes=0.0):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        yield
        e1.record()
        self.records.append((tag, flops, nbytes, e0, e1))

    def summary(self):
        """tag -> dict(calls, ms, flops, bytes); call after torch.cuda.synchronize()."""
        out = {}
        for tag, fl, nb, e0, e1 in self.records:
            d = out.setdefault(tag, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
        return out


def install(timer):
    global _active
    _active = timer


def region(tag, flops=0.0, nbytes=0.0):
    if _active is None:
        return contextlib.nullcontext()
    return _active.region(tag, flops, nbytes)

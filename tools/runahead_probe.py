#!/usr/bin/env python3
"""How far may the host run ahead of the GPU?  Times N un-synchronised fp32 train steps at bs = 64 for several
SAR_MAX_STEPS_IN_FLIGHT settings (0 = unbounded) and prints the caching allocator's state (reserved bytes, retries)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from sar_amd.stgcn import STGCN  # noqa: E402
from sar_amd.train import Trainer, synthetic_clips  # noqa: E402

dev = torch.device("cuda", 0)
mfma = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
eng = STGCN(device=dev, mfma=mfma)
tr = Trainer(eng, batch_size=64)
batches = [synthetic_clips(64, dev, seed=i) for i in range(4)]
for depth in (2, 0, 1, 4, 2):
    tr.run_ahead.set_depth(depth)
    for i in range(5):
        tr.step(*batches[i % 4])
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    r0 = torch.cuda.memory_stats().get("num_alloc_retries", 0)
    t0 = time.perf_counter()
    for i in range(steps):
        tr.step(*batches[i % 4])
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print("%s steps=%d in_flight=%d: %.1f clips/s (%.2f ms/step; host issued everything after %.2f s) peak reserved %.1f GB, "
          "peak allocated %.1f GB, alloc retries %d" % (mfma, steps, depth, 64 * steps / dt, dt / steps * 1e3, t_issue,
          st["reserved_bytes.all.peak"] / 1e9, st["allocated_bytes.all.peak"] / 1e9, st.get("num_alloc_retries", 0) - r0), flush=True)
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""bench.py -- ST-GCN training throughput on MI355X (BASELINE.json metric:
"NTU-xsub clips/sec training (ST-GCN, bs=64/GPU)").

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step = one full train step of the hot path on one synthetic batch resident in HBM: forward (data_bn,
10 ST-GCN blocks, head), softmax-CE, backward of everything, gradient all-reduce (N>1), Nesterov SGD.
Workload = BASELINE.json configs[1]: fp32, synthetic NTU-xsub clips (3,300,25,2), 60 classes, bs=64/GPU.
Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     -- the dominant kernel family (9x1 temporal-conv GEMMs), algorithmic FLOPs / HIP-event time
  cpu_baseline -- the CPU oracle (torch CPU ops, "port") timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 (v_mfma_f32_32x32x16_bf16)
FLOP_PER_CLIP_TRAIN = 102.56e9     # 3 x 34.19 GFLOP
BYTES_PER_CLIP_TRAIN_BF16 = 427.3e6   # block-fused algorithmic HBM traffic per clip and train step, bf16 storage
# What the build EXECUTES per clip and train step: SURVEY's 102.56 GFLOP prices the dense 'nkctv,kvw->nctw' einsum (432
# MMAC / sequence); the kernels apply the 73 non-zeros of A on the Cin side (16.8 MMAC / sequence), so the executed, useful work is
# (1 939.7 + 16.8 + 6 082.6 + 92.2) MMAC / sequence x 2 sequences x 2 FLOP x 3 (fwd + data grad + weight grad) = 97.58 GFLOP.
EXECUTED_FLOP_PER_CLIP_TRAIN = (1939.7 + 16.8 + 6082.6 + 92.2) * 1e6 * 2 * 2 * 3
PEAK_FP64_VALU_TFLOPS = 78.6       # MI355X: packed-free fp64 vector FMA, half the fp32 vector rate of MI355X_MICROARCH.md (157.3 TF)


FAMILY_PREFIX = {"fp32": ("conv_gemm_kernel<1, ",), "bf16": ("conv_gemm_cn8_kernel<", "conv_gemm_cn8_db_kernel<", "conv_gemm_cn8_dma_kernel<"),
                 "bf16_operands": ("conv_gemm_bf16_kernel<",), "pathB": ("conv2d_gemm_kernel",), "pathB_f32_split": ("conv2d_split_kernel<",),
                 "pathB_pad250": ("conv2d_gemm_kernel",),
                 "f32_split": ("conv_gemm_split_kernel<",), "f32_split_bf16x6": ("conv_gemm_split_kernel<",)}
PROFILE_CLOCK = {}   # mode -> the clock the dominant family held in the newest committed profile (GRBM_GUI_ACTIVE pass), or None
BOX = {}        # the box calibration of this process (sar_amd/box.py), filled by main() before the first leg

# What the fields of the line mean (the line itself carries numbers and identifiers of at most 120 characters; DESIGN.md sections 4 and 7
# carry the arithmetic).  Emitted once, in front of the legs.
LEGEND = {
    "value": "whole-job clips/s of the leg: K timed train steps bracketed by barrier + synchronize, MAX over ranks, inputs resident in HBM",
    "roofline.frac": "dominant kernel family: algorithmic FLOPs (split legs: EXECUTED matrix FLOPs) per launch / HIP-event launch time / guide peak",
    "roofline.frac_of_box": "the same numerator over what THIS box sustained in the `box` probe (box_ref names the probe figure)",
    "roofline.timing": "in_step = HIP events inside the timed step; +wgrad_stream = weight gradients overlap on a second stream (inflates it)",
    "roofline.isolated": "the same family over 3 untimed steps with the side stream off: kernel quality without overlap",
    "roofline.traffic": "HBM bytes per launch, (2*FETCH_SIZE+WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of the build named in traffic_src",
    "roofline.clock_ghz": "shader clock held under the dominant family: GRBM_GUI_ACTIVE / 8 XCDs / dispatch time, own PMC pass of the same build",
    "box": "sar_amd/box.py: dense v_mfma loops (f32 32x32x2, f16/bf16 32x32x16; ~125 ms each, held clock = s_memtime/s_memrealtime) + 1 GiB float4 copy",
    "peaks": "MI355X_MICROARCH.md: fp32 MFMA 157.3 TF, fp16/bf16 MFMA 2500 TF dense, HBM 8000 GB/s",
    "f32_split": "fp32 storage, statistics, epilogues, optimizer and parity tolerances; every GEMM product = 3 exact fp16-term products (f16x3a)",
    "bf16": "bf16 CN8 activations in HBM and bf16 MFMA operands; fp32 accumulation, BatchNorm statistics, master weights, optimizer",
    "sustained": "100 further steps behind the contract's K, same protocol",
    "train step": "forward + loss + backward + gradient exchange (N > 1) + optimizer (ST-GCN: Nesterov SGD; Path B: radar -> 256x256 image -> Adam)",
    "summary": "LAST key: legs = {leg: [clips/s, ms/step, roofline.frac, frac_of_box]}; box = [f32 TF, f16 TF, f16 GHz, bf16 TF, copy GB/s]",
}


def measured_traffic(mode="fp32"):
    """(bytes per launch, 'file @ commit') of the dominant kernel family from the newest COMMITTED PMC profile of this same
    workload (profiles/rNN_*kernel_summary.json, written by tools/summarize_profiles.py from separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 on gfx950).  It is a figure of the profiled
    build, not of this run: the file and the commit it was taken at are named next to it, and a summary whose kernels
    are not this mode's kernel family is refused (None)."""
    import glob
    pat = {"pathB": "r[0-9][0-9]_pathB_kernel_summary.json", "bf16": "r[0-9][0-9]_bf16_kernel_summary.json",
           "pathB_pad250": "r[0-9][0-9]_pathB_pad250_kernel_summary.json",
           "f32_split": "r[0-9][0-9]_f32split_kernel_summary.json", "f32_split_bf16x6": "r[0-9][0-9]_f32split_bf16x6_kernel_summary.json",
           "pathB_f32_split": "r[0-9][0-9]_pathB_f32split_kernel_summary.json",
           "bf16_operands": "r[0-9][0-9]_bf16_operands_kernel_summary.json"}.get(mode, "r[0-9][0-9]_kernel_summary.json")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    names = [r.get("kernel", "") for r in d.get("kernels", [])]
    if not any(n.startswith(FAMILY_PREFIX[mode]) for n in names):
        return None, None
    PROFILE_CLOCK[mode] = d.get("dominant_family_clock_ghz")
    return d.get("dominant_family_hbm_bytes_per_launch"), "%s @ %s" % (os.path.basename(files[-1]), d.get("commit", "commit not recorded"))


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(batch=64, fallback_clips=8, budget_s=80.0, classes=60):
    """The CPU oracle (oracle/stgcn.py: torch CPU ops, fp32) -- fwd + bwd + Nesterov SGD -- on the bs = `batch` synthetic batch,
    all physical cores.  A BOUNDED sample: two short steps on `fallback_clips` clips (thread pool / allocator warm-up; the
    second predicts the cost of a full-batch step), then ONE timed full-batch step when it is predicted to fit `budget_s`
    (the GPU box's 128 cores take ~55 s for it), else up to 5 timed small-batch steps; the sample string says which.  (Rounds
    1-3 timed 1 + 2 full-batch steps: 165 s of a 190 s bench run.)  The oracle is used here ONLY as the reported CPU baseline."""
    import torch
    from oracle import stgcn as O
    cores = physical_cores()
    torch.set_num_threads(cores)
    p = O.init_params(classes, seed=0)
    vel = {}

    def one(x, y):
        _, _, grads, new, _ = O.loss_and_grads(p, x, y)
        O.sgd_nesterov_step(p, grads, vel, 0.1)
        p.update(new)

    xs, ys = O.synthetic_batch(fallback_clips, seed=0, T=300, num_classes=classes)
    one(xs, ys)                            # thread-pool / allocator warm-up
    t0 = time.time()
    one(xs, ys)
    small_rate = fallback_clips / (time.time() - t0)
    if 0.75 * batch / small_rate <= budget_s:     # a full batch runs ~1.4x the small batch's rate (better core utilisation)
        x, y = O.synthetic_batch(batch, seed=0, T=300, num_classes=classes)
        t0 = time.time()
        one(x, y)
        dt, n = time.time() - t0, 1
        clips, what = batch, "2 warm-up steps on %d clips + 1 timed step" % fallback_clips
    else:
        t0 = time.time()
        n = 0
        while n < 5 and (n == 0 or time.time() - t0 < 25.0):
            one(xs, ys)
            n += 1
        dt = time.time() - t0
        clips, what = fallback_clips, "FALLBACK (bs=%d would exceed %d s): %d timed steps" % (batch, budget_s, n)
    return {"value": round(clips * n / dt, 3), "unit": "clips/s", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
            "sample": "%s, fwd+bwd+SGD, %d-clip fp32 batch, torch CPU ops, %d threads" % (what, clips, cores)}


def cpu_baseline_spectrogram(sample_clips, budget_s=12.0, num_pad_frames=0):
    """The CPU oracle of Path B (oracle/radar.py numpy + oracle/resnet.py torch CPU ops): [the loader's up-sampling,
    utils.py:134-140 restated in oracle/radar.py:pad_frames, when num_pad_frames > 0,] spectrogram, resnet18 fwd + bwd, Adam
    -- as the reported CPU baseline only."""
    import numpy as np
    import torch
    from oracle import radar as RO
    from oracle import resnet as RN
    cores = physical_cores()
    torch.set_num_threads(cores)
    p = RN.init_params(60, num_filters=64, seed=0)
    g = torch.Generator().manual_seed(0)
    x = (0.12 * torch.randn((sample_clips, 3, 300, 25, 2), generator=g)).clamp_(-1.1, 0.75)
    y = torch.randint(0, 60, (sample_clips,), generator=g)
    T = 300 * num_pad_frames if num_pad_frames else 300
    cols = RO.nearest_columns(T // 16 + 1, 256)
    state = {}

    def one():
        xin = x.numpy()
        if num_pad_frames:       # the reference's loader step, per clip (utils.py:134-140)
            xin = np.stack([RO.pad_frames(c, num_pad_frames) for c in xin])
        spec = RO.virtual_radar(xin, wavelength=5e-4)
        img = torch.from_numpy(np.ascontiguousarray(spec[:, :, cols]))[:, None]
        _, _, grads, new, _ = RN.loss_and_grads(p, img, y)
        RN.adam_step(p, grads, state, 1e-3)
        p.update(new)

    one()
    t0 = time.time()
    n = 0
    while True:
        one()
        n += 1
        if time.time() - t0 > budget_s or n >= 5:
            break
    dt = time.time() - t0
    return {"value": round(sample_clips * n / dt, 3), "unit": "clips/s", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
            "sample": "%d timed steps, %sVirtualRadar + resnet18 fwd+bwd+Adam, %d clips, numpy/scipy/torch CPU ops, %d threads"
                      % (n, "x%d up-sampling + " % num_pad_frames if num_pad_frames else "", sample_clips, cores)}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) through torch.distributed.run
    and relay their output.  Called BEFORE this process imports torch or touches the GPU (a process that has initialised
    the GPU must not be replaced, and the ranks must be fresh processes)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def rank_setup(args):
    """(rank, world, device) of this process; joins the process group when world > 1 (backend nccl = RCCL over xGMI;
    SAR_BENCH_SHARE_GPU=1 -- a test switch for one-GPU boxes -- puts every rank on cuda:0 over gloo)."""
    import torch
    from sar_amd.train import init_distributed
    world = int(os.environ.get("WORLD_SIZE", "1"))
    share = os.environ.get("SAR_BENCH_SHARE_GPU", "0") == "1"
    local_rank = 0 if share else int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank, world = init_distributed(dev, backend="gloo" if share else "nccl")    # SAR_FORCE_DDP=1: also for ONE rank
    return rank, world, dev


def dist_info():
    """what the line reports about the LIVE communicator (not about the arguments)"""
    import torch.distributed as dist
    from sar_amd.train import force_ddp
    live = dist.is_available() and dist.is_initialized()
    return {"rccl_ranks": dist.get_world_size() if live else 1, "dist_backend": dist.get_backend() if live else None,
            "forced_ddp": bool(force_ddp() and live)}


def gather_over_ranks(value, dev):
    """every rank's `value` (a float) as a list, on every rank"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [value]
    on_cpu = dist.get_backend() == "gloo"
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if on_cpu else dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


class Leg:
    """The timing protocol shared by every workload: W untimed steps (and, for the sustained legs, untimed load until
    `warm_seconds` have passed -- the bf16 step draws the most power and the clocks settle after the first second), then
    EXACTLY K timed steps bracketed by barrier + torch.cuda.synchronize() on both sides; the job's time is the MAX over
    ranks; every rank's own time and the mean time of the gradient collective are reported next to it."""

    def __init__(self, world, dev):
        self.world, self.dev = world, dev

    def sync(self):
        import torch
        import torch.distributed as dist
        from sar_amd.train import ddp_active
        torch.cuda.synchronize()
        if ddp_active():
            dist.barrier()
            torch.cuda.synchronize()

    def run(self, step, steps, warmup, warm_seconds=0.0, trainer=None):
        import torch
        import torch.distributed as dist
        for i in range(warmup):
            step(i)
        warm_done = 0.0
        if warm_seconds > 0:
            # the same number of untimed steps on every rank (the step contains the collective): timed on 10 steps, MAX-agreed
            self.sync()
            t0 = time.perf_counter()
            for i in range(10):
                step(i)
            self.sync()
            done = 10
            for _ in range(6):      # the first steps of a cold process are slow: re-estimate until the time has really passed
                left = warm_seconds - (time.perf_counter() - t0)
                left = min(gather_over_ranks(left, self.dev))            # every rank agrees on when to stop
                if left <= 0:
                    break
                per = (time.perf_counter() - t0) / done
                n = int(max(gather_over_ranks(float(max(1, int(left / max(per, 1e-4)))), self.dev)))
                for i in range(n):
                    step(i)
                done += n
                self.sync()
            warm_done = time.perf_counter() - t0
        if trainer is not None:
            trainer.comm_events = []
        self.sync()
        t0 = time.perf_counter()
        last = None
        for i in range(steps):
            last = step(i)
        self.sync()
        dt = time.perf_counter() - t0
        per_rank = gather_over_ranks(dt, self.dev)
        comm_ms = None
        if trainer is not None:
            ev, trainer.comm_events = trainer.comm_events, None
            comm_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1) if ev else 0.0
            comm_ms = max(gather_over_ranks(comm_ms, self.dev))
        return {"dt": max(per_rank), "per_rank_ms": [round(t / max(steps, 1) * 1e3, 3) for t in per_rank],
                "allreduce_ms": None if comm_ms is None else round(comm_ms, 4), "warm_s": round(warm_done, 2), "last": last}


def stgcn_leg(args, mfma, steps, warmup, warm_seconds, rank, world, dev, isolated_pass, first_run=False, instrument_steps=0,
              classes=None, stream="joint", sustained_steps=0):
    """ST-GCN train step (main_gnn.py:219-239) in one arithmetic mode; returns the result dict of rank 0 (None elsewhere).
    instrument_steps = 0: the per-kernel HIP events are recorded inside the timed region (the headline leg, as in every
    round); > 0: the timed region runs bare and that many extra steps are instrumented afterwards (long secondary legs:
    two event records per launch are not free on a 7-15 ms step)."""
    import torch
    import torch.distributed as dist
    from sar_amd import profiler
    from sar_amd.stgcn import STGCN
    from sar_amd.train import Trainer, synthetic_clips
    classes = classes or args.classes
    bone = None
    if stream == "bone":       # data_gen/gen_bone_data.py:7-41, applied on the fly in the data_bn prologue (config 5's second stream)
        from sar_amd.bone import NTU_BONE_PAIRS
        bone = NTU_BONE_PAIRS
    eng = STGCN(num_classes=classes, device=dev, seed=0, mfma=mfma, bone_pairs=bone)  # identical init on every rank
    trainer = Trainer(eng, batch_size=args.batch, world_size=world)
    nb = 4     # a few distinct batches resident in HBM, cycled (per-rank seeds: each rank trains on its own shard)
    batches = [synthetic_clips(args.batch, dev, seed=1000 * rank + i, num_classes=classes) for i in range(nb)]
    leg = Leg(world, dev)

    def step(i):
        return trainer.step(*batches[i % nb])[1]

    first = None
    if first_run:     # the rate of a cool GPU (what a short run sees): 10 warm-up steps (allocator, lazy initialisation) + 20
        r = leg.run(step, 20, 10)   # timed steps, before the sustained leg
        first = args.batch * world * 20 / r["dt"]
    timer = profiler.KernelTimer()
    for i in range(warmup):
        step(i)
    warm_s = leg.run(step, 0, 0, warm_seconds)["warm_s"] if warm_seconds > 0 else 0.0   # untimed load only
    if instrument_steps == 0:
        profiler.install(timer)
    res = leg.run(step, steps, 0, 0.0, trainer)
    profiler.install(None)
    ksteps = steps
    if instrument_steps > 0:
        profiler.install(timer)
        leg.run(step, instrument_steps, 0)
        profiler.install(None)
        ksteps = instrument_steps
    res["warm_s"] = warm_s
    dt = res["dt"]
    loss_val = float(res["last"].item())
    assert loss_val == loss_val, "loss is NaN"
    sustained = None
    if sustained_steps > 0:      # the same protocol over a longer region, right behind the contract's K steps (bare: no per-kernel events)
        r2 = leg.run(step, sustained_steps, 0, 0.0, None)
        sustained = {"steps": sustained_steps, "value": round(args.batch * world * sustained_steps / r2["dt"], 2),
                     "ms_per_step": round(r2["dt"] / sustained_steps * 1e3, 3)}
    # The timed region is the production schedule: weight-gradient kernels run on a second stream and overlap the main
    # chain, which inflates the HIP-event duration of whatever they overlap.  A short untimed pass with that stream off
    # measures the dominant family in isolation (kernel quality); both are reported.
    iso = None
    if eng._side is not None and isolated_pass:      # every rank: the step contains the collective
        side, eng._side = eng._side, None
        iso_timer = profiler.KernelTimer()
        torch.cuda.synchronize()
        profiler.install(iso_timer)
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        profiler.install(None)
        eng._side = side
        iso = iso_timer.summary()
    out = None
    if rank == 0:
        clips = args.batch * world * steps
        value = clips / dt
        summ = timer.summary()
        fam = [k for k in summ if k.startswith("gemm_temporal9")]
        ms = sum(summ[k]["ms"] for k in fam)
        fl = sum(summ[k]["flops"] for k in fam)
        by = sum(summ[k]["bytes"] for k in fam)
        calls = sum(summ[k]["calls"] for k in fam)
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic, traffic_src = measured_traffic(mfma)
        bf16 = mfma in ("bf16", "bf16_operands")
        split = mfma in ("f32_split", "f32_split_bf16x6")
        nprod = 3 if mfma == "f32_split" else 6
        # every prose field of a leg is at most 120 characters (the driver's parser cuts there); LEGEND explains the vocabulary
        what = {"fp32": "ST-GCN fp32", "f32_split": "ST-GCN f32_split (f16x3a)", "f32_split_bf16x6": "ST-GCN f32_split (bf16x6)",
                "bf16": "ST-GCN bf16 (CN8)", "bf16_operands": "ST-GCN bf16-operand"}[mfma]
        overl = eng._side is not None
        common = {"timing": "in_step" + ("+wgrad_stream" if overl else ""), "launches": calls, "avg_launch_ms": round(ms / max(calls, 1), 4),
                  "algorithmic_bytes_per_launch": int(by / max(calls, 1)), "traffic": traffic, "traffic_src": traffic_src,
                  "clock_ghz": PROFILE_CLOCK.get(mfma)}      # held under this family in the profiled build's GRBM_GUI_ACTIVE pass (traffic_src)
        if split:
            # fp32 results on the fp16 / bf16 matrix pipe: the roof is that pipe's, priced on the MFMA FLOPs the kernels EXECUTE
            # (products per fp32 product x the 10 tap slots per 9 taps) -- never against the fp32 MFMA peak
            exe = achieved * nprod * 10.0 / 9.0
            roof = {"bound": "mfma", "kernel": "conv_gemm_split_kernel 9-tap fwd+dgrad (%s)" % ("f16x3a" if nprod == 3 else "bf16x6"),
                    "achieved": round(exe, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(exe / PEAK_BF16_MFMA_TFLOPS, 4),
                    "executed_over_algorithmic_flops": round(nprod * 10.0 / 9.0, 3), "fp32_equivalent_tflops": round(achieved, 2),
                    "step_fp32_equivalent_tflops": round(value / world * EXECUTED_FLOP_PER_CLIP_TRAIN / 1e12, 2),
                    "step_frac_of_hbm_roof": round(value / world * 2 * BYTES_PER_CLIP_TRAIN_BF16 / 8.0e12, 4), **common}
            box_key, box_num = ("f16_mfma_tflops" if nprod == 3 else "bf16_mfma_tflops"), exe
        elif bf16:
            # which roof binds the family: the larger of (algorithmic FLOPs / 2.5 PFLOP/s) and (algorithmic bytes / 8 TB/s)
            gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            step_bytes = BYTES_PER_CLIP_TRAIN_BF16 if mfma == "bf16" else 2 * BYTES_PER_CLIP_TRAIN_BF16
            mfma_bound = fl / (PEAK_BF16_MFMA_TFLOPS * 1e12) >= by / 8.0e12
            roof = {"step_frac_of_hbm_roof": round(value / world * step_bytes / 8.0e12, 4), "step_algorithmic_bytes_per_clip": step_bytes,
                    "bound": "mfma" if mfma_bound else "hbm", "kernel": "conv_gemm_cn8*_kernel 9-tap fwd+dgrad",
                    "achieved": round(achieved, 1) if mfma_bound else round(gbs, 1),
                    "peak": PEAK_BF16_MFMA_TFLOPS if mfma_bound else 8000.0, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                    "frac": round(achieved / PEAK_BF16_MFMA_TFLOPS, 4) if mfma_bound else round(gbs / 8000.0, 4),
                    "hbm_gbps": round(gbs, 1), "hbm_frac": round(gbs / 8000.0, 4), "mfma_tflops": round(achieved, 1), **common}
            box_key, box_num = ("bf16_mfma_tflops", achieved) if mfma_bound else ("copy_gbps", gbs)
        else:
            roof = {"bound": "mfma", "kernel": "conv_gemm_kernel<TEMPORAL,9> fwd+dgrad", "achieved": round(achieved, 2),
                    "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                    "step_frac_of_fp32_roof": round(value / world * FLOP_PER_CLIP_TRAIN / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4),
                    "step_frac_executed_flops": round(value / world * EXECUTED_FLOP_PER_CLIP_TRAIN / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4), **common}
            box_key, box_num = "f32_mfma_tflops", achieved
        if BOX.get(box_key):
            roof["frac_of_box"] = round(box_num / BOX[box_key], 4)
            roof["box_ref"] = box_key
        out = {
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, **dist_info(), "grad_buckets": trainer.buckets_last_step,
            "steps": steps, "warmup": warmup, "warm_s": res["warm_s"], "ms_per_step": round(dt / steps * 1e3, 3),
            "per_rank_ms": res["per_rank_ms"], "allreduce_ms": res["allreduce_ms"], "dtype": "bf16" if bf16 else "f32",
            "config": {"workload": "%s train step, synthetic NTU-xsub clips (3,300,25,2)%s, %d classes, bs=%d/GPU"
                                   % (what, " BONE stream" if bone else "", classes, args.batch),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world},
            "roofline": roof, "final_loss": round(loss_val, 5),
        }
        if first is not None:
            out["first_run_value"] = round(first, 2)
        if sustained is not None:
            out["sustained"] = sustained
        detail = {"kernel_ms_per_step": {k: round(v["ms"] / ksteps, 3) for k, v in sorted(summ.items())},
                  "kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in sorted(summ.items()) if v["ms"] > 0},
                  "executed_flops_per_clip": EXECUTED_FLOP_PER_CLIP_TRAIN, "survey_flops_per_clip": FLOP_PER_CLIP_TRAIN}
        if iso is not None:     # the same family over 3 untimed steps with the weight-gradient side stream off: kernel quality
            ims = sum(iso[k]["ms"] for k in fam if k in iso)
            ifl = sum(iso[k]["flops"] for k in fam if k in iso)
            icalls = sum(iso[k]["calls"] for k in fam if k in iso)
            itf = ifl / (ims * 1e-3) / 1e12 if ims > 0 else 0.0
            iexe = itf * (nprod * 10.0 / 9.0 if split else 1.0)
            ipeak = PEAK_BF16_MFMA_TFLOPS if (split or bf16) else PEAK_FP32_MFMA_TFLOPS
            roof["isolated"] = {"tflops": round(iexe, 1), "frac": round(iexe / ipeak, 4), "avg_launch_ms": round(ims / max(icalls, 1), 4)}
            if split:
                roof["isolated"]["fp32_equivalent_tflops"] = round(itf, 2)
            detail["kernel_ms_per_step_isolated"] = {k: round(v["ms"] / 3, 3) for k, v in sorted(iso.items())}
        if split:
            detail["split_kernels"] = sorted(k for k in summ if k.endswith("_split"))
            detail["fp32_kernels_left"] = sorted(k for k in summ if not k.endswith("_split"))
            nonsplit = sum(v["ms"] for k, v in summ.items() if not k.endswith("_split")) / ksteps
            out["non_split_kernel_ms_per_step"] = round(nonsplit, 3)
        out["_detail"] = detail
    del trainer, eng, batches    # (the cached device blocks stay with torch's allocator: the next leg reuses them instead of paying
    return out                   #  hipMalloc for every tensor of its first steps)


def spectrogram_leg(args, steps, warmup, warm_seconds, rank, world, dev, instrument_steps=0, num_pad_frames=None, mfma="fp32"):
    """Path B: VirtualRadar (signal + STFT/log-magnitude/column select) -> resnet18 fwd+bwd -> Adam; bs = --batch per GPU
    (configs[3] uses 32)."""
    import torch
    import torch.distributed as dist
    from sar_amd import profiler
    from sar_amd.train import SpectrogramTrainer, synthetic_clips
    from models.resnet import Model
    bs = 32 if args.batch == 64 else args.batch
    pad = args.num_pad_frames if num_pad_frames is None else num_pad_frames
    model = Model(num_classes=args.classes, num_filters=64, device=dev, num_pad_frames=pad, mfma=mfma)
    split = mfma != "fp32"
    # the product step of main_spectrogram.py.  SAR_PATHB_GRAPH=1: the step as ONE hipGraph launch (SpectrogramTrainer(graph=True)) --
    # it paid (5.94 -> 4.91 ms) while the host needed 6 ms to issue the split step's ~300 launches; since sar_amd/_lib.py asks torch's
    # C module for the raw stream (a third of the host's time per step was torch.cuda.current_stream()) the eager step is GPU-bound
    # again and faster than the replay (4.64 vs 4.87 ms): off by default.
    use_graph = os.environ.get("SAR_PATHB_GRAPH", "0") == "1" and world == 1
    trainer = SpectrogramTrainer(model, 1e-3, world_size=world, graph=use_graph)
    batches = [synthetic_clips(bs, dev, seed=1000 * rank + i, num_classes=args.classes) for i in range(4)]
    leg = Leg(world, dev)

    def step(i):
        return trainer.step(*batches[i % 4], 1e-3)[1]

    for i in range(warmup):
        step(i)
    warm_s = leg.run(step, 0, 0, warm_seconds)["warm_s"] if warm_seconds > 0 else 0.0
    graphs = None
    if os.environ.get("SAR_BENCH_GRAPH", "0") == "1" and world == 1:
        # one captured hipGraph per resident batch: at bs = 32 the ~250 launches of a step are short enough for the
        # host launch path to show (experiment switch; the kernels read lr / step counters from device memory)
        graphs = []
        leg.sync()
        for i in range(4):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                step(i)
            graphs.append(g)
    timer = profiler.KernelTimer()
    ksteps = steps
    if use_graph and instrument_steps == 0:
        instrument_steps = 3      # a replayed step passes no profiler region: the per-kernel table comes from eager steps behind the timed ones
    if graphs is None:
        if instrument_steps == 0:
            profiler.install(timer)
        res = leg.run(step, steps, 0, 0.0, trainer)
        profiler.install(None)
        if instrument_steps > 0:
            trainer.graph = False
            profiler.install(timer)
            leg.run(step, instrument_steps, 0)
            profiler.install(None)
            trainer.graph = use_graph
            ksteps = instrument_steps
    else:
        res = leg.run(lambda i: graphs[i % 4].replay(), steps, 0, 0.0, None)
        res["last"] = step(0)
    dt, loss = res["dt"], res["last"]
    out = None
    if rank == 0:
        summ = timer.summary()
        fam = [k for k in summ if k.startswith("conv2d_3x3")]
        ms = sum(summ[k]["ms"] for k in fam)
        fl = sum(summ[k]["flops"] for k in fam)
        calls = sum(summ[k]["calls"] for k in fam)
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        value = bs * world * steps / dt
        tmode = ("pathB_f32_split" if split else "pathB") if not pad else (None if split else "pathB_pad250")   # (no profile set of the split pad250 leg)
        traffic, traffic_src = measured_traffic(tmode) if tmode else (None, None)
        nprod = 6 if mfma.endswith("bf16x6") else 3
        front = "VirtualRadar%s" % (" (x%d up-sampling)" % pad if pad else "")
        net = "resnet18 f32_split (%s)" % ("bf16x6" if nprod == 6 else "f16x3a") if split else "resnet18 fp32"
        out = {
            "metric": "spectrogram clips/sec training (VirtualRadar + resnet18, bs=%d/GPU)" % bs,
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, **dist_info(),
            "steps": steps, "warmup": warmup, "warm_s": warm_s,
            "ms_per_step": round(dt / steps * 1e3, 3), "per_rank_ms": res["per_rank_ms"], "allreduce_ms": res["allreduce_ms"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s -> %s train step, synthetic NTU clips, %d classes, bs=%d/GPU"
                                   % (front, net, args.classes, bs),
                       "global_batch": bs * world, "parallelism": "dp%d" % world, "hip_graph_step": bool(use_graph)},
            "final_loss": round(float(loss.item()), 5),
        }
        common = {"timing": "in_step+wgrad_stream", "traffic": traffic, "traffic_src": traffic_src,
                  "clock_ghz": PROFILE_CLOCK.get(tmode) if tmode else None}
        if not split:
            out["roofline"] = {"bound": "mfma", "kernel": "conv2d_gemm_kernel 3x3 fwd+dgrad+wgrad", "achieved": round(achieved, 2),
                               "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                               "launches": calls, "avg_launch_ms": round(ms / max(calls, 1), 4),
                               "step_frac_of_fp32_roof": round(value / world * 13.6e9 / (PEAK_FP32_MFMA_TFLOPS * 1e12), 4), **common}
            box_key, box_num = "f32_mfma_tflops", achieved
        else:
            # the launches on csrc/conv2d_split.hip: roofline on the fp16 / bf16 matrix pipe with the FLOPs actually executed
            # (3 products per fp32 product -- 6 for bf16x6 --, forward / data gradient x 10 / 9: nine taps in five k-steps)
            fs = [k for k in summ if "_split" in k]
            ms_s = sum(summ[k]["ms"] for k in fs)
            fl_s = sum(summ[k]["flops"] for k in fs)
            exe = sum(summ[k]["flops"] * nprod * (1.0 if "wgrad" in k else 10.0 / 9.0) for k in fs) / (ms_s * 1e-3) / 1e12 if ms_s > 0 else 0.0
            calls_s = sum(summ[k]["calls"] for k in fs)
            out["roofline"] = {
                "bound": "mfma", "kernel": "conv2d split kernels 3x3/s1 fwd+dgrad%s" % ("+wgrad" if any("wgrad" in k for k in fs) else ""),
                "achieved": round(exe, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(exe / PEAK_BF16_MFMA_TFLOPS, 4),
                "fp32_equivalent_tflops": round(fl_s / (ms_s * 1e-3) / 1e12, 1) if ms_s > 0 else 0.0,
                "step_fp32_equivalent_tflops": round(value / world * 13.6e9 / 1e12, 2),
                "launches": calls_s, "avg_launch_ms": round(ms_s / max(calls_s, 1), 4), "split_kernels_ms_per_step": round(ms_s / ksteps, 3),
                "fp32_conv_kernels_left_ms_per_step": round(sum(summ[k]["ms"] for k in summ if k.startswith("conv2d") and "_split" not in k) / ksteps, 3),
                **common}
            box_key, box_num = ("bf16_mfma_tflops" if nprod == 6 else "f16_mfma_tflops"), exe
        if BOX.get(box_key):
            out["roofline"]["frac_of_box"] = round(box_num / BOX[box_key], 4)
            out["roofline"]["box_ref"] = box_key
        small = [k for k, v in summ.items() if v["calls"] and v["ms"] / v["calls"] < 0.025]
        out["launch_regions_per_step"] = round(sum(v["calls"] for v in summ.values()) / ksteps, 1)
        out["regions_under_25us_per_step"] = round(sum(summ[k]["calls"] for k in small) / ksteps, 1)
        out["_detail"] = {"kernel_ms_per_step": {k: round(v["ms"] / ksteps, 3) for k, v in sorted(summ.items())},
                          "kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in sorted(summ.items()) if v["ms"] > 0}}
        if pad:
            # The front end that only this variant has: per-clip smoothing + spline pieces, then the radar signal of 300 x pad
            # frames with every frame's 150 coordinates evaluated from the cubic pieces in float64.  Its binding unit is the
            # vector ALU: 48 (edge, body) terms per frame, each with IEEE square roots / divisions and an accurately reduced
            # sin / cos of a ~1e5 rad phase; the float64 spline evaluation is 450 DFMA per frame beside it.
            sig, prep = summ.get("radar_signal_upsampled"), summ.get("radar_upsample_prepare")
            if sig and sig["ms"] > 0:
                frames = bs * 300 * pad
                terms = frames * 24 * 2
                per = sig["ms"] / sig["calls"]
                rr = {"bound": "valu", "kernel": "vr_signal_fast_kernel<SPLINE>", "avg_launch_ms": round(per, 4),
                      "prepare_avg_launch_ms": round(prep["ms"] / prep["calls"], 4) if prep and prep["calls"] else None,
                      "achieved": round(terms / (per * 1e-3) / 1e9, 2), "unit": "G (edge, body) terms/s",
                      "spline_f64_tflops": round(sig["flops"] / sig["calls"] / (per * 1e-3) / 1e12, 3),
                      "share_of_step": round((per + (prep["ms"] / prep["calls"] if prep and prep["calls"] else 0.0)) / (dt / steps * 1e3), 4)}
                import glob
                vf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pathB_pad250_valu.json")))
                if vf:      # SQ_INSTS_VALU of the PROFILED build (its own rocprofv3 --pmc pass), not of this run
                    vk = json.load(open(vf[-1]))["kernels"].get("vr_signal_fast_kernel<1>")
                    if vk:
                        ghz = vk.get("clock_ghz", 2.1)
                        rr.update(peak="1024 SIMDs x launch time x %.2f GHz / 4 cycles per wave64 VALU instruction" % ghz, clock_ghz=ghz,
                                  valu_wave_instructions_per_launch=vk["valu_wave_instructions_per_launch"],
                                  frac=round(vk["valu_wave_instructions_per_launch"] / (1024 * per * 1e-3 * ghz * 1e9 / 4), 4),
                                  frac_src="SQ_INSTS_VALU from %s over this run's launch time" % os.path.basename(vf[-1]))
                out["radar_roofline"] = rr
    del trainer, model, batches
    return out


def strip_detail(leg, details, name):
    """the per-kernel tables of a leg leave the line (they were 1-2 KB per leg and pushed the legs' numbers out of the driver's tail):
    `--detail` prints them under `detail`, and every run writes them to gpurun_out/bench_detail.json when that directory exists"""
    if leg is not None and "_detail" in leg:
        details[name] = leg.pop("_detail")
    return leg


def summary_of(head, sec, world):
    """the compact object at the END of the line: every leg's [clips/s, ms/step, roofline.frac, frac_of_box] and
    the box calibration -- what a reader of the line's last kilobyte needs"""
    legs = {"fp32": head}
    legs.update(sec or {})
    out = {"legs": {}, "box": [BOX.get(k) for k in ("f32_mfma_tflops", "f16_mfma_tflops", "f16_mfma_clock_ghz", "bf16_mfma_tflops", "copy_gbps")]}
    for n, r in legs.items():
        rf = r.get("roofline", {})
        out["legs"][n] = [r["value"], r["ms_per_step"], rf.get("frac"), rf.get("frac_of_box")]
        if isinstance(r.get("bf16"), dict):
            out["legs"][n + ".bf16"] = [r["bf16"]["value"], r["bf16"]["ms_per_step"], None, None]
    if world > 1:
        out["per_rank_ms"] = {n: [round(t, 2) for t in r["per_rank_ms"]] for n, r in legs.items()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU (reference --batch-size)")
    ap.add_argument("--classes", type=int, default=60)
    ap.add_argument("--workload", default="stgcn", choices=["stgcn", "spectrogram"],
                    help="stgcn = BASELINE.json's headline (configs[1]); spectrogram = Path B (configs[3] shape per GPU)")
    ap.add_argument("--num-pad-frames", type=int, default=0,
                    help="spectrogram workload: GPU-side frame up-sampling factor (the reference's loader default is 250)")
    ap.add_argument("--mfma", default="fp32", choices=["fp32", "bf16", "bf16_operands", "f32_split", "f32_split_bf16x6"],
                    help="fp32 = BASELINE.json configs[1] (default, the headline); bf16 = configs[2]: bf16 activations in HBM (CN8 "
                         "layout) + bf16 MFMA operands, fp32 accumulation / BatchNorm statistics / master weights; bf16_operands = "
                         "the round-1 intermediate (bf16 MFMA operands, fp32 activations in HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated-pass", action="store_true",
                    help="skip the 3 extra untimed steps that measure the dominant kernel family with the side stream off "
                         "(profile runs: keeps the launch counts at warmup + steps)")
    ap.add_argument("--no-secondary", action="store_true", help="headline line only (profile runs)")
    ap.add_argument("--secondary", default="f32_split,bf16,pathB,pathB_f32_split,pathB_pad250,pathB_pad250_f32_split,config5",
                    help="comma list of the secondary legs run in the same process BEFORE the fp32 headline and reported under "
                         "'secondary': f32_split = configs[1] with the GEMM contractions on the fp16 matrix pipe (fp32 storage and results), bf16 = configs[2] (sustained: >= 3 s of untimed load first), pathB = configs[3], pathB_pad250 = "
                         "configs[3] on the reference loader's real input (x250 up-sampling on the GPU), config5 = configs[4] (120 "
                         "classes, bone stream)")
    ap.add_argument("--detail", action="store_true",
                    help="print the per-kernel tables (kernel_ms_per_step, kernel_tflops, isolated) of every leg under 'detail' (default: "
                         "they go to gpurun_out/bench_detail.json only, so that the line's tail holds every leg's numbers)")
    ap.add_argument("--quick", action="store_true", help="tests: secondary legs of a few steps without warm-up seconds")
    ap.add_argument("--stream", default="joint", choices=["joint", "bone"],
                    help="input stream of the stgcn workload (bone = data_gen/gen_bone_data.py on the fly; config 5 = --classes 120 "
                         "--stream bone)")
    ap.add_argument("--sustained-steps", type=int, default=100,
                    help="fp32 headline: after the K timed steps of the contract, time this many more with the same protocol and report "
                         "them as 'sustained' (0 = off; skipped when K is already >= this)")
    ap.add_argument("--warm-seconds", type=float, default=3.0,
                    help="untimed load in front of the timed steps of the selected workload, in EVERY mode (with or without secondary "
                         "legs the headline is timed on a GPU that has been under its own load for this long: a cold first process "
                         "measures ~4 %% low, profiles/r03_fp32_first_process.txt; the time really spent is reported as warm_s)")
    ap.add_argument("--cpu-sample", type=int, default=8,
                    help="clips in the CPU-baseline fallback batch (used only when a full --batch step does not fit the budget)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch.distributed as dist
    rank, world, dev = rank_setup(args)
    cpu_ok = rank == 0 and world == 1 and not args.no_cpu_baseline
    # what this box sustains, measured before the first leg on every rank (rank 0's is reported): the denominators of frac_of_box
    from sar_amd import box as box_probe
    BOX.update(box_probe.measure(dev, quick=args.quick))
    details = {}
    if args.workload == "spectrogram":
        assert args.mfma in ("fp32", "f32_split", "f32_split_bf16x6"), "the spectrogram workload has no bf16 engine"
        out = spectrogram_leg(args, args.steps, args.warmup, args.warm_seconds, rank, world, dev, mfma=args.mfma)
        strip_detail(out, details, "headline")
        if rank == 0:
            out["box"] = dict(BOX)
        if cpu_ok:
            out["cpu_baseline"] = cpu_baseline_spectrogram(2 if args.num_pad_frames else 4, num_pad_frames=args.num_pad_frames)
        if rank == 0:
            out["summary"] = summary_of(out, None, world)
    else:
        sec, deferred = None, []
        names = [] if args.no_secondary or args.mfma != "fp32" or args.stream != "joint" else [n for n in args.secondary.split(",") if n]
        if names:
            # The other BASELINE configs in the same process, driver-timed with the headline.  bf16: the SUSTAINED rate (>= 3 s of
            # untimed load first) is `value`; the cold-GPU rate of a short run is first_run_value.  They run BEFORE the headline
            # leg; the headline has its own --warm-seconds of load either way, so its thermal state does not depend on them.
            q = args.quick
            sec = {}
            deferred = []      # the CPU baselines of the Path B legs run BEHIND every GPU leg: their worker threads keep spinning for a
            #                    while and slow the host thread that issues the next leg's launches (pathB_f32_split: 4.7 -> 5.2 ms)
            for n in names:
                if n == "f32_split":
                    # the fp32 engine with its GEMM contractions on the fp16 matrix pipe (same storage, same parity tolerances)
                    r = stgcn_leg(args, "f32_split", 4 if q else 100, 3, 0.0 if q else 3.0, rank, world, dev, not q, first_run=False,
                                  instrument_steps=2 if q else 5)
                elif n == "bf16":
                    r = stgcn_leg(args, "bf16", 4 if q else 150, 3, 0.0 if q else 3.0, rank, world, dev, False, first_run=not q,
                                  instrument_steps=2 if q else 5)
                elif n == "pathB":
                    r = spectrogram_leg(args, 4 if q else 250, 5, 0.0 if q else 1.0, rank, world, dev, instrument_steps=2 if q else 5,
                                        num_pad_frames=0)
                    if r is not None and cpu_ok and not q:
                        deferred.append((n, lambda: cpu_baseline_spectrogram(4)))
                elif n == "pathB_f32_split":
                    # configs[3] with the resnet's 3x3 / stride-1 convolutions on the fp16 matrix pipe (fp32 storage and results)
                    r = spectrogram_leg(args, 4 if q else 250, 5, 0.0 if q else 1.0, rank, world, dev, instrument_steps=2 if q else 5,
                                        num_pad_frames=0, mfma="f32_split")
                elif n == "pathB_pad250":
                    r = spectrogram_leg(args, 4 if q else 120, 5, 0.0 if q else 1.0, rank, world, dev, instrument_steps=2 if q else 5,
                                        num_pad_frames=250)
                    if r is not None and cpu_ok and not q:
                        deferred.append((n, lambda: cpu_baseline_spectrogram(2, budget_s=8.0, num_pad_frames=250)))
                elif n == "pathB_pad250_f32_split":
                    # the reference loader's real input (x250 up-sampling on the GPU) in front of the resnet on the split kernels
                    r = spectrogram_leg(args, 4 if q else 120, 5, 0.0 if q else 1.0, rank, world, dev, instrument_steps=2 if q else 5,
                                        num_pad_frames=250, mfma="f32_split")
                elif n == "config5":
                    # configs[4]: two independently trained ST-GCNs (joint, bone), 120 classes.  The joint stream differs from the
                    # headline only by the 120-class head; the leg times the BONE stream (joint -> bone fused into the data_bn
                    # prologue) in the reference's precision, and the bf16 engine on the same stream beside it.
                    r = stgcn_leg(args, "fp32", 4 if q else 30, 3, 0.0 if q else 1.0, rank, world, dev, False, classes=120, stream="bone",
                                  instrument_steps=2 if q else 3)
                    rb = stgcn_leg(args, "bf16", 4 if q else 100, 3, 0.0 if q else 2.0, rank, world, dev, False, classes=120,
                                   stream="bone", instrument_steps=2 if q else 3)
                    if r is not None:
                        r["bf16"] = {k: rb[k] for k in ("value", "ms_per_step", "warm_s", "steps", "dtype", "per_rank_ms", "allreduce_ms")}
                else:
                    raise SystemExit("unknown secondary leg %r" % n)
                if rank == 0:
                    sec[n] = strip_detail(r, details, n)
        sust = args.sustained_steps if args.steps < args.sustained_steps else 0
        head = stgcn_leg(args, args.mfma, args.steps, args.warmup, args.warm_seconds, rank, world, dev, not args.no_isolated_pass,
                         stream=args.stream, sustained_steps=sust)
        out = None
        if rank == 0:
            strip_detail(head, details, "headline")
            out = {"metric": "NTU-xsub clips/sec training (ST-GCN, bs=64/GPU)", "value": head["value"], "unit": "clips/s",
                   "n_gpus": world, "rccl_ranks": head["rccl_ranks"], "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": head["dtype"], "data": "synthetic", "config": head["config"], "roofline": head["roofline"]}
            if cpu_ok:
                out["cpu_baseline"] = None      # (keeps its place in front of the long objects; filled below)
            out["box"] = dict(BOX)
            out["legend"] = LEGEND
            out.update({k: v for k, v in head.items() if k not in out})
            if sec:
                out["legs_order"] = names + ["fp32 (headline, last)"]     # each leg with its own warm-up and untimed load (warm_s)
                out["secondary"] = sec
        if cpu_ok:
            out["cpu_baseline"] = cpu_baseline(args.batch, args.cpu_sample, classes=args.classes)
            for n, fn in (deferred if sec else []):
                out["secondary"][n]["cpu_baseline"] = fn()
        if rank == 0:
            out["summary"] = summary_of(head, sec, world)        # LAST key: what the driver's tail keeps
    if rank == 0:
        if args.detail:
            out["detail"] = details
            out["summary"] = out.pop("summary")                  # ... also with the tables in the line
        ddir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(ddir) and os.access(ddir, os.W_OK) and world == 1:
            try:
                with open(os.path.join(ddir, "bench_detail.json"), "w") as f:
                    json.dump({"args": vars(args), "box": dict(BOX), "detail": details}, f)
            except OSError:
                pass
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

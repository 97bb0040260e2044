"""Synchronous data-parallel ST-GCN training, one process per GPU (reference: main_gnn.py:219-239 under
tf.distribute.MirroredStrategy, main_gnn.py:257-258,295).

Per step every rank runs the full forward/backward on its own `batch_size` clips with LOCAL BatchNorm
statistics (MirroredStrategy does not sync BN), the loss is scaled by 1/global_batch (main_gnn.py:226),
then the flat fp32 gradient buffer (12.3 MB for ST-GCN-60) is summed over RCCL/xGMI and every rank applies the
identical fused Nesterov-SGD update.  The buffer is exchanged in three contiguous BUCKETS in the order backward
finishes them (last stage l7-l9 + classifier = 75 % of the bytes, l4-l6 = 19 %, the rest): each bucket's all-reduce is issued
asynchronously on a communication stream as soon as its gradients have been issued, so only the last (smallest)
bucket's collective is exposed -- at 58 ms per fp32 step that is irrelevant, at 12 ms per bf16 step it is the
difference between 1-2 % and ~0.3 % of the step.

SAR_FORCE_DDP=1 makes a ONE-rank job take every data-parallel branch (process group on the "nccl" = RCCL backend with
device_id, bucketed asynchronous all-reduces on the communication stream, their events and waits): the rehearsal
of the RCCL path on a one-GPU box (tests/test_gpu_rccl.py).
"""
import os

import torch
import torch.distributed as dist


def force_ddp():
    return os.environ.get("SAR_FORCE_DDP", "0") == "1"


def ddp_active():
    """True when gradients have to be exchanged: a process group exists and it has more than one rank (or the one-rank
    rehearsal switch is set)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or force_ddp()


def lr_schedule(iteration, base_lr=0.1, steps=(10, 50), batch_size=64):
    """main_gnn.py:303-308: PiecewiseConstantDecay, boundaries (step*40000)//batch_size computed from the
    PER-REPLICA batch size; value[i] while iteration <= boundary[i]."""
    boundaries = [(s * 40000) // batch_size for s in steps]
    values = [base_lr * (0.1 ** i) for i in range(len(steps) + 1)]
    for b, v in zip(boundaries, values):
        if iteration <= b:
            return v
    return values[-1]


def shard_indices(perm, rank, world_size, global_batch):
    """Global batches are consecutive slices of the (shared, seeded) permutation; rank r takes elements
    r::world of each global batch; the remainder is dropped (main_gnn.py:293 drop_remainder=True)."""
    n_batches = len(perm) // global_batch
    out = []
    for i in range(n_batches):
        gb = perm[i * global_batch:(i + 1) * global_batch]
        out.append(gb[rank::world_size])
    return out


def allreduce_sum_(flat, group=None):
    """Gradient exchange (the implicit NCCL all-reduce inside apply_gradients, main_gnn.py:234,239).

    Backend "nccl" (= RCCL over xGMI) reduces the device buffer in place on torch's current stream, i.e. ordered after
    the HIP kernels that produced it and before the optimizer kernel that consumes it (both are launched on that same
    stream, sar_amd/_lib.py:stream_ptr).  Under "gloo" (ranks that share one GPU, CPU-only rendezvous) the bucket is
    staged through host memory: the .cpu() copy synchronises with the producing stream."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_ddp()):
        if flat.is_cuda and dist.get_backend(group) == "gloo":
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def init_distributed(device, backend=None):
    """One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns
    (rank, world).  backend: "nccl" (RCCL) by default; SAR_DIST_BACKEND / the argument select "gloo" for ranks that
    share a device."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if (world > 1 or force_ddp()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:            # SAR_FORCE_DDP=1 without a launcher: a one-rank rendezvous on a free local port
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        backend = backend or os.environ.get("SAR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world


class RunAhead:
    """Bounds how far the host may run ahead of the GPU: at most `depth` train steps in flight (SAR_MAX_STEPS_IN_FLIGHT,
    default 2; 0 = unbounded).  A step is ~200 asynchronous launches that the host issues in 1-2 ms while the GPU needs
    13-60 ms for them; nothing in the step synchronises, so a loop without a per-step read-back queues hundreds of steps.
    Every step's activations that the weight-gradient stream reads are `record_stream`-ed and cannot be reused by the caching
    allocator before the GPU has passed them: tens of GB per queued fp32 step -- the allocator runs out of the 288 GB, falls
    into its synchronise-and-free retry path and the rate collapses (measured: 100 un-synchronised fp32 steps at bs = 64 ran at
    291 clips/s instead of 1 076; profiles/r04_runahead.txt).  Waiting on the event of the step before the previous one costs
    the GPU nothing (its queue still holds a full step)."""

    def __init__(self):
        self.depth = int(os.environ.get("SAR_MAX_STEPS_IN_FLIGHT", "2"))
        self._events = []

    def set_depth(self, depth):
        """change the bound and forget the recorded steps (probes: tools/runahead_probe.py)"""
        self.depth, self._events = int(depth), []

    def step_issued(self):
        if self.depth <= 0 or not torch.cuda.is_available():
            return
        ev = torch.cuda.Event()
        ev.record()
        self._events.append(ev)
        if len(self._events) > self.depth:
            self._events.pop(0).synchronize()


_comm_streams = {}


class GradExchange:
    """SUM all-reduce of gradient buckets, overlapped with the backward pass that is still producing the earlier layers'
    gradients.  RCCL: each bucket's collective is issued asynchronously on ONE communication stream that waits for just the
    events that mark the bucket complete (weight gradients on the engine's side stream, BatchNorm / bias gradients on the
    main stream); `wait()` makes the main stream -- not the host -- wait for all of them before the optimizer kernel.
    gloo (ranks that share one device, tests): staged through the host, synchronous."""

    def __init__(self):
        self._stream, self._works = None, []

    def submit(self, flat, events=()):
        if dist.get_backend() == "gloo":
            for e in events:
                torch.cuda.current_stream().wait_event(e)
            allreduce_sum_(flat)
            return
        if self._stream is None:      # one communication stream per device for every trainer of the process (ops.shared_side_stream)
            self._stream = _comm_streams.setdefault(flat.device.index or 0, None) or torch.cuda.Stream(device=flat.device)
            _comm_streams[flat.device.index or 0] = self._stream
        for e in events:
            self._stream.wait_event(e)
        with torch.cuda.stream(self._stream):
            self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        works, self._works = self._works, []
        for w in works:
            w.wait()                                 # the current stream waits for the collective, the host does not
        return len(works)


class Trainer:
    def __init__(self, engine, batch_size=64, base_lr=0.1, steps=(10, 50), momentum=0.9, world_size=1):
        self.engine, self.batch_size, self.base_lr, self.steps = engine, batch_size, base_lr, tuple(steps)
        self.momentum, self.world_size = momentum, world_size
        self.iteration = 0
        self.comm_events = None     # a list while a bench times the exchange: (start, end) events around it
        self.exchange = GradExchange()
        self.run_ahead = RunAhead()
        self.buckets_last_step = 0  # collectives issued by the last step (tests, bench)

    def step(self, x, labels):
        """One train_step (main_gnn.py:219-239).  Returns (logits, loss) as device tensors; nothing in the step reads them back.
        The host does block here on the event of the step before the previous one (RunAhead, SAR_MAX_STEPS_IN_FLIGHT, default
        2): it bounds the queue, not the GPU."""
        gbs = x.shape[0] * self.world_size
        ddp = ddp_active()
        if ddp:     # the loss scale 1 / global batch and the exchange must agree on the number of ranks
            assert self.world_size == dist.get_world_size(), "Trainer(world_size=%d) inside a process group of %d ranks" % (
                self.world_size, dist.get_world_size())
        timing = self.comm_events is not None and ddp
        if not ddp:
            logits, loss = self.engine.loss_and_grad(x, labels, gbs)
            self.buckets_last_step = 0
        else:
            if timing:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = [0]

            def on_bucket(bi, flat, events):
                if timing and n[0] == 0:
                    t0.record()
                n[0] += 1
                self.exchange.submit(flat, events)

            logits, loss = self.engine.loss_and_grad(x, labels, gbs, bucket_cb=on_bucket)
            self.exchange.wait()
            self.buckets_last_step = n[0]
            if timing:
                t1.record()
                self.comm_events.append((t0, t1))    # first bucket ready -> last collective done (overlaps backward)
        self.engine.sgd_step(lr_schedule(self.iteration, self.base_lr, self.steps, self.batch_size), self.momentum)
        self.iteration += 1
        if x.is_cuda:
            self.run_ahead.step_issued()
        return logits, loss


class SpectrogramTrainer:
    """Train step of main_spectrogram.py:124-189 on the HIP engines: VirtualRadar -> spectrogram image -> resnet18
    forward / backward (mean CrossEntropyLoss) -> ONE all-reduce of the flat resnet gradient buffer and ONE of the
    flat radar-parameter bucket (when radar parameters train) -> Adam on both.  Gradients are averaged over ranks
    (each rank's loss is the mean over its own clips), like DataParallel's gather + mean (main_spectrogram.py:118-119):
    the 1 / world factor rides in the loss scale and the exchange is a SUM, bucketed and overlapped with backward."""

    def __init__(self, model, base_lr, world_size=1, graph=False):
        """graph=True: a step on a batch shape seen before is ONE hipGraph launch (its ~300 kernel launches -- on two streams --
        captured once, inputs copied into the captured step's buffers): with the resnet's convolutions on the split kernels a step at
        bs = 32 takes the GPU ~5 ms and the host ~6 ms to ISSUE (20 us of Python per launch).  Single process, frozen radar parameters
        only (the default of main_spectrogram.py): trainable radar parameters take the eager path, a live process group RAISES (the
        RCCL exchange cannot be captured; tests/test_gpu_rccl.py).  The returned (logits, loss) are copies of the captured buffers."""
        self.model, self.eng, self.world_size = model, model.base_model.engine, world_size
        self.graph, self._graphs = bool(graph), {}
        self.radar_params = list(model.virtual_radar.parameters())
        self.radar_opt = torch.optim.Adam(self.radar_params, lr=base_lr)   # main_spectrogram.py:106 hyper-parameters
        self.comm_events = None      # a list while a bench times the exchange: (start, end) events around it
        self.exchange = GradExchange()
        self.run_ahead = RunAhead()
        self._radar_key, self._radar_bucket = None, None

    def train_radar(self):
        return any(p.requires_grad for p in self.radar_params)

    def _radar_grad_bucket(self):
        """ONE flat gradient buffer for the (few) trainable radar parameters; every p.grad is a view of it, so autograd
        accumulates straight into the bucket: no torch.cat before the exchange, no copy back after it."""
        live = [p for p in self.radar_params if p.requires_grad]
        key = tuple(id(p) for p in live)
        if key != self._radar_key:
            n = sum(p.numel() for p in live)
            self._radar_bucket = torch.zeros(n, dtype=torch.float32, device=live[0].device) if live else None
            self._radar_key = key
        # every step: a p.grad that no longer aliases the bucket (zero_grad(set_to_none=True), a user assignment) is re-bound,
        # otherwise the exchange would reduce a stale bucket while the optimizer steps on the un-reduced gradient
        o, base = 0, self._radar_bucket.data_ptr() if self._radar_bucket is not None else 0
        for p in live:
            if p.grad is None or p.grad.data_ptr() != base + 4 * o or p.grad.shape != p.shape:
                view = self._radar_bucket[o:o + p.numel()].view_as(p)
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view
            o += p.numel()
        return self._radar_bucket

    def step(self, x, labels, lr):
        """One train step (eager, or a hipGraph replay: __init__)."""
        if self.graph and ddp_active():
            # no silent fall-back.  The bucketed gradient exchange lives on the communication stream with
            # host-side event bookkeeping per bucket (GradExchange) and RCCL's own stream semantics: a hipGraph capture of the step
            # cannot record it, and a replay without it would train every rank on its own gradients.
            raise RuntimeError("SpectrogramTrainer(graph=True) is a single-process mode: under data parallelism (process group of %d "
                               "ranks) the step's RCCL gradient exchange cannot be captured into a hipGraph -- construct the trainer "
                               "with graph=False" % dist.get_world_size())
        if not (self.graph and x.is_cuda and not self.train_radar()):
            return self._step(x, labels, lr)
        eng = self.eng
        # the capture bakes in: the batch geometry, the engine's parameter / gradient / Adam-state buffers and the radar configuration;
        # any of them changing (load_params into new buffers, another up-sampling factor) re-captures instead of replaying stale pointers
        vr = self.model.virtual_radar
        key = (tuple(x.shape), x.dtype, tuple(labels.shape), labels.dtype, eng.flat.data_ptr(), eng.grad.data_ptr(),
               tuple(p.data_ptr() for p in self.radar_params), getattr(self.model, "num_pad_frames", None), getattr(vr, "sigma", None))
        st = self._graphs.get(key)
        if st is None:
            # this call's step runs eagerly on copies that become the captured step's inputs (it also makes every lazily built
            # table and the learning-rate cell current, so that the capture records kernels only); then the capture (executes nothing)
            sx, sy = x.clone(), labels.clone()
            out = self._step(sx, sy, lr, run_ahead=False)
            out = tuple(t.clone() for t in out)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                cap = self._step(sx, sy, lr, run_ahead=False)
            self._graphs[key] = (g, sx, sy, cap)
            self.run_ahead.step_issued()
            return out
        g, sx, sy, cap = st
        if getattr(eng, "_lr_host", None) != float(lr):      # the captured step reads the rate from device memory
            eng.lr_dev.fill_(float(lr))
            eng._lr_host = float(lr)
        sx.copy_(x, non_blocking=True)
        sy.copy_(labels, non_blocking=True)
        g.replay()
        self.run_ahead.step_issued()
        return tuple(t.clone() for t in cap)     # the captured buffers are overwritten by the next replay: hand out copies

    def _step(self, x, labels, lr, run_ahead=True):
        """Returns (logits, loss) device tensors; nothing in the step reads them back (the host blocks only on the event of the
        step before the previous one: RunAhead).  Under data parallelism every gradient is
        produced already divided by the world size (the loss scale), the flat resnet gradient buffer is exchanged in four
        buckets -- [layer4 + fc], [layer3], [layer2], [conv1 + layer1] -- each as soon as backward has finished it (its
        weight gradients come off the second stream), and the trainable radar parameters share one more small bucket."""
        model, eng, world = self.model, self.eng, self.world_size
        train_radar = self.train_radar()
        ddp = ddp_active()
        if ddp:
            assert self.world_size == dist.get_world_size(), "SpectrogramTrainer(world_size=%d) inside a process group of %d ranks" % (
                self.world_size, dist.get_world_size())
        timing = self.comm_events is not None and ddp
        if timing:
            t0 = torch.cuda.Event(enable_timing=True)

        def on_bucket(bi, flat, events):
            if timing and bi == 0:
                t0.record()
            self.exchange.submit(flat, events)

        kw = dict(grad_scale=1.0 / world, bucket_cb=on_bucket) if ddp else {}
        if hasattr(eng, "prepack"):
            eng.prepack(True)            # the resnet's weight images beside the radar front-end / the stem (sar_amd/resnet.py: SAR_PATHB_DS_STREAM)
        # (Measured and removed, profiles/r06_pathB_fork_ab.txt: the radar front-end of a resident batch on a stream of its own, so
        # that it runs beside the previous step's backward tail -- bit-identical, and SLOWER: 5 225 -> 5 131 / 7 400 -> 7 150 clips/s.)
        with torch.set_grad_enabled(train_radar):
            img = model.spectrogram(x)
        if train_radar:                                  # the image depends on trainable radar parameters
            bucket = self._radar_grad_bucket()
            logits, loss, dimg = eng.loss_and_grad(img.detach(), labels, need_dx=True, **kw)
            self.radar_opt.zero_grad(set_to_none=False)
            img.backward(dimg)                           # dimg carries the 1 / world scale
            if ddp and bucket is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())   # after autograd's kernels on this stream
                self.exchange.submit(bucket, [ev])
        else:
            logits, loss = eng.loss_and_grad(img, labels, **kw)
        self.exchange.wait()                             # the main stream waits for the collectives, the host does not
        if timing:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            self.comm_events.append((t0, t1))            # first bucket ready -> last collective done (overlaps backward)
        eng.adam_step(lr)
        if train_radar:
            for g in self.radar_opt.param_groups:
                g['lr'] = lr
            self.radar_opt.step()
        if run_ahead:
            self.run_ahead.step_issued()
        return logits, loss


def synthetic_clips(n, device, seed=0, T=300, V=25, M=2, C=3, num_classes=60, single_body_frac=0.8):
    """SURVEY 8(d) synthetic NTU-like batch generated ON DEVICE: 0.12*randn clamped to [-1.1, 0.75], second body
    zeroed for ~80 % of the clips, labels uniform."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = (0.12 * torch.randn((n, C, T, V, M), generator=g, device=device)).clamp_(-1.1, 0.75)
    if M > 1:
        drop = torch.rand(n, generator=g, device=device) < single_body_frac
        x[drop, :, :, :, 1:] = 0
    y = torch.randint(0, num_classes, (n,), generator=g, device=device)
    return x, y

"""Synchronous data-parallel ST-GCN training, one process per GPU (reference: main_gnn.py:219-239 under
tf.distribute.MirroredStrategy, main_gnn.py:257-258,295).

Per step every rank runs the full forward/backward on its own `batch_size` clips with LOCAL BatchNorm
statistics (MirroredStrategy does not sync BN), the loss is scaled by 1/global_batch (main_gnn.py:226),
then ONE all-reduce(SUM) of the flat fp32 gradient buffer (12.3 MB for ST-GCN-60) over RCCL/xGMI and an
identical fused Nesterov-SGD update on every rank.  The exchange is a single collective because the
whole gradient lives in one contiguous buffer; at >= 50 ms of compute per step a 12 MB all-reduce over
7 xGMI links (~20-140 us) needs no bucketing/overlap machinery.
"""
import torch
import torch.distributed as dist


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
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
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
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or os.environ.get("SAR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world


class Trainer:
    def __init__(self, engine, batch_size=64, base_lr=0.1, steps=(10, 50), momentum=0.9, world_size=1):
        self.engine, self.batch_size, self.base_lr, self.steps = engine, batch_size, base_lr, tuple(steps)
        self.momentum, self.world_size = momentum, world_size
        self.iteration = 0
        self.comm_events = None     # a list while a bench times the collective: (start, end) events around the all-reduce

    def step(self, x, labels):
        """One train_step (main_gnn.py:219-239).  Returns (logits, loss) as device tensors (no host sync)."""
        gbs = x.shape[0] * self.world_size
        logits, loss = self.engine.loss_and_grad(x, labels, gbs)
        if self.comm_events is not None and self.world_size > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            allreduce_sum_(self.engine.grad)
            e1.record()
            self.comm_events.append((e0, e1))
        else:
            allreduce_sum_(self.engine.grad)
        self.engine.sgd_step(lr_schedule(self.iteration, self.base_lr, self.steps, self.batch_size), self.momentum)
        self.iteration += 1
        return logits, loss


class SpectrogramTrainer:
    """Train step of main_spectrogram.py:124-189 on the HIP engines: VirtualRadar -> spectrogram image -> resnet18
    forward / backward (mean CrossEntropyLoss) -> ONE all-reduce of the flat resnet gradient buffer and ONE of the
    flat radar-parameter bucket (when radar parameters train) -> Adam on both.  Gradients are averaged over ranks
    (each rank's loss is the mean over its own clips), like DataParallel's gather + mean (main_spectrogram.py:118-119):
    the 1 / world factor rides in the loss scale and the exchange is a SUM, bucketed and overlapped with backward."""

    def __init__(self, model, base_lr, world_size=1):
        self.model, self.eng, self.world_size = model, model.base_model.engine, world_size
        self.radar_params = list(model.virtual_radar.parameters())
        self.radar_opt = torch.optim.Adam(self.radar_params, lr=base_lr)   # main_spectrogram.py:106 hyper-parameters
        self.comm_events = None      # a list while a bench times the exchange: (start, end) events around it
        self._comm_stream = None
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
            o = 0
            for p in live:
                p.grad = self._radar_bucket[o:o + p.numel()].view_as(p)
                o += p.numel()
            self._radar_key = key
        return self._radar_bucket

    def _exchange(self, flat, events=()):
        """SUM all-reduce of one gradient bucket once `events` have completed.  RCCL: asynchronously on a communication
        stream (returns the work handle); gloo (ranks sharing a device, tests): staged through the host, synchronous."""
        if dist.get_backend() == "gloo":
            for e in events:
                torch.cuda.current_stream().wait_event(e)
            allreduce_sum_(flat)
            return None
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=flat.device)
        for e in events:
            self._comm_stream.wait_event(e)
        with torch.cuda.stream(self._comm_stream):
            return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def step(self, x, labels, lr):
        """Returns (logits, loss) device tensors; no host synchronisation.  Under data parallelism every gradient is
        produced already divided by the world size (the loss scale), the flat resnet gradient buffer is exchanged in four
        buckets -- [layer4 + fc], [layer3], [layer2], [conv1 + layer1] -- each as soon as backward has finished it (its
        weight gradients come off the second stream), and the trainable radar parameters share one more small bucket."""
        model, eng, world = self.model, self.eng, self.world_size
        train_radar = self.train_radar()
        ddp = world > 1 and dist.is_available() and dist.is_initialized()
        works = []
        timing = self.comm_events is not None and ddp
        if timing:
            t0 = torch.cuda.Event(enable_timing=True)

        def on_bucket(bi, flat, events):
            if timing and bi == 0:
                t0.record()
            works.append(self._exchange(flat, events))

        kw = dict(grad_scale=1.0 / world, bucket_cb=on_bucket) if ddp else {}
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
                works.append(self._exchange(bucket, [ev]))
        else:
            logits, loss = eng.loss_and_grad(img, labels, **kw)
        for w in works:
            if w is not None:
                w.wait()                                 # the main stream waits for the collectives, the host does not
        if timing:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            self.comm_events.append((t0, t1))            # first bucket ready -> last collective done (overlaps backward)
        eng.adam_step(lr)
        if train_radar:
            for g in self.radar_opt.param_groups:
                g['lr'] = lr
            self.radar_opt.step()
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

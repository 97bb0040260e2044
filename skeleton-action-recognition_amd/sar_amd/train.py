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

    def step(self, x, labels):
        """One train_step (main_gnn.py:219-239).  Returns (logits, loss) as device tensors (no host sync)."""
        gbs = x.shape[0] * self.world_size
        logits, loss = self.engine.loss_and_grad(x, labels, gbs)
        allreduce_sum_(self.engine.grad)
        self.engine.sgd_step(lr_schedule(self.iteration, self.base_lr, self.steps, self.batch_size), self.momentum)
        self.iteration += 1
        return logits, loss


class SpectrogramTrainer:
    """Train step of main_spectrogram.py:124-189 on the HIP engines: VirtualRadar -> spectrogram image -> resnet18
    forward / backward (mean CrossEntropyLoss) -> ONE all-reduce of the flat resnet gradient buffer and ONE of the
    flat radar-parameter bucket (when radar parameters train) -> Adam on both.  Gradients are averaged over ranks
    (each rank's loss is the mean over its own clips), like DataParallel's gather + mean (main_spectrogram.py:118-119)."""

    def __init__(self, model, base_lr, world_size=1):
        self.model, self.eng, self.world_size = model, model.base_model.engine, world_size
        self.radar_params = list(model.virtual_radar.parameters())
        self.radar_opt = torch.optim.Adam(self.radar_params, lr=base_lr)   # main_spectrogram.py:106 hyper-parameters

    def train_radar(self):
        return any(p.requires_grad for p in self.radar_params)

    def step(self, x, labels, lr):
        """Returns (logits, loss) device tensors; no host synchronisation."""
        model, eng, world = self.model, self.eng, self.world_size
        train_radar = self.train_radar()
        with torch.set_grad_enabled(train_radar):
            img = model.spectrogram(x)
        if train_radar:                                  # the image depends on trainable radar parameters
            logits, loss, dimg = eng.loss_and_grad(img.detach(), labels, need_dx=True)
            self.radar_opt.zero_grad(set_to_none=False)
            img.backward(dimg)
        else:
            logits, loss = eng.loss_and_grad(img, labels)
        if world > 1:
            allreduce_sum_(eng.grad)
            eng.grad.div_(world)
            live = [p for p in self.radar_params if p.requires_grad and p.grad is not None]
            if live:                                     # one flat bucket for the (few) radar parameters
                bucket = torch.cat([p.grad.reshape(-1) for p in live])
                allreduce_sum_(bucket)
                bucket.div_(world)
                o = 0
                for p in live:
                    p.grad.copy_(bucket[o:o + p.numel()].view_as(p.grad))
                    o += p.numel()
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

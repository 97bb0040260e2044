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
    """Gradient exchange (the implicit NCCL all-reduce inside apply_gradients, main_gnn.py:234,239)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


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

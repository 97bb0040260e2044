"""One rank of the 2-rank product-path check (launched by tests/test_gpu_multirank.py through torch.distributed.run).

Every rank opens the SAME device (cuda:0) with the gloo backend -- the GPU box has one GPU -- and runs the PRODUCT train
step on the HIP engine: sar_amd.train.Trainer.step (main_gnn.py:219-239 under MirroredStrategy :257-258) or
sar_amd.train.SpectrogramTrainer.step (main_spectrogram.py:124-189), each on its own shard of the seeded global batch.
Rank r writes the all-reduced flat gradient, the updated parameters and the loss to <out>/rank<r>.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BLOCKS = [(64, 1, False), (64, 1, True), (128, 2, True)]


def global_batch(workload, n, seed=11):
    g = torch.Generator().manual_seed(seed)
    T = 300 if workload.startswith("spectrogram") else 24
    x = (0.12 * torch.randn((n, 3, T, 25, 2), generator=g)).clamp_(-1.1, 0.75)
    x[1::3, :, :, :, 1] = 0                      # some single-person clips
    y = torch.randint(0, 10, (n,), generator=g)
    return x, y


def make_trainer(workload, dev, world):
    from sar_amd.train import SpectrogramTrainer, Trainer
    if workload == "stgcn":
        from sar_amd.stgcn import STGCN
        eng = STGCN(num_classes=10, device=dev, seed=5, blocks=BLOCKS)
        return eng, Trainer(eng, batch_size=4, world_size=world)
    if workload == "stgcn_bf16":  # BASELINE configs[2]: the bf16-storage engine (CN8 activations) under the same train step
        from sar_amd.stgcn import STGCN
        eng = STGCN(num_classes=10, device=dev, seed=5, blocks=BLOCKS, mfma="bf16")
        return eng, Trainer(eng, batch_size=4, world_size=world)
    if workload == "stgcn_split":  # the fp32 engine with its contractions on the fp16 matrix pipe (bound cells, split weight images) under DDP
        from sar_amd.stgcn import STGCN
        eng = STGCN(num_classes=10, device=dev, seed=5, blocks=BLOCKS, mfma="f32_split")
        return eng, Trainer(eng, batch_size=4, world_size=world)
    if workload == "stgin":       # the sibling model through the same model-agnostic train step (main_gnn.py --model stgin)
        from sar_amd.stgin import STGIN
        eng = STGIN(num_classes=10, device=dev, seed=5, blocks=BLOCKS)
        return eng, Trainer(eng, batch_size=4, world_size=world)
    from models.resnet import Model
    model = Model(num_classes=10, num_filters=8, device=dev, mfma="f32_split" if workload == "spectrogram_split" else "fp32")
    for name, param in model.named_parameters():     # radar parameters train: their flat bucket is exchanged too
        if 'radar_loc' in name or name.endswith('wavelength'):
            param.requires_grad = True
    return model.base_model.engine, SpectrogramTrainer(model, 1e-3, world_size=world)


def run_shard(workload, trainer, eng, x, y, dev):
    if workload in ("stgcn", "stgin", "stgcn_bf16", "stgcn_split"):
        _, loss = trainer.step(x.to(dev), y.to(dev))
        extra = {}
    else:
        _, loss = trainer.step(x.to(dev), y.to(dev), 1e-3)
        vr = trainer.model.virtual_radar
        extra = {"radar_grad": torch.cat([p.grad.reshape(-1) for p in trainer.radar_params]).cpu(),
                 "radar_location": vr.radar_location.detach().cpu().clone(), "wavelength": vr.wavelength.detach().cpu().clone()}
    torch.cuda.synchronize()
    return dict(grad=eng.grad.cpu().clone(), flat=eng.flat.cpu().clone(), loss=loss.cpu().clone(), **extra)


def main():
    workload, out = sys.argv[1], sys.argv[2]
    from sar_amd.train import init_distributed, shard_indices
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rank, world = init_distributed(dev, backend="gloo")
    assert world == 2 and dist.get_world_size() == 2
    n = 4 * world
    x, y = global_batch(workload, n)
    idx = shard_indices(list(range(n)), rank, world, n)[0]
    eng, trainer = make_trainer(workload, dev, world)
    res = run_shard(workload, trainer, eng, x[idx], y[idx], dev)
    torch.save(res, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

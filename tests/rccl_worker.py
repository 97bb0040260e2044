"""ONE rank on the "nccl" (= RCCL) backend, every data-parallel branch forced (SAR_FORCE_DDP=1): the rehearsal of the RCCL
path on a one-GPU box (launched by tests/test_gpu_rccl.py in a fresh process).

init_distributed -> dist.init_process_group("nccl", device_id=cuda:0) with a one-rank rendezvous; Trainer.step /
SpectrogramTrainer.step then issue their bucketed all-reduces asynchronously on the communication stream behind the
bucket events and make the main stream wait for them before the optimizer kernel -- the code the driver's 8-GPU run
executes, on a communicator of size 1.  Writes what each workload produced to <out>/<workload>.pt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "skeleton-action-recognition_amd"), os.path.dirname(os.path.abspath(__file__))):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def run(workload, dev, steps=2):
    """`steps` product train steps on seeded clips; returns the tensors the test compares"""
    import ddp_worker as W
    eng, trainer = W.make_trainer(workload, dev, 1)
    x, y = W.global_batch(workload, 4 * steps)
    losses, nbuckets = [], []
    for i in range(steps):
        xs, ys = x[4 * i:4 * i + 4].to(dev), y[4 * i:4 * i + 4].to(dev)
        if workload.startswith("spectrogram"):
            _, loss = trainer.step(xs, ys, 1e-3)
        else:
            _, loss = trainer.step(xs, ys)
            nbuckets.append(trainer.buckets_last_step)
        losses.append(loss.detach().reshape(-1).cpu().clone())
    torch.cuda.synchronize()
    out = dict(grad=eng.grad.cpu().clone(), flat=eng.flat.cpu().clone(), loss=torch.cat(losses), nbuckets=nbuckets)
    if workload.startswith("spectrogram"):
        vr = trainer.model.virtual_radar
        out.update(radar_grad=torch.cat([p.grad.reshape(-1) for p in trainer.radar_params]).cpu(),
                   radar_location=vr.radar_location.detach().cpu().clone(), wavelength=vr.wavelength.detach().cpu().clone())
    return out


def main():
    out = sys.argv[1]
    from sar_amd.train import ddp_active, init_distributed
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    rank, world = init_distributed(dev)            # default backend: nccl
    assert (rank, world) == (0, 1) and dist.is_initialized() and ddp_active()
    info = dict(backend=dist.get_backend(), world=dist.get_world_size())
    # a bare collective first: the communicator really reduces device memory in place on the current stream
    t = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    info["bare_ok"] = bool(torch.equal(t.cpu(), torch.arange(1 << 20, dtype=torch.float32)))
    # the hipGraph-captured Path B step under a live communicator must REFUSE, not fall back silently
    import ddp_worker as W
    from sar_amd.train import SpectrogramTrainer
    _, tr = W.make_trainer("spectrogram", dev, 1)
    gtr = SpectrogramTrainer(tr.model, 1e-3, world_size=1, graph=True)
    xg, yg = W.global_batch("spectrogram", 4)
    try:
        gtr.step(xg.to(dev), yg.to(dev), 1e-3)
        info["graph_under_ddp"] = "ran"
    except RuntimeError as e:
        info["graph_under_ddp"] = str(e)
    torch.save(info, os.path.join(out, "info.pt"))
    for workload in sys.argv[2:]:
        torch.save(run(workload, dev), os.path.join(out, workload + ".pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

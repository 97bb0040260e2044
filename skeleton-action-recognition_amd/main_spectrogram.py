#!/usr/bin/env python3
"""Drop-in for the reference's main_spectrogram.py on the MI355X-native path: same flags (main_spectrogram.py:13-62),
same loop (main_spectrogram.py:124-189): CrossEntropyLoss, Adam(lr=base_lr), CyclicLR(1e-4 -> base_lr,
step_size_up=lr_cycle, cycle_momentum=False) stepped once per epoch, train / val phases with loss and accuracy.
The reference wraps the model in nn.DataParallel (main_spectrogram.py:118-119); here it is one process per GPU with
one RCCL all-reduce of the flat gradient buffer (launch with torch.distributed.run).
Data: `--data-path data/ntu/xview/{}_data_joint.npy --label-path data/ntu/xview/{}_label.pkl` as in the reference, or
--synthetic.  The reference's CPU-side 250x frame up-sampling (utils.py:105,134-140: Gaussian smoothing + cubic
interpolation in the loader) runs on the GPU, fused into the radar signal kernels (--num-pad-frames, default 250 as in
the reference; 0 feeds the clips at their native T).
Batch semantics under data parallelism: the reference's DataParallel SPLITS --batch-size across the GPUs; here every
rank takes --batch-size clips (global batch = batch_size * world, gradients averaged), like main_gnn.py's
MirroredStrategy.  Divide --batch-size by the number of ranks to reproduce the reference's effective batch."""
import argparse
import inspect
import json
import os
import shutil

import torch
import torch.distributed as dist

from utils import import_class, save_arg


def get_parser():
    parser = argparse.ArgumentParser(description='Skeleton-Based Action Recognition')
    parser.add_argument('--base-lr', type=float, default=1e-1, help='initial learning rate')
    parser.add_argument('--num-classes', type=int, default=60, help='number of classes in dataset')
    parser.add_argument('--batch-size', type=int, default=64, help='training batch size')
    parser.add_argument('--num-epochs', type=int, default=80, help='total epochs to train')
    parser.add_argument('--num-filters', type=int, default=64, help='number of base filters in model')
    parser.add_argument('--log-dir', default="logs/", help='folder to store model-definition/training-logs/hyperparameters')
    parser.add_argument('--data-path', default="data/ntu/xview/{}_data_joint.npy", help='path to data files')
    parser.add_argument('--label-path', default="data/ntu/xview/{}_label.pkl", help='path to label files')
    parser.add_argument('--notes', default="", help='run details')
    parser.add_argument('--model-type', default="resnet", help='model to train')
    parser.add_argument('--lr_cycle', type=int, default=10, help='number of epochs for the cyclic LR cycle')
    parser.add_argument('--lambda-train-epoch', type=int, default=1000, help='epoch to training the radar_lambda')
    parser.add_argument('--loc-train-epoch', type=int, default=1000, help='epoch to training the radar_loc')
    parser.add_argument('--num-pad-frames', type=int, default=250,
                        help='frame-rate up-sampling factor of utils.Dataset (utils.py:105, default 250), applied on the GPU inside '
                             'the radar layer (Gaussian smoothing + cubic interpolation, never materialised); 0 = feed the clips as they are')
    parser.add_argument('--sigma', type=int, default=3, help='sigma of the Gaussian smoothing before up-sampling (utils.py:105)')
    parser.add_argument('--mfma', default='fp32', choices=['fp32', 'f32_split', 'f32_split_bf16x6'],
                        help="(not in the reference) arithmetic of the resnet's 3x3 / stride-1 convolutions: fp32 MFMA, or fp32 results on "
                             "the fp16 / bf16 matrix pipe (csrc/conv2d_split.hip; same parity tolerances)")
    parser.add_argument('--synthetic', action='store_true')
    parser.add_argument('--synthetic-size', type=int, default=2048)
    parser.add_argument('--max-iters', type=int, default=0)
    return parser


def cyclic_lr(epoch, base_lr, max_lr, step_size_up):
    """torch CyclicLR 'triangular', stepped per epoch (main_spectrogram.py:107-111,189)."""
    total = 2 * step_size_up
    cycle = 1 + epoch // total
    xx = 1 + epoch / total - cycle
    scale = xx / 0.5 if xx <= 0.5 else (xx - 1) / (0.5 - 1)
    return base_lr + (max_lr - base_lr) * scale


def main():
    arg = get_parser().parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SAR_DIST_BACKEND") == "gloo":       # ranks sharing one device (tests)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from sar_amd.train import init_distributed
    init_distributed(dev)                                   # RCCL over xGMI unless SAR_DIST_BACKEND says otherwise
    arg.model_type = 'models.' + arg.model_type.strip() + '.Model'
    run_params = {k: v for k, v in vars(arg).items() if k not in ("data_path", "label_path", "log_dir")}
    run_name = str(run_params).replace(" ", "").replace("'", "").replace(",", "-")[1:-1]
    if arg.notes:
        run_name += "-" + arg.notes
    arg.log_dir = os.path.join(arg.log_dir, run_name)
    Model = import_class(arg.model_type)
    if rank == 0:
        save_arg(arg)
        shutil.copy2(inspect.getfile(Model), arg.log_dir)
        shutil.copy2(os.path.abspath(__file__), arg.log_dir)

    from sar_amd.data import NpySkeletonData, SyntheticSkeletonData
    from sar_amd.train import SpectrogramTrainer
    if arg.synthetic:
        data = {"train": SyntheticSkeletonData(arg.synthetic_size, arg.num_classes),
                "val": SyntheticSkeletonData(max(arg.batch_size * 2, 64), arg.num_classes)}
    else:
        data = {x: NpySkeletonData(arg.data_path.format(x), arg.label_path.format(x), num_classes=arg.num_classes)
                for x in ['train', 'val']}
    model = Model(num_classes=arg.num_classes, num_filters=arg.num_filters, device=dev, num_pad_frames=arg.num_pad_frames,
                  sigma=arg.sigma, mfma=arg.mfma)
    eng = model.base_model.engine
    trainer = SpectrogramTrainer(model, arg.base_lr, world_size=world)
    log = open(os.path.join(arg.log_dir, "scalars.jsonl"), "a") if rank == 0 else None

    def scalar(tag, value, step):
        if log:
            log.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    # test hook (tests/test_gpu_multirank.py), see main_gnn.py: clip ids seen by this rank + a digest of its final parameters
    trace_dir, trace = os.environ.get("SAR_TRACE_DIR"), {"ids": []}
    for epoch in range(arg.num_epochs):
        if rank == 0:
            print('Epoch {}/{}'.format(epoch + 1, arg.num_epochs), flush=True)
        lr = cyclic_lr(epoch, 1e-4, arg.base_lr, arg.lr_cycle)
        # main_spectrogram.py:127-136: parameters are un-frozen by NAME while `epoch > ...` ('radar_lambda' matches no
        # parameter of the reference model either -- the wavelength is called `wavelength`)
        if epoch > arg.lambda_train_epoch:
            for name, param in model.named_parameters():
                if 'radar_lambda' in name:
                    param.requires_grad = True
        if epoch > arg.loc_train_epoch:
            for name, param in model.named_parameters():
                if 'radar_loc' in name:
                    param.requires_grad = True
        for phase in ['train', 'val']:
            # per-iteration loss / correct counts stay on the device (one small tensor per iteration) and are read back
            # ONCE per phase: a .item() per iteration would serialise the host with the GPU on a ~8 ms step
            stats, sizes = [], []
            for it, (x, y) in enumerate(data[phase].batches(arg.batch_size, rank if phase == 'train' else 0,
                                                            world if phase == 'train' else 1, dev, shuffle=True, epoch=epoch,
                                                            drop_remainder=phase == 'train')):
                if phase == 'train':
                    logits, loss = trainer.step(x, y, lr)
                    if trace_dir:
                        trace["ids"].append([epoch, x[:, 0, 0, 0, 0].tolist()])
                else:
                    with torch.no_grad():
                        img = model.spectrogram(x)
                    logits = eng.forward(img, training=False)
                    loss = torch.nn.functional.cross_entropy(logits, y).reshape(1)
                stats.append(torch.stack([loss.reshape(()), (logits.argmax(1) == y).sum().float()]))
                sizes.append(len(y))
                if arg.max_iters and it + 1 >= arg.max_iters:
                    break
            host = torch.stack(stats).cpu().tolist() if stats else []
            for it, ((lo, ok), n) in enumerate(zip(host, sizes)):
                scalar('{}_cross_entropy_loss'.format(phase), lo, epoch * 100000 + it)
                scalar('{}_acc'.format(phase), ok / n, epoch * 100000 + it)
            run_loss, run_ok = sum(h[0] for h in host), sum(h[1] for h in host)
            n_it, n_seen = len(host), sum(sizes)
            bad = run_loss != run_loss
            if world > 1 and phase == 'train':
                # every rank must raise together: a rank that kept going would block in its next all-reduce until the
                # collective timed out (the val phase reads the same clips on every rank, so it needs no exchange)
                flag = torch.tensor([1.0 if bad else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                bad = flag.item() > 0
            if bad:
                raise FloatingPointError("NaN loss in phase %s of epoch %d (labels outside [0, --num-classes)?)" % (phase, epoch + 1))
            if rank == 0:
                scalar('{}_epoch_cross_entropy_loss'.format(phase), run_loss / max(n_it, 1), epoch)
                scalar('{}_epoch_acc'.format(phase), run_ok / max(n_seen, 1), epoch)
                print('{} Loss: {:.4f} Acc: {:.4f}'.format(phase, run_loss / max(n_it, 1), run_ok / max(n_seen, 1)), flush=True)
        if rank == 0 and trainer.train_radar():
            print('radar_location {} wavelength {:.6g}'.format([round(v, 6) for v in model.virtual_radar.radar_location.tolist()],
                                                               model.virtual_radar.wavelength.item()), flush=True)
        if log:
            log.flush()
    if trace_dir:
        import hashlib
        trace["digest"] = hashlib.sha256(eng.flat.cpu().numpy().tobytes()).hexdigest()
        trace["has_log"] = log is not None
        with open(os.path.join(trace_dir, "rank%d.json" % rank), "w") as f:
            json.dump(trace, f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Drop-in for the reference's main_gnn.py on the MI355X-native path: same command-line flags
(main_gnn.py:25-77), same train step (main_gnn.py:219-239: softmax-CE summed / global batch, every trainable
variable except `adjacency_matrix`, SGD momentum 0.9 Nesterov, PiecewiseConstantDecay main_gnn.py:303-308),
same synchronous data parallelism (MirroredStrategy, main_gnn.py:257-258 -> one process per GPU + one RCCL
all-reduce of the flat gradient buffer), per-iteration loss / top-1 / top-5, per-epoch test accuracy and a
checkpoint every --save-freq epochs (main_gnn.py:359-428).

Launch:  python main_gnn.py --model stgcn ...                      (1 GPU)
         python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main_gnn.py --model stgcn ...
Data: --train-data-path / --test-data-path point at a directory of `*.tfrecord` shards (the reference's
data_gen/gen_tfrecord_data.py output, read without TensorFlow by sar_amd/tfrecord.py) or at `<prefix>` for which
`<prefix>.npy` and the sibling `*_label.pkl` exist (data_gen/gen_joint_data.py), or pass --synthetic.
TensorBoard is not available in this image: scalars go to <log-dir>/scalars.jsonl with the reference's tags.
"""
import argparse
import inspect
import json
import os

import numpy as np
import shutil
import time

import torch
import torch.distributed as dist

from utils import import_class, save_arg


def get_parser():
    # flags and defaults of the reference, main_gnn.py:31-75
    parser = argparse.ArgumentParser(
        description='Graph Convolutional Neural Network for Skeleton-Based Action Recognition')
    parser.add_argument('--model', required=True, help='model used to train')
    parser.add_argument('--stream', default='joint', choices=['joint', 'bone', 'joint_motion', 'bone_motion'],
                        help='input stream computed on the fly from JOINT data (data_gen/gen_bone_data.py, gen_motion_data.py); '
                             'the reference trains each stream from its own pre-computed file')
    parser.add_argument('--resume', default='', help='checkpoint (ckpt-N.pt of a previous run) to restore model, optimizer '
                                                    'velocity, iteration and epoch from (the reference only saves)')
    parser.add_argument('--save-scores', action='store_true',
                        help='write the test-set class probabilities of every checkpointed epoch (scores-N.npy, for score fusion '
                             'of separately trained streams with tools/fuse_scores.py)')
    parser.add_argument('--mfma', default='fp32', choices=['fp32', 'f32_split', 'f32_split_bf16x6', 'bf16', 'bf16_operands'],
                        help="arithmetic of the convolutions' matrix products: fp32 (the reference's); bf16 = bf16 activations in HBM + bf16 operands, fp32 "
                             "accumulation, BatchNorm statistics and master weights; bf16_operands = bf16 operands only")
    parser.add_argument('--base-lr', type=float, default=1e-1, help='initial learning rate')
    parser.add_argument('--num-classes', type=int, default=60, help='number of classes in dataset')
    parser.add_argument('--batch-size', type=int, default=64, help='training batch size')
    parser.add_argument('--num-epochs', type=int, default=80, help='total epochs to train')
    parser.add_argument('--save-freq', type=int, default=10, help='periodicity of saving model weights')
    parser.add_argument('--freeze-graph-until', type=int, default=80,
                        help='adjacency matrices will be trained only after this epoch')
    parser.add_argument('--log-dir', default="logs/",
                        help='folder to store model-definition/training-logs/hyperparameters')
    parser.add_argument('--train-data-path', default="data/ntu/xview/train_data_joint",
                        help='path prefix of the training data (<prefix>.npy + label pkl)')
    parser.add_argument('--test-data-path', default="data/ntu/xview/val_data_joint",
                        help='path prefix of the testing data')
    parser.add_argument('--notes', default="", help='run details')
    parser.add_argument('--steps', type=int, default=[10, 50], nargs='+',
                        help='the epoch where optimizer reduce the learning rate, eg: 10 50')
    # additions of this implementation
    parser.add_argument('--trainable-adjacency', action='store_true',
                        help="make the stacked adjacency a trainable variable `adjacency_matrix` (models/gcn.py AdjGraphConv); it is "
                             "trained only while epoch > --freeze-graph-until, as in the reference's train_step")
    parser.add_argument('--verify-crc', default='full', choices=['full', 'length', 'off'],
                        help="TFRecord shards: 'full' checks the masked CRC-32C of every length field and payload (what tf.data's "
                             "reader does), 'length' the length fields only, 'off' the framing only")
    parser.add_argument('--synthetic', action='store_true', help='train on synthetic NTU-like clips')
    parser.add_argument('--synthetic-size', type=int, default=40000)
    parser.add_argument('--max-iters', type=int, default=0, help='stop each epoch after this many iterations (0 = all)')
    return parser


def _label_path(prefix):
    d, base = os.path.split(prefix)
    return os.path.join(d, base.replace("_data_joint", "").replace("_data_bone", "") + "_label.pkl")


def topk_correct(logits, labels, k):
    return (logits.topk(k, dim=1).indices == labels[:, None]).any(dim=1).sum()


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
    arg.gpus = world
    global_batch_size = arg.batch_size * world                # main_gnn.py:258

    run_params = {k: v for k, v in vars(arg).items()
                  if k not in ("train_data_path", "test_data_path", "log_dir", "save_freq", "freeze_graph_until", "gpus", "resume",
                               "save_scores", "verify_crc") and not (k == "mfma" and v == "fp32")
                  and not (k == "trainable_adjacency" and not v)}
    run_name = str(run_params).replace(" ", "").replace("'", "").replace(",", "-")[1:-1]
    if arg.notes:
        run_name += "-" + arg.notes
    arg.log_dir = os.path.join(arg.log_dir, run_name)
    ckpt_dir = os.path.join(arg.log_dir, "checkpoints")
    model_mod = import_class('models.' + arg.model)
    if rank == 0:
        save_arg(arg)
        shutil.copy2(inspect.getfile(model_mod), arg.log_dir)   # main_gnn.py:284
        os.makedirs(ckpt_dir, exist_ok=True)

    from sar_amd.data import NpySkeletonData, SyntheticSkeletonData
    from sar_amd.train import Trainer, allreduce_sum_
    if arg.synthetic:
        train_data = SyntheticSkeletonData(arg.synthetic_size, arg.num_classes)
        test_data = SyntheticSkeletonData(max(arg.batch_size * 4, 256), arg.num_classes)
    else:
        def open_data(prefix):
            # a directory of *.tfrecord shards (data_gen/gen_tfrecord_data.py, what the reference's main_gnn.py reads)
            # or the <prefix>.npy + label pkl pair of data_gen/gen_joint_data.py
            if os.path.isdir(prefix) and any(f.endswith("tfrecord") for f in os.listdir(prefix)):
                from sar_amd.tfrecord import TFRecordSkeletonData
                return TFRecordSkeletonData(prefix, verify_crc=arg.verify_crc, num_classes=arg.num_classes)
            return NpySkeletonData(prefix + ".npy", _label_path(prefix), num_classes=arg.num_classes)
        train_data, test_data = open_data(arg.train_data_path), open_data(arg.test_data_path)

    model = model_mod.Model(num_classes=arg.num_classes, device=dev, stream=arg.stream, mfma=arg.mfma,
                            trainable_adjacency=arg.trainable_adjacency)
    eng = model.engine
    trainer = Trainer(eng, batch_size=arg.batch_size, base_lr=arg.base_lr, steps=arg.steps, world_size=world)
    log = open(os.path.join(arg.log_dir, "scalars.jsonl"), "a") if rank == 0 else None

    def scalar(tag, value, step):
        if log:
            log.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    # test hook (tests/test_gpu_multirank.py): SAR_TRACE_DIR=<dir> makes every rank record the first coordinate of the clips
    # it trains on (a test data set stores the clip id there) and a digest of its final parameters.  Not set in production:
    # reading the ids back synchronises the host with the GPU every step.
    trace_dir, trace = os.environ.get("SAR_TRACE_DIR"), {"ids": [], "logged": False}
    train_iter = test_iter = 0
    start_epoch = 0
    if arg.resume:
        ck = torch.load(arg.resume, map_location="cpu")
        eng.load_params(ck["model"])
        eng.velocity.copy_(ck["velocity"].to(dev))
        trainer.iteration = int(ck["iteration"])
        start_epoch = int(ck["epoch"])
        train_iter = trainer.iteration
        if rank == 0:
            print("Resumed from {} (epoch {}, iteration {})".format(arg.resume, start_epoch, trainer.iteration), flush=True)
    next_iter = None      # the loader of the NEXT epoch is created (its threads start parsing) before the current epoch is evaluated
    for epoch in range(start_epoch, arg.num_epochs):
        if rank == 0:
            print("Epoch: {}".format(epoch + 1), flush=True)
        t0 = time.time()
        it = -1
        # `train_adj` (main_gnn.py:228-232,364-365): variables named *adjacency_matrix* receive gradients only while
        # epoch > freeze_graph_until (0-based epoch, as in the reference loop).  For the reference's models.stgcn the adjacency
        # is a non-trainable variable (models/stgcn.py:105-109), so the flag only matters with --trainable-adjacency.
        eng.train_adjacency = epoch > arg.freeze_graph_until
        # per-iteration scalars stay on the device and are exchanged / read back once per epoch (no host sync per step)
        pending = []
        if next_iter is None:
            next_iter = train_data.batches(arg.batch_size, rank, world, dev, shuffle=True, epoch=epoch)
        train_iter_obj, next_iter = next_iter, None
        for it, (x, y) in enumerate(train_iter_obj):
            logits, loss = trainer.step(x, y)
            if trace_dir:
                trace["ids"].append([epoch, x[:, 0, 0, 0, 0].tolist()])
            pending.append(torch.stack([loss.reshape(()) * 1.0, topk_correct(logits, y, 1).float() / global_batch_size,
                                        topk_correct(logits, y, 5).float() / global_batch_size]))
            if arg.max_iters and it + 1 >= arg.max_iters:
                break
        if pending:
            stats = allreduce_sum_(torch.stack(pending)).cpu()     # loss is already divided by the global batch
            if not bool(torch.isfinite(stats[:, 0]).all()):
                raise FloatingPointError("non-finite training loss in epoch %d (labels outside [0, --num-classes)?)" % (epoch + 1))
            for row in stats.tolist():
                scalar("cross_entropy_loss", row[0], train_iter)
                scalar("train_acc", row[1], train_iter)
                scalar("train_acc_top_5", row[2], train_iter)
                train_iter += 1
        if rank == 0:
            torch.cuda.synchronize()
            print("  train: %d iters, %.1f clips/s" % (it + 1, (it + 1) * global_batch_size / (time.time() - t0)),
                  flush=True)
        if epoch + 1 < arg.num_epochs and not arg.max_iters:
            next_iter = train_data.batches(arg.batch_size, rank, world, dev, shuffle=True, epoch=epoch + 1)
        # ---- test (main_gnn.py:381-408), un-distributed like the reference: rank 0 evaluates
        if rank == 0:
            c1 = c5 = n = 0
            checkpointing = (epoch + 1) % arg.save_freq == 0 or epoch + 1 == arg.num_epochs
            cm = torch.zeros((arg.num_classes, arg.num_classes), dtype=torch.int64, device=dev)
            all_probs = []
            for it, (x, y) in enumerate(test_data.batches(arg.batch_size, 0, 1, dev, shuffle=False,
                                                          drop_remainder=False)):
                probs = eng.predict(x)
                if checkpointing:      # main_gnn.py:410-416: confusion matrix of the test set (rows = true class)
                    cm.view(-1).index_add_(0, y * arg.num_classes + probs.argmax(1), torch.ones_like(y))
                    if arg.save_scores:
                        all_probs.append(probs.cpu())
                b1, b5 = topk_correct(probs, y, 1).item(), topk_correct(probs, y, 5).item()
                scalar("test_acc", b1 / len(y), test_iter)
                scalar("test_acc_top_5", b5 / len(y), test_iter)
                test_iter += 1
                c1, c5, n = c1 + b1, c5 + b5, n + len(y)
                if arg.max_iters and it + 1 >= arg.max_iters:
                    break
            scalar("epoch_test_acc", c1 / n, epoch)
            scalar("epoch_test_acc_top_5", c5 / n, epoch)
            print("  test: top1 %.4f top5 %.4f" % (c1 / n, c5 / n), flush=True)
            if checkpointing:
                np.save(os.path.join(arg.log_dir, "confusion_matrix-%d.npy" % (epoch + 1)), cm.cpu().numpy())
                if arg.save_scores:
                    np.save(os.path.join(arg.log_dir, "scores-%d.npy" % (epoch + 1)), torch.cat(all_probs).numpy())
                path = os.path.join(ckpt_dir, "ckpt-%d.pt" % (epoch + 1))
                torch.save({"model": eng.state_dict(), "velocity": eng.velocity.cpu(), "iteration": trainer.iteration,
                            "epoch": epoch + 1}, path)
                print('Saving checkpoint for epoch {} at {}'.format(epoch + 1, path), flush=True)
            log.flush()
        if world > 1:
            dist.barrier()
    if trace_dir:
        import hashlib
        trace["digest"] = hashlib.sha256(eng.flat.cpu().numpy().tobytes()).hexdigest()
        trace["iterations"], trace["has_log"] = trainer.iteration, log is not None
        with open(os.path.join(trace_dir, "rank%d.json" % rank), "w") as f:
            json.dump(trace, f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

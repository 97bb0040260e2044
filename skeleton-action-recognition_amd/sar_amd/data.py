"""Input side of main_gnn.py for the HIP path.

Formats: the reference's `*_data_joint.npy` (N,3,T,25,M) float32 + `*_label.pkl` pair
(data_gen/gen_joint_data.py:138-151), memory-mapped; or synthetic NTU-like clips generated on the device.
(The TFRecord shards of data_gen/gen_tfrecord_data.py are read by sar_amd/tfrecord.py.)
Sharding follows main_gnn.py:290-301 under MirroredStrategy: a global batch of batch_size*world clips per step,
rank r takes elements r::world, remainder dropped; the reference shuffles with a buffer of 1000 BATCHES
(main_gnn.py:189-194) -- here the clip order is a seeded permutation per epoch, identical on every rank.
"""
import pickle

import numpy as np
import torch

from .train import shard_indices, synthetic_clips


class NpySkeletonData:
    def __init__(self, data_path, label_path):
        self.data = np.load(data_path, mmap_mode="r")
        with open(label_path, "rb") as f:
            _, labels = pickle.load(f, encoding="latin1")
        self.labels = np.asarray(labels, dtype=np.int64)
        assert len(self.labels) == len(self.data)

    def __len__(self):
        return len(self.data)

    def batches(self, batch_size, rank, world, device, shuffle, epoch=0, drop_remainder=True):
        n = len(self)
        perm = np.random.default_rng(1234 + epoch).permutation(n) if shuffle else np.arange(n)
        if drop_remainder:
            shards = shard_indices(list(perm), rank, world, batch_size * world)
        else:
            shards = [perm[i:i + batch_size] for i in range(0, n, batch_size)]
        for idx in shards:
            idx = np.sort(np.asarray(idx))
            x = torch.from_numpy(np.ascontiguousarray(self.data[idx])).to(device, non_blocking=True)
            y = torch.from_numpy(self.labels[idx]).to(device, non_blocking=True)
            yield x.float(), y


class SyntheticSkeletonData:
    """N synthetic clips (default 40 000 = the constant in main_gnn.py:303), regenerated on device per batch."""

    def __init__(self, n=40000, num_classes=60, T=300):
        self.n, self.num_classes, self.T = n, num_classes, T

    def __len__(self):
        return self.n

    def batches(self, batch_size, rank, world, device, shuffle, epoch=0, drop_remainder=True):
        steps = self.n // (batch_size * world) if drop_remainder else -(-self.n // batch_size)
        for i in range(steps):
            yield synthetic_clips(batch_size, device, seed=(epoch * 100003 + i) * world + rank, T=self.T,
                                  num_classes=self.num_classes)

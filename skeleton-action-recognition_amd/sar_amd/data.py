"""Input side of main_gnn.py / main_spectrogram.py for the HIP path.

Formats: the reference's `*_data_joint.npy` (N,3,T,25,M) float32 + `*_label.pkl` pair
(data_gen/gen_joint_data.py:138-151), memory-mapped; or synthetic NTU-like clips generated on the device.
(The TFRecord shards of data_gen/gen_tfrecord_data.py are read by sar_amd/tfrecord.py.)
Sharding follows main_gnn.py:290-301 under MirroredStrategy: a global batch of batch_size*world clips per step,
rank r takes elements r::world, remainder dropped; the reference shuffles with a buffer of 1000 BATCHES
(main_gnn.py:189-194) -- here the clip order is a seeded permutation per epoch, identical on every rank.

Host -> device: `prefetch_to_device` is the `dataset.prefetch(AUTOTUNE)` of main_gnn.py:193 / the
`DataLoader(num_workers=10)` of main_spectrogram.py:94-101: a background thread builds the next batches in a small
ring of PINNED host buffers while the GPU runs the current step, and the consumer enqueues the asynchronous
H2D copies (11.5 MB per 64 clips) on the compute stream, so the step never waits for a page-able memcpy or a parse.
"""
import pickle
import queue
import threading

import numpy as np
import torch

from .train import shard_indices, synthetic_clips


def prefetch_to_device(host_batches, device, depth=3):
    """host_batches: iterator of (x float32 ndarray, y int64 ndarray).  Yields (x, y) device tensors.

    The producer thread copies each batch into one of `depth + 1` pinned slots (file reads, memmap gathers, np.stack and
    the CRC / protobuf work all release the GIL or are short); a slot is recycled only after the event recorded behind
    its H2D copy has completed."""
    device = torch.device(device)
    cuda = device.type == "cuda"
    free, ready = queue.Queue(), queue.Queue(maxsize=depth)
    for _ in range(depth + 1):
        free.put({"x": None, "y": None, "event": None})
    stop = threading.Event()

    def slot_buffer(slot, key, arr):
        t = slot[key]
        if t is None or t.numel() < arr.size or t.dtype != torch.from_numpy(arr[:0]).dtype:
            t = torch.empty(arr.size, dtype=torch.from_numpy(arr[:0]).dtype, pin_memory=cuda)
            slot[key] = t
        v = t[:arr.size].view(arr.shape)
        np.copyto(v.numpy(), arr)
        return v

    def producer():
        try:
            for x, y in host_batches:
                slot = free.get()
                if stop.is_set():
                    return
                if slot["event"] is not None:
                    slot["event"].synchronize()          # the previous copy out of this slot has finished
                ready.put((slot, slot_buffer(slot, "x", np.asarray(x, dtype=np.float32)),
                           slot_buffer(slot, "y", np.asarray(y, dtype=np.int64))))
                if stop.is_set():
                    return
            ready.put(None)
        except BaseException as e:                        # surface loader errors in the training thread
            ready.put(e)

    th = threading.Thread(target=producer, name="sar-prefetch", daemon=True)
    th.start()
    try:
        while True:
            item = ready.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            slot, hx, hy = item
            x = hx.to(device, non_blocking=True)
            y = hy.to(device, non_blocking=True)
            if cuda:
                slot["event"] = torch.cuda.Event()
                slot["event"].record()
            free.put(slot)
            yield x, y
    finally:
        stop.set()
        try:                                              # unblock a producer waiting for a slot / queue space
            while True:
                ready.get_nowait()
        except queue.Empty:
            pass
        free.put({"x": None, "y": None, "event": None})


class parallel_batches:
    """Iterator over (x, y) device tensors for `jobs` IN ORDER while `workers` host threads build the next ones.

    build(job, x_view, y_view) fills one batch straight into PINNED memory (x_view float32 of shape_of(job), y_view int64
    of its first dimension) -- no intermediate np.stack, one pass over the clip bytes -- and runs without the GIL where it
    matters (CRC in libsar_hip.so, numpy block copies).  tf.data's `num_parallel_calls` / the reference's
    DataLoader(num_workers=10) (main_spectrogram.py:94-101): one parser thread delivers ~4 300 clips/s with CRC
    verification, about ONE MI355X's bf16 training rate; N threads keep a margin.  The workers START AT CONSTRUCTION (the
    next epoch's loader can be created while the current epoch is being evaluated: its first batches are then ready when
    training resumes).  A slot is recycled only after the event recorded behind its H2D copy has completed.  With a CPU
    `device` the yielded tensors ARE the slot (no copy): they are valid until the next batch is requested."""

    def __init__(self, jobs, build, shape_of, device, workers=4, depth=3):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.jobs, self.build, self.shape_of = list(jobs), build, shape_of
        self.free = queue.Queue()
        for _ in range(max(1, workers) + depth):
            self.free.put({"x": None, "y": None, "event": None})
        self.done, self.cond, self.stop = {}, threading.Condition(), threading.Event()
        self.next_job, self.lock = [0], threading.Lock()
        self.threads = [threading.Thread(target=self._worker, name="sar-loader-%d" % i, daemon=True) for i in range(max(1, workers))]
        for t in self.threads:
            t.start()

    def _pinned(self, slot, key, n, dtype):
        t = slot[key]
        if t is None or t.numel() < n:
            t = torch.empty(n, dtype=dtype, pin_memory=self.cuda)
            slot[key] = t
        return t

    def _worker(self):
        try:
            while not self.stop.is_set():
                # job number AND pinned slot are taken under one lock, i.e. slots are handed out in job order: the job the
                # consumer is waiting for always owns a slot before any later job does (no deadlock with all slots holding
                # finished later batches)
                with self.lock:
                    if self.stop.is_set():
                        return
                    j = self.next_job[0]
                    if j >= len(self.jobs):
                        return
                    self.next_job[0] = j + 1
                    slot = self.free.get()
                if self.stop.is_set():
                    return
                if slot["event"] is not None:
                    slot["event"].synchronize()
                shp = tuple(self.shape_of(self.jobs[j]))
                n = int(np.prod(shp))
                hx = self._pinned(slot, "x", n, torch.float32)[:n].view(shp)
                hy = self._pinned(slot, "y", shp[0], torch.int64)[:shp[0]]
                self.build(self.jobs[j], hx.numpy(), hy.numpy())
                with self.cond:
                    self.done[j] = (slot, hx, hy)
                    self.cond.notify_all()
        except BaseException as e:
            with self.cond:
                self.done["error"] = e
                self.cond.notify_all()

    def close(self):
        self.stop.set()
        for _ in self.threads:
            self.free.put({"x": None, "y": None, "event": None})

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _take(self, j):
        with self.cond:
            while j not in self.done and "error" not in self.done:
                self.cond.wait()
            if "error" in self.done:
                raise self.done["error"]
            return self.done.pop(j)

    def __iter__(self):
        """Host -> device copies run on the device's COPY stream, one batch ahead of the consumer: the copy of batch j + 1 is issued
        before batch j is handed out and overlaps the step that consumes batch j (issued on the consumer's stream -- as until round 5 --
        the 11.5 MB of an ST-GCN batch were stream-ordered IN FRONT of every step: 0.45 ms of a 10.9 ms bf16 step, the whole gap
        between the loader-fed and the synthetic rate).  The consumer's stream waits for the copy's event; the tensors are recorded
        on it for the caching allocator."""
        try:
            n = len(self.jobs)
            if not self.cuda:
                for j in range(n):
                    slot, hx, hy = self._take(j)
                    yield hx, hy       # a CPU consumer gets the slot itself: valid until it asks for the next batch
                    self.free.put(slot)
                return
            copy = copy_stream(self.device)
            pending = None
            for j in range(n + 1):
                nxt = None
                if j < n:
                    slot, hx, hy = self._take(j)
                    with torch.cuda.stream(copy):
                        x = hx.to(self.device, non_blocking=True)
                        y = hy.to(self.device, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy)
                    slot["event"] = ev
                    self.free.put(slot)
                    nxt = (x, y, ev)
                if pending is not None:
                    x, y, ev = pending
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    x.record_stream(cur)
                    y.record_stream(cur)
                    yield x, y
                pending = nxt
        finally:
            self.close()


_copy_streams = {}


def copy_stream(device):
    """ONE host-to-device copy stream per device for every loader of the process (streams share a few hardware queues in creation
    order: ops.shared_side_stream)"""
    key = torch.device(device).index or 0
    st = _copy_streams.get(key)
    if st is None:
        st = _copy_streams[key] = torch.cuda.Stream(device=device)
    return st


LOADER_THREADS = int(__import__("os").environ.get("SAR_LOADER_THREADS", "4"))


class NpySkeletonData:
    def __init__(self, data_path, label_path, num_classes=None):
        self.data = np.load(data_path, mmap_mode="r")
        with open(label_path, "rb") as f:
            _, labels = pickle.load(f, encoding="latin1")
        self.labels = np.asarray(labels, dtype=np.int64)
        assert len(self.labels) == len(self.data)
        if num_classes is not None and len(self.labels) and not (0 <= self.labels.min() and self.labels.max() < num_classes):
            raise ValueError("%s: labels span [%d, %d] but --num-classes is %d" % (label_path, self.labels.min(),
                                                                                   self.labels.max(), num_classes))

    def __len__(self):
        return len(self.data)

    def host_batches(self, batch_size, rank=0, world=1, shuffle=False, epoch=0, drop_remainder=True):
        n = len(self)
        perm = np.random.default_rng(1234 + epoch).permutation(n) if shuffle else np.arange(n)
        if drop_remainder:
            shards = shard_indices(list(perm), rank, world, batch_size * world)
        else:
            shards = [perm[i:i + batch_size] for i in range(0, n, batch_size)]
        for idx in shards:
            idx = np.sort(np.asarray(idx))
            yield self.data[idx], self.labels[idx]

    def batches(self, batch_size, rank, world, device, shuffle, epoch=0, drop_remainder=True, workers=None):
        """device batches; the memmap gather of each batch goes straight into pinned memory on `workers` loader threads"""
        n = len(self)
        perm = np.random.default_rng(1234 + epoch).permutation(n) if shuffle else np.arange(n)
        if drop_remainder:
            shards = shard_indices(list(perm), rank, world, batch_size * world)
        else:
            shards = [perm[i:i + batch_size] for i in range(0, n, batch_size)]
        jobs = [np.sort(np.asarray(idx)) for idx in shards]
        clip = self.data.shape[1:]

        same_dtype = self.data.dtype == np.float32

        def build(idx, x, y):
            if same_dtype:
                np.take(self.data, idx, axis=0, out=x)     # gather straight into the pinned float32 slot
            else:                                          # a float64 / float16 .npy: np.take(out=) refuses the cast
                np.copyto(x, self.data[idx], casting="unsafe")
            y[:] = self.labels[idx]

        return parallel_batches(jobs, build, lambda idx: (len(idx),) + tuple(clip), device,
                                workers=LOADER_THREADS if workers is None else workers)


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

"""TFRecord shards of the reference's data_gen/gen_tfrecord_data.py, read (and written) without TensorFlow.

The reference stores one `tf.train.Example` per clip (data_gen/gen_tfrecord_data.py:25-33):
    features: bytes_list[ tf.io.serialize_tensor(clip.astype(float32)) ]      -- a serialized TensorProto
    label   : int64_list[ label ]
in `<name>-{i}.tfrecord` shards (:76-85: a new shard every len(labels)//num_shards clips), and reads them back with
`tf.data.TFRecordDataset(records, num_parallel_reads=len(records))` -> parse -> batch -> prefetch -> shuffle(1000)
(main_gnn.py:159-194: a cyclic interleave of the shards, one record from each in turn; the shuffle acts on BATCHES).
(The reference's parser reshapes every tensor to (256,256,1), main_gnn.py:180, which cannot hold a 3x300x25x2
skeleton clip -- the shape stored in the TensorProto is used here.)

File framing (TFRecord): uint64 length | uint32 masked_crc32c(length) | data | uint32 masked_crc32c(data), little
endian, masked = ((crc >> 15) | (crc << 17)) + 0xa282ead8 mod 2^32.  Protobuf wire format decoded by hand: only
the fields the reference writes (Example.features=1 -> Features.feature=1 map<string, Feature{bytes_list=1,
float_list=2, int64_list=3}>; TensorProto{dtype=1, tensor_shape=2{dim=2{size=1}}, tensor_content=4, float_val=5}).
"""
import os
import struct

import numpy as np

_MASK_DELTA = 0xA282EAD8
_DT_FLOAT = 1


def _native():
    """libsar_hip.so's host-side input helpers (include/sar_hip.h: sar_crc32c, sar_tfrecord_index).  No Python
    fallback: a 180 KB clip per record needs a CRC at hundreds of MB/s to keep one GPU fed."""
    from . import _lib
    return _lib.load()


def _addr(buf):
    """(address, keep-alive) of a bytes-like object without copying it."""
    arr = np.frombuffer(buf, dtype=np.uint8)
    return arr.ctypes.data, arr


def crc32c(data):
    a, keep = _addr(data)
    return int(_native().sar_crc32c(a, len(keep)))


def masked_crc(data):
    a, keep = _addr(data)
    return int(_native().sar_masked_crc32c(a, len(keep)))


# ---------------------------------------------------------------- protobuf wire format
def _varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """yield (field_number, wire_type, value) -- value is an int (varint / fixed) or a memoryview (length-delimited)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fno, wt, val


def _enc_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _enc_field(fno, payload):
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def parse_tensor(buf):
    """tf.io.parse_tensor(buf, tf.float32) for what tf.io.serialize_tensor(float32 array) writes."""
    dtype, dims, content, fvals = None, [], None, []
    for fno, wt, val in _fields(memoryview(buf)):
        if fno == 1:
            dtype = val
        elif fno == 2:
            for f2, _, v2 in _fields(val):
                if f2 == 2:          # dim
                    size = 0
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            size = v3
                    dims.append(size)
        elif fno == 4:
            content = val                      # memoryview into the record: no copy
        elif fno == 5:
            if wt == 2:
                fvals.extend(np.frombuffer(bytes(val), dtype="<f4").tolist())
            else:
                fvals.append(struct.unpack("<f", struct.pack("<I", val))[0])
    if dtype != _DT_FLOAT:
        raise ValueError("TensorProto dtype %r is not DT_FLOAT" % dtype)
    if content is not None:
        arr = np.frombuffer(content, dtype="<f4")
    else:
        arr = np.asarray(fvals, dtype=np.float32)
        n = int(np.prod(dims)) if dims else 1
        if arr.size == 1 and n > 1:
            arr = np.full(n, arr[0], dtype=np.float32)
    return arr.reshape(dims)


def serialize_tensor(arr):
    """tf.io.serialize_tensor(arr.astype(float32)): TensorProto{dtype, tensor_shape, tensor_content}."""
    arr = np.ascontiguousarray(arr, dtype="<f4")
    shape = b"".join(_enc_field(2, _enc_varint(1 << 3) + _enc_varint(int(d))) for d in arr.shape)
    return _enc_varint(1 << 3) + _enc_varint(_DT_FLOAT) + _enc_field(2, shape) + _enc_field(4, arr.tobytes())


def parse_example(buf):
    """tf.io.parse_single_example for the two features the reference writes -> (float32 array, int label)."""
    feats = {}
    for fno, _, val in _fields(memoryview(buf)):
        if fno != 1:
            continue
        for f2, _, entry in _fields(val):          # Features.feature map entries
            if f2 != 1:
                continue
            key, feature = None, None
            for f3, _, v3 in _fields(entry):
                if f3 == 1:
                    key = bytes(v3).decode()
                elif f3 == 2:
                    feature = v3
            feats[key] = feature
    data = label = None
    for f, _, v in _fields(feats["features"]):
        if f == 1:                                   # bytes_list
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    data = parse_tensor(v2)
    for f, wt, v in _fields(feats["label"]):
        if f == 3:                                   # int64_list
            for f2, wt2, v2 in _fields(v):
                if f2 == 1:
                    label = _varint(v2, 0)[0] if wt2 == 2 else v2     # packed or not
    if data is None or label is None:
        raise ValueError("Example without 'features' / 'label'")
    if label >= 1 << 63:
        label -= 1 << 64
    return data, int(label)


def serialize_example(features, label):
    """data_gen/gen_tfrecord_data.py:25-33."""
    f_feat = _enc_field(1, _enc_field(1, serialize_tensor(features)))                       # Feature{bytes_list{value}}
    f_lab = _enc_field(3, _enc_field(1, _enc_varint(int(label))))                           # Feature{int64_list{packed value}}
    entries = b"".join(_enc_field(1, _enc_field(1, k.encode()) + _enc_field(2, v))
                       for k, v in (("features", f_feat), ("label", f_lab)))
    return _enc_field(1, entries)


# ---------------------------------------------------------------- TFRecord files
_ERRORS = {2: "truncated record header", 3: "corrupt length CRC", 4: "truncated record", 5: "corrupt data CRC"}


def index_records(buf, verify=True):
    """(offsets, lengths) int64 arrays of the record payloads of a whole shard held in `buf` (bytes / mmap / uint8
    array); verify checks the masked CRC-32C of every length field and payload in native code."""
    a, keep = _addr(buf)
    lib = _native()
    level = 2 if verify else 0
    n = lib.sar_tfrecord_index(a, len(keep), 0, None, None, 0)           # framing pass: record count
    if n >= 0:
        off, ln = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
        n = lib.sar_tfrecord_index(a, len(keep), level, off.ctypes.data, ln.ctypes.data, n)
    if n == -1:
        raise IOError("sar_tfrecord_index: bad arguments")
    if n < 0:
        code, rec = 2 + (-n - 2) % 4, (-n - 2) // 4
        raise IOError("%s (record %d)" % (_ERRORS.get(code, "bad arguments"), rec))
    return off, ln


def read_records(path, verify=True):
    """yield the payload of every record of a shard as a zero-copy memoryview into the memory-mapped file"""
    if os.path.getsize(path) == 0:
        return
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    try:
        off, ln = index_records(buf, verify)
    except IOError as e:
        raise IOError("%s: %s" % (path, e)) from None
    view = memoryview(buf)
    for o, l in zip(off.tolist(), ln.tolist()):
        yield view[o:o + l]


def write_records(path, records):
    with open(path, "wb") as f:
        for rec in records:
            head = struct.pack("<Q", len(rec))
            f.write(head + struct.pack("<I", masked_crc(head)) + rec + struct.pack("<I", masked_crc(rec)))


def write_shards(data, labels, dest_folder, name, num_shards):
    """data_gen/gen_tfrecord_data.py:70-85 (shard rule included: a new file every len(labels)//num_shards clips)."""
    os.makedirs(dest_folder, exist_ok=True)
    per = max(len(labels) // num_shards, 1)
    paths = []
    for shard, start in enumerate(range(0, len(labels), per)):
        p = os.path.join(dest_folder, "%s-%d.tfrecord" % (name, shard))
        write_records(p, (serialize_example(data[i], labels[i]) for i in range(start, min(start + per, len(labels)))))
        paths.append(p)
    return paths


class TFRecordSkeletonData:
    """main_gnn.py:159-194 on a directory of shards: cyclic interleave of the files (num_parallel_reads = all of them),
    batch, shuffle of BATCHES with a bounded buffer (seeded here).

    Data parallel: when there are at least as many shards as ranks (and the split is balanced to 20 %), rank r reads ONLY
    shards r::world (tf.data's FILE auto-shard policy) and forms its own per-rank batches; otherwise every rank walks
    the record framing of all shards and keeps records r::world.  Either way a rank parses only the clips it trains on.  Every rank yields the same number of
    batches (the shortest rank's count, computed from the record framing alone) so the collective in the train step
    never deadlocks.  The shuffle buffer is counted in PER-RANK batches and capped by `shuffle_bytes` of host memory.
    Parsing runs in a background thread into pinned buffers (sar_amd/data.py:prefetch_to_device)."""

    def __init__(self, directory, verify_crc=True, num_classes=None):
        """verify_crc: True / "full" = framing + length CRCs + data CRCs (what tf.data's reader does), "length" = framing +
        length CRCs only (a flipped payload byte goes unnoticed; the CRC pass over 180 KB per clip is skipped), False /
        "off" = framing only (main_gnn.py --verify-crc)."""
        self.files = sorted(os.path.join(directory, f) for f in os.listdir(directory) if f.endswith("tfrecord"))
        if not self.files:
            raise FileNotFoundError("no *.tfrecord shard in %s" % directory)
        level = {True: 2, "full": 2, "length": 1, False: 0, "off": 0, None: 0}[verify_crc]
        self.verify_level = level
        self.verify = level == 2           # host_batches(): the single-threaded reference path
        self.num_classes = num_classes
        self._counts = None

    def counts(self):
        if self._counts is None:
            self._counts = [len(index_records(np.memmap(f, dtype=np.uint8, mode="r"), verify=False)[0])
                            if os.path.getsize(f) else 0 for f in self.files]
        return self._counts

    def __len__(self):
        return sum(self.counts())

    def _plan(self, rank, world):
        """(files, record stride, record phase, clips this rank will see, min over ranks of that)"""
        cnt = self.counts()
        if world > 1 and len(self.files) >= world:
            per_rank = [sum(cnt[r::world]) for r in range(world)]
            if min(per_rank) * 10 >= max(per_rank) * 8:       # balanced enough: at most 20 % of a rank's clips go unused
                return self.files[rank::world], 1, 0, per_rank[rank], min(per_rank)
        total = sum(cnt)
        per_rank = [len(range(r, total, world)) for r in range(world)]
        return self.files, world, rank, per_rank[rank], min(per_rank)

    @staticmethod
    def _interleave(files, verify):
        its = [read_records(f, verify) for f in files]
        while its:
            alive = []
            for it in its:
                rec = next(it, None)
                if rec is not None:
                    alive.append(it)
                    yield rec
            its = alive

    def host_batches(self, batch_size, rank=0, world=1, shuffle=False, epoch=0, drop_remainder=True, shuffle_size=1000,
                     shuffle_bytes=2 << 30):
        """numpy (x (n,C,T,V,M) float32, y (n,) int64) per-rank batches"""
        files, stride, phase, mine, fewest = self._plan(rank, world)
        n_batches = fewest // batch_size if drop_remainder else -(-mine // batch_size)

        def batches():
            xs, ys, made = [], [], 0
            if n_batches == 0:
                return
            for i, rec in enumerate(self._interleave(files, self.verify)):
                if i % stride != phase:
                    continue
                x, y = parse_example(rec)
                if self.num_classes is not None and not 0 <= y < self.num_classes:
                    raise ValueError("label %d outside [0, %d)" % (y, self.num_classes))
                xs.append(x)
                ys.append(y)
                if len(xs) == batch_size:
                    yield np.stack(xs), np.asarray(ys, dtype=np.int64)
                    xs, ys, made = [], [], made + 1
                    if made == n_batches:
                        return
            if xs and made < n_batches:
                yield np.stack(xs), np.asarray(ys, dtype=np.int64)

        def shuffled(gen):
            rng = np.random.default_rng(4321 + epoch)      # same seed on every rank; each rank shuffles its own batches
            buf, cap = [], None
            for item in gen:
                if cap is None:
                    cap = max(1, min(shuffle_size, shuffle_bytes // max(item[0].nbytes, 1)))
                buf.append(item)
                if len(buf) > cap:
                    yield buf.pop(int(rng.integers(len(buf))))
            while buf:
                yield buf.pop(int(rng.integers(len(buf))))

        return shuffled(batches()) if shuffle else batches()

    def plan(self, batch_size, rank=0, world=1, shuffle=False, epoch=0, drop_remainder=True, shuffle_size=1000,
             shuffle_bytes=2 << 30):
        """The batches host_batches() would yield, as index work for parallel parsing: a list of jobs, each a list of
        (shard buffer, payload offset, payload length).  Framing (+ length CRCs unless verification is off) of every shard
        this rank reads is checked here, in native code; the data CRC of a record is checked by the thread that parses it."""
        files, stride, phase, mine, fewest = self._plan(rank, world)
        n_batches = fewest // batch_size if drop_remainder else -(-mine // batch_size)
        bufs, offs, lens = [], [], []
        for f in files:
            b, o, ln = self._index(f)
            bufs.append(b), offs.append(o), lens.append(ln)
        recs = []
        for j in range(max([len(o) for o in offs] + [0])):          # cyclic interleave of the shards (main_gnn.py:167-171)
            for i in range(len(files)):
                if j < len(offs[i]):
                    recs.append((bufs[i], offs[i][j], lens[i][j]))
        recs = recs[phase::stride]
        jobs = [recs[b * batch_size:(b + 1) * batch_size] for b in range(n_batches)]
        jobs = [j for j in jobs if j]
        if shuffle and jobs:       # the bounded shuffle buffer of host_batches(), run over batch indices (same draws)
            rng = np.random.default_rng(4321 + epoch)
            x0, _ = parse_example(self._payload(jobs[0][0]))
            cap = max(1, min(shuffle_size, shuffle_bytes // max(x0.nbytes * batch_size, 1)))
            order, buf = [], []
            for b in range(len(jobs)):
                buf.append(b)
                if len(buf) > cap:
                    order.append(buf.pop(int(rng.integers(len(buf)))))
            while buf:
                order.append(buf.pop(int(rng.integers(len(buf)))))
            jobs = [jobs[b] for b in order]
        return jobs

    def _index(self, f):
        """(memory map, payload offsets, payload lengths) of one shard, verified once (framing, + length CRCs unless
        verification is off) and kept: every epoch re-plans from the same index"""
        cache = self.__dict__.setdefault("_index_cache", {})
        if f in cache:
            return cache[f]
        if os.path.getsize(f) == 0:
            cache[f] = (None, [], [])
            return cache[f]
        lib = _native()
        buf = np.memmap(f, dtype=np.uint8, mode="r")
        a, keep = _addr(buf)
        n = lib.sar_tfrecord_index(a, len(keep), 0, None, None, 0)
        if n >= 0:
            off, ln = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
            n = lib.sar_tfrecord_index(a, len(keep), min(self.verify_level, 1), off.ctypes.data, ln.ctypes.data, n)
        if n == -1:
            raise IOError("%s: sar_tfrecord_index: bad arguments" % f)
        if n < 0:
            code, rec = 2 + (-n - 2) % 4, (-n - 2) // 4
            raise IOError("%s: %s (record %d)" % (f, _ERRORS.get(code, "bad arguments"), rec))
        cache[f] = (buf, off.tolist(), ln.tolist())
        return cache[f]

    @staticmethod
    def _payload(rec):
        buf, off, ln = rec
        return memoryview(buf)[off:off + ln]

    def batches(self, batch_size, rank, world, device, shuffle, epoch=0, drop_remainder=True, shuffle_size=1000, workers=None):
        """device batches, parsed by `workers` host threads straight into pinned memory (sar_amd/data.py:parallel_batches)"""
        from .data import LOADER_THREADS, parallel_batches
        jobs = self.plan(batch_size, rank, world, shuffle, epoch, drop_remainder, shuffle_size)
        if not jobs:
            return iter(())
        lib = _native()
        x0, _ = parse_example(self._payload(jobs[0][0]))
        clip = x0.shape

        def build(job, x, y):
            for k, rec in enumerate(job):
                buf, off, ln = rec
                if self.verify_level == 2:      # masked CRC-32C of the payload against the record's footer (native, no GIL)
                    a, keep = _addr(buf[off:off + ln + 4])
                    if lib.sar_masked_crc32c(a, ln) != int.from_bytes(bytes(keep[ln:ln + 4]), "little"):
                        raise IOError("corrupt data CRC (payload at offset %d)" % off)
                xk, yk = parse_example(memoryview(buf)[off:off + ln])
                if self.num_classes is not None and not 0 <= yk < self.num_classes:
                    raise ValueError("label %d outside [0, %d)" % (yk, self.num_classes))
                np.copyto(x[k], xk)
                y[k] = yk

        return parallel_batches(jobs, build, lambda job: (len(job),) + tuple(clip), device,
                                workers=LOADER_THREADS if workers is None else workers)

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


def _crc_table():
    poly = 0x82F63B78        # CRC-32C (Castagnoli), reflected
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        tab.append(c)
    return np.array(tab, dtype=np.uint32)


_TAB = _crc_table()


def crc32c(data):
    c = 0xFFFFFFFF
    tab = _TAB
    for b in bytes(data):
        c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


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
            content = bytes(val)
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
def read_records(path, verify=True):
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError("%s: truncated record header" % path)
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if verify and masked_crc(head[:8]) != lcrc:
                raise IOError("%s: corrupt length CRC" % path)
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise IOError("%s: truncated record" % path)
            if verify and masked_crc(data) != struct.unpack("<I", tail)[0]:
                raise IOError("%s: corrupt data CRC" % path)
            yield data


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
    """main_gnn.py:159-194 on a directory of shards: cyclic interleave of the files, batch, shuffle of BATCHES with a
    bounded buffer (seeded here).  Clips are parsed on the host and moved to the device per batch; with world > 1 every
    rank walks the same batch sequence and takes rows rank::world of each global batch."""

    def __init__(self, directory, verify_crc=True):
        self.files = sorted(os.path.join(directory, f) for f in os.listdir(directory) if f.endswith("tfrecord"))
        if not self.files:
            raise FileNotFoundError("no *.tfrecord shard in %s" % directory)
        self.verify = verify_crc
        self._n = None

    def __len__(self):
        if self._n is None:
            self._n = sum(1 for f in self.files for _ in read_records(f, verify=False))
        return self._n

    def _interleaved(self):
        its = [read_records(f, self.verify) for f in self.files]
        while its:
            alive = []
            for it in its:
                rec = next(it, None)
                if rec is not None:
                    alive.append(it)
                    yield parse_example(rec)
            its = alive

    def batches(self, batch_size, rank, world, device, shuffle, epoch=0, drop_remainder=True, shuffle_size=1000):
        import torch
        gbs = batch_size * world

        def global_batches():
            xs, ys = [], []
            for x, y in self._interleaved():
                xs.append(x)
                ys.append(y)
                if len(xs) == gbs:
                    yield np.stack(xs), np.asarray(ys, dtype=np.int64)
                    xs, ys = [], []
            if xs and not drop_remainder:
                yield np.stack(xs), np.asarray(ys, dtype=np.int64)

        def shuffled(gen):
            rng = np.random.default_rng(4321 + epoch)
            buf = []
            for item in gen:
                buf.append(item)
                if len(buf) > shuffle_size:
                    yield buf.pop(int(rng.integers(len(buf))))
            while buf:
                yield buf.pop(int(rng.integers(len(buf))))

        gen = shuffled(global_batches()) if shuffle else global_batches()
        for x, y in gen:
            x, y = x[rank::world], y[rank::world]
            yield torch.from_numpy(np.ascontiguousarray(x)).to(device, non_blocking=True).float(), torch.from_numpy(y).to(device)

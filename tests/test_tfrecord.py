"""TFRecord reader / writer (SURVEY 8(f) input formats) without TensorFlow.  The hand-written protobuf codec is pinned
against the real protobuf runtime: Example / Feature / TensorProto message classes are built at test time from
descriptors that restate TensorFlow's .proto field numbers, serialized by google.protobuf, and must parse identically
(and vice versa)."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "skeleton-action-recognition_amd"))
from sar_amd import tfrecord as T  # noqa: E402


def _tf_messages():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="sar_tf_subset.proto", package="sartf", syntax="proto3")

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m

    def field(m, name, number, ftype, label=1, type_name=None, packed=None):
        f = m.field.add()
        f.name, f.number, f.type, f.label = name, number, ftype, label
        if type_name:
            f.type_name = type_name
        return f
    F = descriptor_pb2.FieldDescriptorProto
    m = msg("BytesList"); field(m, "value", 1, F.TYPE_BYTES, 3)
    m = msg("FloatList"); field(m, "value", 1, F.TYPE_FLOAT, 3)
    m = msg("Int64List"); field(m, "value", 1, F.TYPE_INT64, 3)
    m = msg("Feature")
    field(m, "bytes_list", 1, F.TYPE_MESSAGE, 1, ".sartf.BytesList")
    field(m, "float_list", 2, F.TYPE_MESSAGE, 1, ".sartf.FloatList")
    field(m, "int64_list", 3, F.TYPE_MESSAGE, 1, ".sartf.Int64List")
    m = msg("FeatureEntry")          # map<string, Feature> is a repeated {key=1, value=2} message on the wire
    field(m, "key", 1, F.TYPE_STRING); field(m, "value", 2, F.TYPE_MESSAGE, 1, ".sartf.Feature")
    m = msg("Features"); field(m, "feature", 1, F.TYPE_MESSAGE, 3, ".sartf.FeatureEntry")
    m = msg("Example"); field(m, "features", 1, F.TYPE_MESSAGE, 1, ".sartf.Features")
    m = msg("Dim"); field(m, "size", 1, F.TYPE_INT64); field(m, "name", 2, F.TYPE_STRING)
    m = msg("TensorShapeProto"); field(m, "dim", 2, F.TYPE_MESSAGE, 3, ".sartf.Dim")
    m = msg("TensorProto")
    field(m, "dtype", 1, F.TYPE_INT32); field(m, "tensor_shape", 2, F.TYPE_MESSAGE, 1, ".sartf.TensorShapeProto")
    field(m, "version_number", 3, F.TYPE_INT32); field(m, "tensor_content", 4, F.TYPE_BYTES)
    field(m, "float_val", 5, F.TYPE_FLOAT, 3)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("sartf." + n))  # noqa: E731
    return {n: get(n) for n in ("Example", "FeatureEntry", "Feature", "TensorProto", "Dim")}


def test_crc32c_known_answers_and_mask():
    assert T.crc32c(b"123456789") == 0xE3069283          # the CRC-32C check value
    assert T.crc32c(b"") == 0
    c = T.crc32c(b"123456789")
    assert T.masked_crc(b"123456789") == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_codec_against_the_protobuf_runtime():
    M = _tf_messages()
    rng = np.random.default_rng(0)
    clip = rng.standard_normal((3, 7, 25, 2)).astype(np.float32)
    # protobuf runtime -> our parser
    tp = M["TensorProto"](dtype=1, tensor_content=clip.tobytes())
    for d in clip.shape:
        tp.tensor_shape.dim.add(size=d)
    ex = M["Example"]()
    e = ex.features.feature.add(key="features"); e.value.bytes_list.value.append(tp.SerializeToString())
    e = ex.features.feature.add(key="label"); e.value.int64_list.value.append(41)
    data, label = T.parse_example(ex.SerializeToString())
    assert label == 41 and data.dtype == np.float32 and np.array_equal(data, clip)
    # our writer -> protobuf runtime
    ex2 = M["Example"].FromString(T.serialize_example(clip, 59))
    feats = {f.key: f.value for f in ex2.features.feature}
    assert list(feats["label"].int64_list.value) == [59]
    tp2 = M["TensorProto"].FromString(feats["features"].bytes_list.value[0])
    assert tp2.dtype == 1 and [d.size for d in tp2.tensor_shape.dim] == list(clip.shape)
    assert np.array_equal(np.frombuffer(tp2.tensor_content, dtype="<f4").reshape(clip.shape), clip)
    # float_val form (what TensorFlow emits for small / constant tensors) and a negative label
    tp3 = M["TensorProto"](dtype=1, float_val=[1.5, -2.0, 3.25, 0.0])
    tp3.tensor_shape.dim.add(size=2); tp3.tensor_shape.dim.add(size=2)
    assert np.array_equal(T.parse_tensor(tp3.SerializeToString()), np.array([[1.5, -2.0], [3.25, 0.0]], dtype=np.float32))
    assert T.parse_example(T.serialize_example(clip, -3))[1] == -3


def test_shards_roundtrip_interleave_and_corruption(tmp_path):
    rng = np.random.default_rng(1)
    n = 23
    data = rng.standard_normal((n, 3, 5, 25, 2)).astype(np.float32)
    labels = rng.integers(0, 60, n)
    paths = T.write_shards(data, labels, str(tmp_path / "train_data_joint"), "train_data_joint", 4)
    assert len(paths) == 5                               # 23 // 4 = 5 clips per shard -> 5 shards (the reference's rule)
    ds = T.TFRecordSkeletonData(str(tmp_path / "train_data_joint"))
    assert len(ds) == n
    got = [T.parse_example(r) for r in ds._interleave(ds.files, True)]
    # cyclic interleave: record j of shard s comes out at position j*n_shards + s while every shard is alive
    order = [s * 5 + j for j in range(5) for s in range(5) if s * 5 + j < n]
    assert [lab for _, lab in got] == [int(labels[i]) for i in order]
    assert all(np.array_equal(x, data[i]) for (x, _), i in zip(got, order))
    import torch
    seen = []
    for rank in range(2):
        for x, y in ds.batches(4, rank, 2, torch.device("cpu"), shuffle=True, epoch=3):
            assert x.shape == (4, 3, 5, 25, 2) and x.dtype == torch.float32
            seen.extend(y.tolist())
    # 5 shards >= 2 ranks: rank 0 reads shards 0,2,4 (13 clips), rank 1 shards 1,3 (10 clips); both yield the shorter
    # rank's 2 full batches
    assert len(seen) == 16
    # a flipped byte is caught by the data CRC
    raw = bytearray(open(paths[0], "rb").read())
    raw[40] ^= 0x01
    open(paths[0], "wb").write(bytes(raw))
    with pytest.raises(IOError):
        list(T.read_records(paths[0]))
    # the length CRC of an intact shard follows the masked CRC-32C rule
    h = open(paths[1], "rb").read(12)
    assert T.masked_crc(h[:8]) == struct.unpack("<I", h[8:])[0]


def test_truncated_shards_are_rejected_at_every_cut(tmp_path):
    """a shard that ends inside a record -- in particular with 12..15 bytes of it (a whole header but not
    header + footer) -- must raise 'truncated', at every verification level, never index past the end of the file."""
    from sar_amd import _lib
    lib = _lib.load()
    recs = [bytes(range(40)), b"", bytes(7)]
    path = str(tmp_path / "t.tfrecord")
    T.write_records(path, recs)
    whole = open(path, "rb").read()
    assert [bytes(r) for r in T.read_records(path)] == recs
    starts = [0, 16 + 40, 16 + 40 + 16]                       # first byte of each record
    for rec_i, st in enumerate(starts):
        for extra in list(range(1, 16)) + [16 + len(recs[rec_i]) - 1]:
            if extra >= 16 + len(recs[rec_i]):
                continue
            cut = np.frombuffer(whole[:st + extra], dtype=np.uint8).copy()
            for level in (0, 1, 2):
                rc = lib.sar_tfrecord_index(cut.ctypes.data, len(cut), level, None, None, 0)
                assert rc < -1, (rec_i, extra, level, rc)
                code, rec = 2 + (-rc - 2) % 4, (-rc - 2) // 4
                assert rec == rec_i and code == (2 if extra < 12 else 4), (rec_i, extra, level, rc)
            with pytest.raises(IOError, match="truncated"):
                T.index_records(cut)
    # bad arguments are reported as such, not decoded as a record error
    assert lib.sar_tfrecord_index(None, 10, 0, None, None, 0) == -1
    with pytest.raises(IOError, match="truncated"):
        open(path, "wb").write(whole[:-2])
        list(T.read_records(path))


def test_crc32c_instruction_path_equals_table_path():
    """sar_crc32c (SSE4.2 crc32 instruction when present) against sar_crc32c_sw (slice-by-8 tables) on every length /
    alignment class, and against the bit-serial definition."""
    from sar_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7)
    buf = rng.integers(0, 256, 5000, dtype=np.uint8)

    def bitwise(b):
        c = 0xFFFFFFFF
        for x in b:
            c ^= int(x)
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        return c ^ 0xFFFFFFFF
    for off in range(9):
        for n in (0, 1, 7, 8, 9, 63, 64, 65, 1000, 4099):
            a = buf[off:off + n]
            hw, sw = lib.sar_crc32c(a.ctypes.data, n), lib.sar_crc32c_sw(a.ctypes.data, n)
            assert hw == sw
            if n <= 65:
                assert hw == bitwise(a)


def test_rank_sharding_disjoint_and_equal_batch_counts(tmp_path):
    """Every clip a rank parses is one it trains on; ranks see disjoint clips and the same number of batches -- with
    fewer shards than ranks (records r::world) and with at least as many (files r::world)."""
    rng = np.random.default_rng(2)
    n = 41
    data = np.arange(n, dtype=np.float32)[:, None, None, None, None] + np.zeros((n, 3, 4, 25, 2), dtype=np.float32)
    labels = rng.integers(0, 60, n)
    for num_shards, world in ((1, 3), (2, 3), (8, 3), (4, 4)):
        d = str(tmp_path / ("s%d_w%d" % (num_shards, world)))
        T.write_shards(data, labels, d, "x", num_shards)
        ds = T.TFRecordSkeletonData(d, num_classes=60)
        per_rank = []
        for r in range(world):
            ids = []
            for x, y in ds.host_batches(5, r, world, shuffle=True, epoch=1):
                assert x.shape == (5, 3, 4, 25, 2)
                ids.extend(int(v) for v in x[:, 0, 0, 0, 0])
                assert all(int(labels[int(i)]) == int(l) for i, l in zip(x[:, 0, 0, 0, 0], y))
            per_rank.append(ids)
        assert len({len(p) for p in per_rank}) == 1 and len(per_rank[0]) > 0
        flat = [i for p in per_rank for i in p]
        assert len(flat) == len(set(flat))
    with pytest.raises(ValueError):
        list(T.TFRecordSkeletonData(d, num_classes=3).host_batches(5))


def _loader_rates(tmp_path, n, device, reps=3):
    import pickle
    import time

    import torch
    from sar_amd.data import NpySkeletonData
    rng = np.random.default_rng(3)
    data = rng.standard_normal((n, 3, 300, 25, 2)).astype(np.float32)
    labels = rng.integers(0, 60, n)
    d = str(tmp_path / "shards")
    T.write_shards(data, labels, d, "train_data_joint", 8)
    np.save(str(tmp_path / "train_data_joint.npy"), data)
    with open(str(tmp_path / "train_label.pkl"), "wb") as f:
        pickle.dump((["s%d" % i for i in range(n)], labels.tolist()), f)
    del data
    rates = {}
    for name, ds in (("tfrecord", T.TFRecordSkeletonData(d, verify_crc="full", num_classes=60)),
                     ("npy", NpySkeletonData(str(tmp_path / "train_data_joint.npy"), str(tmp_path / "train_label.pkl"), 60))):
        best = 0.0
        for rep in range(reps):
            t0 = time.perf_counter()
            clips, checksum = 0, 0.0
            for x, y in ds.batches(64, 0, 1, device, shuffle=True, epoch=rep):
                clips += x.shape[0]
            if device.type == "cuda":
                torch.cuda.synchronize()
            best = max(best, clips / (time.perf_counter() - t0))
        assert clips == n
        rates[name] = best
    return rates


def test_host_loaders_outrun_one_gpu(tmp_path):
    """the host side of the input pipeline (TFRecord shards WITH full CRC verification, and the .npy + .pkl
    pair), through the loader threads.  Full-size NTU clips (3,300,25,2) = 180 KB each.  This CPU-suite version runs
    wherever the suite runs (an 8-vCPU sandbox delivers 4 000-9 000 clips/s, with little gain from threads: it is bound by
    the sandbox's memory system); the >= 10 000 clips/s bar of VERDICT r02 #6 -- twice one MI355X's bf16 training rate --
    is asserted on the GPU box's host by test_host_loaders_have_headroom_on_the_gpu_box below."""
    import torch
    rates = _loader_rates(tmp_path, 1024, torch.device("cpu"))
    print("host loader clips/s:", {k: round(v) for k, v in rates.items()})
    assert rates["tfrecord"] >= 2500 and rates["npy"] >= 2500, rates


@pytest.mark.gpu
def test_host_loaders_have_headroom_on_the_gpu_box(tmp_path):
    """with the loader threads, pinned buffers and asynchronous H2D copies the TFRecord reader (framing,
    length and data CRC-32C verified) and the .npy reader must each deliver >= 10 000 clips/s per rank on the GPU box's
    host -- at least twice the bf16 training rate of one MI355X."""
    import torch
    rates = _loader_rates(tmp_path, 4096, torch.device("cuda:0"))
    print("host loader clips/s into HBM:", {k: round(v) for k, v in rates.items()})
    assert rates["tfrecord"] >= 10000 and rates["npy"] >= 10000, rates


def test_parallel_loader_equals_the_sequential_reader(tmp_path):
    """The threaded loader (plan() + parallel_batches) yields exactly the batches of the single-threaded host_batches()
    generator, in the same order, for every sharding / shuffle / remainder combination; a corrupted payload byte is caught
    by whichever thread parses it; 'length' verification skips exactly that check."""
    import torch
    rng = np.random.default_rng(11)
    n = 53
    data = rng.standard_normal((n, 3, 6, 25, 2)).astype(np.float32)
    labels = rng.integers(0, 60, n)
    d = str(tmp_path / "s")
    paths = T.write_shards(data, labels, d, "x", 4)
    ds = T.TFRecordSkeletonData(d, num_classes=60)
    cpu = torch.device("cpu")
    for world in (1, 2, 3):
        for shuffle in (False, True):
            for drop in (True, False):
                if world > 1 and not drop:
                    continue
                for r in range(world):
                    ref = list(ds.host_batches(5, r, world, shuffle=shuffle, epoch=2, drop_remainder=drop))
                    # (a CPU consumer's batch is valid until the next one is requested: copy)
                    got = [(xg.numpy().copy(), yg.numpy().copy())
                           for xg, yg in ds.batches(5, r, world, cpu, shuffle, epoch=2, drop_remainder=drop, workers=3)]
                    assert len(ref) == len(got) and len(ref) > 0
                    for (xr, yr), (xg, yg) in zip(ref, got):
                        assert np.array_equal(xr, xg) and np.array_equal(yr, yg)
    raw = bytearray(open(paths[1], "rb").read())
    raw[60] ^= 0x40                                   # inside the first payload
    open(paths[1], "wb").write(bytes(raw))
    with pytest.raises(IOError, match="CRC"):
        list(T.TFRecordSkeletonData(d, verify_crc="full").batches(5, 0, 1, cpu, False))
    assert len(list(T.TFRecordSkeletonData(d, verify_crc="length").batches(5, 0, 1, cpu, False))) == n // 5

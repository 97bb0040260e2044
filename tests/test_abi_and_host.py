"""CPU: the C-ABI library loads without a GPU and exports every symbol include/sar_hip.h declares;
host-side logic (LR schedule, sharding, bone table) matches the reference's semantics."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "sar_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(sar_[a-z0-9_]+)\s*\(", hdr))


def test_library_exports_every_declared_symbol():
    from sar_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "skeleton-action-recognition_amd", "csrc"), "-j4"])
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libsar_hip.so does not export %s" % name
    # the Python binding table covers the header exactly (no stale / missing prototypes)
    assert set(_lib.SIGNATURES) == declared
    assert _lib.load().sar_version() >= 100


def test_descriptor_struct_layout_matches_header():
    """ctypes mirrors of sar_conv_desc / sar_wgrad_desc: field order and natural alignment."""
    from sar_amd import _lib
    assert ctypes.sizeof(_lib.ConvDesc) == 18 * 4 + 22 * 8      # round 4: + aux2, ld_aux2, aux_mask (SAR_EPI_ADD_GATE)
    assert ctypes.sizeof(_lib.WgradDesc) == 16 * 4 + 14 * 8
    assert _lib.ConvDesc.src.offset == 72 and _lib.WgradDesc.src.offset == 64
    lib = _lib.load()                                   # the library reports the sizes it was compiled with
    assert lib.sar_struct_size(0) == ctypes.sizeof(_lib.ConvDesc)
    assert lib.sar_struct_size(1) == ctypes.sizeof(_lib.WgradDesc)


def test_argument_errors_without_gpu():
    """Argument validation happens before any GPU work: callable on a CPU-only box."""
    from sar_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    assert lib.sar_conv_gemm_f32(ctypes.byref(d), None) == -1
    assert b"sar_conv_gemm" in lib.sar_last_error_string()
    assert lib.sar_sgd_nesterov_f32(None, None, None, 0, None, 0.9, None) == -1
    assert lib.sar_stft_logmag_f32(None, None, 1, 100, 256, 16, None, 0, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sar_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SarError, match="no CPU fallback"):
        _lib.load()


def test_lr_schedule_matches_reference_piecewise_decay():
    from sar_amd.train import lr_schedule
    # main_gnn.py:303-308: boundaries (step*40000)//batch_size = 6250, 31250 for the defaults
    assert lr_schedule(0) == 0.1 and lr_schedule(6250) == 0.1
    assert lr_schedule(6251) == pytest.approx(0.01) and lr_schedule(31250) == pytest.approx(0.01)
    assert lr_schedule(31251) == pytest.approx(0.001)
    assert lr_schedule(1000, base_lr=0.05, steps=(1,), batch_size=80) == pytest.approx(0.005)


def test_shard_indices_partition_each_global_batch():
    from sar_amd.train import shard_indices
    perm = list(np.random.default_rng(0).permutation(1000))
    shards = [shard_indices(perm, r, 4, 64) for r in range(4)]
    assert all(len(s) == 1000 // 64 for s in shards)           # remainder dropped (drop_remainder=True)
    for b in range(1000 // 64):
        got = sorted(int(i) for r in range(4) for i in shards[r][b])
        assert got == sorted(int(i) for i in perm[b * 64:(b + 1) * 64])
        assert all(len(shards[r][b]) == 16 for r in range(4))


def test_bone_pairs_match_reference_table(golden_dir):
    from sar_amd.bone import NTU_BONE_PAIRS, bone_parent_array
    gold = json.load(open(os.path.join(golden_dir, "bone_pairs.json")))
    assert [list(p) for p in NTU_BONE_PAIRS] == gold["xsub"] == gold["xview"]
    bp = bone_parent_array()
    assert bp[20] == 20 and bp[0] == 1 and (bp >= 0).all()

"""GPU: the drop-in surface -- models.stgcn.Model called the Keras way, autograd + a torch optimizer, and the
main_gnn.py command line on synthetic data."""
import json
import os

import numpy as np
import subprocess
import sys

import pytest
import torch

from oracle import stgcn as O
from util import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_call_signature_and_autograd_match_oracle():
    from models.stgcn import Model
    dev = torch.device("cuda:0")
    model = Model(num_classes=60, device=dev)
    names = [v.name for v in model.trainable_variables]
    assert not any("adjacency_matrix" in n for n in names)              # main_gnn.py:228-232 filter is a no-op
    assert any(v.name == "adjacency_matrix" and not v.trainable for v in model.variables)
    assert sum(p.numel() for p in model.parameters()) == 3080082
    p = O.init_params(60, seed=0, dtype=torch.float64)
    p.update({k: v.detach().cpu().double() for k, v in model.engine.state_dict().items() if k in p})
    x, y = O.synthetic_batch(2, seed=1, T=300)
    logits = model(x.to(dev), training=True)                            # Keras-style call
    loss = torch.nn.functional.cross_entropy(logits, y.to(dev), reduction="sum") / 2
    loss.backward()
    lref, loss_ref, gref, _, _ = O.loss_and_grads(p, x.double(), y)
    assert rel_err(logits.detach().cpu(), lref) < 1e-4
    assert rel_err(loss.detach().cpu().reshape(1), loss_ref.reshape(1)) < 1e-4
    g = model.logits_kernel.grad
    assert g is not None and rel_err(g.cpu(), gref["logits.kernel"]) < 1e-3
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, nesterov=True)
    before = model.l9_tcn_kernel.detach().clone()
    opt.step()
    assert not torch.equal(before, model.l9_tcn_kernel.detach())
    assert torch.equal(model.l9_tcn_kernel.detach(), model.engine.p["l9.tcn.kernel"])   # parameters ARE the flat buffer
    model.eval()
    with torch.no_grad():
        logits_eval = model(x.to(dev), training=False)
    assert logits_eval.shape == (2, 60) and torch.isfinite(logits_eval).all()


def test_main_gnn_cli_trains_on_synthetic_data(tmp_path):
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "skeleton-action-recognition_amd"))
    cmd = [sys.executable, os.path.join(ROOT, "skeleton-action-recognition_amd", "main_gnn.py"), "--model", "stgcn",
           "--synthetic", "--synthetic-size", "64", "--batch-size", "8", "--num-epochs", "2", "--save-freq", "1",
           "--max-iters", "3", "--log-dir", str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Saving checkpoint for epoch 2" in out.stdout
    runs = os.listdir(tmp_path)
    assert len(runs) == 1
    files = os.listdir(os.path.join(tmp_path, runs[0]))
    assert {"config.yaml", "stgcn.py", "scalars.jsonl", "checkpoints"} <= set(files)
    tags = {json.loads(line)["tag"] for line in open(os.path.join(tmp_path, runs[0], "scalars.jsonl"))}
    assert {"cross_entropy_loss", "train_acc", "train_acc_top_5", "epoch_test_acc", "epoch_test_acc_top_5"} <= tags
    # confusion matrix of the test set (main_gnn.py:410-416) and the class scores for stream fusion
    run = os.path.join(tmp_path, runs[0])
    cm = np.load(os.path.join(run, "confusion_matrix-2.npy"))
    assert cm.shape == (60, 60) and cm.sum() == 24
    # resume from the first checkpoint with another stream option: continues at epoch 2, iteration counter restored
    cmd2 = cmd + ["--resume", os.path.join(run, "checkpoints", "ckpt-1.pt"), "--save-scores"]
    out2 = subprocess.run(cmd2, env=env, capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    assert "Resumed from" in out2.stdout and "Epoch: 2" in out2.stdout and "Epoch: 1\n" not in out2.stdout
    assert os.listdir(tmp_path) == runs                  # same run directory (resume / save-scores are not part of its name)
    scores = np.load(os.path.join(run, "scores-2.npy"))
    assert scores.shape == (24, 60) and np.allclose(scores.sum(1), 1, atol=1e-4)


def test_main_gnn_cli_bf16_mode(tmp_path):
    """--mfma bf16 (not a reference flag): the same command line on the bf16-operand kernels; the flag is part of the
    run name only when it is not the default."""
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "skeleton-action-recognition_amd"))
    cmd = [sys.executable, os.path.join(ROOT, "skeleton-action-recognition_amd", "main_gnn.py"), "--model", "stgcn",
           "--synthetic", "--synthetic-size", "32", "--batch-size", "8", "--num-epochs", "1", "--save-freq", "1",
           "--max-iters", "2", "--mfma", "bf16", "--log-dir", str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    runs = os.listdir(tmp_path)
    assert len(runs) == 1 and "mfma:bf16" in runs[0]
    losses = [json.loads(line)["value"] for line in open(os.path.join(tmp_path, runs[0], "scalars.jsonl"))
              if json.loads(line)["tag"] == "cross_entropy_loss"]
    assert len(losses) == 2 and all(np.isfinite(losses))


def test_npy_input_pipeline_keeps_up_with_the_gpu(tmp_path):
    """SURVEY 8(f)-2 / VERDICT r01 #7: main_gnn.py fed from the reference's `<prefix>.npy` + label pkl pair (memory-mapped,
    gathered and copied to pinned buffers by the prefetch thread, asynchronous H2D) must train within 5 % of the rate of the
    same command on on-device synthetic clips.  bf16 configuration = the fastest consumer (~13.5 ms per 64 clips)."""
    import pickle
    import re
    n = 4096      # 64 steps per epoch (1024 until round 5: at 5 850 clips/s the loader's fixed per-epoch start cost crossed the 5 % line of a 0.18 s epoch)
    rng = np.random.default_rng(0)
    d = tmp_path / "xsub"
    d.mkdir()
    data = np.clip(0.12 * rng.standard_normal((n, 3, 300, 25, 2)), -1.1, 0.75).astype(np.float32)
    labels = rng.integers(0, 60, n)
    for split in ("train", "val"):
        np.save(str(d / ("%s_data_joint.npy" % split)), data if split == "train" else data[:128])
        with open(str(d / ("%s_label.pkl" % split)), "wb") as f:
            pickle.dump((["s%d" % i for i in range(n)], (labels if split == "train" else labels[:128]).tolist()), f)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "skeleton-action-recognition_amd"))
    base = [sys.executable, os.path.join(ROOT, "skeleton-action-recognition_amd", "main_gnn.py"), "--model", "stgcn", "--batch-size", "64",
            "--num-epochs", "3", "--mfma", "bf16", "--save-freq", "100", "--log-dir", str(tmp_path / "logs")]
    rates = {}
    for name, extra in (("synthetic", ["--synthetic", "--synthetic-size", str(n)]),
                        ("npy", ["--train-data-path", str(d / "train_data_joint"), "--test-data-path", str(d / "val_data_joint")])):
        out = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        got = [float(v) for v in re.findall(r"train: \d+ iters, ([0-9.]+) clips/s", out.stdout)]
        assert len(got) == 3, out.stdout
        rates[name] = max(got[1:])          # epochs 2-3: the first one pays the cold start
    print("main_gnn.py --mfma bf16 clips/s:", rates)
    assert rates["npy"] >= 0.95 * rates["synthetic"], rates


def test_tfrecord_input_pipeline_keeps_up_with_the_gpu(tmp_path):
    """VERDICT r02 missing #5 / next #6: the TFRecord twin of the test above -- main_gnn.py fed from the reference's shard
    format (data_gen/gen_tfrecord_data.py; framing, length AND data CRC-32C verified like tf.data's reader; parsed by the
    loader threads straight into pinned memory) must train within 5 % of the on-device synthetic rate in the bf16
    configuration, the fastest consumer."""
    import re
    from sar_amd import tfrecord as T
    n = 4096                 # 64 steps per epoch: the loader's per-epoch start (first batch) is a few per cent of it, as in a real epoch
    #                          (2048 until round 5: at 5 870 clips/s -- a fast box -- the fixed start cost of a 0.35 s epoch crossed the 5 % line)
    rng = np.random.default_rng(0)
    data = np.clip(0.12 * rng.standard_normal((n, 3, 300, 25, 2)), -1.1, 0.75).astype(np.float32)
    labels = rng.integers(0, 60, n)
    T.write_shards(data, labels, str(tmp_path / "train"), "train_data_joint", 8)
    T.write_shards(data[:128], labels[:128], str(tmp_path / "val"), "val_data_joint", 2)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "skeleton-action-recognition_amd"))
    base = [sys.executable, os.path.join(ROOT, "skeleton-action-recognition_amd", "main_gnn.py"), "--model", "stgcn", "--batch-size", "64",
            "--num-epochs", "3", "--mfma", "bf16", "--save-freq", "100", "--log-dir", str(tmp_path / "logs")]
    rates = {}
    for name, extra in (("synthetic", ["--synthetic", "--synthetic-size", str(n)]),
                        ("tfrecord", ["--train-data-path", str(tmp_path / "train"), "--test-data-path", str(tmp_path / "val"),
                                      "--verify-crc", "full"])):
        out = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        got = [float(v) for v in re.findall(r"train: \d+ iters, ([0-9.]+) clips/s", out.stdout)]
        assert len(got) == 3, out.stdout
        rates[name] = max(got[1:])
    print("main_gnn.py --mfma bf16 clips/s:", rates)
    assert rates["tfrecord"] >= 0.95 * rates["synthetic"], rates


def test_npy_loader_accepts_float64_and_float16_files(tmp_path):
    """the loader threads gather each batch with np.take(..., out=<pinned float32 slot>), which refuses to cast; a
    `*_data_joint.npy` saved as float64 (numpy's default) or float16 must still load, converted to float32."""
    import pickle
    from sar_amd.data import NpySkeletonData
    rng = np.random.default_rng(1)
    ref = np.clip(0.12 * rng.standard_normal((12, 3, 20, 25, 2)), -1.1, 0.75)
    with open(str(tmp_path / "l.pkl"), "wb") as f:
        pickle.dump((["s%d" % i for i in range(12)], list(range(12))), f)
    dev = torch.device("cuda", 0)
    for dt in (np.float64, np.float16, np.float32):
        np.save(str(tmp_path / "d.npy"), ref.astype(dt))
        data = NpySkeletonData(str(tmp_path / "d.npy"), str(tmp_path / "l.pkl"), num_classes=12)
        got = list(data.batches(4, 0, 1, dev, shuffle=False))
        assert len(got) == 3
        x = torch.cat([b[0] for b in got]).cpu().numpy()
        y = torch.cat([b[1] for b in got]).cpu().numpy()
        assert x.dtype == np.float32 and np.array_equal(x, ref.astype(dt).astype(np.float32)) and np.array_equal(y, np.arange(12))

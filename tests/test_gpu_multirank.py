"""The PRODUCT data-parallel step under world_size 2 on the HIP engines (ADVICE medium): two fresh
processes (torch.distributed.run, gloo rendezvous, both on cuda:0) each run Trainer.step / SpectrogramTrainer.step on
their shard; the all-reduced flat gradient and the updated weights must equal the single-process combination of the two
shards (per-replica BatchNorm statistics, loss scaled by the global batch: main_gnn.py:226,234,257-258), and the ranks'
parameters must be bit-identical after the step.  Also: bench.py --gpus N starts its own ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(workload, out):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "ddp_worker.py"), workload, str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return [torch.load(os.path.join(out, "rank%d.pt" % k)) for k in range(2)]


@pytest.mark.parametrize("workload", ["stgcn", "stgcn_bf16", "stgcn_split", "stgin", "spectrogram", "spectrogram_split"])
def test_two_rank_product_step_equals_single_process_combination(workload, tmp_path):
    import ddp_worker as W
    from sar_amd.train import shard_indices
    ranks = _launch(workload, tmp_path)
    # ranks agree bit for bit after the step (same all-reduced gradient, same optimizer kernel)
    assert torch.equal(ranks[0]["grad"], ranks[1]["grad"]) and torch.equal(ranks[0]["flat"], ranks[1]["flat"])
    assert ranks[0]["grad"].abs().max() > 0
    dev = torch.device("cuda", 0)
    n = 8
    x, y = W.global_batch(workload, n)
    shards = [shard_indices(list(range(n)), r, 2, n)[0] for r in range(2)]
    assert sorted(shards[0] + shards[1]) == list(range(n))
    singles = []
    for r in range(2):
        # world_size=2 without a process group: the step scales exactly as a rank does, the exchange is the identity
        eng, trainer = W.make_trainer(workload, dev, 2)
        eng_before = eng.flat.cpu().clone()
        # capture the un-reduced local gradient: run the engine half of the step only
        if workload in ("stgcn", "stgin", "stgcn_bf16", "stgcn_split"):
            eng.loss_and_grad(x[shards[r]].to(dev), y[shards[r]].to(dev), n)
            singles.append(dict(grad=eng.grad.cpu().clone()))
        else:
            img = trainer.model.spectrogram(x[shards[r]].to(dev))
            _, _, dimg = eng.loss_and_grad(img.detach(), y[shards[r]].to(dev), need_dx=True)
            trainer.radar_opt.zero_grad(set_to_none=False)
            img.backward(dimg)
            singles.append(dict(grad=eng.grad.cpu().clone(),
                                radar_grad=torch.cat([p.grad.reshape(-1) for p in trainer.radar_params]).cpu()))
        torch.cuda.synchronize()
    if workload in ("stgcn", "stgin", "stgcn_bf16", "stgcn_split"):
        total = singles[0]["grad"] + singles[1]["grad"]            # SUM of per-replica gradients (loss / global batch)
        assert torch.equal(ranks[0]["grad"], total)
        # the fused Nesterov step on the summed gradient from the common initial weights
        eng, trainer = W.make_trainer(workload, dev, 2)
        eng.grad.copy_(total.to(dev))
        eng.sgd_step(0.1)
        torch.cuda.synchronize()
        assert torch.equal(ranks[0]["flat"], eng.flat.cpu())
        assert not torch.equal(ranks[0]["flat"], eng_before)
    else:
        mean = (singles[0]["grad"] + singles[1]["grad"]) / 2       # DataParallel semantics: mean over replicas
        assert torch.equal(ranks[0]["grad"], mean)
        rmean = (singles[0]["radar_grad"] + singles[1]["radar_grad"]) / 2
        assert torch.equal(ranks[0]["radar_grad"], rmean) and torch.equal(ranks[1]["radar_grad"], rmean)
        assert rmean.abs().max() > 0
        for k in ("radar_location", "wavelength"):
            assert torch.equal(ranks[0][k], ranks[1][k])


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset must launch 2 ranks itself (before any GPU call in the parent)
    and relay rank 0's JSON line.  One GPU here: SAR_BENCH_SHARE_GPU=1 puts both ranks on cuda:0 over gloo."""
    env = dict(os.environ, SAR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-cpu-baseline", "--secondary", "bf16,pathB", "--sustained-steps", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["global_batch"] == 8 and out["value"] > 0
    # self-diagnosing scaling runs (VERDICT r02 #3c): every rank's own step time, the time of the gradient collective, and the
    # bf16 / Path B configurations measured by the same ranks in the same process
    assert len(out["per_rank_ms"]) == 2 and all(t > 0 for t in out["per_rank_ms"]) and out["allreduce_ms"] is not None
    assert abs(max(out["per_rank_ms"]) - out["ms_per_step"]) < 1e-6
    sec = out["secondary"]
    assert sec["bf16"]["n_gpus"] == 2 and sec["bf16"]["dtype"] == "bf16" and sec["bf16"]["value"] > 0 and sec["bf16"]["warm_s"] >= 2.5
    assert sec["pathB"]["n_gpus"] == 2 and sec["pathB"]["value"] > 0 and len(sec["pathB"]["per_rank_ms"]) == 2


def test_bench_eight_ranks_rehearsal(tmp_path):
    """the driver's 8-GPU command line rehearsed on ONE GPU -- `python bench.py --gpus 8 --quick --batch 2`
    with all eight ranks on cuda:0 over gloo (SAR_BENCH_SHARE_GPU=1): the self-launch, the rendezvous port, the MAX-agreed warm-up
    loop of Leg.run, per_rank_ms of length 8, every secondary leg at world 8, a clean exit.  (What it cannot rehearse is RCCL
    over xGMI itself: tests/test_gpu_rccl.py runs the RCCL branches on a one-rank communicator.)"""
    env = dict(os.environ, SAR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--quick", "--batch", "2", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline", "--sustained-steps", "0", "--warm-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["config"]["global_batch"] == 16 and out["value"] > 0
    assert len(out["per_rank_ms"]) == 8 and all(t > 0 for t in out["per_rank_ms"]) and out["allreduce_ms"] is not None
    assert out["grad_buckets"] == 3 and out["config"]["parallelism"] == "dp8"
    sec = out["secondary"]
    for name in ("f32_split", "bf16", "pathB", "pathB_pad250", "config5"):
        assert sec[name]["n_gpus"] == 8 and sec[name]["value"] > 0 and len(sec[name]["per_rank_ms"]) == 8, name
    assert sec["f32_split"]["dtype"] == "f32" and "f16x3a" in sec["f32_split"]["config"]["workload"]
    # the compact object at the END of the line carries every leg -- value, ms, fractions -- the box
    # calibration, and at world > 1 every rank's time per leg
    assert list(out)[-1] == "summary" and len(json.dumps(out["summary"])) < 2600
    summ = out["summary"]
    assert set(summ["legs"]) >= {"fp32", "f32_split", "bf16", "pathB", "pathB_f32_split", "pathB_pad250", "config5"}
    for name, (v, ms, frac, frac_box) in summ["legs"].items():
        assert v > 0 and ms > 0, name
    assert all(len(summ["per_rank_ms"][name]) == 8 for name in ("fp32", "f32_split", "bf16", "pathB", "pathB_f32_split", "config5"))
    assert len(summ["box"]) == 5 and all(b and b > 0 for b in summ["box"]) and out["box"]["copy_gbps"] > 0
    assert all(len(v) <= 160 for v in out["legend"].values())


def _write_npy_dataset(d, n, T, classes, split):
    """<d>/<split>_data_joint.npy + <d>/<split>_label.pkl (data_gen/gen_joint_data.py:138-151); clip i carries its id in
    its first coordinate so that a rank's trace says which clips it trained on."""
    import pickle

    import numpy as np
    rng = np.random.default_rng(5)
    x = (0.12 * rng.standard_normal((n, 3, T, 25, 2))).astype(np.float32).clip(-1.1, 0.75)
    x[:, 0, 0, 0, 0] = np.arange(n, dtype=np.float32) / 1024.0
    np.save(os.path.join(d, "%s_data_joint.npy" % split), x)
    with open(os.path.join(d, "%s_label.pkl" % split), "wb") as f:
        pickle.dump((["c%d" % i for i in range(n)], rng.integers(0, classes, n).tolist()), f)


def _run_cli(script, extra, tmp_path, world=2):
    trace = tmp_path / "trace"
    trace.mkdir()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SAR_DIST_BACKEND="gloo", SAR_TRACE_DIR=str(trace))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    pkg = os.path.join(ROOT, "skeleton-action-recognition_amd")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(pkg, script)] + extra + ["--log-dir", str(tmp_path / "logs")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=pkg)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return [json.load(open(trace / ("rank%d.json" % k))) for k in range(world)], r.stdout


def _check_cli_traces(traces, n_clips, bs, epochs):
    """disjoint shards covering each epoch's global batches, the same number of steps on every rank, identical parameters
    after training, log files on rank 0 only (main_gnn.py:257-258,290-301; main_spectrogram.py:118-121)"""
    assert traces[0]["digest"] == traces[1]["digest"]
    assert traces[0]["has_log"] and not traces[1]["has_log"]
    per_epoch = n_clips // (bs * 2)
    for ep in range(epochs):
        seen = []
        for t in traces:
            batches = [ids for e, ids in t["ids"] if e == ep]
            assert len(batches) == per_epoch and all(len(b) == bs for b in batches)
            seen.append(sorted(round(v * 1024) for b in batches for v in b))
        assert not set(seen[0]) & set(seen[1])                                   # disjoint shards
        assert len(set(seen[0]) | set(seen[1])) == per_epoch * bs * 2            # no clip twice in an epoch
        assert set(seen[0]) | set(seen[1]) <= set(range(n_clips))
    e0 = [sorted(round(v * 1024) for e, b in traces[0]["ids"] if e == 0 for v in b)]
    e1 = [sorted(round(v * 1024) for e, b in traces[0]["ids"] if e == 1 for v in b)]
    assert e0 != e1 or per_epoch * bs * 2 == n_clips                            # the shuffle depends on the epoch


@pytest.mark.parametrize("mfma", ["fp32", "bf16", "f32_split"])
def test_main_gnn_cli_under_two_ranks(mfma, tmp_path):
    """the REAL main_gnn.py as two ranks (gloo, both on cuda:0): rank > 0 branches -- sharded batches(),
    rank-0-only logging / evaluation / checkpoint, the per-epoch barrier -- for the fp32, the bf16 and (VERDICT r05 next #1d) the
    f32_split engine."""
    d = tmp_path / "data"
    d.mkdir()
    _write_npy_dataset(str(d), 64, 32, 10, "train")
    _write_npy_dataset(str(d), 16, 32, 10, "val")
    traces, out = _run_cli("main_gnn.py", ["--model", "stgcn", "--mfma", mfma, "--num-classes", "10", "--batch-size", "8", "--num-epochs", "2",
                                           "--save-freq", "2", "--train-data-path", str(d / "train_data_joint"),
                                           "--test-data-path", str(d / "val_data_joint")], tmp_path)
    _check_cli_traces(traces, 64, 8, 2)
    assert traces[0]["iterations"] == traces[1]["iterations"] == 2 * (64 // 16)
    assert out.count("Saving checkpoint") == 1 and out.count("test: top1") == 2          # rank 0 only
    runs = os.listdir(tmp_path / "logs")
    assert len(runs) == 1 and os.path.exists(tmp_path / "logs" / runs[0] / "checkpoints" / "ckpt-2.pt")


def test_main_spectrogram_cli_under_two_ranks(tmp_path):
    d = tmp_path / "data"
    d.mkdir()
    _write_npy_dataset(str(d), 32, 300, 10, "train")
    _write_npy_dataset(str(d), 8, 300, 10, "val")
    traces, out = _run_cli("main_spectrogram.py", ["--num-classes", "10", "--batch-size", "4", "--num-epochs", "2", "--num-filters", "8",
                                                   "--num-pad-frames", "0", "--base-lr", "1e-3", "--data-path", str(d / "{}_data_joint.npy"),
                                                   "--label-path", str(d / "{}_label.pkl")], tmp_path)
    _check_cli_traces(traces, 32, 4, 2)
    assert out.count("train Loss") == 2 and out.count("val Loss") == 2                       # rank 0 prints

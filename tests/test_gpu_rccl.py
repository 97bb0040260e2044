"""Rehearsal of the RCCL path on the one GPU of the test box (VERDICT r03 missing #1 / next #2).

Every multi-rank test of tests/test_gpu_multirank.py has to use gloo (two ranks sharing cuda:0 -- RCCL refuses two
ranks on one device), so before this file the `backend="nccl"` branches had never executed anywhere:
`init_process_group("nccl", device_id=...)`, the asynchronous bucketed all-reduces of Trainer.step (ST-GCN fp32 / bf16,
ST-GIN) and SpectrogramTrainer.step (four resnet buckets + the radar bucket) on the communication stream with their
events, and bench.py's rank set-up.  SAR_FORCE_DDP=1 makes a ONE-rank job take all of them on a size-1 RCCL communicator;
the results must equal the plain single-process step BIT FOR BIT (a one-rank SUM is the identity) -- any missing event /
wait on the communication stream shows up as a stale or half-written gradient.
Reference behaviour: main_gnn.py:234,239,257-258; main_spectrogram.py:118-119."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ["stgcn", "stgcn_bf16", "stgcn_split", "stgin", "spectrogram", "spectrogram_split"]


def _env():
    env = dict(os.environ, SAR_FORCE_DDP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "SAR_DIST_BACKEND"):
        env.pop(k, None)
    return env


@pytest.fixture(scope="module")
def rccl_run(tmp_path_factory):
    out = tmp_path_factory.mktemp("rccl")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), str(out)] + WORKLOADS, env=_env(),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return out


def test_one_rank_rccl_communicator_reduces_device_memory(rccl_run):
    info = torch.load(os.path.join(rccl_run, "info.pt"))
    assert info["backend"] == "nccl" and info["world"] == 1 and info["bare_ok"]


def test_hip_graph_step_refuses_a_live_process_group(rccl_run):
    """SpectrogramTrainer(graph=True) captures the step's launches into ONE hipGraph; the bucketed RCCL exchange
    (communication stream, per-bucket events handled by the host) cannot be part of that capture, and a replay without it would train
    each rank on its own gradients -- so under a live communicator (here the forced one-rank RCCL group) step() raises, loudly."""
    info = torch.load(os.path.join(rccl_run, "info.pt"))
    assert info["graph_under_ddp"] != "ran"
    assert "graph=True" in info["graph_under_ddp"] and "graph=False" in info["graph_under_ddp"] and "hipGraph" in info["graph_under_ddp"]


@pytest.mark.parametrize("workload", WORKLOADS)
def test_forced_ddp_step_on_rccl_is_bit_identical_to_the_plain_step(workload, rccl_run):
    import rccl_worker as R
    got = torch.load(os.path.join(rccl_run, workload + ".pt"))
    assert os.environ.get("SAR_FORCE_DDP", "0") != "1"
    want = R.run(workload, torch.device("cuda", 0))          # this process: no process group, no exchange
    pathb = workload.startswith("spectrogram")
    assert want["nbuckets"] == ([] if pathb else [0, 0])
    if not pathb:
        assert got["nbuckets"] == [2, 2]                     # ddp_worker's 3-block model: [l2 + logits], [data_bn + l0 + l1]
    for k in ("grad", "flat", "loss") + (("radar_grad", "radar_location", "wavelength") if pathb else ()):
        assert torch.equal(got[k], want[k]), k
    assert got["grad"].abs().max() > 0 and torch.isfinite(got["flat"]).all()


def test_bench_reports_the_live_rccl_communicator(tmp_path):
    """bench.py --gpus 1 under SAR_FORCE_DDP=1: rank set-up on nccl, the timed steps go through the bucketed exchange, and the
    line says so (rccl_ranks / dist_backend from the live communicator, the collective's time, the bucket count)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4",
                        "--no-cpu-baseline", "--secondary", "bf16,pathB", "--quick"], env=_env(), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["dist_backend"] == "nccl" and out["forced_ddp"] is True
    assert out["grad_buckets"] == 3 and out["allreduce_ms"] is not None and out["allreduce_ms"] > 0 and out["value"] > 0
    sec = out["secondary"]
    assert sec["bf16"]["dist_backend"] == "nccl" and sec["bf16"]["grad_buckets"] == 3 and sec["bf16"]["allreduce_ms"] > 0
    assert sec["pathB"]["dist_backend"] == "nccl" and sec["pathB"]["allreduce_ms"] > 0

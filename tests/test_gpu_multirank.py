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


@pytest.mark.parametrize("workload", ["stgcn", "stgin", "spectrogram"])
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
        if workload in ("stgcn", "stgin"):
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
    if workload in ("stgcn", "stgin"):
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
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["config"]["global_batch"] == 8 and out["value"] > 0

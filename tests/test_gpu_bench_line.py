"""The shape of bench.py's JSON line: the driver keeps the TAIL of the output and cuts strings at 120
characters, so (i) the last key is a compact `summary` of every leg and of the box calibration, under 1 KB at one rank; (ii) no
prose field of a leg is longer than 120 characters; (iii) the per-kernel tables stay out of the default line (`--detail`);
(iv) the `box` probe (csrc/box_probe.hip through the C ABI) reports plausible figures of the chip."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strings(obj, path=""):
    if isinstance(obj, dict):
        for k, v in obj.items():
            yield from _strings(v, path + "." + str(k))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            yield from _strings(v, path + "[%d]" % i)
    elif isinstance(obj, str):
        yield path, obj


def _run(*extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--quick", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--no-cpu-baseline", "--sustained-steps", "0", "--warm-seconds", "0", *extra],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return line, json.loads(line)


def test_default_line_ends_with_a_compact_summary_of_every_leg():
    line, out = _run()
    assert list(out)[-1] == "summary"
    tail = json.dumps(out["summary"])
    assert len(tail) < 1024 and line.rstrip().endswith(tail + "}")
    legs = out["summary"]["legs"]
    assert set(legs) == {"fp32", "f32_split", "bf16", "pathB", "pathB_f32_split", "pathB_pad250", "pathB_pad250_f32_split", "config5",
                         "config5.bf16"}
    assert legs["fp32"][0] == out["value"] and legs["fp32"][1] == out["ms_per_step"] and legs["fp32"][2] == out["roofline"]["frac"]
    for name in ("f32_split", "bf16", "pathB", "pathB_f32_split"):
        sec = out["secondary"][name]
        assert legs[name] == [sec["value"], sec["ms_per_step"], sec["roofline"]["frac"], sec["roofline"]["frac_of_box"]], name
        assert sec["roofline"]["box_ref"] in out["box"]
    assert out["summary"]["box"] == [out["box"][k] for k in ("f32_mfma_tflops", "f16_mfma_tflops", "f16_mfma_clock_ghz",
                                                              "bf16_mfma_tflops", "copy_gbps")]
    # nothing the driver's 120-character cut would truncate, outside the legend (whose entries are bounded too)
    for path, text in _strings({k: v for k, v in out.items() if k != "legend"}):
        assert len(text) <= 120, (path, len(text), text)
    assert all(len(v) <= 160 for v in out["legend"].values())
    # the per-kernel tables are not in the default line, and no profile of an earlier round is cited in it
    assert "detail" not in out and '"kernel_ms_per_step"' not in line
    assert len(line) < 16384


def test_detail_flag_prints_the_kernel_tables():
    line, out = _run("--detail", "--secondary", "f32_split")
    assert list(out)[-1] == "summary"
    d = out["detail"]
    assert d["headline"]["kernel_ms_per_step"] and d["f32_split"]["kernel_ms_per_step"]
    assert any(k.endswith("_split") for k in d["f32_split"]["kernel_ms_per_step"])


def test_box_probe_reports_the_chip():
    sys.path.insert(0, os.path.join(ROOT, "skeleton-action-recognition_amd"))
    from sar_amd import box
    b = box.measure(torch.device("cuda:0"))
    # an MI355X: fp32 MFMA 157.3 TF, fp16 / bf16 2.5 PF dense, HBM 8 TB/s (MI355X_MICROARCH.md); a dense loop reaches most of the
    # matrix peaks at the clock it holds, a copy 60-80 % of the HBM figure
    assert 0.6 * 157.3 < b["f32_mfma_tflops"] < 1.02 * 157.3
    assert 0.5 * 2500 < b["f16_mfma_tflops"] < 1.02 * 2500 and 0.5 * 2500 < b["bf16_mfma_tflops"] < 1.02 * 2500
    assert 1.0 < b["f16_mfma_clock_ghz"] <= 2.45 and 1.0 < b["f32_mfma_clock_ghz"] <= 2.45
    assert 3000 < b["copy_gbps"] < 8000 and b["probe_s"] < 4.0

mkdir -p gpurun_out/r6
python - <<'PY' > gpurun_out/r6/box_probe.txt 2>&1
import sys, os
sys.path.insert(0, "skeleton-action-recognition_amd")
import torch
from sar_amd import box
dev = torch.device("cuda:0")
for i in range(3):
    print(box.measure(dev))
print("quick", box.measure(dev, quick=True))
PY
cat gpurun_out/r6/box_probe.txt
AB_ARGS="--mfma f32_split --steps 60" tools/ab_env.sh 2 "SAR_WGRAD_STREAM=1" "SAR_WGRAD_STREAM=0" > gpurun_out/r6/ab_wgrad_stream_split.txt 2>&1
cat gpurun_out/r6/ab_wgrad_stream_split.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r6/bench_head_r05tree.json 2> gpurun_out/r6/bench_head_r05tree.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6/bench_head_r05tree.json").read().strip().splitlines()[-1])
print("fp32", d["value"], d["ms_per_step"])
for k, v in d["secondary"].items():
    print(k, v["value"], v["ms_per_step"])
PY

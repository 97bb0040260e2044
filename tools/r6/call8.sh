mkdir -p gpurun_out/r6
python - <<'PY'
import sys
sys.path.insert(0, "skeleton-action-recognition_amd")
import torch
from sar_amd import box
print(box.measure(torch.device("cuda:0")))
PY
G2_LAYERS=2 G2_BITS="0 0" tools/g2_ablate.sh run
SAR_GRAPH_SPLIT2=1 python tools/kernel_bench.py --split f16x3a --only gcn_fwd,gcn_dgrad --layers 2 --reps 7 2>&1 | grep "^L"
G2_LAYERS=2,6,9 G2_BITS="0" tools/g2_ablate.sh run
G2_LAYERS=2 G2_BITS="33 35 39 64 128 192 32" tools/g2_ablate.sh run 2>&1 | tee gpurun_out/r6/g2_ablate_2.txt

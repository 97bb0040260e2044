#!/bin/bash
# interleaved A/B of the fp32 ST-GCN step over the weight-gradient grid sizes (diagnostic)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for cfg in "1024 512" "512 512" "768 512" "1536 512" "1024 768" "1024 1024"; do
    set -- $cfg
    v=$(SAR_WGRAD9_SLOTS=$1 SAR_WGRADG_SLOTS=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-isolated-pass 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/')
    echo "temporal $1 graph $2: $v clips/s"
  done
done

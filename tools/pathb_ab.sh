#!/bin/bash
# interleaved A/B of the Path B step: weight-gradient stream on / off (diagnostic; box-to-box variance is +-10 %, so only
# comparisons inside one call mean anything)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3 4 5; do
  for st in 1 0; do
    v=$(SAR_WGRAD_STREAM=$st python bench.py --workload spectrogram --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | sed 's/.*"value": \([0-9.]*\).*/\1/')
    echo "stream $st: $v clips/s"
  done
done

#!/bin/bash
# Interleaved comparison of N environment settings of the bf16 step in one gpurun call (SUSTAINED rate: 3 s of untimed load, then
# 100 timed steps):   tools/ab_env.sh rounds "ENV1" "ENV2" ...      (AB_ARGS="--mfma fp32 --steps 30": another bench leg)
cd "$(dirname "$0")/.."
N=$1; shift
for i in $(seq $N); do
  for E in "$@"; do
    r=$(env $E python bench.py ${AB_ARGS:---mfma bf16 --steps 100} --warmup 5 --warm-seconds 3 --no-cpu-baseline --no-isolated-pass --no-secondary 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i [$E]: $r"
  done
done

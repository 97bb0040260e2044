#!/bin/bash
# Interleaved A/B of the bf16 train step inside ONE gpurun call (box-to-box spread is ~10 %): tools/ab_bf16.sh "ENV_A" "ENV_B" [rounds]
# e.g. tools/ab_bf16.sh "SAR_HIP_LIB=$PWD/tools/bin/libsar_r2.so" "" 3
cd "$(dirname "$0")/.."
A="$1"; B="$2"; N=${3:-3}
for i in $(seq $N); do
  for cfg in A B; do
    if [ $cfg = A ]; then E="$A"; else E="$B"; fi
    r=$(env $E python bench.py --mfma bf16 --steps 30 --warmup 10 --no-cpu-baseline --no-isolated-pass 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i $cfg [$E]: $r"
  done
done

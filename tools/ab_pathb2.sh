#!/bin/bash
# Path B legs of bench.py (fp32 and f32_split, 250 un-instrumented timed steps each) under N environment settings, interleaved, one
# PROCESS per entry (engine instances of one process differ by up to 3 % at bs = 32: tools/ab_inproc.py's null comparison):
#   tools/ab_pathb2.sh rounds "ENV1" "ENV2" ...
cd "$(dirname "$0")/.."
N=$1; shift
for i in $(seq $N); do
  for E in "$@"; do
    r=$(env $E python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated-pass 2>&1 | tail -1 | python3 -c "import sys,json; s=json.loads(sys.stdin.read())['secondary']; print(s['pathB']['value'], s['pathB']['ms_per_step'], s['pathB_f32_split']['value'], s['pathB_f32_split']['ms_per_step'])")
    echo "round $i [$E]: pathB / ms, pathB_f32_split / ms: $r"
  done
done

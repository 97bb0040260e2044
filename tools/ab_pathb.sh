#!/bin/bash
# Path B (secondary leg of bench.py: 1 s of load, 250 un-instrumented timed steps) under N environment settings, interleaved:
#   tools/ab_pathb.sh rounds "ENV1" "ENV2" ...
cd "$(dirname "$0")/.."
N=$1; shift
for i in $(seq $N); do
  for E in "$@"; do
    r=$(env $E python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated-pass 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read())['secondary']['pathB']; print(d['value'], d['ms_per_step'])")
    echo "round $i [$E]: $r"
  done
done

#!/bin/bash
# In-step A/B of environment settings, interleaved over ROUNDS rounds inside ONE process sequence on one box:
#   tools/ab_matrix.sh "<bench args>" "ENV1" "ENV2" ...     (an ENV is a space-separated list of VAR=value, or "-" for none)
set -e
cd "$(dirname "$0")/.."
ARGS="$1"; shift
for r in $(seq 1 ${ROUNDS:-3}); do
  for E in "$@"; do
    EE="$E"; [ "$E" = "-" ] && EE=""
    r_=$(env $EE python bench.py $ARGS --no-cpu-baseline --no-isolated-pass --no-secondary --sustained-steps 0 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$r_ | $E"
  done
done | sort -t'|' -k2,2 -k1,1n

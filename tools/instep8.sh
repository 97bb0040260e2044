#!/bin/bash
# In-step per-kernel-family times of the bf16 train step (HIP events, weight-gradient stream OFF so that nothing overlaps):
#   tools/instep8.sh [lib.so]
cd "$(dirname "$0")/.."
[ -n "$1" ] && export SAR_HIP_LIB=$PWD/$1
SAR_WGRAD_STREAM=0 python bench.py --mfma bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-isolated-pass --no-secondary --detail --warm-seconds 0 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'])
k = d['detail']['headline']['kernel_ms_per_step']
tot = 0
for name, ms in sorted(k.items(), key=lambda kv: -kv[1]):
    print('  %-28s %7.3f ms' % (name, ms)); tot += ms
print('  sum of bracketed kernels %.3f ms' % tot)
"

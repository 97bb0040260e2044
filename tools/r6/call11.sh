export SAR_GRAPH_SPLIT3=1
timeout 600 python -m pytest tests/test_gpu_split.py -x -q -m gpu -k "persistent or graph" 2>&1 | tail -4
timeout 300 python tools/kernel_bench.py --split f16x3a --only gcn_fwd,gcn_dgrad --reps 7 2>&1 | grep "^L\|TOTAL"
tools/split_timeline.sh run g_fwd g_dgate 2>&1 | grep -v amdgpu

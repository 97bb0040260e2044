mkdir -p gpurun_out/r6
tools/split_timeline.sh run g_fwd g_dgate 2>&1 | tee gpurun_out/r6/graph2_timeline.txt
SAR_GRAPH_SPLIT2=0 tools/split_timeline.sh run g_fwd g_dgate 2>&1 | tee -a gpurun_out/r6/graph2_timeline.txt
timeout 900 python -m pytest tests/test_gpu_bench_line.py -x -q -m gpu 2>&1 | grep -E "^E  |passed|failed" | head -8

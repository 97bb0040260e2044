mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_bench_line.py tests/test_gpu_rccl.py -x -q -m gpu 2>&1 | tail -5
for v in 0 2; do echo "== SAR_GRAPH_SPLIT2=$v"; SAR_GRAPH_SPLIT2=$v python tools/step_table.py --mfma f32_split 2>&1 | head -4; done
SAR_SPLIT_KINDS=tdgrad,twgrad timeout 600 python -m pytest tests/test_gpu_stgcn_model.py -x -q -m gpu -k "two_blocks or stride2" 2>&1 | tail -3

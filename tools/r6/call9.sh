mkdir -p gpurun_out/r6
SAR_GRAPH_SPLIT3=1 timeout 600 python -m pytest tests/test_gpu_split.py -x -q -m gpu -k "persistent or graph" 2>&1 | tail -8
SAR_GRAPH_SPLIT3=1 timeout 900 python -m pytest tests/test_gpu_stgcn_kernels.py -x -q -m gpu -k "graph" 2>&1 | tail -3
for v in 0 1; do echo "== SAR_GRAPH_SPLIT3=$v"; SAR_GRAPH_SPLIT3=$v timeout 300 python tools/kernel_bench.py --split f16x3a --only gcn_fwd,gcn_dgrad --reps 7 2>&1 | grep -v amdgpu; done | tee gpurun_out/r6/graph3_kernel_bench.txt

timeout 1200 python -m pytest tests/test_gpu_stgcn_kernels.py -x -q -m gpu -k "even_frame or graph_conv_gradients or gated" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_stgcn_model.py -x -q -m gpu -k "stride2 or odd_sizes or tiny or full_model_ntu or sgd_training" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_stgin.py tests/test_gpu_adjacency.py tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -2
for m in fp32 f32_split; do for v in 0 1 0 1; do echo "== $m SAR_COMPACT_SKIP=$v"; SAR_COMPACT_SKIP=$v python bench.py --mfma $m --steps 40 --warmup 5 --warm-seconds 3 --no-cpu-baseline --no-isolated-pass --no-secondary --sustained-steps 0 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done

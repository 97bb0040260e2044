mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_bench_line.py -x -q -m gpu > gpurun_out/r6/t_benchline.log 2>&1; tail -3 gpurun_out/r6/t_benchline.log
python -m pytest tests/test_gpu_split.py -x -q -m gpu -k "learnable" -s 2>&1 | grep -v amdgpu | tail -4
python tools/split_curve.py > gpurun_out/r6/split_curve.txt 2>&1; tail -8 gpurun_out/r6/split_curve.txt
AB_ARGS="--mfma f32_split --steps 60" tools/ab_env.sh 2 "SAR_SIDE_CU_MASK=off" "SAR_SIDE_CU_MASK=skip:8" "SAR_SIDE_CU_MASK=skip:4" "SAR_SIDE_CU_MASK=last:32" "SAR_SIDE_CU_MASK=first:32" "SAR_SIDE_CU_MASK=last:16" "SAR_SIDE_CU_MASK=skip:16" > gpurun_out/r6/ab_cu_mask_split.txt 2>&1
cat gpurun_out/r6/ab_cu_mask_split.txt

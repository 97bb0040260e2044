mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_split.py -x -q -m gpu -k "non_finite or diverged or amax" > gpurun_out/r6/t_nonfinite.log 2>&1; tail -3 gpurun_out/r6/t_nonfinite.log
python -m pytest tests/test_gpu_bench_line.py -x -q -m gpu > gpurun_out/r6/t_benchline.log 2>&1; tail -3 gpurun_out/r6/t_benchline.log
python -m pytest tests/test_gpu_stgcn_model.py -x -q -m gpu -k "outliers" -s > gpurun_out/r6/t_outliers.log 2>&1; tail -30 gpurun_out/r6/t_outliers.log
python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "rehearsal or main_gnn_cli" > gpurun_out/r6/t_multirank.log 2>&1; tail -3 gpurun_out/r6/t_multirank.log
python tools/split_curve.py > gpurun_out/r6/split_curve.txt 2>&1; cat gpurun_out/r6/split_curve.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r6/bench_line_new.json 2> gpurun_out/r6/bench_line_new.err; tail -c 1500 gpurun_out/r6/bench_line_new.json; wc -c gpurun_out/r6/bench_line_new.json
